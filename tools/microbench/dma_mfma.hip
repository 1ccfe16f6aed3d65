// Micro-benchmark: do LDS-DMA operand streaming and MFMAs overlap when the SAME waves issue both?
// Every wave runs `iters` k-steps of NM MFMAs (independent accumulators) with NP LDS-DMA pieces (1 KiB each) dropped
// between them, ring of two slots per wave, one workgroup barrier per step — the shape of the GEMM main loop without
// its LDS reads.  Reported: time per step for MFMA only, DMA only, and both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NP, bool DO_MFMA, bool DO_DMA, int KIND, int STAG>
__global__ __launch_bounds__(512) void k(const char* src, unsigned span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    char* lds = smem + wave * NP * 1024 * 2;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    unsigned off = wave * NP * 1024;  // 32-bit offsets and a power-of-two span: no 64-bit modulo on the streaming path
    for (int it = 0; it < iters; ++it) {
        char* dst = lds + (it & 1) * NP * 1024;
        // STAG 0: piece p right after MFMA p (every wave at the same point of the step)
        // STAG 1: pieces spread evenly over the step, and wave w shifted by w * NM / nw MFMAs (round robin over the waves)
        const int shift = STAG ? (wave * NM) / nw : 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (DO_MFMA) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
            if (DO_DMA) {
                const int mm = (m + NM - shift) % NM;  // position in this wave's own schedule
                constexpr int SPREAD = STAG ? NM / NP : 1;
                if (mm % SPREAD == 0 && mm / SPREAD < NP) {
                    const int p = mm / SPREAD;
                    __builtin_amdgcn_sched_barrier(0);
                    const char* g = src + ((off + p * 1024) & (span - 1)) + lane * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (DO_DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
        off += nw * NP * 1024;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][3];
    if (s == 1.2345f) sink[0] = s + smem[threadIdx.x];
}

template <int NM, int NP, bool M, bool D, int KIND, int STAG>
static float run(int nw, const char* buf, float* sink) {
    const int iters = 2000;
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k<NM, NP, M, D, KIND, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NM, NP, M, D, KIND, STAG><<<256, nw * 64, lds>>>(buf, 4u << 20, 50, sink);
    (void)hipEventRecord(e0);
    k<NM, NP, M, D, KIND, STAG><<<256, nw * 64, lds>>>(buf, 4u << 20, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;  // us per step
}

template <int NM, int NP, int STAG = 0>
static void report(int nw, const char* buf, float* sink) {
    const float m = run<NM, NP, true, false, 0, STAG>(nw, buf, sink), d = run<NM, NP, false, true, 0, STAG>(nw, buf, sink), b = run<NM, NP, true, true, 0, STAG>(nw, buf, sink);
    printf("%s waves %d,", STAG ? "staggered:   " : "synchronous: ", nw);
    printf(" %2d MFMA + %d KiB DMA per wave-step: MFMA only %.3f us, DMA only %.3f us (%.1f TB/s), both %.3f us  (max %.3f, sum %.3f)\n", NM,
           NP, m, d, 256.0 * nw * NP * 1024 / d / 1e6, b, m > d ? m : d, m + d);
}

int main() {
    char* buf; float* sink;
    (void)hipMalloc(&buf, 8 << 20); (void)hipMemset(buf, 1, 8 << 20); (void)hipMalloc(&sink, 64);
    report<24, 5>(8, buf, sink);   // split-bf16 layer tail: 24 MFMAs and 5 pieces per wave per k-step, 2 waves per SIMD
    report<24, 6>(4, buf, sink);   // attention-layer projections: 24 MFMAs and 6 pieces, 1 wave per SIMD
    report<24, 3>(8, buf, sink);
    report<24, 10>(8, buf, sink);
    report<48, 5>(8, buf, sink);
    report<24, 5, 1>(8, buf, sink);
    report<24, 6, 1>(4, buf, sink);
    report<24, 3, 1>(8, buf, sink);
    report<48, 5, 1>(8, buf, sink);
    return 0;
}
