// Micro-benchmark: producer / consumer wave specialisation.  An 8-wave workgroup, one per CU: waves 0-3 only stream
// (LDS-DMA, NP pieces of 1 KiB per step each), waves 4-7 only multiply (NM MFMAs per step each); one barrier per step.
// Compared with the same total work issued by 4 waves that do both (tools/microbench/dma_mfma.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NP, int MODE, int NSW>  // MODE 0: both roles, 1: only the streaming waves work, 2: only the multiplying waves work; NSW streaming waves
__global__ __launch_bounds__(512) void k(const char* src, unsigned span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    if (wave < NSW) {
        char* lds = smem + wave * NP * 1024 * 2;
        unsigned off = wave * NP * 1024;  // 32-bit offsets, power-of-two span: the streaming waves must not burn VALU on address math
        for (int it = 0; it < iters; ++it) {
            if (MODE != 2) {
                char* dst = lds + (it & 1) * NP * 1024;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const char* g = src + ((off + p * 1024) & (span - 1)) + lane * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            off += NSW * NP * 1024;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MODE != 1) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][3];
    if (s == 1.2345f) sink[0] = s + smem[threadIdx.x];
}

template <int NM, int NP, int MODE, int NSW, int NW>
static float run(const char* buf, float* sink) {
    const int iters = 2000;
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k<NM, NP, MODE, NSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NM, NP, MODE, NSW><<<256, NW * 64, lds>>>(buf, 4u << 20, 50, sink);
    (void)hipEventRecord(e0);
    k<NM, NP, MODE, NSW><<<256, NW * 64, lds>>>(buf, 4u << 20, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int NM, int NP, int NSW = 4, int NW = 8>
static void report(const char* buf, float* sink) {
    const float d = run<NM, NP, 1, NSW, NW>(buf, sink), m = run<NM, NP, 2, NSW, NW>(buf, sink), b = run<NM, NP, 0, NSW, NW>(buf, sink);
    printf("%d streaming waves x %2d KiB + %d multiplying waves x %2d MFMA per step: DMA only %.3f us (%.1f TB/s), MFMA only %.3f us, both %.3f us (max %.3f, sum %.3f)\n",
           NSW, NP, NW - NSW, NM, d, 256.0 * NSW * NP * 1024 / d / 1e6, m, b, m > d ? m : d, m + d);
}

int main() {
    char* buf; float* sink;
    (void)hipMalloc(&buf, 8 << 20); (void)hipMemset(buf, 1, 8 << 20); (void)hipMalloc(&sink, 64);
    report<24, 6>(buf, sink);
    report<12, 4>(buf, sink);
    report<24, 10>(buf, sink);
    report<48, 10>(buf, sink);
    report<48, 12, 2, 4>(buf, sink);  // 4 waves: SIMDs 0-1 stream, SIMDs 2-3 multiply (no SIMD shared)
    report<24, 6, 2, 4>(buf, sink);
    return 0;
}
