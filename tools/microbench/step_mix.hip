// Micro-benchmark: the instruction mix of one k-step of the split-bf16 layer-tail main loop, component by component.
// 8 waves per CU (2 per SIMD), per wave and step: NM MFMAs, NR ds_read_b128 (1 KiB per wave-instruction) whose results feed
// the MFMAs, NP LDS-DMA pieces (1 KiB each) into a ring, one barrier.  Which components are on is a template mask.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NW, int NM, int NR, int NP, int MASK>  // MASK: 1 = MFMA, 2 = LDS reads, 4 = DMA
__global__ __launch_bounds__(NW * 64) void k(const char* src, unsigned span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fr[NR > 0 ? NR : 1];
    for (int i = 0; i < (NR > 0 ? NR : 1); ++i)
        for (int e = 0; e < 8; ++e) fr[i][e] = (__bf16)(float)(lane + e + i);
    char* ring = smem + 48 * 1024;  // DMA target, separate from the region the reads walk
    unsigned off = wave * NP * 1024;
    for (int it = 0; it < iters; ++it) {
        if (MASK & 2) {
#pragma unroll
            for (int r = 0; r < NR; ++r) fr[r] = *(const bf16x8*)(smem + ((it + r * 5 + wave * 3) % 48) * 1024 + lane * 16);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (MASK & 1) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % (NR > 0 ? NR : 1)], fr[(m + 1) % (NR > 0 ? NR : 1)], acc[m & 7], 0, 0, 0);
            if ((MASK & 4) && m < NP) {
                __builtin_amdgcn_sched_barrier(0);
                const char* g = src + ((off + m * 1024) & (span - 1)) + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(ring + ((it & 1) * NW * NP + wave * NP + m) * 1024), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MASK & 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
        off += NW * NP * 1024;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][3];
    for (int i = 0; i < (NR > 0 ? NR : 1); ++i) s += (float)fr[i][1];
    if (s == 1.2345f) sink[0] = s;
}

template <int NW, int NM, int NR, int NP, int MASK>
static float run(const char* buf, float* sink) {
    const int iters = 2000;
    const size_t lds = 48 * 1024 + 2 * NW * NP * 1024;
    (void)hipFuncSetAttribute((const void*)k<NW, NM, NR, NP, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NW, NM, NR, NP, MASK><<<256, NW * 64, lds>>>(buf, 4u << 20, 50, sink);
    (void)hipEventRecord(e0);
    k<NW, NM, NR, NP, MASK><<<256, NW * 64, lds>>>(buf, 4u << 20, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int NW, int NM, int NR, int NP>
static void report(const char* what, const char* buf, float* sink) {
    printf("%s (%d waves; per wave-step %d MFMA, %d ds_read_b128, %d DMA KiB):\n", what, NW, NM, NR, NP);
    printf("   MFMA %.3f | reads %.3f | DMA %.3f | MFMA+reads %.3f | MFMA+DMA %.3f | reads+DMA %.3f | all %.3f us per step\n",
           run<NW, NM, NR, NP, 1>(buf, sink), run<NW, NM, NR, NP, 2>(buf, sink), run<NW, NM, NR, NP, 4>(buf, sink), run<NW, NM, NR, NP, 3>(buf, sink),
           run<NW, NM, NR, NP, 5>(buf, sink), run<NW, NM, NR, NP, 6>(buf, sink), run<NW, NM, NR, NP, 7>(buf, sink));
}

int main() {
    char* buf; float* sink;
    (void)hipMalloc(&buf, 8 << 20); (void)hipMemset(buf, 1, 8 << 20); (void)hipMalloc(&sink, 64);
    report<8, 24, 12, 5>("layer tail, split-bf16", buf, sink);
    report<4, 24, 12, 6>("attention-layer projections", buf, sink);
    return 0;
}
