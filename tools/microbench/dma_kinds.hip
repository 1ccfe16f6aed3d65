// Micro-benchmark: SIMD issue cost of one 1-KiB LDS-DMA wave-instruction in three encodings.
//   0: global_load_lds_dwordx4 (64-bit per-lane address in two VGPRs)
//   1: buffer_load_dwordx4 ... offen lds (SGPR resource + SGPR offset + one 32-bit VGPR lane offset)
//   2: buffer_load_dwordx4 ... off lds with ADD_TID_ENABLE in the resource (no VGPR at all: address = base + soffset + 16 * lane)
// 4 waves per CU (one per SIMD), each streaming `NP` pieces per step from an L2-resident buffer, two slots per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NP, int KIND>
__global__ __launch_bounds__(256) void k(const char* src, unsigned span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    char* lds = smem + wave * NP * 1024 * 2;
    unsigned off = wave * NP * 1024;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)src, 16, 0x7fffffff, 0x00020000 | (1 << 23));
    for (int it = 0; it < iters; ++it) {
        char* dst = lds + (it & 1) * NP * 1024;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned so = (off + p * 1024) & (span - 1);
            if (KIND == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + so + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            else if (KIND == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, lane * 16, so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, so, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        off += 4 * NP * 1024;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((float*)smem)[threadIdx.x] == 1.2345f) sink[0] = 1.f;
}

template <int NP, int KIND>
static void run(const char* name, const char* buf, float* sink) {
    const int iters = 4000;
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k<NP, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NP, KIND><<<256, 256, lds>>>(buf, 4u << 20, 50, sink);
    (void)hipEventRecord(e0);
    k<NP, KIND><<<256, 256, lds>>>(buf, 4u << 20, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    printf("%-44s %d pieces/wave/step: %.3f us/step = %.0f ns per piece per wave, %.1f TB/s\n", name, NP, us, us * 1e3 / NP, 256.0 * 4 * NP * 1024 / us / 1e6);
}

int main() {
    char* buf; float* sink;
    (void)hipMalloc(&buf, 8 << 20); (void)hipMemset(buf, 1, 8 << 20); (void)hipMalloc(&sink, 64);
    run<6, 0>("global_load_lds_dwordx4", buf, sink);
    run<6, 1>("buffer_load_dwordx4 offen lds", buf, sink);
    run<6, 2>("buffer_load_dwordx4 off lds (ADD_TID_ENABLE)", buf, sink);
    return 0;
}
