// Micro-benchmark: cycles per MFMA for one wave per SIMD (4 waves/CU) and two (8 waves/CU), independent accumulators,
// with the shader clock measured in-kernel (s_memtime ticks / 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool I8>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* stamps, float* sink) {
    const int lane = threadIdx.x & 63;
    using AccT = typename std::conditional<I8, i32x16, f32x16>::type;
    AccT acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    const i32x4 ia = {lane, lane + 1, lane + 2, lane + 3}, ib = {lane * 3, lane * 5, lane * 7, lane * 9};
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if constexpr (I8) acc[m % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, acc[m % NACC], 0, 0, 0);
            else acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += (float)acc[i][3];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
    if (s == 1.2345f) sink[0] = s;
}

template <int NACC, bool I8>
static void run(int nw, unsigned long long* stamps, float* sink) {
    const int iters = 4000;
    (void)hipMemset(stamps, 0, 16);
    k<NACC, I8><<<256, nw * 64>>>(50, stamps, sink);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<NACC, I8><<<256, nw * 64>>>(iters, stamps, sink);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    unsigned long long h[2];
    (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0], us = (double)h[1] / 100.0;
    const double mfma_per_simd = (double)iters * 24 * (nw / 4);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double total = 256.0 * nw * iters * 24;  // MFMAs of the whole grid
    printf("%s, %d waves/CU, %d independent accumulators: %.1f cycles per MFMA per SIMD at %.0f MHz  (%.1f ns); whole grid %.3f ms -> %.2f P(FL)OP/s\n", I8 ? "i32_32x32x32_i8  " : "f32_32x32x16_bf16",
           nw, NACC, cyc / mfma_per_simd, cyc / us, us * 1e3 / mfma_per_simd, ms, total * (I8 ? 65536.0 : 32768.0) / (ms * 1e-3) / 1e15);
}

int main() {
    unsigned long long* stamps; float* sink;
    (void)hipMalloc(&stamps, 64); (void)hipMalloc(&sink, 64);
    run<8, false>(4, stamps, sink);
    run<8, false>(8, stamps, sink);
    run<4, false>(4, stamps, sink);
    run<8, true>(4, stamps, sink);
    run<8, true>(8, stamps, sink);
    run<2, true>(4, stamps, sink);
    return 0;
}
