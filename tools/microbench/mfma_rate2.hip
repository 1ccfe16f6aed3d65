// Micro-benchmark: what an MFMA-only loop sustains when it looks like a real k-step — 16 independent accumulators,
// 8 weight-fragment registers x 4 (or 8) activation-fragment registers, operands holding random data or small
// integers — one wave per SIMD on every CU.  Reports cycles per MFMA, the shader clock and ns per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA, int NB, bool RANDOM>
__global__ __launch_bounds__(256, 1) void k(int iters, const i32x4* data, unsigned long long* stamps, float* sink) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NA / 2][NB];
    for (int i = 0; i < NA / 2; ++i)
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    i32x4 a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = RANDOM ? data[(i * 64 + lane) & 4095] : i32x4{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    for (int j = 0; j < NB; ++j) b[j] = RANDOM ? data[((j + 17) * 64 + lane) & 4095] : i32x4{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int i = 0; i < NA / 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * i + (part == 1)]),
                                                                       __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    float s = 0.f;
    for (int i = 0; i < NA / 2; ++i)
        for (int j = 0; j < NB; ++j) s += acc[i][j][3];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
    if (s == 1.2345f) sink[0] = s;
}

template <int NA, int NB, bool RANDOM>
static void run(const i32x4* data, unsigned long long* stamps, float* sink, const char* what) {
    const int iters = 2000;
    k<NA, NB, RANDOM><<<256, 256>>>(50, data, stamps, sink);
    k<NA, NB, RANDOM><<<256, 256>>>(iters, data, stamps, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    unsigned long long h[2];
    (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
    const double n = (double)iters * 3 * (NA / 2) * NB;
    printf("%-44s %2d accumulators: %.1f cycles per MFMA at %.0f MHz = %.1f ns\n", what, (NA / 2) * NB, (double)h[0] / n,
           (double)h[0] / ((double)h[1] / 100.0), (double)h[1] * 10.0 / n);
}

int main() {
    unsigned long long* stamps; float* sink; i32x4* data;
    (void)hipMalloc(&stamps, 64); (void)hipMalloc(&sink, 64); (void)hipMalloc(&data, 4096 * 16);
    int* h = (int*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 4; ++i) {  // random bf16 pairs in [-2, 2]: sign, exponent 0x3f/0x40 area, random mantissa
        unsigned lo = (rand() & 0x80ff) | 0x3f00, hi = (rand() & 0x80ff) | 0x3f00;
        h[i] = (int)(lo | (hi << 16));
    }
    (void)hipMemcpy(data, h, 4096 * 16, hipMemcpyHostToDevice);
    run<8, 4, false>(data, stamps, sink, "4x4 tiles, operands all ones");
    run<8, 4, true>(data, stamps, sink, "4x4 tiles, random operands");
    run<4, 4, false>(data, stamps, sink, "2x4 tiles, operands all ones");
    run<4, 4, true>(data, stamps, sink, "2x4 tiles, random operands");
    run<8, 2, true>(data, stamps, sink, "4x2 tiles, random operands");
    return 0;
}
