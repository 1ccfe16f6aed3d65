// Micro-benchmark: the library's own GEMM main loop (gemm.h GemmBody::mainloop) in isolation, on synthetic operands:
// 256 workgroups (one per CU), a [512 x K] weight (L2-resident, shared) against [32768 x K] activations (one 128-row block
// per workgroup), no epilogue.  A fast harness for main-loop experiments: compile this file alone.
// Caveat (measured): operands here are constant bytes and nothing surrounds the loop, so rankings can differ from the real
// kernels — a 2-stage ring wins here (1.03 vs 1.21 us per k-step) and loses inside layer_tail_kernel (240 vs 226 us on the same
// node).  Confirm every finding with an A/B of the real kernel (EGOEGO_HIP_LIB).
#include "../../egoego_release_amd/csrc/gemm.h"
#include <cstdio>

struct NoEpiB {};

template <class C, class AccT>
__global__ __launch_bounds__(C::NT, C::MINW) void k(GemmOperands g, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AccT acc[C::FT][C::TT];
    GemmBody<C, NoEpiB>::mainloop(g, 0, (int)blockIdx.x, smem, acc);
    float s = 0.f;
    for (int i = 0; i < C::FT; ++i)
        for (int j = 0; j < C::TT; ++j) {
            if constexpr (std::is_same<AccT, f32x16>::value) s += acc[i][j][3];
            else s += (float)(acc[i][j].h[3] + acc[i][j].m[5]);
        }
    if (s == 1.2345f) sink[0] = s;
}

template <class C, class AccT>
static void run(const char* name, int K, int kdiv, const __bf16* w, const __bf16* a, size_t w_plane, size_t a_plane, float* sink) {
    GemmOperands g{w, w_plane, a, a_plane, K / kdiv, 1, 256, 0, 0, nullptr};
    (void)hipFuncSetAttribute((const void*)k<C, AccT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 32768 / C::BT;
    g.ntb = grid;
    k<C, AccT><<<grid, C::NT, C::SMEM_BYTES>>>(g, sink);
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<C, AccT><<<grid, C::NT, C::SMEM_BYTES>>>(g, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, steps = (double)(K / kdiv) / C::KS * C::KS;
    printf("%-52s K=%d: %.1f us per launch, %.3f us per k-step per 128 tokens\n", name, K, us, us / steps * 128.0 / C::BT * (grid > 256 ? 256.0 / grid * C::BT / 128.0 * (double)grid / 256.0 : 1.0));
}

#include <type_traits>
int main() {
    const size_t w_elems = (size_t)512 * 1024, a_elems = (size_t)32768 * 1024;
    __bf16 *w, *a; float* sink;
    (void)hipMalloc(&w, 2 * w_elems * 2); (void)hipMalloc(&a, 2 * a_elems * 2); (void)hipMalloc(&sink, 64);
    (void)hipMemset(w, 0x11, 2 * w_elems * 2); (void)hipMemset(a, 0x11, 2 * a_elems * 2);
    using CfgB = GemmCfg<4, 2, 4, 2, 1, 2, false, 1, 3>;
    run<CfgB, f32x16>("tail tile 512f x 128t, 8 waves, split-bf16, 3-stage", 1024, 16, w, a, w_elems, a_elems, sink);
    run<CfgB, f32x16>("tail tile 512f x 128t, 8 waves, split-bf16, 3-stage", 512, 16, w, a, w_elems, a_elems, sink);
    run<GemmCfg<4, 2, 4, 2, 1, 2, false, 1, 4>, f32x16>("  same, 4-stage ring (160 KiB)", 1024, 16, w, a, w_elems, a_elems, sink);
    run<GemmCfg<4, 2, 4, 2, 1, 2, false, 1, 2>, f32x16>("  same, 2-stage ring", 1024, 16, w, a, w_elems, a_elems, sink);
    run<GemmCfg<2, 4, 8, 1, 1, 2, false, 1, 3>, f32x16>("  waves 8f x 1t (64f x 128t per wave)", 1024, 16, w, a, w_elems, a_elems, sink);
    run<GemmCfg<8, 1, 2, 4, 1, 2, false, 1, 3>, f32x16>("  waves 2f x 4t (256f x 32t per wave)", 1024, 16, w, a, w_elems, a_elems, sink);
    run<GemmCfg<4, 2, 4, 1, 1, 2, false, 2, 2>, f32x16>("  512f x 64t, 4 waves, 2 WG/CU (512 WGs)", 1024, 16, w, a, w_elems, a_elems, sink);
    using CfgI = GemmCfg<4, 2, 2, 2, 1, 2, false, 1, 3>;
    run<CfgI, I8Acc>("attention projection 256f x 128t, 4 waves, int8 slices", 512, 32, w, a, w_elems, a_elems, sink);
    return 0;
}
