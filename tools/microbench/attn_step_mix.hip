// Micro-benchmark (round 4): one k-step of the attention layer's int8 projection main loop under three operand paths, all CUs busy,
// eight 256-register waves per CU (two per SIMD), 12 v_mfma_i32_32x32x32_i8 per wave and k-step (a 64f x 64t wave tile, three MFMAs
// per tile pair) on operands that really arrive through the path under test:
//   ring   (shipped: attn_layer_i8w.h / gemm.h)  8 ds_read_b128 + 3 LDS-DMA pieces per wave-step, one barrier per k-step, 3-stage ring
//   wdir42  4(f) x 2(t) waves, weights global -> VGPR (4 buffer_load_b128 per wave-step, register ring 3 steps
//                           ahead), activations through LDS chunks (4 ds_read_b128 + 1 DMA piece per wave-step, one barrier per 4 steps)
//   wdir81                  8(f) x 1(t) waves (32f x 128t tiles): 2 weight loads + 8 ds_read_b128 + 1 DMA piece, barrier per 4 steps
//   alldir (round 3's loser) 8 buffer loads, no LDS
// MASK removes components (1 = MFMA, 2 = LDS reads, 4 = DMA pieces, 8 = global->VGPR loads) to price them.
// Operand bytes come from an L2-resident span with random bits (the MFMA clock depends on the data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

template <int VM, int LGKM>
__device__ __forceinline__ void wait_counts() {
    __builtin_amdgcn_s_waitcnt((VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (LGKM << 8));
}

// NRL ds_read_b128, NGL global->VGPR b128 loads, NP LDS-DMA pieces per wave-step; NB k-steps per barrier; PD prefetch distance (steps)
template <int NRL, int NGL, int NP, int NB, int MASK>
__global__ __launch_bounds__(512, 2) void k(const char* src, unsigned span, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PD = 3, NM = 12, NF = 8;  // 8 operand fragments feed the 12 MFMAs (2 slices x (2 weight + 2 activation tiles))
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    i32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    i32x4 fl[NF];                       // fragments from LDS (or constants when that path is off)
    i32x4 fg[PD + 1][NGL > 0 ? NGL : 1];  // register ring of the global -> VGPR fragments
    for (int i = 0; i < NF; ++i) fl[i] = i32x4{lane * 3 + i, lane + 7 * i, lane ^ i, 11 * i};
    for (int s = 0; s <= PD; ++s)
        for (int i = 0; i < (NGL > 0 ? NGL : 1); ++i) fg[s][i] = i32x4{lane + s, i, lane * s, 5};
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)span, 0x00020000);
    char* img = smem;                   // 64 KiB the reads walk
    char* ring = smem + 64 * 1024;      // DMA target (3 x 24 KiB at most)
    unsigned off = (blockIdx.x * 8 + wave) * 4096u;
    auto gload = [&](int slot, unsigned o) {
#pragma unroll
        for (int i = 0; i < NGL; ++i)
            if (MASK & 8) fg[slot][i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (o + i * 1024) & (span - 1), 0));
    };
    auto pieces = [&](int it, unsigned o) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (MASK & 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + (((it % 3) * 8 + wave) * NP + p) * 1024), 16, lane * 16,
                                                         (o + 65536 + p * 1024) & (span - 1), 0, 0);
    };
    for (int s = 0; s < PD; ++s) {
        pieces(s, off + s * 8192);
        gload(s, off + s * 8192);
    }
    for (int it4 = 0; it4 < iters; it4 += PD + 1)
#pragma unroll
    for (int u = 0; u <= PD; ++u) {  // (unrolled by the ring depth: every register-ring index is a compile-time constant)
        const int it = it4 + u;
        // operands of this step have landed: everything but the PD - 1 younger steps' loads
        asm volatile("" ::: "memory");
        wait_counts<((PD - 1) * (NGL + NP) < 63 ? (PD - 1) * (NGL + NP) : 63), 15>();
        if (u % NB == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pieces(it + PD, off + (it + PD) * 8192);
        gload((u + PD) % (PD + 1), off + (it + PD) * 8192);
        if (MASK & 2) {
#pragma unroll
            for (int r = 0; r < NRL; ++r) fl[r] = *(const i32x4*)(img + ((it * 7 + r * 5 + wave * 3) & 63) * 1024 + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int cur = u;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            // operand A from the weight side (global ring if present, else LDS), operand B from the activation side (LDS)
            const i32x4 a = NGL > 0 ? fg[cur][m % NGL] : fl[m % 4];
            const i32x4 b = fl[NRL > 4 ? 4 + m % 4 : (NRL > 0 ? m % NRL : m % NF)];
            if (MASK & 1) acc[m & 7] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[m & 7], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][3];
    for (int i = 0; i < NF; ++i) s += fl[i][1];
    for (int i = 0; i < (NGL > 0 ? NGL : 1); ++i) s += fg[0][i][2];
    if (s == 0x12345678) sink[0] = s;
}

template <int NRL, int NGL, int NP, int NB, int MASK>
static float run(const char* buf, int* sink) {
    const int iters = 3000;  // (a multiple of the ring depth)
    const size_t lds = 148 * 1024;  // one workgroup per CU, as the kernel
    (void)hipFuncSetAttribute((const void*)k<NRL, NGL, NP, NB, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NRL, NGL, NP, NB, MASK><<<256, 512, lds>>>(buf, 4u << 20, 100, sink);
    (void)hipEventRecord(e0);
    k<NRL, NGL, NP, NB, MASK><<<256, 512, lds>>>(buf, 4u << 20, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int NRL, int NGL, int NP, int NB>
static void report(const char* what, const char* buf, int* sink) {
    printf("%-8s per wave-step: 12 MFMA, %d ds_read_b128, %d global b128, %d DMA KiB, barrier every %d:", what, NRL, NGL, NP, NB);
    printf("  MFMA only %.3f | all %.3f | no MFMA %.3f | no LDS reads %.3f | no DMA %.3f | no global loads %.3f  us per k-step\n",
           run<NRL, NGL, NP, NB, 1>(buf, sink), run<NRL, NGL, NP, NB, 15>(buf, sink), run<NRL, NGL, NP, NB, 14>(buf, sink), run<NRL, NGL, NP, NB, 13>(buf, sink),
           run<NRL, NGL, NP, NB, 11>(buf, sink), run<NRL, NGL, NP, NB, 7>(buf, sink));
}

int main() {
    char* buf; int* sink;
    const size_t n = 8 << 20;
    (void)hipMalloc(&buf, n); (void)hipMalloc(&sink, 64);
    char* h = (char*)malloc(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (char)(rand() >> 7);
    (void)hipMemcpy(buf, h, n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        report<8, 0, 3, 1>("ring", buf, sink);
        report<4, 4, 1, 4>("wdir42", buf, sink);
        report<8, 2, 1, 4>("wdir81", buf, sink);
    }
    return 0;
}
