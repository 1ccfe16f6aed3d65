// tail_i8.h — the layer tail of the i8x3 precision: fc + residual + LayerNorm -> FFN w_1 + ReLU -> FFN w_2 + residual + LayerNorm
// (TM:92-93, 107-116, 135, 139) for one 64-token block per workgroup, the two FFN contractions on int8 slices.
//
// Geometry.  EIGHT waves side by side along the 512 features, each owning 64 features x 64 tokens, 256 registers per wave
// (two waves per SIMD), one workgroup per CU: an int8 accumulator tile is two int32 registers per value, so the 128f x 64t
// wave tile of layer_tail_kernel (4 waves) would need the whole register file — at 64f x 64t every GEMM of the tail is ONE
// pass, no value is carried between feature passes, and the row-wide epilogues (LayerNorm, row maximum) see all their
// operands in the accumulators as before.  The operand ring is gemm.h's (two stages of two k-steps, 144 KiB).
//
// What is quantised where (one scale per token row, |q| <= 32639 as two int8 slices, common.h):
//   fc          split-bf16 (its input, the attention output, has one natural scale per row AND head)
//   LayerNorm-1 epilogue writes y1 twice: split-bf16 (the residual of LayerNorm-2) and int8 rows + scales (FFN-1's operand)
//   FFN-1       int8 MFMAs (K = 512: 16 k-steps of 32), dequantise, + bias, ReLU, row maximum over the 8 waves through LDS,
//               int8 rows + scales (FFN-2's operand) — the hidden activations never exist in floating point in memory
//   FFN-2       int8 MFMAs, dequantise, then gemm.h's EpiResLN (+ the next layer's int8 rows) unchanged
// Weights w_1, w_2: int8 slices with one scale per output feature, K in accumulator order (k_pack_rows_i8).
// The same kernel serves every batch size (bit-identical results whatever the sharding).
//
// STATUS: experiment of round 2, NOT part of the build.  Correct (all 102 GPU tests green with it in place of the shipped tails,
// forward error 1.2e-4), measured per launch: B=128 110 us (shipped tail_kernel<2>: 129), B=256 249 us (layer_tail_kernel: 222-227),
// B=32 77 us (tail_kernel<1>: 58), T=196 441 us (410).  One 8-wave workgroup per CU means two rounds at B=256 with every epilogue
// exposed; ablations at B=128: MFMAs + epilogues alone 67 us, + LDS reads and barriers 79, + the LDS-DMA stream 110.  To try it
// again: copy next to gemm.h, include it from egoego_hip.hip, pack w_1 / w_2 with k_pack_rows_i8 and launch it where
// layer_tail_kernel / tail_kernel are launched (DESIGN.md section 6b has the dispatch that was used).
#pragma once
#include "attn_layer_i8.h"

using T8 = GemmCfg<2, 2, 8, 1, 2, 2, false, 2, 2>;  // 512f x 64t, 8 waves of 64f x 64t, 2 k-steps per stage, 2 stages

// FFN-1 epilogue: int32 pairs -> fp32 (row scale x feature scale), + bias, ReLU, quantise per row.
struct EpiReluQ8 {
    const float* w_scale;  // [512] feature scales of w_1
    const float* a_scale;  // [Mp] row scales of the operand (LayerNorm-1's int8 rows)
    const float* bias;     // [512]
    int8_t* q8;            // [Mp][512] fragment-tiled int8 rows (accumulator order), slice 2 at + q8_plane
    size_t q8_plane;
    float* q8_scale;       // [Mp]
    __bf16* dbg;           // optional split-bf16 copy of the hidden activations (debug tap EGOEGO_DBG_FFN_HIDDEN)
    size_t dbg_plane;
    __device__ void run(I8Acc (&q)[2][2], int f0, int t0, int lane, int wf, char* smem) const {
        const int hf = lane >> 5, col = lane & 31;
        float* red = (float*)smem;  // [8 waves][64 tokens]
        f32x16 v[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = t0 + j * 32 + col;
            const float sa = a_scale[m];
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                i8_dequant(q[i][j], v[i][j], w_scale + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *(const float4*)(bias + f0 + i * 32 + 8 * gq + 4 * hf);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float h = fmaxf(v[i][j][4 * gq + c] + bb[c], 0.f);
                        v[i][j][4 * gq + c] = h;
                        amax = fmaxf(amax, h);
                    }
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            if (hf == 0) red[wf * 64 + j * 32 + col] = amax;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = t0 + j * 32 + col;
            float rmax = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) rmax = fmaxf(rmax, red[w * 64 + j * 32 + col]);
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) q8_scale[m] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                const size_t idx = acc_slot_i8(m, f0 + i * 32, hf, 16);
                *(u32x4*)(q8 + idx) = s1;
                *(u32x4*)(q8 + q8_plane + idx) = s2;
                if (dbg) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        u32x4 hi, lo;
                        split8(t + 8 * jj, hi, lo);
                        const size_t di = acc_slot(m, f0 + i * 32, jj, hf, 32);
                        *(u32x4*)(dbg + di) = hi;
                        *(u32x4*)(dbg + dbg_plane + di) = lo;
                    }
                }
            }
        }
    }
};

struct Tail8Args {
    GemmOperands g_fc;            // split-bf16: w_fc x attention output
    EpiResLN<2, 8, 64> e_fc;      // -> y1 split-bf16 + int8 rows / scales
    GemmOperands g_1;             // int8: w_1 x y1 rows (K16 counts 32-wide k blocks)
    EpiReluQ8 e_1;
    GemmOperands g_2;             // int8: w_2 x hidden rows
    const float* w2_scale;        // [512]
    const float* h_scale;         // [Mp] row scales of the hidden rows (= e_1.q8_scale)
    EpiResLN<2, 8, 64> e_2;
    int stop;                     // debug taps: 1 = after LayerNorm-1, 2 = after FFN-1
};

__global__ __launch_bounds__(T8::NT, T8::MINW) void tail8_kernel(Tail8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tblk = (int)blockIdx.x + a.g_fc.tblk0;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int f0 = wave * 64, t0 = tblk * 64;
    GemmBody<T8, EpiResLN<2, 8, 64>>::run(a.g_fc, a.e_fc, 0, tblk, smem);
    if (a.stop == 1) return;
    // the rows this workgroup just wrote are its next operand: they must have reached L2 before the LDS-DMA reads them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    I8Acc q[2][2];
    GemmBody<T8, NoEpi>::mainloop(a.g_1, 0, tblk, smem, q);
    a.e_1.run(q, f0, t0, lane, wave, smem);
    if (a.stop == 2) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GemmBody<T8, NoEpi>::mainloop(a.g_2, 0, tblk, smem, q);
    f32x16 v[2][2];
    const int hf = lane >> 5, col = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float sa = a.h_scale[t0 + j * 32 + col];
#pragma unroll
        for (int i = 0; i < 2; ++i) i8_dequant(q[i][j], v[i][j], a.w2_scale + f0 + i * 32 + 4 * hf, sa);
    }
    a.e_2.template run<2, 2>(v, f0, t0, lane, wave, 0, smem);
}
