// attn_layer_i8.h — one decoder layer's attention front end (TM:71-88) in ONE kernel, nothing but h in and O out.
//
// One workgroup per (window, head), one workgroup per CU:
//   1. K_h = h W_k^T + b      int8-slice GEMM (gemm.h "i8x3"), quantised per key row -> LDS (64 KiB)
//   2. Q_h = (h W_q^T + b)/sqrt(d_k)   quantised per query row into an LDS image over the idle operand ring, then read into
//                                      registers (64 VGPRs) by the wave that owns the query tile
//   3. S^T = K_h Q_h^T        int8 MFMAs, K fragments from LDS, Q from registers; softmax over keys in-lane;
//                             P = exp(s - max) in [0, 1] quantised with the fixed scale 1/32639 -> registers
//   4. V_h = h W_v^T + b      un-swapped accumulator (lane owns a feature), quantised per feature column over
//                             the window's keys -> LDS, transposed, over the K image (dead by now)
//   5. O^T = V_h^T P          int8 MFMAs; 1/rowsum, split-bf16 store into the fc GEMM's operand
// Q, K, V and the probabilities never leave the CU: per launch the kernel reads h (int8 slices) and the
// weights and writes O — the split-bf16 fused kernel moves ~1.1 GB of Q/K/V through L2/HBM instead.
//
// Fragment orders.  An int8 MFMA operand fragment is [half][32 rows][16 bytes]; a lane of an accumulator
// holds, for a 32-wide tile, elements 8g + 4hf + c in register 4g + c — exactly the 16 bytes of position
// 16hf + 4g + c ("acc32" order, common.h).  So every operand produced by an accumulator (K and Q along d_k,
// P and V^T along the keys) is stored by one 16-byte write per lane per tile, and because both operands of
// a product are permuted identically the contraction is unchanged.
#pragma once
#include "gemm.h"

struct AttnLayerArgs {
    const int8_t* w8;  // [3*HD][512] two slices, rows in natural order, K in acc32 order
    size_t w_plane;    // bytes between slices
    const float* w_scale;  // [3*HD]
    const float* bias;     // [3*HD]
    const int8_t* h8;      // [Mp][512] two slices
    size_t h_plane;
    const float* h_scale;  // [Mp]
    __bf16* o;             // [Mp][HD] split-bf16, fragment-tiled, accumulator order
    size_t o_plane;
    int HD16;
    float qscale;
    int H, L, bh0;
    EG_DBG(unsigned long long* trace;)  // perf-debug build: [grid][16] phase timestamps or nullptr
    // o8 != nullptr: O goes out as int8 slices instead ([Mp][HD] fragment-tiled, K in acc32 order, slices o8_plane bytes
    // apart) with one scale per row AND head, o_scale[row * H + head] — the operand of the int8 fc contraction
    int8_t* o8;
    size_t o8_plane;
    float* o_scale;
};

using AL8K = GemmCfg<4, 2, 2, 2, 1, 2, false, 1, 3>;
using AL8V = GemmCfg<4, 2, 2, 2, 1, 2, true, 1, 3>;
using AL8Q = GemmCfg<8, 1, 1, 4, 1, 2, false, 1, 3>;
static constexpr int AL_KV_BYTES = 65536, AL_SLICE = 32768, AL_MISC_BYTES = 12288;
static constexpr int AL_SMEM_BYTES = AL_KV_BYTES + AL_MISC_BYTES + AL8K::SMEM_BYTES;
struct NoEpi {};

EG_D i32x4 lds_frag(const char* p) { return *(const i32x4*)p; }

// ---- the probabilities as THREE int8 slices (round 4) -----------------------------------------------------------------------
// p = exp2(s - rowmax) lies in (0, 1] with the row's largest entry exactly 1, so one fixed scale serves every row — but two slices
// (steps of 1 / 32639) leave every small probability an ABSOLUTE error of 1.5e-5, and a trained, peaked attention sums ~120 of them
// against a small row sum: int8_site_study.py (round-4/5 experiment, removed; results: HISTORY.md) puts the P image at 4.5e-4 of a trained-like chain's error where
// everything else together (prepared packing) makes 2.6e-4, and Q, K, V images add nothing.  So P gets a third slice:
//     q = rint(p * 8355711) = 65536 a1 + 256 a2 + a3   (a1 in 0..127, a2, a3 signed bytes; 8355711 = 127 * 65536 + 127 * 256 + 127:
//     the top slice keeps the 7 bits the two-slice form had, so the dropped a2 v2 term is no larger than its b2 v2 was — with a1
//     only 6 bits wide, the first version, the DIFFUSE attention of the initialisation got WORSE: 4.0e-4 against 2.9e-4 at the
//     attention output),
// and PV a third int32 accumulator for the new level:  H += v1 a1,  M += v2 a1 + v1 a2,  L += v1 a3   (v = 256 v1 + v2; the terms
// a2 v2 and a3 v2 are dropped like s2 s2 everywhere else) — four MFMAs per (d_v tile, key block) instead of three, in the PV phase
// only (1/16 of the kernel's MFMAs).  Exact integers; the sum 2^8 (2^8 (2^8 H + M) + L) is formed as
// fma(float(256 H + M), 256, float(L)), one function for every kernel form (same bits in all of them).
static constexpr float P_QMAX = 8355711.0f;
struct PVAcc {
    i32x16 h, m, l;
};
EG_D void acc_zero(PVAcc& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c.h[r] = c.m[r] = c.l[r] = 0;
}
// 16 values v with 0 <= v * k <= P_QMAX -> three slices of 16 bytes.  q = rint(v * k) < 2^23 sits in the mantissa of v * k + 2^23 (one
// fma; the values are not negative, so the adder trick of common.h quant16 reaches one bit further than there); byte 0 of q is a3,
// byte 1 of q + 128 is a2, byte 2 of q + 128 + 32768 is a1 (the float's exponent byte is never selected, and no carry reaches it:
// q + 32896 <= 0x7fffff).
EG_D void quant_p(const float v[16], float k, u32x4& s1, u32x4& s2, u32x4& s3) {
    uint32_t q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[i], k, 8388608.0f));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t a = q[4 * w], b = q[4 * w + 1], c = q[4 * w + 2], d = q[4 * w + 3];
        s3[w] = __builtin_amdgcn_perm(b, a, 0x0c0c0400u) | __builtin_amdgcn_perm(d, c, 0x04000c0cu);
        const uint32_t a1 = a + 128u, b1 = b + 128u, c1 = c + 128u, d1 = d + 128u;
        s2[w] = __builtin_amdgcn_perm(b1, a1, 0x0c0c0501u) | __builtin_amdgcn_perm(d1, c1, 0x05010c0cu);
        const uint32_t a2 = a1 + 32768u, b2 = b1 + 32768u, c2 = c1 + 32768u, d2 = d1 + 32768u;
        s1[w] = __builtin_amdgcn_perm(b2, a2, 0x0c0c0602u) | __builtin_amdgcn_perm(d2, c2, 0x06020c0cu);
    }
}
// the value of one accumulator element in units of s_v / P_QMAX / 256 ... i.e. O = pv_value(..) * (s_v * (1 / rowsum) * 256 / P_QMAX)
EG_D float pv_value(int h, int m, int l) { return __builtin_fmaf((float)i8_combine(h, m), 256.0f, (float)l); }

// (The kernel itself is attn_layer_i8w.h: eight waves, two per SIMD.  Round 1-2's four-wave form — one 512-register wave per SIMD, Q
// projected with a lane owning all 256 d_k of its query — ran 206-210 us per launch at B=256 where the eight-wave form runs 191.)

// ---- Q/K/V projections on int8 slices for windows the one-kernel form does not cover (T + 1 <= 64 or > 128) ----------
// The same int8 main loop and dequantisation, one 256-feature x 128-token block per workgroup, handing fp32 tiles to the
// split-bf16 epilogues of gemm.h (EpiQK / EpiV): Q, K, V go to memory in attention.h's operand layouts and attn_kernel
// runs the attention core.  Feature blocks 0 .. n_qk-1 are Q/K (swapped accumulator), the rest V (un-swapped).
struct QkvI8Args {
    const int8_t* w8;      // [3*HD][512] two slices
    size_t w_plane;        // bytes between slices
    const float* w_scale;  // [3*HD]
    const int8_t* h8;      // [Mp][512] two slices
    size_t h_plane;
    const float* h_scale;  // [Mp]
    int ntb, n_qk;         // token blocks in the grid; number of Q/K feature blocks
};

template <class EQK, class EV>
__global__ __launch_bounds__(256, 1) void qkv_i8_kernel(QkvI8Args a, EQK eqk, EV ev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int fblk, tblk;
    grouped_map(lid, (int)gridDim.x / a.ntb, a.ntb, fblk, tblk);
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int wf = wave & 1, wt = wave >> 1;
    const int f0 = fblk * 256 + wf * 128, t0 = tblk * 128 + wt * 64;
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    I8Acc q[4][2];
    f32x16 v[4][2];
    if (fblk < a.n_qk) {
        GemmBody<AL8K, NoEpi>::mainloop(g, fblk, tblk, smem, q);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sa = a.h_scale[t0 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) i8_dequant(q[i][j], v[i][j], a.w_scale + f0 + i * 32 + 4 * hf, sa);
        }
        eqk.template run<4, 2>(v, f0, t0, lane, wf, wt, smem);
    } else {
        GemmBody<AL8V, NoEpi>::mainloop(g, fblk, tblk, smem, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sw = a.w_scale[f0 + i * 32 + col];
#pragma unroll
            for (int j = 0; j < 2; ++j) i8_dequant_rows(q[i][j], v[i][j], sw, a.h_scale + t0 + j * 32 + 4 * hf);
        }
        ev.template run<4, 2>(v, f0, t0, lane, wf, wt, smem);
    }
}
