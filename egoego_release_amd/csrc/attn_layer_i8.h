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
