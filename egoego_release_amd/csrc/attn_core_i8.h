// attn_core_i8.h — the attention front end on int8 slices for windows the one-kernel form (attn_layer_i8.h, <= 128 tokens)
// cannot hold on chip: up to 224 tokens (BASELINE configs[3]: T = 196).  Two kernels per layer:
//
//   qkv_i8q_kernel       Q/K/V projections (TM:71-73) on int8 slices, one 256-feature (= one head of Q, K or V) x 64-token
//                        block per workgroup, two workgroups per CU, results QUANTISED in the epilogue and written as the int8 operand images
//                        of the core kernel: Q and K per row (query / key) with one scale each, V transposed with one
//                        scale per key.  2 bytes per value go to memory instead of the 4 of the split-bf16 form — at
//                        T = 196 the Q/K/V round trip (0.7 GB written, 1.2 GB read per layer at B = 256) is what bounds
//                        the split-bf16 pair.
//   attn_core_i8_kernel  one 4-wave workgroup per (window, head, block of 4 query tiles): the head's K image
//                        (<= 112 KiB) comes through LDS in two halves by LDS-DMA, S^T = K Q^T on int8 MFMAs with the queries' Q
//                        fragments in registers, softmax in-lane (TM:76-82), P * (V's key scales) quantised per query
//                        into registers, the V^T image over the K image, O^T = V^T P, 1/rowsum, split-bf16 store.
//
// Scales.  K, Q: one per row (maximum over the 256 d_k of the head).  V: one per KEY row (maximum over the 256 d_v);
// it multiplies the probabilities before they are quantised, so that the PV contraction over all keys of the window is
// ONE integer accumulation: O[q,d] = (1/sum_q) sum_k (p[q,k] s_v[k]) vq[k,d].  (The one-kernel form scales V per feature
// column over the window's keys; here a 128-token block may straddle two windows.)
// Fragment orders are those of attn_layer_i8.h: every operand produced by an accumulator is 16 consecutive bytes per
// lane and tile ("acc32" order), both operands of a product permuted alike.
#pragma once
#include "attn_layer_i8.h"

struct Qkv8Out {
    int8_t *q8, *k8, *v8;  // [B*H][KT][8][1 KiB] (Q, K: token tile, d_k block) / [B*H][8][KT][1 KiB] (V^T: d_v tile, key block); slice 2 at +plane
    size_t plane;          // bytes between the two slices of each tensor
    float *sq, *sk, *sv;   // [B*H][Lp] row scales
    const float* bias;     // [3*HD]
    float qscale;          // 1 / sqrt(d_k)
    int Lp, KT, H, HD, Mvalid;
    int Lr;                // token rows per window in the ROW space (a multiple of 16, <= Lp = 32 KT): window b owns rows b Lr .. b Lr + Lr - 1
    EG_DBG(unsigned long long* trace;)  // perf-debug build: [grid][8] phase timestamps or nullptr (tools/long_window_trace.py)
};

// Quantise 16 values, each with its own inverse scale, into the two slices (cf. common.h quant16).
EG_D void quant16v(const float v[16], const float inv[16], u32x4& s1, u32x4& s2) {
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[i], inv[i], 12582912.0f));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t a = u[4 * w], b = u[4 * w + 1], c = u[4 * w + 2], d = u[4 * w + 3];
        s2[w] = __builtin_amdgcn_perm(b, a, 0x0c0c0400u) | __builtin_amdgcn_perm(d, c, 0x04000c0cu);
        const uint32_t a1 = a + 128u, b1 = b + 128u, c1 = c + 128u, d1 = d + 128u;
        s1[w] = __builtin_amdgcn_perm(b1, a1, 0x0c0c0501u) | __builtin_amdgcn_perm(d1, c1, 0x05010c0cu);
    }
}

// 256 features (one head of Q, K or V) x 64 NWT tokens per workgroup, waves of 64 features x 64 tokens, 4 (features) x NWT (tokens):
//   NWT = 1  four waves, two workgroups per CU (one's prologue / epilogue meets the other's main loop) — row counts that are not a
//            multiple of 128;
//   NWT = 2  eight waves, one workgroup per CU (the projection tiling of attn_layer_i8w.h): the 256 KB of a feature block's weights
//            cross L2 -> LDS once per 128 tokens instead of once per 64.  Round 4: at B = 256 x T = 196 the NWT = 1 grid moves 3.2 GB
//            per launch through that path (9.7 TB/s at 330 us) — it, not the matrix pipe (40 % busy), is what the kernel waits for.
template <int NWT> using Q8K = GemmCfg<2, 2, 4, NWT, 1, 2, false, 2, 3>;
template <int NWT> using Q8V = GemmCfg<2, 2, 4, NWT, 1, 2, true, 2, 3>;

template <int NWT>
__global__ __launch_bounds__(256 * NWT, NWT == 1 ? 2 : 1) void qkv_i8q_kernel(QkvI8Args a, Qkv8Out o) {
    constexpr int BT = 64 * NWT;  // tokens per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int fblk, tblk;
    grouped_map(lid, (int)gridDim.x / a.ntb, a.ntb, fblk, tblk);
    const int wave = wave_id_uniform();
    const int wf = wave & 3, wt = wave >> 2;  // feature quarter of the head, token half of the block
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int f0 = fblk * 256 + wf * 64, t0 = tblk * BT + wt * 64;
    const int which = fblk * 256 / o.HD, h = (fblk * 256 % o.HD) >> 8;  // 0 = Q, 1 = K, 2 = V; head
    EG_DBG(unsigned long long* tr = o.trace ? o.trace + 131072 + (size_t)blockIdx.x * 8 : nullptr;)
    auto mark = [&](int i) {
        EG_DBG(if (tr && threadIdx.x == 0) tr[i] = wall_clock64();)
        (void)i;
    };
    mark(0);
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    float* red = (float*)smem;  // [4][BT] cross-wave maxima; the main loop's ring is dead when it is used
    // the block's parameters (weight row scales and biases of its 256 features, row scales of its tokens) staged in LDS behind the
    // ring: the epilogue is a chain of dependent loads, and an LDS read costs a tenth of an L2 round trip
    float* const p_ws = (float*)(smem + Q8K<NWT>::SMEM_BYTES);  // [256]
    float* const p_b = p_ws + 256;                              // [256]
    float* const p_hs = p_b + 256;                              // [BT]
    if (threadIdx.x < 256) {
        p_ws[threadIdx.x] = a.w_scale[fblk * 256 + threadIdx.x];
        p_b[threadIdx.x] = o.bias[fblk * 256 + threadIdx.x];
    }
    if (threadIdx.x < BT) p_hs[threadIdx.x] = a.h_scale[tblk * BT + threadIdx.x];
    // (visible after the first barrier of the main loop)
    I8Acc q[2][2];
    f32x16 v[2][2];
    if (which < 2) {
        // ---- Q_h / K_h: lane owns a token; one scale per row; int8 image [token tile][d_k block]
        GemmBody<Q8K<NWT>, NoEpi>::mainloop(g, fblk, tblk, smem, q);
        mark(1);
        const float sc = which == 0 ? o.qscale : 1.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sa = p_hs[wt * 64 + j * 32 + col];
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                i8_dequant(q[i][j], v[i][j], p_ws + wf * 64 + i * 32 + 4 * hf, sa);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *(const float4*)(p_b + wf * 64 + i * 32 + 8 * gq + 4 * hf);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[i][j][4 * gq + c] = (v[i][j][4 * gq + c] + bb[c]) * sc;
                        amax = fmaxf(amax, fabsf(v[i][j][4 * gq + c]));
                    }
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            if (hf == 0) red[wf * BT + wt * 64 + j * 32 + col] = amax;
        }
        __syncthreads();
        mark(2);
        int8_t* dst8 = which == 0 ? o.q8 : o.k8;
        float* dsts = which == 0 ? o.sq : o.sk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // the lane's token: row m of the row space = token l of window b (windows start on 16-row boundaries, so a
            // 32-row tile may hold the end of one window and the start of the next: the destination is per lane)
            const int m = t0 + j * 32 + col;
            if (m >= o.Mvalid) continue;
            const int b = m / o.Lr, l = m - b * o.Lr, bh = b * o.H + h;
            const int tokb = wt * 64 + j * 32 + col;
            const float rmax = fmaxf(fmaxf(red[tokb], red[BT + tokb]), fmaxf(red[2 * BT + tokb], red[3 * BT + tokb]));
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) dsts[(size_t)bh * o.Lp + l] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                int8_t* p = dst8 + ((((size_t)bh * o.KT + (l >> 5)) * 8 + wf * 2 + i) << 10) + (hf * 32 + (l & 31)) * 16;
                *(u32x4*)p = s1;
                *(u32x4*)(p + o.plane) = s2;
            }
        }
    } else {
        // ---- V_h: un-swapped accumulator (lane owns a feature, registers walk the tokens); one scale per KEY row
        // (maximum over the head's 256 features: two tiles in-lane, 32 lanes by shuffles, four waves through LDS);
        // stored transposed [d_v tile][key block]
        GemmBody<Q8V<NWT>, NoEpi>::mainloop(g, fblk, tblk, smem, q);
        mark(1);
        float tmax[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax[j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sw = p_ws[wf * 64 + i * 32 + col], bf = p_b[wf * 64 + i * 32 + col];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                i8_dequant_rows(q[i][j], v[i][j], sw, p_hs + wt * 64 + j * 32 + 4 * hf);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[i][j][r] += bf;
                    tmax[j][r] = fmaxf(tmax[j][r], fabsf(v[i][j][r]));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m = tmax[j][r];
#pragma unroll
                for (int s = 1; s < 32; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
                // token of register r: 8 (r >> 2) + 4 hf + (r & 3) within tile j
                if (col == 0) red[wf * BT + wt * 64 + j * 32 + mfma32_row(r, hf)] = m;
            }
        __syncthreads();
        mark(2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float inv[16], rm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tokb = wt * 64 + j * 32 + mfma32_row(r, hf);
                rm[r] = fmaxf(fmaxf(red[tokb], red[BT + tokb]), fmaxf(red[2 * BT + tokb], red[3 * BT + tokb]));
                inv[r] = rm[r] > 0.f ? I8_QMAX / rm[r] : 0.f;
            }
            float t[2][16];
            u32x4 s1[2], s2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) t[i][r] = v[i][j][r];
                quant16v(t[i], inv, s1[i], s2[i]);
            }
            // A lane's 16 bytes of a tile are two runs of 8 keys: registers 0..7 = the tile's rows 0..15, 8..15 = rows 16..31.
            // Windows start on 16-row boundaries, so each run lies in ONE window: it goes, as 8 bytes, to the half of that
            // window's key tile it belongs to (acc32 order: keys 0..15 of a 32-key tile are bytes 0..7, keys 16..31 bytes 8..15).
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int m0 = t0 + j * 32 + 16 * g2;
                if (m0 >= o.Mvalid) continue;
                const int b = m0 / o.Lr, l0 = m0 - b * o.Lr, bh = b * o.H + h;
                if (wf == 0 && col == 0) {
#pragma unroll
                    for (int r = 8 * g2; r < 8 * g2 + 8; ++r)
                        o.sv[(size_t)bh * o.Lp + l0 + (mfma32_row(r, hf) & 15)] = rm[r] > 0.f ? rm[r] / I8_QMAX : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int8_t* p = o.v8 + ((((size_t)bh * 8 + wf * 2 + i) * o.KT + (l0 >> 5)) << 10) + lane * 16 + ((l0 >> 4) & 1) * 8;
                    *(uint2*)p = make_uint2(s1[i][2 * g2], s1[i][2 * g2 + 1]);
                    *(uint2*)(p + o.plane) = make_uint2(s2[i][2 * g2], s2[i][2 * g2 + 1]);
                }
            }
        }
    }
    mark(3);
}

struct AttnCore8Args {
    const int8_t *q8, *k8, *v8;
    size_t plane;
    const float *sq, *sk, *sv;
    __bf16* o;  // [Mp][HD] split-bf16 fragment-tiled (accumulator order): the fc GEMM's operand
    size_t o_plane;
    int HD16, H, L, Lp;
    int Lr;     // token rows per window in the row space (<= Lp); query l of window b is row b Lr + l
    // kernel template O8: O as int8 slices with one scale per row and head (see AttnLayerArgs), else split-bf16 into `o`
    int8_t* o8;
    size_t o8_plane;
    float* o_scale;
    int BH;     // (window, head) pairs of the launch; the grid is min(BH * ceil(KT / 4), compute units) persistent workgroups
    EG_DBG(unsigned long long* trace;)  // perf-debug build: [item][8] phase timestamps or nullptr
};

// The K image (d_k halves) and the V^T image (d_v halves) pass through two LDS buffers of KT * 8 KiB x 2 slices each.  S accumulates
// over the two d_k halves in the same int32 accumulators (the row scales of K and Q cover the whole d_k).
//
// PERSISTENT workgroups (round 4).  An item = one (window, head, block of 4 query tiles); a workgroup walks items lid, lid + grid, ...
// and requests the NEXT item's operands while the current one computes: its Q fragments (64 registers, dead after S^T) and scales
// before the softmax, its first K half into buffer 0 the moment the PV phase is done with the first V^T half, its second K half
// into buffer 1 after the PV phase.  The round-4 trace of the one-item-per-workgroup form (one wave per SIMD: nothing else on
// the CU to cover a wait) showed 5.7 of a workgroup's 26 us waiting for the first 120 KB (K half + Q: the whole chip starts a
// round at once and gets ~11 B/clk per CU) and the second K half arriving behind the first half's MFMAs.  Every wait is a full
// vmcnt(0): what is waited for was requested a phase earlier.  Items are computed exactly as before: same bits.
template <int KT, bool O8>
__global__ __launch_bounds__(256, 1) void attn_core_i8_kernel(AttnCore8Args a) {
    constexpr int HALF = KT * 4 * 1024;  // bytes of one slice of half an image (4 of the 8 d blocks x KT tiles)
    constexpr int BUF = 2 * HALF;        // one buffer: both slices of a half image
    constexpr int NQB = (KT + 3) / 4;    // query blocks (4 tiles, one per wave) per (window, head)
    constexpr int NPIECE = KT * 8 / 4;   // 1-KiB pieces of a half image (both slices) per wave: KT*4 blocks x 2 slices / 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                           // [2 buffers][slice][half image]
    // The key scales live in their OWN LDS object: hipcc puts a full vmcnt(0) in front of any LDS read it cannot prove disjoint from a
    // pending LDS-DMA destination, and a read at smem + constant + 16 hf is such a read — the softmax then waited for the V^T half
    // requested just before it (round 4: 6.1 of an item's 23 us).  The fragment reads are at compile-time offsets and are not affected.
    __shared__ float sk[KT * 32];              // key scales of K
    __shared__ float sv[KT * 32];              // key scales of V
    const int wave = wave_id_uniform();
    // lane, hf, col are re-derived at the top of every item from a value the compiler cannot see through: left loop-invariant, ~90
    // registers of per-lane address arithmetic were hoisted in front of the item loop and spilled to scratch — and a scratch reload
    // behind in-flight LDS-DMA waits for it (vmcnt is in order)
    int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int n_items = a.BH * NQB;
    // consecutive items (the query blocks of one (window, head), then the next head) run at the same time on one XCD: its L2 serves the
    // second reader of a K / V^T image
    int item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (item >= n_items) return;

    // ---- what an item needs before its first MFMA
    struct Front {
        int bh, qt, tile_active;
        float sq, skr, svr;
        int touch, touch2;
    };
    auto locate = [&](int it, Front& f) {
        f.bh = it / NQB;
        const int qt_raw = (it - f.bh * NQB) * 4 + wave;
        f.tile_active = qt_raw < KT;
        f.qt = f.tile_active ? qt_raw : KT - 1;
    };
    // (resources based at THIS (window, head)'s image, so that the 32-bit offsets — slice stride + a few hundred KiB — stay far below
    // the 2^31 - 1 bytes a buffer resource can span whatever the batch)
    auto k_rsrc = [&](int bh) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.k8 + (((size_t)bh * KT * 8) << 10)), 0, 0x7fffffff, 0x00020000); };
    auto v_rsrc = [&](int bh) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.v8 + (((size_t)bh * 8 * KT) << 10)), 0, 0x7fffffff, 0x00020000); };
    // K image [kt][8 d blocks]: half `hh` = d blocks 4hh .. 4hh+3 of every key tile -> buffer layout [kt][4]
    auto dma_k_piece = [&](__amdgpu_buffer_rsrc_t kr, int hh, int buf, int n) {
        const int pc = n * 4 + wave;
        const int sl = pc / (KT * 4), blk = pc - sl * KT * 4, kt = blk >> 2, i = blk & 3;
        const unsigned src = (unsigned)(sl * a.plane) + (unsigned)((kt * 8 + 4 * hh + i) << 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, (__attribute__((address_space(3))) void*)(kv + buf * BUF + sl * HALF + (blk << 10)), 16, lane * 16,
                                                 src, 0, 0);
    };
    // V^T image [8 d_v tiles][kt]: half `hh` = d_v tiles 4hh .. 4hh+3 -> contiguous
    auto dma_v_piece = [&](__amdgpu_buffer_rsrc_t vr, int hh, int buf, int n) {
        const int pc = n * 4 + wave;
        const int sl = pc / (KT * 4), blk = pc - sl * KT * 4;
        const unsigned src = (unsigned)(sl * a.plane) + (unsigned)((4 * hh * KT + blk) << 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (__attribute__((address_space(3))) void*)(kv + buf * BUF + sl * HALF + (blk << 10)), 16, lane * 16,
                                                 src, 0, 0);
    };
    i32x4 qs1[8], qs2[8];
    auto load_q = [&](const Front& f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int8_t* p = a.q8 + ((((size_t)f.bh * KT + f.qt) * 8 + i) << 10) + lane * 16;
            qs1[i] = *(const i32x4*)p;
            qs2[i] = *(const i32x4*)(p + a.plane);
        }
    };
    // the item's scales into three registers, and one dword of every 128-byte line of its Q fragments (8 KiB per slice and wave) so
    // that the fragment loads at the item's start find them in L2 (64 registers of fragments across the PV phase do not fit)
    auto load_scales = [&](Front& f) {
        f.sq = a.sq[(size_t)f.bh * a.Lp + f.qt * 32 + col];
        const int ks = min((int)threadIdx.x, KT * 32 - 1);
        f.skr = a.sk[(size_t)f.bh * a.Lp + ks];
        f.svr = a.sv[(size_t)f.bh * a.Lp + ks];
        const int8_t* q = a.q8 + ((((size_t)f.bh * KT + f.qt) * 8) << 10) + lane * 128;
        f.touch = *(const int*)q;  // (no arithmetic on them before the next item's top: a use is a wait behind the V^T DMA)
        f.touch2 = *(const int*)(q + a.plane);
    };

    Front cur;
    locate(item, cur);
    {
        const __amdgpu_buffer_rsrc_t kr = k_rsrc(cur.bh);
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) dma_k_piece(kr, 0, 0, n);
        load_scales(cur);
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) dma_k_piece(kr, 1, 1, n);
    }

    for (;;) {
        asm volatile("" : "+v"(lane));
        hf = lane >> 5;
        col = lane & 31;
        const int next = item + (int)gridDim.x;
        const bool has_next = next < n_items;  // workgroup-uniform
        const int bh = cur.bh, qt = cur.qt;
        const int b = bh / a.H, h = bh - b * a.H;
        const bool active = cur.tile_active && qt * 32 + col < a.Lr;  // per lane: the last query tile may reach beyond the window's rows
        const float sq = cur.sq;
        const __amdgpu_buffer_rsrc_t vr = v_rsrc(bh);
        EG_DBG(unsigned long long* tr = a.trace ? a.trace + 131072 + (size_t)item * 8 : nullptr;)
        auto mark = [&](int i) {
            EG_DBG(if (tr && threadIdx.x == 0) {
                tr[i] = wall_clock64();
                if (i == 1 || i == 2) tr[5 + i] = __builtin_readcyclecounter();  // shader cycles over the S^T phase
            })
            (void)i;
        };
        mark(0);
        // both K halves and the scales of this item have been on their way since the previous item's PV phase (or the prologue above),
        // the Q fragments' lines are in L2; the previous item's stores are in the same counter
        load_q(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (cur.touch == 0x7fffffff && cur.touch2 == 0x7ffffffe) __builtin_amdgcn_s_sleep(1);  // (keeps the touching loads alive)
        if (threadIdx.x < KT * 32) {
            sk[threadIdx.x] = cur.skr;
            sv[threadIdx.x] = cur.svr;
        }
        __syncthreads();
        mark(1);

        // ---- S^T = K Q^T over the two d_k halves, softmax over keys (TM:76-82), P * s_v quantised per query
        i32x4 ps1[KT], ps2[KT], ps3[KT];  // three slices (attn_layer_i8.h quant_p): small probabilities keep their relative precision
        float oscale;
        Front nxt = cur;
        {
            I8Acc s[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc_zero(s[kt]);
            // One unit = one d_k block x a PAIR of key tiles: 4 fragment reads, 6 MFMAs (the second MFMA into an `m` accumulator four
            // MFMAs behind the first).  The fragments of unit u + 2 are requested before unit u's MFMAs (a ring of three): with one wave
            // per SIMD nothing else covers the LDS latency, and hipcc left to itself sinks every read to its use.  Integer
            // accumulation: any order gives the same bits.
            constexpr int NU = (KT + 1) / 2, NS = 4 * NU;
            i32x4 f[3][4];
            auto load_unit = [&](const char* img, int u, i32x4(&d)[4]) {
                const int i = u / NU, kt0 = 2 * (u % NU);
                const char* src = img + ((kt0 * 4 + i) << 10) + lane * 16;
                d[0] = lds_frag(src);
                d[1] = lds_frag(src + HALF);
                if (kt0 + 1 < KT) {
                    d[2] = lds_frag(src + 4096);
                    d[3] = lds_frag(src + 4096 + HALF);
                }
            };
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const char* img = kv + hh * BUF;
                load_unit(img, 0, f[0]);
                if (NS > 1) load_unit(img, 1, f[1]);
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    if (u + 2 < NS) load_unit(img, u + 2, f[(u + 2) % 3]);
                    // second d_k half: the first V^T half streams into the buffer the barrier below freed, one piece per unit
                    if (hh == 1 && u < NPIECE) dma_v_piece(vr, 0, 0, u);
                    const int i = u / NU, kt0 = 2 * (u % NU);
                    const bool two = kt0 + 1 < KT;
                    const i32x4 q1 = qs1[4 * hh + i], q2 = qs2[4 * hh + i];
                    i32x4(&c)[4] = f[u % 3];
                    s[kt0].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[1], q1, s[kt0].m, 0, 0, 0);
                    if (two) s[kt0 + 1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[3], q1, s[kt0 + 1].m, 0, 0, 0);
                    s[kt0].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], q1, s[kt0].h, 0, 0, 0);
                    if (two) s[kt0 + 1].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], q1, s[kt0 + 1].h, 0, 0, 0);
                    s[kt0].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], q2, s[kt0].m, 0, 0, 0);
                    if (two) s[kt0 + 1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], q2, s[kt0 + 1].m, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // every wave has finished with this buffer (its fragment reads were waited for by the MFMAs that used them): buffer 0
                // takes the first V^T half (during the second d_k half), buffer 1 the second.  A RAW s_barrier: __syncthreads() carries a
                // fence that drains vmcnt, i.e. waits for the V^T pieces already under way (gemm.h mainloop has the same note)
                __builtin_amdgcn_s_barrier();
                if (hh == 1) {
#pragma unroll
                    for (int n = 0; n < NPIECE; ++n) dma_v_piece(vr, 1, 1, n);
                }
            }
            mark(2);
            // the next item's scales and the lines of its Q fragments start their trip now
            if (has_next) {
                locate(next, nxt);
                load_scales(nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            float p[KT][16];
            float mx = -INFINITY;
            const float sq256 = sq * 256.0f * 1.44269504088896f;  // logits in units of log2(e): softmax through v_exp_f32
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 k4 = *(const float4*)(sk + kt * 32 + 8 * gq + 4 * hf);
                    const float ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * gq + c;
                        float val = (float)i8_combine(s[kt].h[r], s[kt].m[r]) * (sq256 * ks[c]);
                        if (kt * 32 + 8 * gq + 4 * hf + c >= a.L) val = -INFINITY;
                        p[kt][r] = val;
                        mx = fmaxf(mx, val);
                    }
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // row sum in the order of the eight-wave form (attn_core_i8w.h): tiles 0..3, then the rest, each half summed across the two
            // half-waves first — so that the two forms give the same bits
            float sum2[2] = {0.f, 0.f}, pmax = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v4 = *(const float4*)(sv + kt * 32 + 8 * gq + 4 * hf);
                    const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * gq + c;
                        const float e = __builtin_amdgcn_exp2f(p[kt][r] - mx);
                        sum2[kt >= 4] += e;
                        p[kt][r] = e * vs[c];  // the key's V scale rides on the probability
                        pmax = fmaxf(pmax, p[kt][r]);
                    }
                }
            sum2[0] += __shfl_xor(sum2[0], 32);
            sum2[1] += __shfl_xor(sum2[1], 32);
            const float sum = sum2[0] + sum2[1];
            pmax = fmaxf(pmax, __shfl_xor(pmax, 32));
            // q = rint(p * pk) <= P_QMAX for every p <= pmax: the quarter unit of slack covers the rounding of pk and of the product
            const float pk = pmax > 1e-30f ? (P_QMAX - 0.25f) / pmax : 0.f;  // (below: every V row of the window is zero to fp32 — O is 0)
            oscale = (1.0f / sum) * (pmax / (P_QMAX - 0.25f)) * 256.0f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                u32x4 s1, s2, s3;
                quant_p(p[kt], pk, s1, s2, s3);
                ps1[kt] = __builtin_bit_cast(i32x4, s1);
                ps2[kt] = __builtin_bit_cast(i32x4, s2);
                ps3[kt] = __builtin_bit_cast(i32x4, s3);
            }
        }

        mark(3);
        // ---- O^T = V^T P (TM:83-88), heads merged on store
        const int m = b * a.Lr + qt * 32 + col;
        float t8[8][16];  // int8 output: the values wait for the row maximum over the head's 256 features
        float amax = 0.f;
        // both V^T halves were requested before the softmax: one wait, one barrier, then 4 groups (d_v half x tile pair) x KT key blocks
        // of 4 fragment reads + 8 MFMAs, the reads two units ahead like above
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int mo = m;  // (the split-bf16 store addresses are formed from here on, not at the item's start: see `lane` above)
        asm volatile("" : "+v"(mo));
        {
            constexpr int NPV = 4 * KT;
            const __amdgpu_buffer_rsrc_t nkr = k_rsrc(nxt.bh);
            i32x4 f[3][4];
            auto load_unit = [&](int u, i32x4(&d)[4]) {
                const int g = u / KT, kb = u - g * KT, dvh = g >> 1, dp = g & 1;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* src = kv + dvh * BUF + (((2 * dp + dt) * KT + kb) << 10) + lane * 16;
                    d[2 * dt] = lds_frag(src);
                    d[2 * dt + 1] = lds_frag(src + HALF);
                }
            };
            load_unit(0, f[0]);
            load_unit(1, f[1]);
            PVAcc o[2];
#pragma unroll
            for (int u = 0; u < NPV; ++u) {
                const int g = u / KT, kb = u - g * KT;
                if (kb == 0) {
                    acc_zero(o[0]);
                    acc_zero(o[1]);
                }
                if (u == 2 * KT) __builtin_amdgcn_s_barrier();  // the first V^T half is spent in every wave: buffer 0 takes the next item's first K half
                if (u + 2 < NPV) load_unit(u + 2, f[(u + 2) % 3]);
                if (u >= 2 * KT && u - 2 * KT < NPIECE && has_next) dma_k_piece(nkr, 0, 0, u - 2 * KT);
                i32x4(&c)[4] = f[u % 3];
                o[0].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[1], ps1[kb], o[0].m, 0, 0, 0);
                o[1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[3], ps1[kb], o[1].m, 0, 0, 0);
                o[0].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], ps1[kb], o[0].h, 0, 0, 0);
                o[1].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], ps1[kb], o[1].h, 0, 0, 0);
                o[0].l = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], ps3[kb], o[0].l, 0, 0, 0);
                o[1].l = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], ps3[kb], o[1].l, 0, 0, 0);
                o[0].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], ps2[kb], o[0].m, 0, 0, 0);
                o[1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], ps2[kb], o[1].m, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb == KT - 1) {
                    if (O8) {
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = pv_value(o[dt].h[r], o[dt].m[r], o[dt].l[r]) * oscale;
                                t8[2 * g + dt][r] = v;
                                amax = fmaxf(amax, fabsf(v));
                            }
                        // (pins the conversion HERE: its only consumer is the `active` branch after the loop, and hipcc would sink it
                        // there, keeping every group's raw accumulators — 384 registers — alive until then)
                        asm volatile("" : "+v"(amax));
                    } else if (active) {
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const int tile = 2 * g + dt;
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                float t[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) t[e] = pv_value(o[dt].h[8 * jj + e], o[dt].m[8 * jj + e], o[dt].l[8 * jj + e]) * oscale;
                                u32x4 hi, lo;
                                split8(t, hi, lo);
                                const size_t idx = acc_slot(mo, h * 256 + tile * 32, jj, hf, a.HD16);
                                *(u32x4*)(a.o + idx) = hi;
                                *(u32x4*)(a.o + a.o_plane + idx) = lo;
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the second V^T half is spent in every wave: buffer 1 takes the next item's second K half (raw barrier: the first K half stays in flight)
            __builtin_amdgcn_s_barrier();
            if (has_next) {
#pragma unroll
                for (int n = 0; n < NPIECE; ++n) dma_k_piece(nkr, 1, 1, n);
            }
        }
        mark(4);
        if (O8 && active) {
            int ml = m;
            asm volatile("" : "+v"(ml));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
            if (hf == 0) a.o_scale[(size_t)ml * a.H + h] = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
            for (int tile = 0; tile < 8; ++tile) {
                u32x4 s1, s2;
                quant16(t8[tile], inv, s1, s2);
                const size_t idx = acc_slot_i8(ml, h * 256 + tile * 32, hf, a.HD16 / 2);
                *(u32x4*)(a.o8 + idx) = s1;
                *(u32x4*)(a.o8 + a.o8_plane + idx) = s2;
            }
        }
        mark(5);
        if (!has_next) break;
        item = next;
        cur = nxt;
    }
}
