// attn_split_i8.h — the eight-wave attention layer (attn_layer_i8w.h, TM:71-88) as TWO launches, for grids that leave most of the chip
// idle: a (window, head) workgroup of the one-kernel form is a serial chain of ~38 us whatever the batch, 29 of them its three
// projections.  Here the projections of a (window, head) are THREE workgroups running side by side,
//   attn_proj_i8_kernel   one eight-wave workgroup per (window, head, Q | K | V): the projection phase of attn_layer_i8w_kernel —
//                         same configuration, same quantising epilogue — writing its int8 image (the 64 KiB that kernel keeps in
//                         LDS, same layout) and its scales to memory,
//   attn_core_s_kernel    one eight-wave workgroup per (window, head): the three images come back by LDS-DMA (K into the image
//                         region, Q over the ring region, V over K once S^T is done) and phases 3 and 5 of attn_layer_i8w_kernel
//                         run unchanged,
// so up to 21 windows x 4 heads the projections take ONE main loop's time instead of three.  Same integers, the same float operations in
// the same order: a window's bits do not depend on which form computed it.  The images cross L2 (192 KiB per window x head: nothing
// at these sizes).
#pragma once
#include "attn_layer_i8w.h"
#include "attn_layer_i8h.h"

struct AttnSplitBufs {
    int8_t* img;   // [B*H][3][64 KiB]: Q, K (token tile, d_k block), V^T (d_v tile, key block); slice 2 at +32 KiB inside each image
    float* sq;     // [B*H][128] query row scales
    float* sk;     // [B*H][128] key row scales
    float* sv;     // [B*H][256] V column scales
};

__global__ __launch_bounds__(512, 2) void attn_proj_i8_kernel(AttnLayerArgs a, AttnSplitBufs o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;                     // [512] cross-wave maxima
    float* p_ws = red + 512;                       // [256] weight row scales of this projection
    float* p_b = p_ws + 256;                       // [256] biases
    float* p_hs = p_b + 256;                       // [128] row scales of the window's int8 input rows
    char* ring = smem + 8192;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the three projections of all H heads of a window share an XCD
    const int bhl = lid / 3, which = lid - bhl * 3;              // 0 = Q, 1 = K, 2 = V
    const int bh = bhl + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int wf = wave & 3, wt = wave >> 2;
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    {
        const int HD = a.H * 256;
        if (threadIdx.x < 256) {
            const int src = which * HD + h * 256 + threadIdx.x;
            p_ws[threadIdx.x] = a.w_scale[src];
            p_b[threadIdx.x] = a.bias[src];
        } else if (threadIdx.x < 384) {
            p_hs[threadIdx.x - 256] = a.h_scale[b * 128 + threadIdx.x - 256];
        }
    }  // visible after the first barrier of the main loop
    int8_t* const img = o.img + ((size_t)bh * 3 + which) * 65536;
    const int t0 = wt * 64;
    if (which < 2) {
        // ---- Q_h / K_h: attn_layer_i8w_kernel's rows_epilogue, the image to memory
        I8Acc q[2][2];
        GemmBody<AW8K, NoEpi>::mainloop(g, which * a.H + h, b, ring, q);
        const float qs = which == 0 ? a.qscale : 1.0f;
        float* const scales = (which == 0 ? o.sq : o.sk) + (size_t)bh * 128;
        const int f0 = wf * 64;
        f32x16 v[2][2];
        float amax[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sa = p_hs[t0 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                i8_dequant(q[i][j], v[i][j], p_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *(const float4*)(p_b + f0 + i * 32 + 8 * gq + 4 * hf);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[i][j][4 * gq + c] = (v[i][j][4 * gq + c] + bb[c]) * qs;
                        amax[j] = fmaxf(amax[j], fabsf(v[i][j][4 * gq + c]));
                    }
                }
            }
            amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
            if (hf == 0) red[wf * 128 + t0 + j * 32 + col] = amax[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tok = t0 + j * 32 + col;
            const float rmax = fmaxf(fmaxf(red[tok], red[128 + tok]), fmaxf(red[256 + tok], red[384 + tok]));
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) scales[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                int8_t* dst = img + (((wt * 2 + j) * 8 + wf * 2 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    } else {
        // ---- V_h: attn_layer_i8w_kernel's phase-4 epilogue, the transposed image to memory
        I8Acc q[2][2];
        GemmBody<AW8V, NoEpi>::mainloop(g, 2 * a.H + h, b, ring, q);
        const int f0 = wf * 64;
        f32x16 v[2][2];
        float amax[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sw = p_ws[f0 + i * 32 + col], bf = p_b[f0 + i * 32 + col];
            amax[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                i8_dequant_rows(q[i][j], v[i][j], sw, p_hs + t0 + j * 32 + 4 * hf);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[i][j][r] += bf;
                    amax[i] = fmaxf(amax[i], fabsf(v[i][j][r]));
                }
            }
            amax[i] = fmaxf(amax[i], __shfl_xor(amax[i], 32));
            if (hf == 0) red[wt * 256 + wf * 64 + i * 32 + col] = amax[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int dv = wf * 64 + i * 32 + col;
            const float cmax = fmaxf(red[dv], red[256 + dv]);
            const float inv = cmax > 0.f ? I8_QMAX / cmax : 0.f;
            if (wt == 0 && hf == 0) o.sv[(size_t)bh * 256 + dv] = cmax > 0.f ? cmax / I8_QMAX : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                int8_t* dst = img + (((wf * 2 + i) * 4 + wt * 2 + j) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    }
}
static constexpr int ATTN_PROJ_SMEM = 8192 + AW8K::SMEM_BYTES;

// The same projections in SIX four-wave workgroups per (window, head): Q and K by token halves (their scales are per token row),
// V by feature halves (its scales are per feature column over the window's keys) — for calls of at most 10 windows, where each of
// the six still gets a CU of its own and the projection time halves once more (B=1: 0.232 -> 0.222 ms per step).  256 registers per
// wave, 68 KiB of LDS.  Per wave the same 64f x 64t tile and the same integer sums; the same epilogue code per value; same images.
// (Two per CU in more than one round — 22 windows and more — it loses to the one-kernel forms: measured, egoego_hip.hip.)
using AP6K = GemmCfg<2, 2, 4, 1, 1, 2, false, 2, 3>;  // 256 features x 64 tokens (Q / K half)
using AP6V = GemmCfg<2, 2, 2, 2, 1, 2, true, 2, 3>;   // 128 features x 128 tokens (V half), un-swapped accumulator
static constexpr int ATTN_PROJ6_SMEM = 8192 + (AP6K::SMEM_BYTES > AP6V::SMEM_BYTES ? AP6K::SMEM_BYTES : AP6V::SMEM_BYTES);

__global__ __launch_bounds__(256, 2) void attn_proj6_i8_kernel(AttnLayerArgs a, AttnSplitBufs o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;                     // [256] cross-wave maxima
    float* p_ws = red + 512;                       // [256] weight row scales of this projection
    float* p_b = p_ws + 256;                       // [256] biases
    float* p_hs = p_b + 256;                       // [128] row scales of the window's int8 input rows
    char* ring = smem + 8192;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the six workgroups of all H heads of a window share an XCD
    const int bhl = lid / 6, part = lid - bhl * 6;               // 0, 1: Q token halves; 2, 3: K token halves; 4, 5: V feature halves
    const int which = part >> 1, half = part & 1;
    const int bh = bhl + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    {
        const int src = which * a.H * 256 + h * 256 + threadIdx.x;
        p_ws[threadIdx.x] = a.w_scale[src];
        p_b[threadIdx.x] = a.bias[src];
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    }  // visible after the first barrier of the main loop
    int8_t* const img = o.img + ((size_t)bh * 3 + which) * 65536;
    if (which < 2) {
        // ---- one token half of Q_h / K_h: four waves side by side along the features
        const int wf = wave, t0 = half * 64;
        I8Acc q[2][2];
        GemmBody<AP6K, NoEpi>::mainloop(g, which * a.H + h, b * 2 + half, ring, q);
        const float qs = which == 0 ? a.qscale : 1.0f;
        float* const scales = (which == 0 ? o.sq : o.sk) + (size_t)bh * 128;
        const int f0 = wf * 64;
        f32x16 v[2][2];
        float amax[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sa = p_hs[t0 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                i8_dequant(q[i][j], v[i][j], p_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *(const float4*)(p_b + f0 + i * 32 + 8 * gq + 4 * hf);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[i][j][4 * gq + c] = (v[i][j][4 * gq + c] + bb[c]) * qs;
                        amax[j] = fmaxf(amax[j], fabsf(v[i][j][4 * gq + c]));
                    }
                }
            }
            amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
            if (hf == 0) red[wf * 64 + j * 32 + col] = amax[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tl = j * 32 + col, tok = t0 + tl;
            const float rmax = fmaxf(fmaxf(red[tl], red[64 + tl]), fmaxf(red[128 + tl], red[192 + tl]));
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) scales[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                int8_t* dst = img + (((half * 2 + j) * 8 + wf * 2 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    } else {
        // ---- one feature half of V_h: waves 2 (features) x 2 (tokens)
        const int wf = wave & 1, wt = wave >> 1, t0 = wt * 64;
        I8Acc q[2][2];
        GemmBody<AP6V, NoEpi>::mainloop(g, (2 * a.H + h) * 2 + half, b, ring, q);
        const int f0 = half * 128 + wf * 64;  // feature of the head
        f32x16 v[2][2];
        float amax[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sw = p_ws[f0 + i * 32 + col], bf = p_b[f0 + i * 32 + col];
            amax[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                i8_dequant_rows(q[i][j], v[i][j], sw, p_hs + t0 + j * 32 + 4 * hf);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[i][j][r] += bf;
                    amax[i] = fmaxf(amax[i], fabsf(v[i][j][r]));
                }
            }
            amax[i] = fmaxf(amax[i], __shfl_xor(amax[i], 32));
            if (hf == 0) red[wt * 128 + wf * 64 + i * 32 + col] = amax[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int dl = wf * 64 + i * 32 + col, dv = half * 128 + dl;
            const float cmax = fmaxf(red[dl], red[128 + dl]);
            const float inv = cmax > 0.f ? I8_QMAX / cmax : 0.f;
            if (wt == 0 && hf == 0) o.sv[(size_t)bh * 256 + dv] = cmax > 0.f ? cmax / I8_QMAX : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                int8_t* dst = img + ((((half * 2 + wf) * 2 + i) * 4 + wt * 2 + j) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    }
}

// The projections as TWO eight-wave workgroups per (window, head), 1.5 projections each: part 0 = K_h (all 128 keys) + Q_h of tokens
// 0..63, part 1 = V_h (all keys) + Q_h of tokens 64..127 — the halves attn_proj6_i8_kernel already cuts Q into.  For 22..32 windows
// per GPU (the per-rank shard of the 8-GPU split of BASELINE configs[2]): 256 workgroups = ONE round of the chip, where the
// three-workgroup form needs 1.5 rounds and the one-kernel form leaves half the CUs idle behind a 38-us serial chain.
// Same configurations, same epilogue code per value, same images as the other split forms: same bits (round 4).
__global__ __launch_bounds__(512, 2) void attn_proj2_i8_kernel(AttnLayerArgs a, AttnSplitBufs o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;                     // [512] cross-wave maxima
    float* p_ws = red + 512;                       // [2][256] weight row scales: the main projection (K or V), Q
    float* p_b = p_ws + 512;                       // [2][256] biases
    float* p_hs = p_b + 512;                       // [128] row scales of the window's int8 input rows
    char* ring = smem + 8192;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // both parts of all H heads of a window share an XCD
    const int bhl = lid >> 1, part = lid & 1;
    const int which = 1 + part;                                  // the full projection of this part: 1 = K, 2 = V
    const int bh = bhl + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int wf = wave & 3, wt = wave >> 2;
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    {
        const int HD = a.H * 256;
        const int sel = threadIdx.x >> 8, f = threadIdx.x & 255;  // 0: the main projection, 1: Q
        const int src = (sel ? 0 : which) * HD + h * 256 + f;
        p_ws[threadIdx.x] = a.w_scale[src];
        p_b[threadIdx.x] = a.bias[src];
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    }  // visible after the first barrier of the main loop
    int8_t* const img = o.img + ((size_t)bh * 3 + which) * 65536;
    {
        const int t0 = wt * 64, f0 = wf * 64;
        if (which == 1) {
            // ---- K_h: attn_proj_i8_kernel's Q / K branch
            I8Acc q[2][2];
            GemmBody<AW8K, NoEpi>::mainloop(g, a.H + h, b, ring, q);
            float* const scales = o.sk + (size_t)bh * 128;
            f32x16 v[2][2];
            float amax[2] = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float sa = p_hs[t0 + j * 32 + col];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    i8_dequant(q[i][j], v[i][j], p_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 b4 = *(const float4*)(p_b + f0 + i * 32 + 8 * gq + 4 * hf);
                        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            v[i][j][4 * gq + c] = (v[i][j][4 * gq + c] + bb[c]) * 1.0f;
                            amax[j] = fmaxf(amax[j], fabsf(v[i][j][4 * gq + c]));
                        }
                    }
                }
                amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
                if (hf == 0) red[wf * 128 + t0 + j * 32 + col] = amax[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tok = t0 + j * 32 + col;
                const float rmax = fmaxf(fmaxf(red[tok], red[128 + tok]), fmaxf(red[256 + tok], red[384 + tok]));
                const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
                if (wf == 0 && hf == 0) scales[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                    u32x4 s1, s2;
                    quant16(t, inv, s1, s2);
                    int8_t* dst = img + (((wt * 2 + j) * 8 + wf * 2 + i) << 10) + lane * 16;
                    *(u32x4*)dst = s1;
                    *(u32x4*)(dst + AL_SLICE) = s2;
                }
            }
        } else {
            // ---- V_h: attn_proj_i8_kernel's V branch
            I8Acc q[2][2];
            GemmBody<AW8V, NoEpi>::mainloop(g, 2 * a.H + h, b, ring, q);
            f32x16 v[2][2];
            float amax[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float sw = p_ws[f0 + i * 32 + col], bf = p_b[f0 + i * 32 + col];
                amax[i] = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    i8_dequant_rows(q[i][j], v[i][j], sw, p_hs + t0 + j * 32 + 4 * hf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        v[i][j][r] += bf;
                        amax[i] = fmaxf(amax[i], fabsf(v[i][j][r]));
                    }
                }
                amax[i] = fmaxf(amax[i], __shfl_xor(amax[i], 32));
                if (hf == 0) red[wt * 256 + wf * 64 + i * 32 + col] = amax[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int dv = wf * 64 + i * 32 + col;
                const float cmax = fmaxf(red[dv], red[256 + dv]);
                const float inv = cmax > 0.f ? I8_QMAX / cmax : 0.f;
                if (wt == 0 && hf == 0) o.sv[(size_t)bh * 256 + dv] = cmax > 0.f ? cmax / I8_QMAX : 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                    u32x4 s1, s2;
                    quant16(t, inv, s1, s2);
                    int8_t* dst = img + (((wf * 2 + i) * 4 + wt * 2 + j) << 10) + lane * 16;
                    *(u32x4*)dst = s1;
                    *(u32x4*)(dst + AL_SLICE) = s2;
                }
            }
        }
    }
    // ---- Q_h of this part's 64 tokens: waves 4 (features) x 2 (token tiles), 64f x 32t per wave (attn_layer_i8h_kernel's phase 2),
    // the image rows and scales to memory.  (The main loop's first barrier also separates the reads of `red` above from the writes below.)
    {
        const int qh = part;
        I8Acc q[2][1];
        GemmBody<AH8Q, NoEpi>::mainloop(g, h, b * 2 + qh, ring, q);
        const int f0 = wf * 64;
        const float* const q_ws = p_ws + 256;
        const float* const q_b = p_b + 256;
        f32x16 v[2];
        float amax = 0.f;
        const float sa = p_hs[qh * 64 + wt * 32 + col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            i8_dequant(q[i][0], v[i], q_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 b4 = *(const float4*)(q_b + f0 + i * 32 + 8 * gq + 4 * hf);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[i][4 * gq + c] = (v[i][4 * gq + c] + bb[c]) * a.qscale;
                    amax = fmaxf(amax, fabsf(v[i][4 * gq + c]));
                }
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if (hf == 0) red[wf * 64 + wt * 32 + col] = amax;
        __syncthreads();
        const int tl = wt * 32 + col, tok = qh * 64 + tl;
        const float rmax = fmaxf(fmaxf(red[tl], red[64 + tl]), fmaxf(red[128 + tl], red[192 + tl]));
        const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
        if (wf == 0 && hf == 0) o.sq[(size_t)bh * 128 + tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
        int8_t* const qimg = o.img + (size_t)bh * 3 * 65536;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = v[i][r];
            u32x4 s1, s2;
            quant16(t, inv, s1, s2);
            int8_t* dst = qimg + (((qh * 2 + wt) * 8 + wf * 2 + i) << 10) + lane * 16;
            *(u32x4*)dst = s1;
            *(u32x4*)(dst + AL_SLICE) = s2;
        }
    }
}

__global__ __launch_bounds__(512, 2) void attn_core_s_kernel(AttnLayerArgs a, AttnSplitBufs o) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                               // K image, later V^T image
    float* sk = (float*)(smem + AL_KV_BYTES);      // [128] key row scales
    float* sv = sk + 128;                          // [256] V column scales
    float* red = sv + 256;                         // [512] cross-wave maxima
    float* sqv = red + 512;                        // [128] query row scales
    float* psum = sqv + 128;                       // [2][128] half-row sums of the probabilities
    char* ring = smem + AL_KV_BYTES + AL_MISC_BYTES;  // the Q image, later the P image
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int bh = lid + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int8_t* const img = o.img + (size_t)bh * 3 * 65536;
    // an image = 64 one-KiB pieces, 8 per wave (both slices)
    const __amdgpu_buffer_rsrc_t ir = gemm_rsrc(img);  // (buffer form, common.h: the three images of this (window, head), 192 KiB)
    auto dma_image = [&](const int8_t* src, char* dst) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int pc = n * 8 + wave;
            if constexpr (EGOEGO_GEMM_BUFFER_DMA != 0)
                gemm_dma_piece(ir, dst + (pc << 10), (unsigned)(src - img) + (unsigned)(pc << 10), lane);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)pc << 10) + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + (pc << 10)), 16, 0, 0);
        }
    };
    dma_image(img + 65536, kv);    // K
    dma_image(img, ring);          // Q
    if (threadIdx.x < 128) {
        sk[threadIdx.x] = o.sk[(size_t)bh * 128 + threadIdx.x];
        sqv[threadIdx.x] = o.sq[(size_t)bh * 128 + threadIdx.x];
    } else if (threadIdx.x < 384) {
        sv[threadIdx.x - 128] = o.sv[(size_t)bh * 256 + threadIdx.x - 128];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    i32x4 qs1[8], qs2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const char* src = ring + (((wave & 3) * 8 + i) << 10) + lane * 16;
        qs1[i] = lds_frag(src);
        qs2[i] = lds_frag(src + AL_SLICE);
    }
    const float sq = sqv[(wave & 3) * 32 + col];
    // ---- 3. S^T = K Q^T, softmax over keys: wave (query tile wave & 3, key half wave >> 2) — attn_layer_i8w_kernel phase 3
    i32x4 ps1[2], ps2[2], ps3[2];
    {
        const int qt3 = wave & 3, kh = wave >> 2;
        I8Acc s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) acc_zero(s[kt]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            i32x4 k1[2], k2[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const char* src = kv + (((2 * kh + kt) * 8 + i) << 10) + lane * 16;
                k1[kt] = lds_frag(src);
                k2[kt] = lds_frag(src + AL_SLICE);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k2[kt], qs1[i], s[kt].m, 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1[kt], qs2[i], s[kt].m, 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1[kt], qs1[i], s[kt].h, 0, 0, 0);
        }
        __syncthreads();              // every wave is done with the K image and holds its Q fragments
        dma_image(img + 131072, kv);  // V^T over K; lands during the softmax
        float p[2][16];
        float mx = -INFINITY;
        const float sq256 = sq * 256.0f * 1.44269504088896f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 k4 = *(const float4*)(sk + (2 * kh + kt) * 32 + 8 * gq + 4 * hf);
                const float ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * gq + c;
                    float val = (float)i8_combine(s[kt].h[r], s[kt].m[r]) * (sq256 * ks[c]);
                    if ((2 * kh + kt) * 32 + 8 * gq + 4 * hf + c >= a.L) val = -INFINITY;
                    p[kt][r] = val;
                    mx = fmaxf(mx, val);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (hf == 0) red[kh * 128 + qt3 * 32 + col] = mx;
        __syncthreads();
        mx = fmaxf(red[qt3 * 32 + col], red[128 + qt3 * 32 + col]);
        float st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(p[kt][r] - mx);
                s1 += p[kt][r];
            }
            st[kt] = s1 + __shfl_xor(s1, 32);
        }
        const float sum = st[0] + st[1];
        if (hf == 0) psum[kh * 128 + qt3 * 32 + col] = sum;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            u32x4 s1, s2, s3;
            quant_p(p[kt], P_QMAX, s1, s2, s3);
            ps1[kt] = __builtin_bit_cast(i32x4, s1);
            ps2[kt] = __builtin_bit_cast(i32x4, s2);
            ps3[kt] = __builtin_bit_cast(i32x4, s3);
        }
        // the probabilities of the four query tiles -> LDS over the Q image (every wave read its Q fragments before the barrier above)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            char* dst = ring + (((wave & 3) * 4 + 2 * (wave >> 2) + kt) << 10) + lane * 16;
            *(i32x4*)dst = ps1[kt];
            *(i32x4*)(dst + 16384) = ps2[kt];
            *(i32x4*)(dst + 32768) = ps3[kt];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the V^T image
    __syncthreads();
    // ---- 5. O^T = V^T P: wave (query tile qt, d_v half dvh) — attn_layer_i8w_kernel phase 5
    const int qt = wave & 3, dvh = wave >> 2;
    const int m = b * 128 + qt * 32 + col;
    i32x4 pa1[4], pa2[4], pa3[4];  // all four key blocks of this wave's query tile, three slices (attn_layer_i8.h quant_p)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const char* src = ring + ((qt * 4 + kb) << 10) + lane * 16;
        pa1[kb] = lds_frag(src);
        pa2[kb] = lds_frag(src + 16384);
        pa3[kb] = lds_frag(src + 32768);
    }
    const float oscale = (1.0f / (psum[qt * 32 + col] + psum[128 + qt * 32 + col])) * (256.0f / P_QMAX);
    float t[4][16];
    float amax = 0.f;
    // two d_v tiles at a time (three int32 accumulators per tile: the four at once would not leave room for their float results)
#pragma unroll
    for (int dp = 0; dp < 2; ++dp) {
        PVAcc oa[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acc_zero(oa[dt]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            i32x4 v1[2], v2[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* src = kv + (((dvh * 4 + 2 * dp + dt) * 4 + kb) << 10) + lane * 16;
                v1[dt] = lds_frag(src);
                v2[dt] = lds_frag(src + AL_SLICE);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oa[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2[dt], pa1[kb], oa[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oa[dt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa1[kb], oa[dt].h, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oa[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa2[kb], oa[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oa[dt].l = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa3[kb], oa[dt].l, 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 s4 = *(const float4*)(sv + (dvh * 4 + 2 * dp + dt) * 32 + 8 * gq + 4 * hf);
                const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * gq + c;
                    const float val = pv_value(oa[dt].h[r], oa[dt].m[r], oa[dt].l[r]) * (ss[c] * oscale);
                    t[2 * dp + dt][r] = val;
                    amax = fmaxf(amax, fabsf(val));
                }
            }
    }
    if (a.o8) {
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if (hf == 0) red[dvh * 128 + qt * 32 + col] = amax;
        __syncthreads();
        amax = fmaxf(red[qt * 32 + col], red[128 + qt * 32 + col]);
        const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
        if (dvh == 0 && hf == 0) a.o_scale[(size_t)m * a.H + h] = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x4 s1, s2;
            quant16(t[dt], inv, s1, s2);
            const size_t idx = acc_slot_i8(m, h * 256 + (dvh * 4 + dt) * 32, hf, a.HD16 / 2);
            *(u32x4*)(a.o8 + idx) = s1;
            *(u32x4*)(a.o8 + a.o8_plane + idx) = s2;
        }
    } else {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                u32x4 hi, lo;
                split8(t[dt] + 8 * jj, hi, lo);
                const size_t idx = acc_slot(m, h * 256 + (dvh * 4 + dt) * 32, jj, hf, a.HD16);
                *(u32x4*)(a.o + idx) = hi;
                *(u32x4*)(a.o + a.o_plane + idx) = lo;
            }
    }
}
static constexpr int ATTN_CORE_S_SMEM = AL_KV_BYTES + AL_MISC_BYTES + 65536;
