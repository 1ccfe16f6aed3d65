// attn_layer_i8w.h — the one-kernel attention layer (attn_layer_i8.h: TM:71-88 for one window x head per workgroup, K, V, Q and the
// probabilities never leaving the CU) on EIGHT waves, two per SIMD, 256 registers each.
//
// The 4-wave form spends 40 % of a workgroup's 50 us in VALU-only phases (three quantising epilogues, the softmax, the O store) during
// which the matrix pipe idles, and its main loops leave the pipe a third idle behind each wave's own LDS reads and waits: a lone wave per
// SIMD hides nothing.  Here two waves share a SIMD:
//   * projections: 8 waves side by side 4 (features) x 2 (tokens), 64f x 64t per wave (I8Acc pairs of 2 x 2 tiles = 128 registers), the
//     same LDS ring; one wave's fragment reads and waits sit in the other's MFMAs, and every epilogue is split over twice the waves;
//   * Q: projected in that same layout, quantised per query (row maximum across the 4 feature waves through LDS) into an LDS image that
//     overlays the idle ring, and read back by the four waves that own a query tile as the B fragments of S^T = K Q^T;
//   * S^T, softmax: wave (query tile, key half): 2 of the 4 key tiles each; the row maximum and the row sum of a query cross the
//     pair through LDS;
//   * PV: wave (query tile, d_v half); the probabilities of a query tile go through LDS (over the ring, idle again after the V main
//     loop), the row maximum of the int8 output across the two halves through LDS.
// Same integers as the 4-wave form (the row sum of the probabilities is now the sum of two half-row sums).
#pragma once
#include "attn_layer_i8.h"

using AW8K = GemmCfg<2, 2, 4, 2, 1, 2, false, 2, 3>;
using AW8V = GemmCfg<2, 2, 4, 2, 1, 2, true, 2, 3>;
static_assert(AW8K::SMEM_BYTES == AL8K::SMEM_BYTES, "same ring as the 4-wave form");

// EGOEGO_ATTN_PREFETCH (round 5): the Q projection's pipeline fill is requested BEFORE the K epilogue and the V projection's before S^T +
// softmax (GemmBody::prefetch), so that two of the three fills' round trips run behind arithmetic instead of in front of a main loop.
// For that the K / V^T image and the parameter block are STATIC LDS objects and only the operand ring is dynamic: hipcc puts a full
// vmcnt(0) in front of any LDS access it cannot prove disjoint from a pending LDS-DMA destination, and distinct objects are what it can prove.
// MEASURED (profiles/r05_attn_prefetch_ab.txt, B=256 and B=32, two runs each inside one gpurun): 199.2 / 199.6 us per launch with, 201.1 /
// 197.5 without; steps 1.3805 / 1.3824 against 1.3821 / 1.3804 ms — nothing: with two waves per SIMD the fills were already covered.  Off
// by default (round 4's layout); kept as an A/B knob of variant builds.
#ifndef EGOEGO_ATTN_PREFETCH
#define EGOEGO_ATTN_PREFETCH 0
#endif
static constexpr int AW_DYN_SMEM_BYTES = EGOEGO_ATTN_PREFETCH ? (int)GemmCfg<2, 2, 4, 2, 1, 2, false, 2, 3>::SMEM_BYTES : AL_SMEM_BYTES;

__global__ __launch_bounds__(512, 2) void attn_layer_i8w_kernel(AttnLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if EGOEGO_ATTN_PREFETCH
    __shared__ __attribute__((aligned(16))) char kv_img[AL_KV_BYTES];
    __shared__ __attribute__((aligned(16))) float misc[AL_MISC_BYTES / 4];
    char* kv = kv_img;                             // K image, later V^T image: [slice][tile][k32 block][1 KiB]
    float* sk = misc;                              // [128] key row scales
#else
    char* kv = smem;                               // K image, later V^T image: [slice][tile][k32 block][1 KiB]
    float* sk = (float*)(smem + AL_KV_BYTES);      // [128] key row scales
#endif
    float* sv = sk + 128;                          // [256] V column scales
    float* red = sv + 256;                         // [512] cross-wave maxima
    float* p_ws = red + 512;                       // [3][256] weight row scales of Q_h, K_h, V_h
    float* p_b = p_ws + 768;                       // [3][256] biases
    float* p_hs = p_b + 768;                       // [128] row scales of the window's int8 input rows
    float* sqv = p_hs + 128;                       // [128] query row scales
    float* psum = sqv + 128;                       // [2][128] half-row sums of the probabilities (key half, query)
    static_assert((128 + 256 + 512 + 768 + 768 + 128 + 128 + 256) * 4 <= AL_MISC_BYTES, "parameter block");
#if EGOEGO_ATTN_PREFETCH
    char* ring = smem;                                // operand ring (the only dynamic LDS); between main loops: the Q image, then the P image
#else
    char* ring = smem + AL_KV_BYTES + AL_MISC_BYTES;  // operand ring; between main loops: the Q image, then the P image
#endif
    using GK = GemmBody<AW8K, NoEpi>;
    using GV = GemmBody<AW8V, NoEpi>;
    constexpr bool PF = EGOEGO_ATTN_PREFETCH != 0;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the H heads of a window share an XCD (and its L2)
    const int bh = lid + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int wf = wave & 3, wt = wave >> 2;       // projections: feature quarter (64 features), token half (64 tokens)
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    EG_DBG(unsigned long long* tr = a.trace ? a.trace + 131072 + (size_t)blockIdx.x * 16 : nullptr;)
    auto mark = [&](int i) {  // perf-debug build: phase timestamps (tools/attn_layer_trace.py)
        EG_DBG(if (tr && threadIdx.x == 0) {
            tr[i] = wall_clock64();
            if (i < 2) tr[12 + i] = __builtin_readcyclecounter();
        })
        (void)i;
    };
    mark(0);
    // the per-feature parameters of the head and the window's row scales -> LDS; issued from inside the K projection's prologue, behind its
    // pipeline-fill LDS-DMA (in front of it, their round trip preceded the fill's: round 4); visible after that main loop's first barrier
    auto stage_params = [&] {
        const int HD = a.H * 256;
        for (int i = threadIdx.x; i < 768; i += 512) {
            const int src = (i >> 8) * HD + h * 256 + (i & 255);
            p_ws[i] = a.w_scale[src];
            p_b[i] = a.bias[src];
        }
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    };

    // One row-quantising projection epilogue for K and Q: dequantise, bias (x qs), maximum over the head's 256 features of each
    // token (in-lane over the wave's 64, then across the 4 feature waves through LDS), two int8 slices into `img`.
    auto rows_epilogue = [&](I8Acc (&q)[2][2], int which, float qs, float* scales, char* img) {
        const int f0 = which * 256 + wf * 64, t0 = wt * 64;
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
                char* dst = img + (((wt * 2 + j) * 8 + wf * 2 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    };

    // ---- 1. K_h -> LDS ------------------------------------------------------------------------------
    {
        I8Acc q[2][2];
        GK::mainloop(g, a.H + h, b, ring, q, stage_params);
        mark(1);
        if (PF) GK::prefetch(g, h, b, ring);  // (the main loop ended with a barrier: the ring is idle)
        // (the 4-wave form adds the bias without a scale: x 1.0f is exact)
        rows_epilogue(q, 1, 1.0f, sk, kv);
    }
    mark(2);
    // ---- 2. Q_h -> LDS image over the ring -> B fragments of the query-tile waves -------------------------
    i32x4 qs1[8], qs2[8];
    float sq = 0.f;
    {
        I8Acc q[2][2];
        GK::template mainloop<I8Acc, true, GK::NoPre, PF>(g, h, b, ring, q);  // (ends with a barrier: the ring is idle, every wave is past the K image writes)
        mark(3);
        rows_epilogue(q, 0, a.qscale, sqv, ring);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* src = ring + (((wave & 3) * 8 + i) << 10) + lane * 16;
            qs1[i] = lds_frag(src);
            qs2[i] = lds_frag(src + AL_SLICE);
        }
        sq = sqv[(wave & 3) * 32 + col];
        __syncthreads();  // the Q image is in registers: the ring may be refilled (V projection)
        if (PF) GV::prefetch(g, 2 * a.H + h, b, ring);
    }
    mark(4);
    // ---- 3. S^T = K Q^T, softmax over keys (TM:76-82): wave (query tile wave & 3, key half wave >> 2) -----------------------
    i32x4 ps1[2], ps2[2], ps3[2];  // this wave's two key blocks of the probabilities, three slices
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
        float p[2][16];
        float mx = -INFINITY;
        const float sq256 = sq * 256.0f * 1.44269504088896f;  // logits in units of log2(e): softmax through v_exp_f32
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
        mx = fmaxf(red[qt3 * 32 + col], red[128 + qt3 * 32 + col]);  // (key 0 always exists: finite)
        // row sum of the probabilities, in one order for every form of this kernel: per key tile (16 values of a lane in register
        // order + the other half-wave's), then (tile 0 + tile 1) + (tile 2 + tile 3)
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
        if (hf == 0) psum[kh * 128 + qt3 * 32 + col] = sum;  // read in phase 5, behind the V projection's barriers
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            u32x4 s1, s2, s3;
            quant_p(p[kt], P_QMAX, s1, s2, s3);
            ps1[kt] = __builtin_bit_cast(i32x4, s1);
            ps2[kt] = __builtin_bit_cast(i32x4, s2);
            ps3[kt] = __builtin_bit_cast(i32x4, s3);
        }
    }
    mark(5);  // (thread 0 belongs to wave 0: after its S^T + softmax)
    // ---- 4. V_h -> LDS (transposed, over the K image) ---------------------------------------------------
    {
        I8Acc q[2][2];
        GV::template mainloop<I8Acc, true, GV::NoPre, PF>(g, 2 * a.H + h, b, ring, q);  // its first barrier: every wave is past S^T (the K image is dead)
        mark(6);
        // the probabilities of the four query tiles -> LDS (over the idle ring) for the d_v-half waves of phase 5
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            char* dst = ring + (((wave & 3) * 4 + 2 * (wave >> 2) + kt) << 10) + lane * 16;
            *(i32x4*)dst = ps1[kt];
            *(i32x4*)(dst + 16384) = ps2[kt];
            *(i32x4*)(dst + 32768) = ps3[kt];
        }
        const int f0 = 512 + wf * 64, t0 = wt * 64;
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
            if (wt == 0 && hf == 0) sv[dv] = cmax > 0.f ? cmax / I8_QMAX : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                char* dst = kv + (((wf * 2 + i) * 4 + wt * 2 + j) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
        __syncthreads();
    }
    mark(7);
    // ---- 5. O^T = V^T P (TM:83-88): wave (query tile qt, d_v half dvh), heads merged on store -----------------------
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
        PVAcc o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acc_zero(o[dt]);
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
            for (int dt = 0; dt < 2; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2[dt], pa1[kb], o[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa1[kb], o[dt].h, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa2[kb], o[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt].l = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa3[kb], o[dt].l, 0, 0, 0);
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
                    const float val = pv_value(o[dt].h[r], o[dt].m[r], o[dt].l[r]) * (ss[c] * oscale);
                    t[2 * dp + dt][r] = val;
                    amax = fmaxf(amax, fabsf(val));
                }
            }
    }
    if (a.o8) {
        // int8 rows for the int8 fc: one scale per row and head = the maximum over both d_v halves (the partner wave's through LDS)
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
    EG_DBG(if (tr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(8);
    })
}
