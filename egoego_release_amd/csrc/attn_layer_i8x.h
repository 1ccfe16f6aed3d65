// attn_layer_i8x.h — the one-kernel attention layer (attn_layer_i8w.h: TM:71-88 for one window x head per workgroup, K, V, Q and the
// probabilities never leaving the CU) with the projections' operands on the path round 4's micro-benchmark picked
// (tools/microbench/attn_step_mix.hip, per k-step with all CUs busy: LDS ring 0.60-0.68 us, this path 0.54-0.56):
//   * eight waves side by side along the FEATURES, 8 (f) x 1 (t): a wave owns one 32-feature tile of the head for all 128 tokens
//     (four token tiles: I8Acc pairs of 1 x 4 tiles = 128 registers, as before);
//   * its WEIGHT fragments are needed by no other wave, so they go global -> VGPR through a buffer resource (two 16-byte loads per
//     k-step, prefetched three k-steps ahead in a register ring): no LDS write, no LDS read, no barrier for them, and — unlike the
//     4 (f) x 2 (t) layout with direct weights (0.59 us) — no fragment crosses the CU's L1 path twice;
//   * the ACTIVATIONS (every wave needs all four token tiles) stream by LDS-DMA in chunks of 4 k-steps, double-buffered (2 x 32 KiB
//     where the ring took 72): one piece per wave and k-step instead of three, one barrier per FOUR k-steps instead of one per step.
// LDS fragment reads per MFMA are what they were (8 per 12: fixed by the 128-register accumulator tile); what goes away is two
// thirds of the LDS-DMA traffic, three quarters of the barriers and the LDS round trip of the weights.
// Phases 3 (S^T, softmax) and 5 (PV) are attn_layer_i8w.h's, verbatim.  Same integers, the same float operations per value in the
// same order (row / column maxima are order-independent): bit-identical to the ring form and to the split forms of
// attn_split_i8.h, which keep the ring (test_small_grid_kernels_give_the_bits_of_the_large_grid_ones).
#pragma once
#include "attn_layer_i8w.h"
#include "tail_fused.h"

static constexpr int ALX_MISC_BYTES = 14336;                // + 512 floats: the row maxima of K / Q cross EIGHT feature waves
static constexpr int ALX_CHUNK_BYTES = 32768;               // [slice][token tile 4][k-block 4][1 KiB]
static constexpr int ALX_SMEM_BYTES = AL_KV_BYTES + ALX_MISC_BYTES + 2 * ALX_CHUNK_BYTES;

// One projection: q[j] = W[32 features of this wave][512] x h[token tile j][512]^T over 16 k-blocks of 32, int8 slices
// (three MFMAs per tile pair: s2*s1 and s1*s2 into .m, s1*s1 into .h, the order of gemm.h's mma_part).  ACT_ROWS: un-swapped
// accumulator (rows = tokens) for V.  `wrow`: the weight row tile (32 rows) of this wave; `trow0`: the window's first token tile.
template <bool ACT_ROWS>
struct Proj81 {
    static constexpr int CK = 4, NQ = 4, RING = 4, PD = 3, DP = 4;  // chunk length, chunks, weight ring, prefetch distance, DMA pieces per wave and chunk
    // vector-memory operations issued after the weight loads of (chunk-local) k-step ks: what may still be in flight when they are needed
    static constexpr int allowed(int ks, bool last) {
        if (!last) return ks == 0 ? 4 : 4 + DP;          // every step prefetches 2 loads; step 0 also issues the DP pieces (after its loads)
        return ks == 0 ? 4 : (ks == 1 ? 4 : (ks == 2 ? 2 : 0));  // last chunk: only step 0 prefetches (k-step 3), no pieces
    }
    template <int G>
    static EG_D void mma(i32x4 w, i32x4 a, I8Acc& c) {
        i32x16& d = G == 2 ? c.h : c.m;
        d = ACT_ROWS ? __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, d, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, d, 0, 0, 0);
    }
    template <int KS, bool LAST, class Dma>
    static EG_D void step(I8Acc (&q)[4], i32x4 (&wq)[RING][2], i32x4 (&ah)[2][4], i32x4 (&al)[4], tail_rsrc wr, unsigned w0, unsigned w1, int kbase,
                          const char* act, int lane, Dma& dma) {
        constexpr int cur = KS & 1;
        asm volatile("" ::: "memory");
        wait_counts<allowed(KS, LAST), (KS == 0 ? 0 : 4)>();  // this k-step's weights (counted: younger loads stay in flight) and hi-slice activations
        __builtin_amdgcn_sched_barrier(0);
        if (!LAST || KS + PD < CK) {
            const unsigned kb = (unsigned)(kbase + KS + PD) << 10;
            wq[(KS + PD) % RING][0] = tail_load(wr, lane * 16, w0 + kb);
            wq[(KS + PD) % RING][1] = tail_load(wr, lane * 16, w1 + kb);
        }
        if (KS == 0 && !LAST) {
#pragma unroll
            for (int d = 0; d < DP; ++d) dma(d);
        }
        if (KS < CK - 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ah[cur ^ 1][j] = *(const i32x4*)(act + ((j * CK + KS + 1) << 10) + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const i32x4 wh = wq[KS % RING][0], wl = wq[KS % RING][1];
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<0>(wl, ah[cur][j], q[j]);
        asm volatile("" ::: "memory");
        wait_counts<63, (KS < CK - 1 ? 4 : 0)>();  // the lo-slice activations of this k-step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<1>(wh, al[j], q[j]);
        __builtin_amdgcn_sched_barrier(0);
        if (KS < CK - 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) al[j] = *(const i32x4*)(act + 16384 + ((j * CK + KS + 1) << 10) + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<2>(wh, ah[cur][j], q[j]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // chunks: the 64 KiB double buffer.  Ends with a workgroup barrier (every wave is done with the buffers).
    static EG_D void run(I8Acc (&q)[4], const int8_t* w8, size_t w_plane, int wrow, const int8_t* h8, size_t h_plane, int trow0, char* chunks,
                         int wave, int lane) {
        const tail_rsrc wr = tail_make_rsrc(w8, 2 * w_plane);
        const tail_rsrc hr = tail_make_rsrc(h8 + (((size_t)trow0 * 16) << 10), 0x7fffffffu);
        const unsigned w0 = (unsigned)((wrow * 16) << 10), w1 = w0 + (unsigned)w_plane;
        const unsigned hp = (unsigned)h_plane;
        i32x4 wq[RING][2];
        auto dma_piece = [&](int c, int piece) {  // flat piece index over [slice][token tile][k-block]
            const int x = wave * DP + piece;
            const int s = x >> 4, j = (x >> 2) & 3, kb = x & 3;
            const unsigned src = s * hp + (unsigned)((j * 16 + c * CK + kb) << 10);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (__attribute__((address_space(3))) void*)(chunks + (c & 1) * ALX_CHUNK_BYTES + (x << 10)), 16,
                                                     lane * 16, src, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < DP; ++d) dma_piece(0, d);
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            wq[k][0] = tail_load(wr, lane * 16, w0 + ((unsigned)k << 10));
            wq[k][1] = tail_load(wr, lane * 16, w1 + ((unsigned)k << 10));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_zero(q[j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        i32x4 ah[2][4], al[4];
        auto chunk = [&](int c, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const char* act = chunks + (c & 1) * ALX_CHUNK_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ah[0][j] = *(const i32x4*)(act + ((j * CK) << 10) + lane * 16);
                al[j] = *(const i32x4*)(act + 16384 + ((j * CK) << 10) + lane * 16);
            }
            auto dma = [&](int piece) { dma_piece(c + 1, piece); };
            step<0, LAST>(q, wq, ah, al, wr, w0, w1, c * CK, act, lane, dma);
            step<1, LAST>(q, wq, ah, al, wr, w0, w1, c * CK, act, lane, dma);
            step<2, LAST>(q, wq, ah, al, wr, w0, w1, c * CK, act, lane, dma);
            step<3, LAST>(q, wq, ah, al, wr, w0, w1, c * CK, act, lane, dma);
        };
        for (int c = 0; c + 1 < NQ; ++c) {
            chunk(c, std::false_type{});
            // the next chunk has landed (this wave's pieces: the 6 weight loads issued after them may stay in flight) and everyone is
            // done with this one
            asm volatile("" ::: "memory");
            wait_counts<6, 15>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        chunk(NQ - 1, std::true_type{});
        __syncthreads();
    }
};

__global__ __launch_bounds__(512, 2) void attn_layer_i8x_kernel(AttnLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                               // K image, later V^T image: [slice][tile][k32 block][1 KiB]
    float* sk = (float*)(smem + AL_KV_BYTES);      // [128] key row scales
    float* sv = sk + 128;                          // [256] V column scales
    float* red = sv + 256;                         // [1024] cross-wave maxima: [8 feature waves][128 tokens]
    float* p_ws = red + 1024;                      // [3][256] weight row scales of Q_h, K_h, V_h
    float* p_b = p_ws + 768;                       // [3][256] biases
    float* p_hs = p_b + 768;                       // [128] row scales of the window's int8 input rows
    float* sqv = p_hs + 128;                       // [128] query row scales
    float* psum = sqv + 128;                       // [2][128] half-row sums of the probabilities (key half, query)
    static_assert((128 + 256 + 1024 + 768 + 768 + 128 + 128 + 256) * 4 <= ALX_MISC_BYTES, "parameter block");
    char* ring = smem + AL_KV_BYTES + ALX_MISC_BYTES;  // the activation chunk double buffer; between main loops: the Q image, then the P image
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the H heads of a window share an XCD (and its L2)
    const int bh = lid + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    EG_DBG(unsigned long long* tr = a.trace ? a.trace + 131072 + (size_t)blockIdx.x * 16 : nullptr;)
    auto mark = [&](int i) {  // perf-debug build: phase timestamps (tools/attn_layer_trace.py)
        EG_DBG(if (tr && threadIdx.x == 0) {
            tr[i] = wall_clock64();
            if (i < 2) tr[12 + i] = __builtin_readcyclecounter();
        })
        (void)i;
    };
    mark(0);
    {
        const int HD = a.H * 256;
        for (int i = threadIdx.x; i < 768; i += 512) {
            const int src = (i >> 8) * HD + h * 256 + (i & 255);
            p_ws[i] = a.w_scale[src];
            p_b[i] = a.bias[src];
        }
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    }  // visible after the first projection's prologue barrier

    // Row-quantising epilogue of K and Q: the wave holds 32 features (tile `wave` of the head) of all 128 tokens.  Dequantise, bias
    // (x qs), maximum over the head's 256 features of each token (in-lane over the wave's 16, the other half-wave's, then across the
    // 8 feature waves through LDS), two int8 slices into `img` ([token tile][k32 block = wave][1 KiB], slice 2 at + AL_SLICE).
    auto rows_epilogue = [&](I8Acc (&q)[4], int which, float qs, float* scales, char* img) {
        const int f0 = which * 256 + wave * 32;
        f32x16 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (one token tile at a time: left alone hipcc hoists every tile's parameter loads to the top and spills)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float sa = p_hs[j * 32 + col];
            float amax = 0.f;
            i8_dequant(q[j], v[j], p_ws + f0 + 4 * hf, sa);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 b4 = *(const float4*)(p_b + f0 + 8 * gq + 4 * hf);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[j][4 * gq + c] = (v[j][4 * gq + c] + bb[c]) * qs;
                    amax = fmaxf(amax, fabsf(v[j][4 * gq + c]));
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            if (hf == 0) red[wave * 128 + j * 32 + col] = amax;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = j * 32 + col;
            float rmax = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) rmax = fmaxf(rmax, red[w * 128 + tok]);
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wave == 0 && hf == 0) scales[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = v[j][r];
            u32x4 s1, s2;
            quant16(t, inv, s1, s2);
            char* dst = img + ((j * 8 + wave) << 10) + lane * 16;
            *(u32x4*)dst = s1;
            *(u32x4*)(dst + AL_SLICE) = s2;
        }
    };

    // ---- 1. K_h -> LDS ------------------------------------------------------------------------------
    {
        I8Acc q[4];
        Proj81<false>::run(q, a.w8, a.w_plane, (a.H + h) * 8 + wave, a.h8, a.h_plane, b * 4, ring, wave, lane);
        mark(1);
        rows_epilogue(q, 1, 1.0f, sk, kv);
    }
    mark(2);
    // ---- 2. Q_h -> LDS image over the chunk buffers -> B fragments of the query-tile waves -------------------------
    i32x4 qs1[8], qs2[8];
    float sq = 0.f;
    {
        I8Acc q[4];
        Proj81<false>::run(q, a.w8, a.w_plane, h * 8 + wave, a.h8, a.h_plane, b * 4, ring, wave, lane);  // (ends with a barrier: the buffers are idle)
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
        __syncthreads();  // the Q image is in registers: the buffers may be refilled (V projection)
    }
    mark(4);
    // ---- 3. S^T = K Q^T, softmax over keys (TM:76-82): wave (query tile wave & 3, key half wave >> 2) — attn_layer_i8w.h phase 3 -------
    i32x4 ps1[2], ps2[2];  // this wave's two key blocks of the probabilities
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
            u32x4 s1, s2;
            quant16(p[kt], I8_QMAX, s1, s2);
            ps1[kt] = __builtin_bit_cast(i32x4, s1);
            ps2[kt] = __builtin_bit_cast(i32x4, s2);
        }
    }
    mark(5);
    // ---- 4. V_h -> LDS (transposed, over the K image): un-swapped accumulator, a lane owns ONE feature for all 128 keys -----------
    {
        I8Acc q[4];
        Proj81<true>::run(q, a.w8, a.w_plane, (2 * a.H + h) * 8 + wave, a.h8, a.h_plane, b * 4, ring, wave, lane);  // its prologue barrier: every wave is past S^T (the K image is dead)
        mark(6);
        // the probabilities of the four query tiles -> LDS (over the idle chunk buffers) for the d_v-half waves of phase 5
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            char* dst = ring + (((wave & 3) * 4 + 2 * (wave >> 2) + kt) << 10) + lane * 16;
            *(i32x4*)dst = ps1[kt];
            *(i32x4*)(dst + 16384) = ps2[kt];
        }
        const int f0 = 512 + wave * 32;
        const float sw = p_ws[f0 + col], bf = p_b[f0 + col];
        f32x16 v[4];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            i8_dequant_rows(q[j], v[j], sw, p_hs + j * 32 + 4 * hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[j][r] += bf;
                amax = fmaxf(amax, fabsf(v[j][r]));
            }
        }
        const float cmax = fmaxf(amax, __shfl_xor(amax, 32));  // the column's maximum over all 128 keys: the whole column is in this wave
        const float inv = cmax > 0.f ? I8_QMAX / cmax : 0.f;
        if (hf == 0) sv[wave * 32 + col] = cmax > 0.f ? cmax / I8_QMAX : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = v[j][r];
            u32x4 s1, s2;
            quant16(t, inv, s1, s2);
            char* dst = kv + ((wave * 4 + j) << 10) + lane * 16;
            *(u32x4*)dst = s1;
            *(u32x4*)(dst + AL_SLICE) = s2;
        }
        __syncthreads();
    }
    mark(7);
    // ---- 5. O^T = V^T P (TM:83-88): wave (query tile qt, d_v half dvh), heads merged on store — attn_layer_i8w.h phase 5 ----------
    const int qt = wave & 3, dvh = wave >> 2;
    const int m = b * 128 + qt * 32 + col;
    i32x4 pa1[4], pa2[4];  // all four key blocks of this wave's query tile
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const char* src = ring + ((qt * 4 + kb) << 10) + lane * 16;
        pa1[kb] = lds_frag(src);
        pa2[kb] = lds_frag(src + 16384);
    }
    const float oscale = (1.0f / (psum[qt * 32 + col] + psum[128 + qt * 32 + col])) * (256.0f / I8_QMAX);
    I8Acc o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_zero(o[dt]);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        i32x4 v1[4], v2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* src = kv + (((dvh * 4 + dt) * 4 + kb) << 10) + lane * 16;
            v1[dt] = lds_frag(src);
            v2[dt] = lds_frag(src + AL_SLICE);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2[dt], pa1[kb], o[dt].m, 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa2[kb], o[dt].m, 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], pa1[kb], o[dt].h, 0, 0, 0);
    }
    float t[4][16];
    float amax = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 s4 = *(const float4*)(sv + (dvh * 4 + dt) * 32 + 8 * gq + 4 * hf);
            const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float val = (float)i8_combine(o[dt].h[4 * gq + c], o[dt].m[4 * gq + c]) * (ss[c] * oscale);
                t[dt][4 * gq + c] = val;
                amax = fmaxf(amax, fabsf(val));
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
