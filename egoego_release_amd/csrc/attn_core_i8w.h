// attn_core_i8w.h — the long-window attention core (attn_core_i8.h: TM:75-88 for windows of 129..224 tokens) on EIGHT waves, two per SIMD
// (round 5).
//
// The four-wave form gives a query tile to ONE 512-register wave per SIMD: its 20 us per item are 6.4 us of MFMAs, 4.4 of softmax /
// quantisation VALU and ~5 of accumulator conversions, one after the other, with nothing else on the SIMD to run meanwhile
// (HISTORY.md R4; 0.048 of the int8 peak).  Here a query tile belongs to a PAIR of 256-register waves on one SIMD, (query tile qt3 =
// wave & 3, half kh = wave >> 2), like the T <= 127 layer kernel (attn_layer_i8w.h):
//   * S^T = K Q^T: the pair splits the KEYS, 4 + (KT - 4) tiles (4 x 32 accumulator registers + the 64 of the Q fragments); the row
//     maximum, the row sum and the largest scaled probability of a query cross the pair through LDS;
//   * the probabilities (three slices per key block) cross the pair through an LDS image — 12 KT KiB over the second K buffer (dead by
//     then) and the 4 KT KiB behind it — and each wave ends up with all KT blocks of its query tile in registers (12 KT = 84);
//   * O^T = V^T P: the pair splits the d_v tiles of each V^T half (2 + 2), one tile at a time (one PVAcc = 48 registers), so both waves
//     work on the same half and the buffers turn over as before: V^T half 0 streams in during the second d_k half of S^T, half 1 behind the
//     probabilities' hand-over, the NEXT item's K halves as the V^T halves are spent (persistent workgroups, attn_core_i8.h);
//   * int8 output: the row maximum of the head's 256 features crosses the pair through LDS.
// Measured and NOT kept (round 5, one gpurun each): the logits as I * (s_k 256 log2 e) with the query's scale applied inside the exponent's fma —
// one multiply and one subtraction less per probability — 149.7 us per launch against 147.0: the softmax phase (5.5 of an item's 18 us) does
// not follow the instruction count; a wave-uniform branch per key tile for the key mask instead of a select per value: no change.
// What one wave's VALU phases and LDS waits cost, the other wave's MFMAs now cover.  Same integers as the four-wave form; the row sum of the
// probabilities is the sum of the pair's two partial sums (tiles 0..3, then the rest).
#pragma once
#include "attn_core_i8.h"

template <int KT>
static constexpr int attn_core8w_smem_bytes() { return 2 * KT * 4 * 1024 + 3 * 4 * KT * 1024; }  // K/V buffer 0 + the P image (over buffer 1 and beyond)

template <int KT, bool O8>
__global__ __launch_bounds__(512, 2) void attn_core_i8w_kernel(AttnCore8Args a) {
    static_assert(KT > 4 && KT <= 8, "the pair splits the key tiles 4 + (KT - 4)");
    constexpr int HALF = KT * 4 * 1024;  // bytes of one slice of half an image (4 of the 8 d blocks x KT tiles)
    constexpr int BUF = 2 * HALF;        // one buffer: both slices of a half image
    constexpr int NQB = (KT + 3) / 4;    // query blocks (4 tiles, one per wave pair) per (window, head)
    constexpr int NPIECE = KT;           // 1-KiB pieces of a half image (both slices) per wave: KT*4 blocks x 2 slices / 8 waves
    constexpr int PSL = 4 * KT * 1024;   // one slice of the P image: [4 query tiles][KT key blocks][1 KiB]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;          // [2 buffers][slice][half image]
    char* pimg = smem + BUF;  // [3 slices][4 query tiles][KT key blocks][1 KiB]: over buffer 1 (8 KT KiB) + 4 KT KiB behind it
    // static LDS objects: reads of them are provably disjoint from pending LDS-DMA destinations (see attn_core_i8.h)
    __shared__ float sk[KT * 32];  // key scales of K
    __shared__ float sv[KT * 32];  // key scales of V
    __shared__ float xmax[2][128];  // cross-pair exchanges: [key half][query of the block]
    __shared__ float xsum[2][128];
    __shared__ float xpmx[2][128];
    const int wave = wave_id_uniform();
    const int qt3 = wave & 3;
    int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int n_items = a.BH * NQB;
    int item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (item >= n_items) return;

    struct Front {
        int bh, qt, tile_active;
        float sq, skr, svr;
        int touch, touch2;
    };
    auto locate = [&](int it, Front& f) {
        f.bh = it / NQB;
        const int qt_raw = (it - f.bh * NQB) * 4 + qt3;
        f.tile_active = qt_raw < KT;
        f.qt = f.tile_active ? qt_raw : KT - 1;
    };
    auto k_rsrc = [&](int bh) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.k8 + (((size_t)bh * KT * 8) << 10)), 0, 0x7fffffff, 0x00020000); };
    auto v_rsrc = [&](int bh) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.v8 + (((size_t)bh * 8 * KT) << 10)), 0, 0x7fffffff, 0x00020000); };
    auto dma_k_piece = [&](__amdgpu_buffer_rsrc_t kr, int hh, int buf, int n) {
        const int pc = n * 8 + wave;
        const int sl = pc / (KT * 4), blk = pc - sl * KT * 4, kt = blk >> 2, i = blk & 3;
        const unsigned src = (unsigned)(sl * a.plane) + (unsigned)((kt * 8 + 4 * hh + i) << 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, (__attribute__((address_space(3))) void*)(kv + buf * BUF + sl * HALF + (blk << 10)), 16, lane * 16,
                                                 src, 0, 0);
    };
    auto dma_v_piece = [&](__amdgpu_buffer_rsrc_t vr, int hh, int buf, int n) {
        const int pc = n * 8 + wave;
        const int sl = pc / (KT * 4), blk = pc - sl * KT * 4;
        const unsigned src = (unsigned)(sl * a.plane) + (unsigned)((4 * hh * KT + blk) << 10);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (__attribute__((address_space(3))) void*)(kv + buf * BUF + sl * HALF + (blk << 10)), 16, lane * 16,
                                                 src, 0, 0);
    };
    i32x4 qs1[8], qs2[8];
    // the Q fragments of d_k blocks i0 .. i1 - 1 (both slices: 8 registers per block)
    constexpr int QPRE = 2;  // blocks requested across the item boundary (all eight do not fit the registers next to the S^T accumulators)
    auto load_q = [&](const Front& f, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int8_t* p = a.q8 + ((((size_t)f.bh * KT + f.qt) * 8 + i) << 10) + lane * 16;
            qs1[i] = *(const i32x4*)p;
            qs2[i] = *(const i32x4*)(p + a.plane);
        }
    };
    auto load_scales = [&](Front& f) {
        f.sq = a.sq[(size_t)f.bh * a.Lp + f.qt * 32 + col];
        const int ks = min((int)threadIdx.x, KT * 32 - 1);
        f.skr = a.sk[(size_t)f.bh * a.Lp + ks];
        f.svr = a.sv[(size_t)f.bh * a.Lp + ks];
        const int8_t* q = a.q8 + ((((size_t)f.bh * KT + f.qt) * 8) << 10) + lane * 128;
        f.touch = *(const int*)q;
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
        load_q(cur, 0, QPRE);
    }

    // one item for a wave of key half KH (compile-time: every fragment address below is then a constant offset from the buffers, which is
    // what lets hipcc keep LDS reads clear of the LDS-DMA still under way into the OTHER buffer); returns false after the last item
    auto run_item = [&](auto kh_c) -> bool {
        constexpr int KH = decltype(kh_c)::value;
        constexpr int T0 = 4 * KH, NT = KH ? KT - 4 : 4;  // this wave's key tiles T0 .. T0 + NT - 1
        constexpr int NU = (NT + 1) / 2, NS = 4 * NU;     // S^T units per d_k half: (d_k block, pair of key tiles)
        static_assert(NS >= NPIECE, "one V^T piece per S^T unit of the second d_k half");
        asm volatile("" : "+v"(lane));
        hf = lane >> 5;
        col = lane & 31;
        const int next = item + (int)gridDim.x;
        const bool has_next = next < n_items;  // workgroup-uniform
        const int bh = cur.bh, qt = cur.qt;
        const int b = bh / a.H, h = bh - b * a.H;
        const bool active = cur.tile_active && qt * 32 + col < a.Lr;
        const float sq = cur.sq;
        const __amdgpu_buffer_rsrc_t vr = v_rsrc(bh);
        const int qrow = qt3 * 32 + col;  // this lane's query within the block: the index of the cross-pair exchanges
        EG_DBG(unsigned long long* tr = a.trace ? a.trace + 131072 + (size_t)item * 8 : nullptr;)
        auto mark = [&](int i) {
            EG_DBG(if (tr && threadIdx.x == 0) {
                tr[i] = wall_clock64();
                if (i == 1 || i == 2) tr[5 + i] = __builtin_readcyclecounter();
            })
            (void)i;
        };
        mark(0);
        // both K halves, the scales and the first Q fragments of this item have been on their way since the previous item's PV phase /
        // epilogue (or the prologue above); the previous item's stores are in the same counter.  The other Q fragments are requested now
        // and arrive behind the first blocks' MFMAs.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        load_q(cur, QPRE, 8);
        if (cur.touch == 0x7fffffff && cur.touch2 == 0x7ffffffe) __builtin_amdgcn_s_sleep(1);  // (keeps the touching loads alive)
        if (threadIdx.x < KT * 32) {
            sk[threadIdx.x] = cur.skr;
            sv[threadIdx.x] = cur.svr;
        }
        __syncthreads();
        mark(1);

        // ---- S^T = K Q^T over the two d_k halves for this wave's key tiles
        i32x4 pa1[KT], pa2[KT], pa3[KT];  // the probabilities of the query tile, all key blocks, three slices (own blocks first, the partner's after the hand-over)
        float oscale;
        Front nxt = cur;
        {
            I8Acc s[NT + 1];  // (+1: the odd tile count's pair code names a tile that is never touched)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) acc_zero(s[kt]);
            i32x4 f[2][4];
            auto load_unit = [&](const char* img, int u, i32x4(&d)[4]) {
                const int i = u / NU, kl = 2 * (u % NU);
                const char* src = img + (((T0 + kl) * 4 + i) << 10) + lane * 16;
                d[0] = lds_frag(src);
                d[1] = lds_frag(src + HALF);
                if (kl + 1 < NT) {
                    d[2] = lds_frag(src + 4096);
                    d[3] = lds_frag(src + 4096 + HALF);
                }
            };
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const char* img = kv + hh * BUF;
                load_unit(img, 0, f[0]);
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    if (u + 1 < NS) load_unit(img, u + 1, f[(u + 1) & 1]);
                    if (hh == 1 && u < NPIECE) dma_v_piece(vr, 0, 0, u);  // the first V^T half into the buffer the barrier below freed
                    const int i = u / NU, kl = 2 * (u % NU);
                    const bool two = kl + 1 < NT;
                    const i32x4 q1 = qs1[4 * hh + i], q2 = qs2[4 * hh + i];
                    i32x4(&c)[4] = f[u & 1];
                    s[kl].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[1], q1, s[kl].m, 0, 0, 0);
                    if (two) s[kl + 1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[3], q1, s[kl + 1].m, 0, 0, 0);
                    s[kl].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], q1, s[kl].h, 0, 0, 0);
                    if (two) s[kl + 1].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], q1, s[kl + 1].h, 0, 0, 0);
                    s[kl].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], q2, s[kl].m, 0, 0, 0);
                    if (two) s[kl + 1].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[2], q2, s[kl + 1].m, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();  // every wave is done with this buffer (raw: the V^T pieces under way stay in flight)
            }
            mark(2);
            if (has_next) {
                locate(next, nxt);
                load_scales(nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- softmax over ALL keys of the query (TM:76-82): row maximum, row sum and the largest scaled probability across the pair
            float p[NT][16];
            float mx = -INFINITY;
            const float sq256 = sq * 256.0f * 1.44269504088896f;
            // (only the window's last key tile reaches beyond its keys: a wave-uniform BRANCH per tile — as a select per value the mask
            // costs three of the ~20 VALU instructions a probability takes)
            auto logits = [&](int kt, auto masked_c) {
                constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 k4 = *(const float4*)(sk + (T0 + kt) * 32 + 8 * gq + 4 * hf);
                    const float ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * gq + c;
                        float val = (float)i8_combine(s[kt].h[r], s[kt].m[r]) * (sq256 * ks[c]);
                        if (MASKED && (T0 + kt) * 32 + 8 * gq + 4 * hf + c >= a.L) val = -INFINITY;
                        p[kt][r] = val;
                        mx = fmaxf(mx, val);
                    }
                }
            };
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if ((T0 + kt) * 32 + 32 <= a.L) logits(kt, std::false_type{});
                else logits(kt, std::true_type{});
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (hf == 0) xmax[KH][qrow] = mx;
            __syncthreads();
            mx = fmaxf(xmax[0][qrow], xmax[1][qrow]);  // (key 0 always exists: finite)
            float sum = 0.f, pmax = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v4 = *(const float4*)(sv + (T0 + kt) * 32 + 8 * gq + 4 * hf);
                    const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * gq + c;
                        const float e = __builtin_amdgcn_exp2f(p[kt][r] - mx);
                        sum += e;
                        p[kt][r] = e * vs[c];  // the key's V scale rides on the probability
                        pmax = fmaxf(pmax, p[kt][r]);
                    }
                }
            sum += __shfl_xor(sum, 32);
            pmax = fmaxf(pmax, __shfl_xor(pmax, 32));
            if (hf == 0) {
                xsum[KH][qrow] = sum;
                xpmx[KH][qrow] = pmax;
            }
            __syncthreads();
            sum = xsum[0][qrow] + xsum[1][qrow];  // one order for both waves of the pair
            pmax = fmaxf(xpmx[0][qrow], xpmx[1][qrow]);
            const float pk = pmax > 1e-30f ? (P_QMAX - 0.25f) / pmax : 0.f;
            oscale = (1.0f / sum) * (pmax / (P_QMAX - 0.25f)) * 256.0f;
            // own key blocks: quantised into registers and into the P image for the partner
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                u32x4 s1, s2, s3;
                quant_p(p[kt], pk, s1, s2, s3);
                pa1[T0 + kt] = __builtin_bit_cast(i32x4, s1);
                pa2[T0 + kt] = __builtin_bit_cast(i32x4, s2);
                pa3[T0 + kt] = __builtin_bit_cast(i32x4, s3);
                char* dst = pimg + ((qt3 * KT + T0 + kt) << 10) + lane * 16;
                *(i32x4*)dst = pa1[T0 + kt];
                *(i32x4*)(dst + PSL) = pa2[T0 + kt];
                *(i32x4*)(dst + 2 * PSL) = pa3[T0 + kt];
            }
        }
        __syncthreads();
        {
            constexpr int O0 = KH ? 0 : 4, NO = KH ? 4 : KT - 4;  // the partner's key blocks
#pragma unroll
            for (int kt = 0; kt < NO; ++kt) {
                const char* src = pimg + ((qt3 * KT + O0 + kt) << 10) + lane * 16;
                pa1[O0 + kt] = lds_frag(src);
                pa2[O0 + kt] = lds_frag(src + PSL);
                pa3[O0 + kt] = lds_frag(src + 2 * PSL);
            }
        }
        __syncthreads();  // every wave holds its probabilities: the second V^T half may overwrite the image
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) dma_v_piece(vr, 1, 1, n);
        mark(3);

        // ---- O^T = V^T P (TM:83-88): per V^T half, this wave's two d_v tiles (2 KH, 2 KH + 1 of the half's four), one at a time
        const int m = b * a.Lr + qt * 32 + col;
        float t8[4][16];  // int8 output: this wave's four tiles wait for the row maximum over the head's 256 features
        float amax = 0.f;
        int mo = m;
        asm volatile("" : "+v"(mo));
        const __amdgpu_buffer_rsrc_t nkr = k_rsrc(nxt.bh);
#pragma unroll
        for (int dvh = 0; dvh < 2; ++dvh) {
            if (dvh == 1) {
                // the second V^T half has landed everywhere, and every wave is done with the first: buffer 0 takes the next item's first K half
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (has_next) {
#pragma unroll
                    for (int n = 0; n < NPIECE; ++n) dma_k_piece(nkr, 0, 0, n);
                }
            }
            const char* img = kv + dvh * BUF;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int dt = 2 * KH + tl;  // d_v tile within the half
                PVAcc o;
                acc_zero(o);
                i32x4 f[2][2];
                auto load_unit = [&](int kb, i32x4(&d)[2]) {
                    const char* src = img + ((dt * KT + kb) << 10) + lane * 16;
                    d[0] = lds_frag(src);
                    d[1] = lds_frag(src + HALF);
                };
                load_unit(0, f[0]);
#pragma unroll
                for (int kb = 0; kb < KT; ++kb) {
                    if (kb + 1 < KT) load_unit(kb + 1, f[(kb + 1) & 1]);
                    i32x4(&c)[2] = f[kb & 1];
                    o.m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[1], pa1[kb], o.m, 0, 0, 0);
                    o.h = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], pa1[kb], o.h, 0, 0, 0);
                    o.l = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], pa3[kb], o.l, 0, 0, 0);
                    o.m = __builtin_amdgcn_mfma_i32_32x32x32_i8(c[0], pa2[kb], o.m, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int tile = 4 * dvh + dt;  // feature tile of the head
                if (O8) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = pv_value(o.h[r], o.m[r], o.l[r]) * oscale;
                        t8[2 * dvh + tl][r] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                    asm volatile("" : "+v"(amax));  // (pins the conversion here: attn_core_i8.h)
                } else if (active) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = pv_value(o.h[8 * jj + e], o.m[8 * jj + e], o.l[8 * jj + e]) * oscale;
                        u32x4 hi, lo;
                        split8(t, hi, lo);
                        const size_t idx = acc_slot(mo, h * 256 + tile * 32, jj, hf, a.HD16);
                        *(u32x4*)(a.o + idx) = hi;
                        *(u32x4*)(a.o + a.o_plane + idx) = lo;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the second V^T half is spent in every wave: buffer 1 takes the next item's second K half
        __builtin_amdgcn_s_barrier();
        if (has_next) {
#pragma unroll
            for (int n = 0; n < NPIECE; ++n) dma_k_piece(nkr, 1, 1, n);
        }
        mark(4);
        // the probabilities are spent: their registers take the NEXT item's first Q fragments (their lines were touched into L2 before the
        // softmax), which then arrive behind the epilogue below instead of in front of the next S^T
        if (has_next) load_q(nxt, 0, QPRE);
        if (O8) {
            // one scale per row and head: the maximum over both waves of the pair (every wave takes part in the barrier)
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            if (hf == 0) xmax[KH][qrow] = amax;
            asm volatile("" ::: "memory");
            wait_lds();  // the LDS write above is done before the barrier (a RAW barrier: the next item's K pieces stay in flight)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            amax = fmaxf(xmax[0][qrow], xmax[1][qrow]);
            if (active) {
                int ml = m;
                asm volatile("" : "+v"(ml));
                const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
                if (KH == 0 && hf == 0) a.o_scale[(size_t)ml * a.H + h] = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int tile = 4 * (t4 >> 1) + 2 * KH + (t4 & 1);
                    u32x4 s1, s2;
                    quant16(t8[t4], inv, s1, s2);
                    const size_t idx = acc_slot_i8(ml, h * 256 + tile * 32, hf, a.HD16 / 2);
                    *(u32x4*)(a.o8 + idx) = s1;
                    *(u32x4*)(a.o8 + a.o8_plane + idx) = s2;
                }
            }
        }
        mark(5);
        if (!has_next) return false;
        item = next;
        cur = nxt;
        return true;
    };
    if (wave < 4) {
        while (run_item(std::integral_constant<int, 0>{})) {
        }
    } else {
        while (run_item(std::integral_constant<int, 1>{})) {
        }
    }
}
