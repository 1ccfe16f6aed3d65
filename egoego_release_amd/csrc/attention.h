// attention.h — full (unmasked) multi-head self-attention over one window, split-bf16 MFMA.
// (The i8x3 precision replaces this and the QKV projection by attn_layer_i8.h for windows of 65..128 tokens;
// this kernel serves the split-bf16 precision and every other window length.)
//
// Replaces TM:75-88 (bmm, /temperature, softmax, bmm, head merge) for use_full_attention=True.
// One workgroup (4 waves) per (batch, head, group of 4 query tiles); wave w owns 32 queries.
//
//   S^T[key][query] = K_frag (A, rows = keys) x Q_frag (B, cols = queries)   ("swapped" QK^T)
// puts a whole softmax row in ONE lane pair (lane, lane^32): the row max / sum are in-lane
// reductions plus one cross-half shuffle, and the probabilities never leave registers: the
// S^T accumulator registers 8jj..8jj+7 of key tile kt ARE the B-operand fragment of
//   O^T[d][query]  = V^T_frag (A, rows = d) x P_frag (B, cols = queries)
// for key group kg = 2*kt + jj, because the V projection's epilogue (gemm.h EpiV) already stored V
// transposed with the keys of each group of 16 in exactly that (half, slot) order.
// The attention matrix itself is never written (the reference returns and discards it, TM:95,223).
//
// K (16 / CH chunks of 16 CH of d_k) and then V^T (16 / CH chunks of 16 CH of d_v) stream through a double-buffered
// LDS ring as contiguous kilobyte fragments; Q fragments go global -> registers (each is used by one
// wave only).  32 / CH phases, one barrier each; the next phase's loads are in flight during the MFMAs.
// CH = 4: 64-wide chunks, 8 phases (<= 128 keys: two workgroups per CU; the fused T = 120 kernel).  CH = 2: 32-wide chunks, 16 phases —
// the long window (7 key tiles), whose 64-wide stages (2 x 56 KiB) left room for ONE four-wave workgroup per CU, one wave per SIMD with
// nothing to run in its waits; at 2 x 28 KiB two workgroups share a CU.  Same k order, same bits.
#pragma once
#include <type_traits>

#include "common.h"

#ifndef EGOEGO_ATTN_LIBM_SOFTMAX
#define EGOEGO_ATTN_LIBM_SOFTMAX 0  // (A/B knob of variant builds: 1 = libm expf and a division per probability, the form up to round 5)
#endif

struct AttnArgs {
    const __bf16* q;  // [B][H][L/32][16][2][32][8], pre-scaled by 1/temperature
    const __bf16* k;  // same layout
    const __bf16* v;  // [B][H][8][L/16][2][32][8], transposed + key-permuted
    size_t plane;     // elements between hi and lo planes of q/k/v
    __bf16* o;        // fragment-tiled [Mp][HD]
    size_t o_plane;
    int HD16;  // H*256/16
    int H;
    int L;     // valid keys (T + 1); keys >= L are padding and get probability 0
    int bh0;   // first (batch, head) pair of this launch (window-chunked launches)
};

// QREG: the Q fragments are not read from global memory but handed over in registers by the caller
// (the fused kernel computes the Q projection last and keeps it): qreg_h/qreg_l[2*i + jj] is the B-operand
// fragment of k-step 2*i + jj in accumulator order, the order in which K is stored.
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_phases(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_phases<N, F, I + 1>(f);
    }
}

template <int KT, int NP, bool QREG = false, int CH = 4>
__device__ __forceinline__ void attn_body(const AttnArgs& a, int bh, int qblock, char* smem,
                                          const bf16x8* qreg_h = nullptr, const bf16x8* qreg_l = nullptr) {
    static_assert((CH == 4 || CH == 2) && (KT * NP * CH) % 4 == 0, "chunk width");
    constexpr int NCH = KT * NP * CH / 4;            // 16-byte chunks per thread per phase
    constexpr int STAGE_BYTES = KT * NP * CH * 1024;  // K chunk: KT tiles x NP planes x CH k-steps x 1 KiB
    constexpr int NKP = 16 / CH;                      // K phases (then as many V^T phases)
    constexpr int DT = CH / 2;                        // d_v tiles per V^T phase
    constexpr int Lp = KT * 32;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int hf = lane >> 5, col = lane & 31;
    const int qt_raw = qblock * 4 + wave;
    const bool active = qt_raw < KT;
    const int qt = active ? qt_raw : KT - 1;

    auto stage_src = [&](int ph, int j) -> const u32x4* {
        const int blk = j * 4 + wave;
        if (ph < NKP) {
            const int ks = blk % CH, t2 = blk / CH;
            const int p = t2 / KT, kt = t2 % KT;
            return (const u32x4*)(a.k + (size_t)p * a.plane) + (((size_t)bh * KT + kt) * 16 + CH * ph + ks) * 64 + lane;
        } else {
            const int kg = blk % (2 * KT), t2 = blk / (2 * KT);
            const int p = t2 / DT, dt2 = t2 % DT;
            return (const u32x4*)(a.v + (size_t)p * a.plane) +
                   (((size_t)bh * 8 + DT * (ph - NKP) + dt2) * (2 * KT) + kg) * 64 + lane;
        }
    };
    auto q_src = [&](int dc, int ks, int p) -> const u32x4* {
        return (const u32x4*)(a.q + (size_t)p * a.plane) + (((size_t)bh * KT + qt) * 16 + CH * dc + ks) * 64 + lane;
    };

    f32x16 s[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
    bf16x8 phi[KT][2], plo[KT][2];

    u32x4 qb[2][CH][NP];  // Q fragments of the current / next d_k chunk (static ping-pong)
    // K / V^T chunks travel global -> LDS by LDS-DMA (16 B per lane at wave-uniform base + 16*lane: the
    // fragment image itself), one phase ahead of the MFMAs; two slots, so two workgroups fit per CU.
    // (buffer form, common.h: one resource per piece, based at this (window, head)'s image in the piece's PLANE — a plane of 8192 windows is
    // 2 GiB, beyond a 32-bit offset —; stage_off = stage_src's offset inside that image)
    auto stage_base = [&](int ph, int j) -> const __bf16* {
        const int blk = j * 4 + wave;
        if (ph < NKP) return a.k + (size_t)((blk / CH) / KT) * a.plane + (((size_t)bh * KT * 16) << 9);
        return a.v + (size_t)((blk / (2 * KT)) / DT) * a.plane + (((size_t)bh * 8 * 2 * KT) << 9);
    };
    auto stage_off = [&](int ph, int j) -> unsigned {
        const int blk = j * 4 + wave;
        if (ph < NKP) {
            const int ks = blk % CH, kt = (blk / CH) % KT;
            return (unsigned)((kt * 16 + CH * ph + ks) << 10);
        }
        const int kg = blk % (2 * KT), dt2 = (blk / (2 * KT)) % DT;
        return (unsigned)((((DT * (ph - NKP) + dt2) * (2 * KT)) + kg) << 10);
    };
    auto dma_phase = [&](int ph, int slot) {
        char* dst = smem + (size_t)slot * STAGE_BYTES + (size_t)wave * 1024;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if constexpr (EGOEGO_GEMM_BUFFER_DMA != 0)
                gemm_dma_piece(gemm_rsrc(stage_base(ph, j)), dst + (size_t)j * 4096, stage_off(ph, j), lane);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)stage_src(ph, j),
                                                 (__attribute__((address_space(3))) void*)(dst + (size_t)j * 4096), 16, 0, 0);
        }
    };
    dma_phase(0, 0);
    if constexpr (!QREG) {
#pragma unroll
        for (int ks = 0; ks < CH; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) qb[0][ks][p] = *q_src(0, ks, p);
    }

    auto phase = [&](auto PHC) {
        constexpr int ph = decltype(PHC)::value;
        constexpr int buf = ph & 1;
        // my share of this phase's chunk has landed; after the barrier everyone's has, and everyone is
        // done reading the other slot, which the next DMA may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (ph < 2 * NKP - 1) dma_phase(ph + 1, buf ^ 1);
        if constexpr (ph < NKP - 1 && !QREG) {
#pragma unroll
            for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) qb[(ph + 1) & 1][ks][p] = *q_src(ph + 1, ks, p);
        }
        __builtin_amdgcn_sched_barrier(0);  // loads stay above the MFMAs (hipcc would sink them to their use)
        const char* sb = smem + (size_t)buf * STAGE_BYTES + lane * 16;
        if constexpr (ph < NKP) {
            // S^T += K_chunk x Q_chunk^T
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                bf16x8 qh, ql;
                if constexpr (QREG) {
                    qh = qreg_h[CH * (ph < NKP ? ph : 0) + ks];
                    ql = qreg_l[CH * (ph < NKP ? ph : 0) + ks];
                } else {
                    qh = __builtin_bit_cast(bf16x8, qb[ph & 1][ks][0]);
                    ql = __builtin_bit_cast(bf16x8, qb[ph & 1][ks][NP - 1]);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const bf16x8 kh = *(const bf16x8*)(sb + ((0 * KT + kt) * CH + ks) * 1024);
                    if constexpr (NP == 2) {
                        const bf16x8 kl = *(const bf16x8*)(sb + ((1 * KT + kt) * CH + ks) * 1024);
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, s[kt], 0, 0, 0);
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, s[kt], 0, 0, 0);
                    }
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, s[kt], 0, 0, 0);
                }
            }
        }
        if constexpr (ph == NKP - 1) {
            // softmax over keys (TM:82); lane (col, hf) holds keys 32kt + 8(r>>2) + 4hf + (r&3)
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + mfma32_row(r, hf);
                    if (key >= a.L) s[kt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // e^(s - max) as one fma and one v_exp_f32 (relative error ~1e-6 for |s - max| < 30 — an eighth of the step of the
            // split-bf16 probabilities it feeds), and ONE reciprocal of the row sum instead of a division per probability: libm's
            // expf and IEEE division are ~20 instructions per probability, 2 x 112 of them per lane on the long window
            constexpr float LOG2E = 1.4426950408889634f;
            const float mxl = mx * LOG2E;
            (void)mxl;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#if EGOEGO_ATTN_LIBM_SOFTMAX
                    s[kt][r] = expf(s[kt][r] - mx);
#else
                    s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], LOG2E, -mxl));  // (masked keys: exp2(-inf) = 0)
#endif
                    sum += s[kt][r];
                }
            sum += __shfl_xor(sum, 32);
            const float inv_sum = 1.0f / sum;
            (void)inv_sum;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        __bf16 x, y;
#if EGOEGO_ATTN_LIBM_SOFTMAX
                        split_bf16(s[kt][8 * jj + e] / sum, x, y);
#else
                        split_bf16(s[kt][8 * jj + e] * inv_sum, x, y);
#endif
                        phi[kt][jj][e] = x;
                        plo[kt][jj][e] = y;
                    }
        }
        if constexpr (ph >= NKP) {
            // O^T[32 DT of d_v][32 queries] = V^T_chunk x P
            f32x16 o[DT];
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
            for (int kg = 0; kg < 2 * KT; ++kg) {
                const bf16x8 ph_ = phi[kg >> 1][kg & 1];
                const bf16x8 pl_ = plo[kg >> 1][kg & 1];
#pragma unroll
                for (int dt2 = 0; dt2 < DT; ++dt2) {
                    const bf16x8 vh = *(const bf16x8*)(sb + ((0 * DT + dt2) * (2 * KT) + kg) * 1024);
                    if constexpr (NP == 2) {
                        const bf16x8 vl = *(const bf16x8*)(sb + ((1 * DT + dt2) * (2 * KT) + kg) * 1024);
                        o[dt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph_, o[dt2], 0, 0, 0);
                        o[dt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl_, o[dt2], 0, 0, 0);
                    }
                    o[dt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph_, o[dt2], 0, 0, 0);
                }
            }
            if (active) {
                const int b = bh / a.H, h = bh % a.H;
                const int m = b * Lp + qt * 32 + col;
#pragma unroll
                for (int dt2 = 0; dt2 < DT; ++dt2)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float v[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = o[dt2][8 * jj + c];
                        u32x4 hi, lo;
                        split8(v, hi, lo);
                        const size_t idx = acc_slot(m, h * 256 + (DT * (ph - NKP) + dt2) * 32, jj, hf, a.HD16);
                        *(u32x4*)(a.o + idx) = hi;
                        if constexpr (NP == 2) *(u32x4*)(a.o + a.o_plane + idx) = lo;
                    }
            }
        }
    };
    static_phases<2 * NKP>(phase);
}

#ifndef EGOEGO_ATTN_CH4
#define EGOEGO_ATTN_CH4 0  // (A/B knob of variant builds: 1 = 64-wide chunks for the long window too, one workgroup per CU — the form up to round 5)
#endif
template <int KT, int NP> constexpr int attn_chunk() { return (KT <= 4 || EGOEGO_ATTN_CH4 || (KT * NP * 2) % 4 != 0) ? 4 : 2; }
template <int KT, int NP> constexpr int attn_smem() { return 2 * KT * NP * attn_chunk<KT, NP>() * 1024; }

template <int KT, int NP>
__global__ __launch_bounds__(256, (attn_smem<KT, NP>() <= 80 * 1024 && !(EGOEGO_ATTN_CH4 && KT > 4) ? 2 : 1)) void attn_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef EGOEGO_ATTN_NOREMAP
#define EGOEGO_ATTN_NOREMAP 0  // (A/B knob of variant builds: 1 = hardware block order, the query blocks of a (window, head) on different XCDs)
#endif
    // the query blocks of one (window, head) stream the same K / V^T images: consecutive remapped ids = one XCD, started together, so
    // the second read of a chunk comes out of that XCD's L2
    const int nqb = (int)gridDim.x;
    const int lid = EGOEGO_ATTN_NOREMAP ? (int)(blockIdx.y * nqb + blockIdx.x) : xcd_remap((int)(blockIdx.y * nqb + blockIdx.x), nqb * (int)gridDim.y);
    attn_body<KT, NP, false, attn_chunk<KT, NP>()>(a, lid / nqb + a.bh0, lid % nqb, smem);
}

// ---- The long window (KT = 7 key tiles, <= 224 tokens) on EIGHT waves: one workgroup per (window, head) ------------------------------
// attn_kernel<7> runs two four-wave workgroups per (window, head) — each streams the whole K and V^T image (2 x 448 KiB through L2 -> LDS
// per pair) with ONE chunk of prefetch distance, and waits for it in every one of its sixteen phases (round 5's ablations: HISTORY R5).
// Here wave w owns query tile w (wave 7 only helps with the loads), the images cross L2 -> LDS once, and the ring has FOUR 28-KiB slots:
// the chunk of phase p + 3 is requested at the start of phase p, so a wave waits with `vmcnt` counted for the chunks still allowed in
// flight (gemm.h's ring discipline: counted vmcnt + lgkmcnt(0), raw s_barrier, never __syncthreads()).  The Q fragments of the next d_k
// chunk come global -> VGPR by INLINE ASSEMBLY: hipcc drains vmcnt to 0 in front of the first use of an ordinary load's result while an
// LDS-DMA is in flight, which would undo the ring; the asm loads are ordered by hand — issued BEFORE the phase's LDS-DMA requests, so the
// wait that covers them leaves exactly those requests in flight (tools/check_untracked_loads.py verifies on the generated assembly that
// nothing touches a destination register of such a load before a vmcnt wait).  Same k order per accumulator as attn_body: same bits.
// What bounds it (timing-only ablations, B=256 x T=196, HISTORY R5): 200 us per launch as is, 188 without its MFMAs, 204 without its LDS
// fragment reads, 112 without its global loads — the kernel streams Q, K, V^T in and O out, 4 bytes per value, 0.9 GB per launch = 4.6 TB/s:
// HBM.  (At T = 120 the fused kernel keeps K, V^T and Q on the CU; a 224-key split-bf16 image pair does not fit a CU's LDS.)
EG_D u32x4 load16_untracked(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int KT, int NP>
__global__ __launch_bounds__(512, 1) void attn8_kernel(AttnArgs a) {
    static_assert(NP == 2 && KT >= 5 && KT <= 8, "the split-bf16 long window");
    constexpr int CH = 2, NKP = 16 / CH, NPH = 2 * NKP, NSLOT = 4;
    constexpr int NBLK = KT * NP * CH;        // 1-KiB fragment blocks per chunk (K: [plane][key tile][k-step], V^T: [plane][key group])
    constexpr int NCH = (NBLK + 7) / 8;       // LDS-DMA requests per wave and phase (the last block is requested more than once when 8 does not divide NBLK)
    constexpr int STAGE_BYTES = NBLK * 1024;
    constexpr int NQ = CH * NP;               // Q fragments per d_k chunk
    constexpr int NST = 2 * NP;               // output stores per V^T phase
    constexpr int Lp = KT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = (int)blockIdx.x + a.bh0;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const bool active = wave < KT;
    const int qt = active ? wave : KT - 1;

    auto stage_src = [&](int ph, int j) -> const u32x4* {
        const int blk = min(j * 8 + wave, NBLK - 1);
        if (ph < NKP) {
            const int ks = blk % CH, t2 = blk / CH;
            const int p = t2 / KT, kt = t2 % KT;
            return (const u32x4*)(a.k + (size_t)p * a.plane) + (((size_t)bh * KT + kt) * 16 + CH * ph + ks) * 64 + lane;
        } else {
            const int kg = blk % (2 * KT), p = blk / (2 * KT);
            return (const u32x4*)(a.v + (size_t)p * a.plane) + (((size_t)bh * 8 + (ph - NKP)) * (2 * KT) + kg) * 64 + lane;
        }
    };
    auto stage_base = [&](int ph, int j) -> const __bf16* {  // (buffer form, common.h) this (window, head)'s image in the piece's plane
        const int blk = min(j * 8 + wave, NBLK - 1);
        if (ph < NKP) return a.k + (size_t)((blk / CH) / KT) * a.plane + (((size_t)bh * KT * 16) << 9);
        return a.v + (size_t)(blk / (2 * KT)) * a.plane + (((size_t)bh * 8 * 2 * KT) << 9);
    };
    auto stage_off = [&](int ph, int j) -> unsigned {  // stage_src's byte offset inside that image
        const int blk = min(j * 8 + wave, NBLK - 1);
        if (ph < NKP) {
            const int ks = blk % CH, kt = (blk / CH) % KT;
            return (unsigned)((kt * 16 + CH * ph + ks) << 10);
        }
        return (unsigned)((((ph - NKP) * (2 * KT)) + blk % (2 * KT)) << 10);
    };
    auto dma_phase = [&](int ph) {
        char* dst = smem + (size_t)(ph % NSLOT) * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            char* d = dst + (size_t)min(j * 8 + wave, NBLK - 1) * 1024;
            if constexpr (EGOEGO_GEMM_BUFFER_DMA != 0)
                gemm_dma_piece(gemm_rsrc(stage_base(ph, j)), d, stage_off(ph, j), lane);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)stage_src(ph, j), (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        }
    };
    u32x4 qb[2][CH][NP];
    auto q_load = [&](int dc) {
#pragma unroll
        for (int ks = 0; ks < CH; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                qb[dc & 1][ks][p] = load16_untracked((const u32x4*)(a.q + (size_t)p * a.plane) + (((size_t)bh * KT + qt) * 16 + CH * dc + ks) * 64 + lane);
    };

    f32x16 s[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
    bf16x8 phi[KT][2], plo[KT][2];

    q_load(0);
    dma_phase(0);
    dma_phase(1);
    dma_phase(2);

    auto phase = [&](auto PHC) {
        constexpr int ph = decltype(PHC)::value;
        // What this wave issued AFTER the requests it needs now (its share of chunk ph; in a K phase also the Q fragments of chunk ph,
        // which were issued in front of the requests of phase ph - 1) may stay in flight.
        //   K phase ph >= 1: the requests of chunk ph + 2 (issued in phase ph - 1 behind the Q loads);  phase 0: chunks 1, 2
        //   V phase: the requests of chunks ph + 1, ph + 2 and the stores of the V phases among ph - 3 .. ph - 1 (none for wave 7)
        constexpr int d_after = (ph + 1 < NPH ? 1 : 0) + (ph + 2 < NPH ? 1 : 0);
        constexpr int st_after = (ph - 3 >= NKP ? 1 : 0) + (ph - 2 >= NKP ? 1 : 0) + (ph - 1 >= NKP ? 1 : 0);
        asm volatile("" ::: "memory");
        if constexpr (ph == 0) wait_counts<2 * NCH, 0>();
        else if constexpr (ph < NKP) wait_counts<NCH, 0>();
        else if constexpr (st_after == 0) wait_counts<d_after * NCH, 0>();
        else {
            if (active) wait_counts<d_after * NCH + st_after * NST, 0>();
            else wait_counts<d_after * NCH, 0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (ph < NKP) {
            // the Q fragments of this chunk are in their registers now: from here on the compiler may read them
#pragma unroll
            for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(qb[ph & 1][ks][p]));
        }
        if constexpr (ph + 1 < NKP) q_load(ph + 1);
        if constexpr (ph + 3 < NPH) dma_phase(ph + 3);
        __builtin_amdgcn_sched_barrier(0);
        const char* sb = smem + (size_t)(ph % NSLOT) * STAGE_BYTES + lane * 16;
        if constexpr (ph < NKP) {
            if (active) {
                // the K fragments of step i + 1 are requested before the MFMAs of step i and waited for behind them (gemm.h's discipline: an
                // LDS wait only ever finds reads that were issued a batch of MFMAs ago; on its own hipcc reads, waits and multiplies in turn,
                // one exposed LDS round trip per 96 MFMA clocks)
                bf16x8 fk[2][2];
                auto k_read = [&](int i) {
                    const int ks = i / KT, kt = i % KT;
                    fk[i & 1][0] = *(const bf16x8*)(sb + ((0 * KT + kt) * CH + ks) * 1024);
                    fk[i & 1][1] = *(const bf16x8*)(sb + ((1 * KT + kt) * CH + ks) * 1024);
                };
                k_read(0);
#pragma unroll
                for (int i = 0; i < CH * KT; ++i) {
                    const int ks = i / KT, kt = i % KT;
                    wait_lds();  // the fragments of step i (requested in front of the previous step's MFMAs) are here ...
                    if (i + 1 < CH * KT) k_read(i + 1);  // ... and those of step i + 1 travel behind this step's
                    const bf16x8 qh = __builtin_bit_cast(bf16x8, qb[ph & 1][ks][0]);
                    const bf16x8 ql = __builtin_bit_cast(bf16x8, qb[ph & 1][ks][1]);
                    __builtin_amdgcn_sched_barrier(0);  // (hipcc would sink the reads to their use and wait lgkmcnt(0) there)
                    const bf16x8 kh = fk[i & 1][0], kl = fk[i & 1][1];
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, s[kt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (ph == NKP - 1) {
            if (active) {  // softmax over keys (TM:82), as in attn_body
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * kt + mfma32_row(r, hf);
                        if (key >= a.L) s[kt][r] = -INFINITY;
                        mx = fmaxf(mx, s[kt][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                constexpr float LOG2E = 1.4426950408889634f;
                const float mxl = mx * LOG2E;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], LOG2E, -mxl));
                        sum += s[kt][r];
                    }
                sum += __shfl_xor(sum, 32);
                const float inv_sum = 1.0f / sum;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            __bf16 x, y;
                            split_bf16(s[kt][8 * jj + e] * inv_sum, x, y);
                            phi[kt][jj][e] = x;
                            plo[kt][jj][e] = y;
                        }
            }
        }
        if constexpr (ph >= NKP) {
            if (active) {
                // O^T[32 of d_v][32 queries] = V^T_chunk x P
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
                bf16x8 fv[2][2];
                auto v_read = [&](int kg) {
                    fv[kg & 1][0] = *(const bf16x8*)(sb + (0 * (2 * KT) + kg) * 1024);
                    fv[kg & 1][1] = *(const bf16x8*)(sb + (1 * (2 * KT) + kg) * 1024);
                };
                v_read(0);
#pragma unroll
                for (int kg = 0; kg < 2 * KT; ++kg) {
                    wait_lds();
                    if (kg + 1 < 2 * KT) v_read(kg + 1);
                    const bf16x8 ph_ = phi[kg >> 1][kg & 1];
                    const bf16x8 pl_ = plo[kg >> 1][kg & 1];
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 vh = fv[kg & 1][0], vl = fv[kg & 1][1];
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph_, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl_, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph_, o, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int b = bh / a.H, h = bh % a.H;
                const int m = b * Lp + qt * 32 + col;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = o[8 * jj + c];
                    u32x4 hi, lo;
                    split8(v, hi, lo);
                    const size_t idx = acc_slot(m, h * 256 + (ph - NKP) * 32, jj, hf, a.HD16);
                    *(u32x4*)(a.o + idx) = hi;
                    *(u32x4*)(a.o + a.o_plane + idx) = lo;
                }
            }
        }
    };
    static_phases<NPH>(phase);
}
