// gemm.h — MFMA GEMM over fragment-tiled operands (split-bf16, or int8 slices), with fused epilogues.
//
// Every Linear / Conv1d(k=1) of the denoiser (TM:45-47,55,102-103,179; M:102) is
//     out[token][feature] = sum_k act[token][k] * W[feature][k]  (+ bias, + fused tail)
// Both operands are K-contiguous, so both are stored fragment-tiled (common.h) and both are
// loaded with the same lane-linear 16-byte reads.  The MFMA is issued "swapped":
//     D[feature][token] = W_frag (A operand, rows = features) x act_frag (B operand, cols = tokens)
// so that in the accumulator a lane owns ONE token (lane & 31) and its registers walk the
// features in groups of 4 consecutive ones.  Consequences:
//   * LayerNorm's reduction over features is in-lane (+1 cross-half shuffle, + LDS across the
//     waves that split the feature dimension) — no 32-lane butterflies;
//   * epilogues store 16 contiguous bytes per lane (8 bf16 of one plane, or 16 int8 of one slice) straight into
//     the next kernel's fragment-tiled operand, whose K axis is kept in accumulator order (common.h);
//   * the V projection alone is issued un-swapped (ACT_ROWS) so that V lands transposed and
//     key-permuted exactly as the PV MFMA of attention.h consumes it.
//
// Pipeline (GemmBody::mainloop): an LDS ring of NSTAGE stages filled by LDS-DMA (global_load_lds_dwordx4, no VGPR
// staging), counted s_waitcnt vmcnt + one raw s_barrier per stage, fragments read from LDS one half-tile ahead of the
// MFMAs that consume them.  With NP == 2 each fragment pair costs three MFMAs (lo*hi, hi*lo, hi*hi); the int8-slice
// form (accumulator type I8Acc) issues s2*s1, s1*s2 into one int32 accumulator and s1*s1 into a second.
#pragma once
#include <type_traits>

#include "common.h"

#ifndef EGOEGO_ABLATE_MAINLOOP
#define EGOEGO_ABLATE_MAINLOOP 0
#endif
// Per-block timestamps and epilogue/attention ablation exist only in the perf-debug build
// (`python -m egoego_release_amd.build --perfdebug` -> libegoego_hip_perfdebug.so, used by tools/*_trace.py);
// the product library carries neither the fields nor the branches.
#ifdef EGOEGO_PERFDEBUG
#define EG_DBG(...) __VA_ARGS__
#else
#define EG_DBG(...)
#endif

struct GemmOperands {
    const __bf16* w;  // weights, fragment-tiled [N][K]; lo plane at w + w_plane
    size_t w_plane;
    const __bf16* a;  // activations, fragment-tiled [Mp][K]; lo plane at a + a_plane
    size_t a_plane;
    int K16;  // K / 16
    int nfb;  // feature blocks in the grid
    int ntb;  // token blocks in the grid
    int tblk0;  // first token block of this launch (window-chunked launches)
    EG_DBG(int ablate;                  // 2 = skip the epilogue, 4 = skip attention
           unsigned long long* trace;)  // [nblocks][4] = {t_start, t_mainloop_end, t_end, hw_id | xcc_id << 32} or nullptr
    int kcount;  // k-blocks to contract, starting at w / a (K16 stays the row stride of both operands); 0 = all K16
};

// Block-id -> (feature block, token block).  After xcd_remap every XCD owns a contiguous range of ids;
// inside it, ids walk groups of 8 token blocks x all feature blocks (token fastest), so the ~32 blocks
// resident on an XCD at any time form an 8-token-block x 4-feature-block patch whose operand tiles are
// shared through that XCD's L2 while the blocks stream K in near lockstep.
EG_D void grouped_map(int lid, int nfb, int ntb, int& fblk, int& tblk) {
    constexpr int GT = 8;
    const int per_group = GT * nfb;
    const int grp = lid / per_group;
    const int first_t = grp * GT;
    const int gsz = min(GT, ntb - first_t);
    const int r = lid - grp * per_group;
    tblk = first_t + r % gsz;
    fblk = r / gsz;
}

template <int FT_, int TT_, int NWF_, int NWT_, int KS_, int NP_, bool ACT_ROWS_, int MINW_ = 1, int NSTAGE_ = 2>
struct GemmCfg {
    static constexpr int MINW = MINW_;      // min waves per SIMD the register allocation must allow
    static constexpr int NSTAGE = NSTAGE_;  // LDS ring depth; NSTAGE-1 stages are in flight (LDS-DMA) during the MFMAs
    static constexpr int FT = FT_, TT = TT_, NWF = NWF_, NWT = NWT_, KS = KS_, NP = NP_;
    static constexpr bool ACT_ROWS = ACT_ROWS_;
    static constexpr int NW = NWF * NWT, NT = NW * 64;
    static constexpr int WT = NWF * FT;  // weight (feature) tiles per block
    static constexpr int AT = NWT * TT;  // activation (token) tiles per block
    static constexpr int BF = WT * 32, BT = AT * 32;
    static constexpr int NBLK = (WT + AT) * NP * KS;  // 1 KiB fragment blocks per stage
    static constexpr int NCH = (NBLK + NW - 1) / NW;  // 16-byte chunks per thread per stage
    // (NBLK % NW != 0: the waves that would run out of blocks stage the stage's LAST block once more — same bytes to the same
    // place — so that every wave has the same number of loads in flight, which is what the counted waits assume)
    static constexpr int STAGE_BYTES = NBLK * 1024;
    static constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && 2 * NCH < 64, "ring depth / vmcnt range");
};

// accumulate one (feature tile, token tile) pair for one k-step: bf16 -> fp32 accumulator, int8 -> two int32
EG_D void acc_zero(f32x16& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
}
EG_D void acc_zero(I8Acc& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c.h[r] = c.m[r] = 0;
}
// One of the NP == 2 ? 3 : 1 MFMAs of a product.  The main loop issues part p of ALL of a half's tiles before
// part p + 1, so two MFMAs on the same accumulator are never back to back.
template <int NP, bool ACT_ROWS>
EG_D void mma_part(int part, f32x16& c, bf16x8 wh, bf16x8 wl, bf16x8 ah, bf16x8 al) {
    if (NP == 1) {
        c = ACT_ROWS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, c, 0, 0, 0);
        return;
    }
    const bf16x8 w = part == 0 ? wl : wh;
    const bf16x8 a = part == 1 ? al : ah;
    c = ACT_ROWS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0);
}
template <int NP, bool ACT_ROWS>
EG_D void mma_part(int part, I8Acc& c, bf16x8 wh, bf16x8 wl, bf16x8 ah, bf16x8 al) {
    const i32x4 w = __builtin_bit_cast(i32x4, part == 0 ? wl : wh);
    const i32x4 a = __builtin_bit_cast(i32x4, part == 1 ? al : ah);
    i32x16& d = part == 2 ? c.h : c.m;
    d = ACT_ROWS ? __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, d, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, d, 0, 0, 0);
}

// int8 slices into ONE int32 accumulator, in two passes over K ("i8x3" where the register file has no room for the
// I8Acc pair): pass 1 (a configuration with NP == 1: only the high slices are staged) accumulates s1*s1; the caller
// shifts the sums left by 8; pass 2 (NP == 2, both slices staged) adds s2*s1 + s1*s2 on top.  Integer arithmetic is
// exact, so the result equals i8_combine(h, m) of the one-pass form bit for bit (same bound: K <= 512).
struct I8One {
    i32x16 v;
};
EG_D void acc_zero(I8One& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c.v[r] = 0;
}
template <int NP, bool ACT_ROWS>
EG_D void mma_part(int part, I8One& c, bf16x8 wh, bf16x8 wl, bf16x8 ah, bf16x8 al) {
    const i32x4 w = __builtin_bit_cast(i32x4, (NP == 2 && part == 0) ? wl : wh);
    const i32x4 a = __builtin_bit_cast(i32x4, (NP == 2 && part == 1) ? al : ah);
    c.v = ACT_ROWS ? __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, c.v, 0, 0, 0) : __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, c.v, 0, 0, 0);
}
// MFMAs per (feature tile, token tile, k-step)
template <class AccT, int NP> struct AccParts { static constexpr int N = NP == 2 ? 3 : 1; };
template <int NP> struct AccParts<I8One, NP> { static constexpr int N = NP == 2 ? 2 : 1; };

// Epilogues that offer run_block (EpiOut: the workgroup's x rows staged through LDS) declare BLOCK_ROWS
template <class E, class = void> struct epi_block_rows : std::false_type {};
template <class E> struct epi_block_rows<E, std::enable_if_t<E::BLOCK_ROWS>> : std::true_type {};

template <class C, class Epi>
struct GemmBody {
    // main loop only: on return acc[ft][tt] holds the pre-epilogue sums of this wave's tile.
    // AccT = f32x16 (bf16 operands) or I8Acc (int8 slices: g.K16 then counts 32-wide k blocks).
    // ZERO = false: accumulate on top of what acc holds (second pass of the I8One form)
    struct NoPre {
        EG_D void operator()() const {}
    };
    // pre(): called once, after the pipeline-fill LDS-DMA of the prologue has been requested and before it is waited for — work with a
    // memory round trip of its own (a kernel's parameter staging) then shares the fill's instead of preceding it
    // prefetch(): the pipeline-fill requests of mainloop<..., PREFETCHED = true> on the same (g, fblk, tblk, smem), issued EARLY — by every
    // wave, once the ring is idle (after the previous main loop's closing barrier) — so that the fill's round trip runs behind whatever
    // the caller does in between (an epilogue that touches neither the ring nor global memory: the counted waits of the main loop assume
    // that nothing else of this wave is in the vector-memory queue behind these requests).
    static __device__ void prefetch(const GemmOperands& g, int fblk, int tblk, char* smem) {
        constexpr int KS = C::KS, NP = C::NP, WT = C::WT, AT = C::AT, NW = C::NW, NCH = C::NCH, D = C::NSTAGE - 1;
        const int wave = wave_id_uniform();
        const int lane = threadIdx.x & 63;
        const int ns = (g.kcount ? g.kcount : g.K16) / KS;
        const u32x4* gp[NCH];
        int dsto[NCH];
        unsigned boff[NCH];       // (buffer form) byte offset of chunk j's block inside its operand PLANE: wave-uniform, lives in an SGPR
        const __bf16* bbase[NCH]; // (buffer form) that plane's base (weights) / this token block's first row tile in it (activations): wave-uniform
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int blk = min(j * NW + wave, C::NBLK - 1);
            dsto[j] = blk * 1024;
            const int ks = blk % KS, t2 = blk / KS;
            const u32x4* base;
            if (t2 < WT * NP) {
                const int p = t2 / WT, i = t2 % WT;
                base = (const u32x4*)(g.w + (size_t)p * g.w_plane) + ((size_t)(fblk * WT + i) * g.K16 + ks) * 64;
                boff[j] = (unsigned)(((fblk * WT + i) * g.K16 + ks) << 10);
                bbase[j] = g.w + (size_t)p * g.w_plane;
            } else {
                const int t3 = t2 - WT * NP;
                const int p = t3 / AT, i = t3 % AT;
                base = (const u32x4*)(g.a + (size_t)p * g.a_plane) + ((size_t)(tblk * AT + i) * g.K16 + ks) * 64;
                boff[j] = (unsigned)((i * g.K16 + ks) << 10);
                bbase[j] = g.a + (size_t)p * g.a_plane + ((size_t)tblk * AT * g.K16 << 9);
            }
            gp[j] = base + lane;
        }
        // (stage 0's requests before stage 1's: the main loop's first counted wait leaves exactly the newest stage in flight)
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < ns) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    if constexpr (EGOEGO_GEMM_BUFFER_DMA != 0) {
                        char* dst = smem + (size_t)d * C::STAGE_BYTES + dsto[j];
                        gemm_dma_piece(gemm_rsrc(bbase[j]), dst, boff[j] + (unsigned)(d * KS * 1024), lane);
                    } else
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp[j] + (size_t)d * KS * 64),
                                                         (__attribute__((address_space(3))) void*)(smem + (size_t)d * C::STAGE_BYTES + dsto[j]), 16, 0, 0);
                }
            }
    }
    template <class AccT, bool ZERO = true, class Pre = NoPre, bool PREFETCHED = false>
    static __device__ void mainloop(const GemmOperands& g, int fblk, int tblk, char* smem, AccT (&acc)[C::FT][C::TT], Pre pre = Pre{}) {
        constexpr int FT = C::FT, TT = C::TT, KS = C::KS, NP = C::NP, WT = C::WT, AT = C::AT, NW = C::NW;
        constexpr int NCH = C::NCH;
        const int wave = wave_id_uniform();
        const int lane = threadIdx.x & 63;
        const int wf = wave % C::NWF, wt = wave / C::NWF;
        EG_DBG(if (g.trace && threadIdx.x == 0) {
            g.trace[(size_t)blockIdx.x * 4 + 0] = wall_clock64();
            g.trace[65536 + (size_t)blockIdx.x * 2] = __builtin_readcyclecounter();
            g.trace[(size_t)blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                  ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        })

        // Per-thread source pointers of this stage's chunks (wave-uniform base + lane).
        const u32x4* gp[NCH];
        int dsto[NCH];  // byte offset of chunk j's block inside a stage
        unsigned boff[NCH];       // (buffer form) byte offset of chunk j's block inside its operand PLANE: wave-uniform, lives in an SGPR
        const __bf16* bbase[NCH]; // (buffer form) that plane's base (weights) / this token block's first row tile in it (activations): wave-uniform
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int blk = min(j * NW + wave, C::NBLK - 1);
            dsto[j] = blk * 1024;
            const int ks = blk % KS, t2 = blk / KS;
            const u32x4* base;
            if (t2 < WT * NP) {
                const int p = t2 / WT, i = t2 % WT;
                base = (const u32x4*)(g.w + (size_t)p * g.w_plane) + ((size_t)(fblk * WT + i) * g.K16 + ks) * 64;
                boff[j] = (unsigned)(((fblk * WT + i) * g.K16 + ks) << 10);
                bbase[j] = g.w + (size_t)p * g.w_plane;
            } else {
                const int t3 = t2 - WT * NP;
                const int p = t3 / AT, i = t3 % AT;
                base = (const u32x4*)(g.a + (size_t)p * g.a_plane) + ((size_t)(tblk * AT + i) * g.K16 + ks) * 64;
                boff[j] = (unsigned)((i * g.K16 + ks) << 10);
                bbase[j] = g.a + (size_t)p * g.a_plane + ((size_t)tblk * AT * g.K16 << 9);
            }
            gp[j] = base + lane;
        }
        // (buffer form: one resource per chunk, built from its plane's base at issue time — a few SALU operations — so that the 32-bit offsets
        // stay inside ONE plane of ONE token block / the weight matrix whatever the batch: a plane of 8192 windows' attention output is 2 GiB)

        if (ZERO) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) acc_zero(acc[i][j]);
        }

        const int ns = (g.kcount ? g.kcount : g.K16) / KS;
        // ---- software pipeline -------------------------------------------------------------------
        // Ring of NS LDS stages filled by LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight
        // into the fragment-tiled image at wave-uniform base + 16*lane; no VGPR staging, no ds_write).
        // Iteration s:  counted vmcnt (my share of stage s landed) -> raw s_barrier (everyone's landed,
        // everyone finished reading stage s-1) -> ds_read fragments of stage s into register set s&1
        // -> MFMAs of stage s-1 from the OTHER register set, with the DMA instructions of stage s+D
        // dropped between them.  So the matrix pipe has work the moment the barrier opens, LDS read
        // latency and DMA issue cost sit in the MFMA shadow, and D = NS-1 stages stay in flight.
        // (__syncthreads() is avoided on purpose: with a DMA in flight it would drain vmcnt to 0.)
        constexpr int NS = C::NSTAGE, D = NS - 1;
        constexpr int FH = FT / 2;  // feature tiles per half
        static_assert(FT % 2 == 0, "the pipeline alternates two halves of the wave's feature tiles");
        struct ActFr { bf16x8 h[TT], l[TT]; };
        struct WFr { bf16x8 h[FH], l[FH]; };
        // perf-debug, compile-time so the shipped loop carries no branches: -DEGOEGO_ABLATE_MAINLOOP=bits
        // (1 = no global->LDS loads, 8 = no LDS fragment reads, 16 = no waits / barriers, 32 = no MFMAs)
        constexpr bool loads_on = !(EGOEGO_ABLATE_MAINLOOP & 1), reads_on = !(EGOEGO_ABLATE_MAINLOOP & 8), sync_on = !(EGOEGO_ABLATE_MAINLOOP & 16);
        auto read_act = [&](int slot, int ks, ActFr& f) {
            if (!reads_on) return;
            const char* sb = smem + (size_t)slot * C::STAGE_BYTES + lane * 16;
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                f.h[j] = *(const bf16x8*)(sb + ((WT * NP + 0 * AT + wt * TT + j) * KS + ks) * 1024);
                if (NP == 2) f.l[j] = *(const bf16x8*)(sb + ((WT * NP + 1 * AT + wt * TT + j) * KS + ks) * 1024);
            }
        };
        auto read_w = [&](int slot, int ks, int half, WFr& f) {
            if (!reads_on) return;
            const char* sb = smem + (size_t)slot * C::STAGE_BYTES + lane * 16;
#pragma unroll
            for (int i = 0; i < FH; ++i) {
                f.h[i] = *(const bf16x8*)(sb + ((0 * WT + wf * FT + half * FH + i) * KS + ks) * 1024);
                if (NP == 2) f.l[i] = *(const bf16x8*)(sb + ((1 * WT + wf * FT + half * FH + i) * KS + ks) * 1024);
            }
        };
        auto issue_one = [&](int stage, int slot, int j) {
            char* dst = smem + (size_t)slot * C::STAGE_BYTES;
            if constexpr (EGOEGO_GEMM_BUFFER_DMA != 0) {
                gemm_dma_piece(gemm_rsrc(bbase[j]), dst + dsto[j], boff[j] + (unsigned)(stage * KS * 1024), lane);
            } else
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(gp[j] + (size_t)stage * KS * 64),
                    (__attribute__((address_space(3))) void*)(dst + dsto[j]), 16, 0, 0);
        };
        // MFMAs of one half (FH x TT accumulator triples, part-major).  DMA instruction q of (stage, slot) is issued
        // after MFMA q - q0, so the DMA issue cost is paid in the shadow of the matrix pipe.
        auto mfmas = [&](const ActFr& a, const WFr& w, int half, bool dma, int stage, int slot, int q0, auto&& after_first, auto&& reads) {
            constexpr int NPART = AccParts<AccT, NP>::N;
            constexpr int NMMA = NPART * FH * TT;
            constexpr int PER = (NCH + 2 * NMMA - 1) / (2 * NMMA);
#pragma unroll
            for (int part = 0; part < NPART; ++part)
#pragma unroll
                for (int i = 0; i < FH; ++i)
#pragma unroll
                    for (int j = 0; j < TT; ++j) {
                        if constexpr (!(EGOEGO_ABLATE_MAINLOOP & 32)) mma_part<NP, C::ACT_ROWS>(part, acc[half * FH + i][j], w.h[i], w.l[i], a.h[j], a.l[j]);
                        else asm volatile("" ::"v"(w.h[i]), "v"(w.l[i]), "v"(a.h[j]), "v"(a.l[j]));
                        const int n = (part * FH + i) * TT + j;
                        if (n == 0) after_first();
                        reads(n);
#pragma unroll
                        for (int q = 0; q < PER; ++q)
                            if (q0 + n * PER + q < NCH && dma) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_one(stage, slot, q0 + n * PER + q);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
        };
        constexpr int NMMA_HALF = AccParts<AccT, NP>::N * FH * TT;
        constexpr int DMA_PER_HALF = ((NCH + 2 * NMMA_HALF - 1) / (2 * NMMA_HALF)) * NMMA_HALF;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < ns && loads_on && !PREFETCHED) {
#pragma unroll
                for (int j = 0; j < NCH; ++j) issue_one(d, d, j);
            }
        auto wait_stage = [&](int st) {  // stages st+1 .. min(st+D-1, ns-1) may stay in flight
            if (!sync_on) return;
            asm volatile("" ::: "memory");
            const int ahead = min(D - 1, ns - 1 - st);
            // vmcnt: my share of stage st landed; lgkmcnt(0): my reads of the slot about to be refilled are done
            if (ahead >= 2) wait_counts<2 * NCH, 0>();
            else if (ahead == 1) wait_counts<NCH, 0>();
            else wait_counts<0, 0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");  // compiler-level fence: no LDS read moves above the barrier
        };
        // ---- software pipeline over k-steps kk = stage*KS + ks ------------------------------------
        // iteration kk:  [new stage: counted vmcnt + barrier]  read act(kk), W-half0(kk)
        //                MFMA half1(kk-1)  (+ the DMA instructions refilling the slot of the previous stage)
        //                read W-half1(kk)
        //                MFMA half0(kk)
        // Every batch of ds_reads is followed by 3*FH*TT MFMAs on operands that are ALREADY in registers,
        // so the matrix pipe has work the moment the barrier opens and LDS latency stays in its shadow.
        ActFr a0{}, a1{};
        WFr w0{}, w1{};
        const int nu = ns * KS;
        int slot = 0;   // slot of the stage being read
        int pslot = 0;  // slot of the previous stage (the one being refilled)
        pre();
        wait_stage(0);
        if (D < ns && loads_on) {  // stage D goes into the one slot the prologue left empty
#pragma unroll
            for (int j = 0; j < NCH; ++j) issue_one(D, D, j);
        }
        auto nothing = [] {};
        auto noreads = [](int) {};
        read_act(0, 0, a0);
        read_w(0, 0, 0, w0);
        read_w(0, 0, 1, w1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a0, w0, 0, false, 0, 0, 0, nothing, noreads);
        // LDS reads are only ever waited for with NOTHING younger in flight: a batch is issued, a full batch of
        // MFMAs on operands already in registers runs, and the wait (hipcc puts an lgkmcnt(0) before the first
        // use, or the explicit one below) then finds the data there.  W-half1 is therefore read right AFTER the
        // first MFMA of half 0, whose operand wait has just drained the counter.
        auto unit = [&](int kk, ActFr& acur, const ActFr& aprev) {
            const int st = kk / KS, ks = kk - st * KS;
            bool dma = false;
            if (ks == 0) {
                wait_stage(st);
                pslot = slot;
                if (++slot == NS) slot = 0;
                dma = (st - 1 + NS < ns) && loads_on;
            } else {
                wait_lds();  // W-half1 of the previous k-step (issued 11 MFMAs ago): tell hipcc it has landed
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // The k-step's first fragments (activations + first half of the weight tiles) go out two at a time behind the first
                // MFMAs of the previous half instead of as one burst in front of them: a lone wave per SIMD issues nothing else
                // while a burst of ds_reads goes out (measured at B=256: attention layer -2.6 %, tail -0.7 %; one or three per
                // MFMA, or spreading the second half's reads as well, measured no better).
                constexpr int NRD = NP * (TT + FH), PER = 2;
                const char* sb = smem + (size_t)slot * C::STAGE_BYTES + lane * 16;
                auto one_read = [&](int r) {
                    if (!reads_on) return;
                    if (r < NP * TT) {
                        const int p = r / TT, j = r - p * TT;
                        const bf16x8 v = *(const bf16x8*)(sb + ((WT * NP + p * AT + wt * TT + j) * KS + ks) * 1024);
                        if (p == 0) acur.h[j] = v; else acur.l[j] = v;
                    } else {
                        const int q = r - NP * TT, p = q / FH, i = q - p * FH;
                        const bf16x8 v = *(const bf16x8*)(sb + ((p * WT + wf * FT + i) * KS + ks) * 1024);
                        if (p == 0) w0.h[i] = v; else w0.l[i] = v;
                    }
                };
                mfmas(aprev, w1, 1, dma, st - 1 + NS, pslot, 0, nothing, [&](int n) {
                    if (PER * n < NRD) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < PER; ++e)
                            if (PER * n + e < NRD) one_read(PER * n + e);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(acur, w0, 0, dma, st - 1 + NS, pslot, DMA_PER_HALF, [&] {
                __builtin_amdgcn_sched_barrier(0);
                read_w(slot, ks, 1, w1);
                __builtin_amdgcn_sched_barrier(0);
            }, noreads);
            __builtin_amdgcn_sched_barrier(0);
        };
        int kk = 1;
        for (; kk + 1 < nu; kk += 2) {
            unit(kk, a1, a0);
            unit(kk + 1, a0, a1);
        }
        if (kk < nu) {
            unit(kk, a1, a0);
            mfmas(a1, w1, 1, false, 0, 0, 0, nothing, noreads);
        } else {
            mfmas(a0, w1, 1, false, 0, 0, 0, nothing, noreads);
        }
        __syncthreads();
        EG_DBG(if (g.trace && threadIdx.x == 0) {
            g.trace[(size_t)blockIdx.x * 4 + 1] = wall_clock64();
            g.trace[65536 + (size_t)blockIdx.x * 2 + 1] = __builtin_readcyclecounter();
        })
    }

    static __device__ void run(const GemmOperands& g, const Epi& epi, int fblk, int tblk, char* smem) {
        f32x16 acc[C::FT][C::TT];
        [[clang::always_inline]] mainloop(g, fblk, tblk, smem, acc);
        const int wave = wave_id_uniform();
        const int lane = threadIdx.x & 63;
        const int wf = wave % C::NWF, wt = wave / C::NWF;
        const int f0 = (fblk * C::WT + wf * C::FT) * 32;
        const int t0 = (tblk * C::AT + wt * C::TT) * 32;
        EG_DBG(if (g.ablate & 2) {
            if (acc[0][0][0] == 123.456f) *(float*)smem = acc[C::FT - 1][C::TT - 1][7];  // keep the MFMAs alive
            return;
        })
        if constexpr (epi_block_rows<Epi>::value && C::WT * 32 >= 256 && C::SMEM_BYTES >= C::BT * 1024 + 16) {
            epi.template run_block<C::FT, C::TT, C::BT, C::NT>(acc, f0, t0, tblk * C::BT, lane, smem);  // (the workgroup owns every output feature of its rows)
        } else {
            epi.template run<C::FT, C::TT>(acc, f0, t0, lane, wf, wt, smem);
        }
        EG_DBG(if (g.trace) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) g.trace[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
        })
    }
};

template <class C, class Epi>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm_kernel(GemmOperands g, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int fblk, tblk;
    grouped_map(lid, g.nfb, g.ntb, fblk, tblk);
    [[clang::always_inline]] GemmBody<C, Epi>::run(g, epi, fblk, tblk + g.tblk0, smem);
}

// Q/K feature blocks run swapped, V feature blocks un-swapped; the branch is block-uniform.
template <class CQK, class EpiQK, class CV, class EpiV>
__global__ __launch_bounds__(CQK::NT, CQK::MINW) void qkv_kernel(GemmOperands g, EpiQK eqk, EpiV ev, int n_qk_fblocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int fblk, tblk;
    grouped_map(lid, g.nfb, g.ntb, fblk, tblk);
    tblk += g.tblk0;
    if (fblk < n_qk_fblocks)
        GemmBody<CQK, EpiQK>::run(g, eqk, fblk, tblk, smem);
    else
        GemmBody<CV, EpiV>::run(g, ev, fblk, tblk, smem);
}

// ------------------------------------------------------------------------------------ int8-slice helpers
// "i8x3": both operands are integers q = rint(v / scale) with |q| <= 32639 (one scale per token row, one per
// weight row), stored as two int8 slices q = 256*s1 + s2.  A product costs three v_mfma_i32_32x32x32_i8
// (s1*s1 into one int32 accumulator, s1*s2 + s2*s1 into a second; s2*s2 is dropped like lo*lo in split-bf16),
// each covering K = 32: half the matrix-pipe time of the three bf16 MFMAs, on operands of half the bytes.

// 256 * H + M in one int32: |sum| <= K * (127*127*256 + 2*127*128), which fits for K <= 512 — every contraction
// here (d_model = 512, d_k = 256, <= 128 keys).  The result is the integer dot product / 256.
EG_D int i8_combine(int h, int m) { return (h << 8) + m; }

// int32 pair -> fp32, swapped accumulator (lane owns a token)
EG_D void i8_dequant(const I8Acc& q, f32x16& o, const float* sw8, float sa) {
    const float sa256 = sa * 256.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w4 = *(const float4*)(sw8 + 8 * g);  // features 8g + 4hf + (0..3); sw8 already includes 4*hf
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) o[4 * g + c] = (float)i8_combine(q.h[4 * g + c], q.m[4 * g + c]) * (sa256 * ws[c]);
    }
}
// the same from the one-accumulator form (its integer IS i8_combine(h, m)): same float operations, same bits
EG_D void i8_dequant(const I8One& q, f32x16& o, const float* sw8, float sa) {
    const float sa256 = sa * 256.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w4 = *(const float4*)(sw8 + 8 * g);
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) o[4 * g + c] = (float)q.v[4 * g + c] * (sa256 * ws[c]);
    }
}
// A whole wave tile: acc[i][j] = dequantised sums of feature tile i (weight row scales sw[f0 + 32 i ..]) and token tile j
// (activation row scales sa[t0 + 32 j + col]); what an fp32 epilogue (EpiResLN, EpiReluQ8) then takes.
// FENCE: for waves with 256 registers (two per SIMD) — see below; a 512-register wave lets hipcc batch all the loads.
template <bool FENCE, class QT, int FT, int TT>
EG_D void i8_dequant_tile(const QT (&q)[FT][TT], f32x16 (&acc)[FT][TT], const float* sw, const float* sa, int f0, int t0, int lane) {
    const int hf = lane >> 5, col = lane & 31;
    float s[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) s[j] = sa[t0 + j * 32 + col];
    // one feature tile at a time (its 16 weight scales are loaded once for all token tiles); the compiler-level fences keep
    // hipcc from hoisting every tile's scale loads to the top, which costs more registers than the wave has
#pragma unroll
    for (int i = 0; i < FT; ++i) {
        if (FENCE) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < TT; ++j) i8_dequant(q[i][j], acc[i][j], sw + f0 + i * 32 + 4 * hf, s[j]);
    }
    if (FENCE) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // ... and the epilogue's loads from moving up beside the still-live integer sums
    }
}
// The same over int8-slice operands (one scale per weight row, one per activation row): integer main loop, dequantised tile, fp32 epilogue.
template <class C, class Epi>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm_i8_kernel(GemmOperands g, const float* sw, const float* sa, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int fblk, tblk;
    grouped_map(lid, g.nfb, g.ntb, fblk, tblk);
    tblk += g.tblk0;
    const int wave = wave_id_uniform(), lane = threadIdx.x & 63;
    const int wf = wave % C::NWF, wt = wave / C::NWF;
    const int f0 = (fblk * C::WT + wf * C::FT) * 32, t0 = (tblk * C::AT + wt * C::TT) * 32;
    f32x16 acc[C::FT][C::TT];
    {
        I8Acc q[C::FT][C::TT];
        [[clang::always_inline]] GemmBody<C, Epi>::mainloop(g, fblk, tblk, smem, q);  // (with run_block's copy loops inlined below, hipcc would otherwise call it)
        i8_dequant_tile<true>(q, acc, sw, sa, f0, t0, lane);
    }
    if constexpr (epi_block_rows<Epi>::value && C::WT * 32 >= 256) {  // (the workgroup owns every output feature of its rows)
        epi.template run_block<C::FT, C::TT, C::BT, C::NT>(acc, f0, t0, tblk * C::BT, lane, smem);
    } else {
        epi.template run<C::FT, C::TT>(acc, f0, t0, lane, wf, wt, smem);
    }
    EG_DBG(if (g.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) g.trace[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
    })
}

// un-swapped accumulator (lane owns a feature, registers walk tokens)
EG_D void i8_dequant_rows(const I8Acc& q, f32x16& o, float sw, const float* sa8) {
    const float sw256 = sw * 256.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 a4 = *(const float4*)(sa8 + 8 * g);
        const float as[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) o[4 * g + c] = (float)i8_combine(q.h[4 * g + c], q.m[4 * g + c]) * (sw256 * as[c]);
    }
}

// ---- int8 contraction whose activation rows carry one scale per K CHUNK (the attention output: one scale per row and head,
// K = H x 256).  Each chunk is its own exact integer chain; chunk by chunk the sums are folded into an fp32 running sum,
//     run <- fma(float(I_chunk), 256 * s_act[row, chunk], run)      (chunks in ascending order, run = 0 before the first)
// and the weight row scale is applied once at the end.  Both tail kernels call exactly these two functions.
EG_D void i8_fold(const I8Acc& q, f32x16& run, float sa256) {
#pragma unroll
    for (int r = 0; r < 16; ++r) run[r] = __builtin_fmaf((float)i8_combine(q.h[r], q.m[r]), sa256, run[r]);
}
EG_D void i8_fold_finish(f32x16& run, const float* sw8) {  // sw8 = weight row scales + first feature of the tile + 4 * hf
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w4 = *(const float4*)(sw8 + 8 * g);
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) run[4 * g + c] *= ws[c];
    }
}

// =================================================================================== epilogues
// A lane's 16 bytes of (k-block kb = 32 features, token tile j) in an LDS-resident int8 operand: the chunk buffers of
// tail_fused.h's DirectGemm, [chunk = kb / 8][slice][token tile][kb % 8][1 KiB]; the second slice TT * 8 KiB further.
template <int TT>
EG_D char* lds_chunk_slot(char* base, int kb, int j, int lane) {
    return base + (kb >> 3) * (2 * TT * 8192) + ((j * 8 + (kb & 7)) << 10) + lane * 16;
}
// int8 rows of a wave tile (FT feature tiles x one token tile) held in registers: the all-int8 tail keeps LayerNorm-1's rows
// as LayerNorm-2's residual this way
template <int FT>
struct Rows8 {
    u32x4 s1[FT], s2[FT];
    float scale;
};
struct NoRows {};
// Swapped accumulator geometry used below: for acc[ft][tt][r] of lane l (hf = l >> 5, col = l & 31)
//   token   m = t0 + tt*32 + col
//   feature f = f0 + ft*32 + 8*(r>>2) + 4*hf + (r&3)          (4 consecutive features per r>>2)

// bias (+ReLU) -> fragment-tiled split-bf16 [Mp][N].  FFN first conv (TM:111 inner).
template <bool RELU, int NP>
struct EpiTiled {
    const float* bias;
    __bf16* out;
    size_t out_plane;
    int N16;
    template <int FT, int TT>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int, int, char*) const {
        const int hf = lane >> 5, col = lane & 31;
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int fb = f0 + i * 32 + 16 * jj + 4 * hf;  // features fb..fb+3 and fb+8..fb+11
                const float4 b0 = *(const float4*)(bias + fb);
                const float4 b1 = *(const float4*)(bias + fb + 8);
#pragma unroll
                for (int j = 0; j < TT; ++j) {
                    const int m = t0 + j * 32 + col;
                    float v[8] = {acc[i][j][8 * jj + 0] + b0.x, acc[i][j][8 * jj + 1] + b0.y, acc[i][j][8 * jj + 2] + b0.z,
                                  acc[i][j][8 * jj + 3] + b0.w, acc[i][j][8 * jj + 4] + b1.x, acc[i][j][8 * jj + 5] + b1.y,
                                  acc[i][j][8 * jj + 6] + b1.z, acc[i][j][8 * jj + 7] + b1.w};
                    if (RELU) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
                    }
                    u32x4 hi, lo;
                    split8(v, hi, lo);
                    const size_t idx = acc_slot(m, f0 + i * 32, jj, hf, N16);
                    *(u32x4*)(out + idx) = hi;
                    if (NP == 2) *(u32x4*)(out + out_plane + idx) = lo;
                }
            }
    }
};

// bias + ReLU -> the rows as int8 slices with one scale per row (i8x3 FFN: the hidden activations of TM:111 are the
// int8 operand of the second conv).  The block must span all 512 hidden features (NWF * FT * 32 == 512): the row maximum
// is taken in-lane over the wave's tiles, across the two halves by a shuffle and across the NWF waves through LDS.
// lds_q8 != nullptr (the LDS-resident tail, tail_fused.h): the rows go to LDS instead, as the operand chunks of the next contraction
// ([chunk = k-block / 8][slice][token tile][k-block % 8][1 KiB], token tile j of the block = token tile j of the wave tile), and
// the row scales to lds_scale[token of the block]; q8 / q8_scale may then be null.
template <int NWF, int BT>
struct EpiReluQ8 {
    const float* bias;   // [512]
    int8_t* q8;          // fragment-tiled int8 rows [Mp][512], K in acc32 order; second slice at + q8_plane
    size_t q8_plane;
    float* q8_scale;     // [Mp]
    char* lds_q8;
    float* lds_scale;
    int tok_base;        // added to t0 for the MEMORY rows only (the LDS-resident tail calls with block-local t0 = 0; its debug tap also stores to memory)
    // q: the integer sums of an int8-slice contraction (I8Acc or I8One); sw / sa: weight-row and activation-row scales.
    // LEAN (256-register waves, two per SIMD): two sweeps over the INTEGER sums, one tile at a time (a tile's 16 scales and
    // 16 biases in registers; the fences keep hipcc from hoisting every tile's loads to the top): the first only takes the
    // row maximum of relu(dequant + bias), the second recomputes those values and quantises them straight into the
    // stores — no fp32 copy of the wave tile ever exists, which is what lets such a wave (128 registers of integer sums)
    // run this without spilling.  !LEAN (512-register waves, one per SIMD, every load latency exposed): one sweep, the
    // values kept, all parameter loads in flight at once.  Same float operations per value: same bits.
    template <bool LEAN, class QT, int FT, int TT>
    __device__ void run(const QT (&q)[FT][TT], const float* sw, const float* sa, int f0, int t0, int lane, int wf, int wt, char* smem) const {
        static_assert(NWF * FT * 32 == 512, "the row maximum needs the whole 512-wide row in the block");
        const int hf = lane >> 5, col = lane & 31;
        float* red = (float*)smem;  // [TT][NWF][BT]
        float amax[TT], s[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            amax[j] = 0.f;
            s[j] = sa[t0 + j * 32 + col];
        }
        auto tile = [&](int i, int j, f32x16& v) {  // relu(dequant + bias) of one tile
            i8_dequant(q[i][j], v, sw + f0 + i * 32 + 4 * hf, s[j]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *(const float4*)(bias + f0 + i * 32 + 8 * g + 4 * hf);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 * g + c] = fmaxf(v[4 * g + c] + bb[c], 0.f);
            }
        };
        f32x16 keep[LEAN ? 1 : FT][LEAN ? 1 : TT];
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                if (LEAN) {
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                f32x16 v;
                tile(i, j, v);
#pragma unroll
                for (int r = 0; r < 16; ++r) amax[j] = fmaxf(amax[j], v[r]);
                if (!LEAN) keep[LEAN ? 0 : i][LEAN ? 0 : j] = v;
            }
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
            if (hf == 0) red[(j * NWF + wf) * BT + (wt * TT + j) * 32 + col] = amax[j];
        }
        __syncthreads();
        float inv[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            float rmax = 0.f;
#pragma unroll
            for (int w = 0; w < NWF; ++w) rmax = fmaxf(rmax, red[(j * NWF + w) * BT + (wt * TT + j) * 32 + col]);
            inv[j] = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) {
                const float sc = rmax > 0.f ? rmax / I8_QMAX : 0.f;
                if (q8_scale) q8_scale[tok_base + t0 + j * 32 + col] = sc;
                if (lds_scale) lds_scale[j * 32 + col] = sc;
            }
        }
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                f32x16 v;
                if (LEAN) {
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    tile(i, j, v);
                } else {
                    v = keep[LEAN ? 0 : i][LEAN ? 0 : j];
                }
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[r];
                u32x4 s1, s2;
                quant16(t, inv[j], s1, s2);
                if (q8) {
                    const size_t idx = acc_slot_i8(tok_base + t0 + j * 32 + col, f0 + i * 32, hf, 16);
                    *(u32x4*)(q8 + idx) = s1;
                    *(u32x4*)(q8 + q8_plane + idx) = s2;
                }
                if (lds_q8) {
                    char* d = lds_chunk_slot<TT>(lds_q8, (f0 >> 5) + i, j, lane);
                    *(u32x4*)d = s1;
                    *(u32x4*)(d + TT * 8192) = s2;
                }
            }
    }
};

// Q and K projections (TM:71-72): bias, Q pre-scaled by 1/temperature (TM:52,76), written per
// (batch, head) as the attention kernel's fragment-tiled operands [b][h][L/32][dk/16][2][32][8].
template <int NP>
struct EpiQK {
    const float* bias;  // [3*HD]
    __bf16* q;
    __bf16* k;
    size_t plane;
    float qscale;
    int Lp, H, HD, Mvalid;
    template <int FT, int TT>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int, int, char*) const {
        const int hf = lane >> 5, col = lane & 31;
        const int LT = Lp >> 5;
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int m0 = t0 + j * 32;
            if (m0 >= Mvalid) continue;
            const int b = m0 / Lp, lt = (m0 % Lp) >> 5;
#pragma unroll
            for (int i = 0; i < FT; ++i) {
                const int fb = f0 + i * 32;
                const int which = fb / HD, fh = fb % HD;
                const int h = fh >> 8, d0 = fh & 255;
                __bf16* dst = which ? k : q;
                const float sc = which ? 1.0f : qscale;
                const size_t blk0 = ((size_t)(b * H + h) * LT + lt) * 16;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float4 b0 = *(const float4*)(bias + fb + 16 * jj + 4 * hf);
                    const float4 b1 = *(const float4*)(bias + fb + 16 * jj + 4 * hf + 8);
                    float v[8] = {(acc[i][j][8 * jj + 0] + b0.x) * sc, (acc[i][j][8 * jj + 1] + b0.y) * sc,
                                  (acc[i][j][8 * jj + 2] + b0.z) * sc, (acc[i][j][8 * jj + 3] + b0.w) * sc,
                                  (acc[i][j][8 * jj + 4] + b1.x) * sc, (acc[i][j][8 * jj + 5] + b1.y) * sc,
                                  (acc[i][j][8 * jj + 6] + b1.z) * sc, (acc[i][j][8 * jj + 7] + b1.w) * sc};
                    u32x4 hi, lo;
                    split8(v, hi, lo);
                    // d_k kept in accumulator order (common.h swap23): one 16-byte store per lane
                    const size_t idx = (((blk0 + (size_t)((d0 >> 4) + jj)) * 2 + (size_t)hf) << 8) + col * 8;
                    *(u32x4*)(dst + idx) = hi;
                    if (NP == 2) *(u32x4*)(dst + plane + idx) = lo;
                }
            }
        }
    }
};

// V projection (TM:73), un-swapped accumulator: lane owns feature (col), registers walk tokens.
// Written transposed per (batch, head): [b][h][dv/32][L/16][2][32][8] where inside each group of 16
// keys, key = 8a + 4b + c sits at slot (hfk = b, e = 4a + c) — the order in which the PV MFMA's
// B operand (the softmax probabilities, straight out of the S^T accumulator) presents its keys.
template <int NP>
struct EpiV {
    const float* bias;  // [3*HD]
    __bf16* v;
    size_t plane;
    int Lp, H, HD, Mvalid;
    template <int FT, int TT>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int, int, char*) const {
        const int hf = lane >> 5, col = lane & 31;
        const int KG = Lp >> 4;
#pragma unroll
        for (int i = 0; i < FT; ++i) {
            const int f = f0 + i * 32 + col;
            const int fv = f - 2 * HD;
            const int h = fv >> 8, dt = (fv & 255) >> 5;
            const float bf = bias[f];
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                const int m0 = t0 + j * 32;
                if (m0 >= Mvalid) continue;
                const int b = m0 / Lp, lt = (m0 % Lp) >> 5;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        __bf16 x, y;
                        split_bf16(acc[i][j][8 * jj + e] + bf, x, y);
                        hi[e] = x;
                        lo[e] = y;
                    }
                    const size_t idx = ((((size_t)((b * H + h) * 8 + dt) * KG + (2 * lt + jj)) * 2 + hf) << 8) + col * 8;
                    *(uint4*)(v + idx) = __builtin_bit_cast(uint4, hi);
                    if (NP == 2) *(uint4*)(v + plane + idx) = __builtin_bit_cast(uint4, lo);
                }
            }
        }
    }
};

// bias + residual + LayerNorm(512) (+ padding-mask row multiply) -> fragment-tiled split-bf16.
// Output projection of attention (TM:92-93, 135) and second FFN conv (TM:111-114, 139).
// The block must span all 512 features (NWF * FT * 32 == 512).
// Summation order of the two row reductions (sum, centred squares), the same for every tiling so that a row's bits do not depend on
// the kernel that computes it: one partial per PAIR of consecutive feature tiles (64 features: the 32 values of a lane in tile, register
// order, + the other half-wave's), the 8 pair partials of a row combined as ((q0 + q1) + q2) + q3 with q_w = pair 2w + pair 2w+1.
// A wave holding 4 tiles (NWF = 4) owns one q, a wave holding 2 (NWF = 8: the eight-wave small-grid tail) one pair.
template <int NP, int NWF, int BT>
struct EpiResLN {
    static_assert(NWF == 4 || NWF == 8, "4 waves of 4 feature tiles or 8 waves of 2");
    // the row total from the per-wave partials in LDS
    static EG_D float combine(const float* red, int slot) {
        float q[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) q[w] = NWF == 4 ? red[slot + w * BT] : red[slot + 2 * w * BT] + red[slot + (2 * w + 1) * BT];
        return ((q[0] + q[1]) + q[2]) + q[3];
    }
    const float* bias;
    const __bf16* res;  // fragment-tiled [Mp][512]
    size_t res_plane;
    const float* gamma;
    const float* beta;
    const float* row_mask;  // [Mp] or nullptr
    __bf16* out;
    size_t out_plane;
    float eps;
    int8_t* q8;       // optional: the same rows as int8 slices (fragment-tiled, accumulator order) for an i8x3 consumer
    size_t q8_plane;  // bytes between the two slices
    float* q8_scale;  // [Mp] row scales
    // optional: the residual as int8 rows (the same layout: a lane's 16 accumulator values of a tile are 16 consecutive bytes per
    // slice) instead of `res` — precision 9 keeps every inter-kernel activation of a layer as int8 rows only; `out` may then be null
    const int8_t* res8;
    size_t res8_plane;
    const float* res8_scale;
    // optional (the LDS-resident tail): the int8 rows ALSO / ONLY into LDS as the next contraction's operand chunks (see EpiReluQ8)
    char* lds_q8;
    float* lds_scale;
    // optional: outlier monitor (common.h StepState::ln_max) — the largest row maximum this launch quantises, one atomicMax per workgroup
    unsigned* outlier;
    float* outlier_park;  // ... or an LDS slot that receives the maximum instead (mid-kernel epilogues: an atomic in the vector-memory queue
                          // would sit in front of the next contraction's counted waits; the kernel flushes the slot at its end)
    int outlier_rows;  // token tiles reaching beyond this row hold the padding of the last token block (uninitialised inputs): not recorded
    // rr: the residual as int8 rows in registers (instead of res / res8); keep: receives the int8 rows this call produces
    template <int FT, int TT, class ResR = NoRows, class KeepR = NoRows>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int wf, int wt, char* smem, const ResR* rr = nullptr,
                        KeepR* keep = nullptr) const {
        constexpr bool RES_REG = !std::is_same<ResR, NoRows>::value, KEEP = !std::is_same<KeepR, NoRows>::value;
        static_assert((!RES_REG && !KEEP) || TT == 1, "register rows: one token tile");
        static_assert(NWF * FT * 32 == 512 && (FT == 2 || FT == 4), "LayerNorm epilogue needs the whole 512-wide row in the block");
        const int hf = lane >> 5, col = lane & 31;
        float* red1 = (float*)smem;     // [TT][NWF][BT]  per-wave partial sums
        float* red2 = red1 + TT * NWF * BT;
        // A token is one lane (col) of one token tile j.  The three reductions over the 512 features (sum, centred
        // squares, |max| for the int8 side output) are each done for ALL token tiles at once: one LDS exchange and
        // one barrier per reduction, whatever TT is.
        float* red3 = red2 + TT * NWF * BT;
        int slot[TT];
        float mean[TT], rstd[TT], amax[TT];
        // ---- y = acc + bias + residual, row sums
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int m = t0 + j * 32 + col;
            slot[j] = (j * NWF) * BT + (wt * TT + j) * 32 + col;
            float s1 = 0.f, pp[FT / 2];
            if (RES_REG || res8) {
                float rs;
                if constexpr (RES_REG) rs = rr->scale; else rs = res8_scale[m];
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    float r[16];
                    if constexpr (RES_REG) {
                        dequant16(rr->s1[i], rr->s2[i], rs, r);
                    } else {
                        const size_t idx8 = acc_slot_i8(m, f0 + i * 32, hf, 16);
                        dequant16(*(const u32x4*)(res8 + idx8), *(const u32x4*)(res8 + res8_plane + idx8), rs, r);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *(const float4*)(bias + f0 + i * 32 + 8 * g + 4 * hf);
                        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[i][j][4 * g + c] += bb[c] + r[4 * g + c];
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int fb = f0 + i * 32 + 16 * jj + 4 * hf;
                    const float4 b0 = *(const float4*)(bias + fb);
                    const float4 b1 = *(const float4*)(bias + fb + 8);
                    const size_t idx = acc_slot(m, f0 + i * 32, jj, hf, 32);
                    float r[8];
                    const u32x4 rh = *(const u32x4*)(res + idx);
                    if (NP == 2) {
                        const u32x4 rl = *(const u32x4*)(res + res_plane + idx);
                        unpack8(rh, rl, r);
                    } else {
                        unpack8_hi(rh, r);
                    }
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[i][j][8 * jj + c] += bb[c] + r[c];
                }
#pragma unroll
            for (int i2 = 0; i2 < FT / 2; ++i2) {
                float p = 0.f;
#pragma unroll
                for (int i = 2 * i2; i < 2 * i2 + 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) p += acc[i][j][r];
                pp[i2] = p + __shfl_xor(p, 32);
            }
            s1 = pp[0];
            if (FT == 4) s1 += pp[FT / 2 - 1];
            if (hf == 0) red1[slot[j] + wf * BT] = s1;
        }
        __syncthreads();
        // ---- biased variance of the centred values (two-pass, like the reference's LayerNorm)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            mean[j] = combine(red1, slot[j]) * (1.0f / 512.0f);
            float pp[FT / 2];
#pragma unroll
            for (int i2 = 0; i2 < FT / 2; ++i2) {
                float p = 0.f;
#pragma unroll
                for (int i = 2 * i2; i < 2 * i2 + 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[i][j][r] - mean[j];
                        p += d * d;
                    }
                pp[i2] = p + __shfl_xor(p, 32);
            }
            float s2 = pp[0];
            if (FT == 4) s2 += pp[FT / 2 - 1];
            if (hf == 0) red2[slot[j] + wf * BT] = s2;
        }
        __syncthreads();
        // ---- normalise, scale/shift, padding mask, store
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int m = t0 + j * 32 + col;
            const float var = combine(red2, slot[j]);
            rstd[j] = 1.0f / sqrtf(var * (1.0f / 512.0f) + eps);
            const float mk = row_mask ? row_mask[m] : 1.0f;
            amax[j] = 0.f;
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int fb = f0 + i * 32 + 16 * jj + 4 * hf;
                    const float4 g0 = *(const float4*)(gamma + fb), g1 = *(const float4*)(gamma + fb + 8);
                    const float4 e0 = *(const float4*)(beta + fb), e1 = *(const float4*)(beta + fb + 8);
                    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    const float be[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        v[c] = ((acc[i][j][8 * jj + c] - mean[j]) * rstd[j] * ga[c] + be[c]) * mk;
                        acc[i][j][8 * jj + c] = v[c];
                        amax[j] = fmaxf(amax[j], fabsf(v[c]));
                    }
                    if (out) {
                        u32x4 hi, lo;
                        split8(v, hi, lo);
                        const size_t idx = acc_slot(m, f0 + i * 32, jj, hf, 32);
                        *(u32x4*)(out + idx) = hi;
                        if (NP == 2) *(u32x4*)(out + out_plane + idx) = lo;
                    }
                }
            if (q8 || lds_q8) {
                amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
                if (hf == 0) red3[slot[j] + wf * BT] = amax[j];
            }
        }
        if (q8 || lds_q8) {
            // row maximum over the 512 features -> one scale per token -> two int8 slices per value
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                const int m = t0 + j * 32 + col;
                float rmax = 0.f;
#pragma unroll
                for (int w = 0; w < NWF; ++w) rmax = fmaxf(rmax, red3[slot[j] + w * BT]);
                const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
                const float sc = rmax > 0.f ? rmax / I8_QMAX : 0.f;
                if (wf == 0 && hf == 0) {
                    if (q8_scale) q8_scale[m] = sc;
                    if (lds_scale) lds_scale[(wt * TT + j) * 32 + col] = sc;
                }
                if constexpr (KEEP) keep->scale = sc;
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                    u32x4 s1, s2;
                    quant16(v, inv, s1, s2);
                    if (q8) {
                        const size_t idx = acc_slot_i8(m, f0 + i * 32, hf, 16);
                        *(u32x4*)(q8 + idx) = s1;
                        *(u32x4*)(q8 + q8_plane + idx) = s2;
                    }
                    if (lds_q8) {
                        char* d = lds_chunk_slot<TT>(lds_q8, (f0 >> 5) + i, j, lane);
                        *(u32x4*)d = s1;
                        *(u32x4*)(d + TT * 8192) = s2;
                    }
                    if constexpr (KEEP) {
                        keep->s1[i] = s1;
                        keep->s2[i] = s2;
                    }
                }
            }
#ifndef EGOEGO_NO_MONITOR
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);  // (after the stores: the monitor must not lengthen any live range above)
            if ((outlier || outlier_park) && wf == 0) {
                // the row maxima once more from LDS (nothing is kept live across the quantisation above; red3 stays intact until
                // the next barrier); both half-waves hold the same 32 tokens
                float omax = 0.f;
#pragma unroll
                for (int j = 0; j < TT; ++j) {
                    float rmax = 0.f;
#pragma unroll
                    for (int w = 0; w < NWF; ++w) rmax = fmaxf(rmax, red3[slot[j] + w * BT]);
                    if (t0 + j * 32 + 32 <= outlier_rows) omax = fmaxf(omax, rmax);  // (wave-uniform: whole token tiles only)
                }
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
                if (lane == 0) {
                    if (outlier_park) *outlier_park = omax;
                    else atomicMax(outlier, __builtin_bit_cast(unsigned, omax));
                }
            }
#endif
        }
    }
};

// start_conv bias + frozen position embedding, time token in row 0 of every window, zeros in the
// padding rows (TM:199-216; M:122-123,133).  Row l of a window gets position id l + 1.
// NWF / BT (optional): waves along the features and tokens per block of the launching tile.  When the block spans all
// 512 features (NWF * FT * 32 == 512) and q8 is set, the rows are also written as int8 slices with one scale per row
// (the i8x3 attention-layer kernel's operand), like the LayerNorm epilogues do for the later layers.
template <int NP, int NWF = 0, int BT = 0>
struct EpiEmbed {
    const float* bias;      // [512]
    const float* pe;        // [max_timesteps + 1][512]
    const float* tt_table;  // [S][512]: time_mlp(t) + pe[1]
    const int* t_idx;       // [B]
    __bf16* out;
    size_t out_plane;
    int Lp, T, B;
    int8_t* q8;
    size_t q8_plane;
    float* q8_scale;
    // multi-step loops (nullptr: single step, timesteps from t_idx): the step state of pointwise.h — every window runs
    // the timestep of step state->embed_step; this kernel then publishes embed_step + 1 as the out kernel's counter
    StepState* state;
    const int* ts;          // explicit timestep list (strided samplers) or nullptr: t = t_start - step
    template <int FT, int TT>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int wf, int wt, char* smem) const {
        const int hf = lane >> 5, col = lane & 31;
        constexpr bool Q8 = NWF > 0 && NWF * FT * 32 == 512;
        float* red = (float*)smem;  // [TT][NWF][BT] row maxima per wave (Q8 only)
        int t_loop = 0;
        if (state) {
            const int i = state->embed_step;
            t_loop = ts ? ts[i] : state->t_start - i;
            // no block of THIS launch reads out_step; the step's out kernel (a later launch on the stream) does
            if (blockIdx.x == 0 && threadIdx.x == 0) state->out_step = i + 1;
        }
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int m = t0 + j * 32 + col;
            const int b = m / Lp, lw = m % Lp;
            const int kind = (b >= B || lw > T) ? 0 : (lw == 0 ? 1 : 2);
            const float* trow = (kind == 1) ? tt_table + (size_t)(state ? t_loop : t_idx[b]) * 512 : nullptr;
            const float* prow = (kind == 2) ? pe + (size_t)(lw + 1) * 512 : nullptr;
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int fb = f0 + i * 32 + 16 * jj + 4 * hf;
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (kind == 1) {
                        const float4 t0v = *(const float4*)(trow + fb), t1v = *(const float4*)(trow + fb + 8);
                        v[0] = t0v.x; v[1] = t0v.y; v[2] = t0v.z; v[3] = t0v.w;
                        v[4] = t1v.x; v[5] = t1v.y; v[6] = t1v.z; v[7] = t1v.w;
                    } else if (kind == 2) {
                        const float4 b0 = *(const float4*)(bias + fb), b1 = *(const float4*)(bias + fb + 8);
                        const float4 p0 = *(const float4*)(prow + fb), p1 = *(const float4*)(prow + fb + 8);
                        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                        const float pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = (acc[i][j][8 * jj + c] + bb[c]) + pp[c];
                    }
                    if (Q8) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            acc[i][j][8 * jj + c] = v[c];
                            amax = fmaxf(amax, fabsf(v[c]));
                        }
                    }
                    if (out) {  // (null in precision 9 outside debug runs: layer 0 reads the int8 rows only)
                        u32x4 hi, lo;
                        split8(v, hi, lo);
                        const size_t idx = acc_slot(m, f0 + i * 32, jj, hf, 32);
                        *(u32x4*)(out + idx) = hi;
                        if (NP == 2) *(u32x4*)(out + out_plane + idx) = lo;
                    }
                }
            if (Q8 && q8) {
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                if (hf == 0) red[(j * NWF + wf) * BT + (wt * TT + j) * 32 + col] = amax;
            }
        }
        if (Q8 && q8) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                const int m = t0 + j * 32 + col;
                float rmax = 0.f;
#pragma unroll
                for (int w = 0; w < NWF; ++w) rmax = fmaxf(rmax, red[(j * NWF + w) * BT + (wt * TT + j) * 32 + col]);
                const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
                if (wf == 0 && hf == 0) q8_scale[m] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
                for (int i = 0; i < FT; ++i) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
                    u32x4 s1, s2;
                    quant16(v, inv, s1, s2);
                    const size_t idx = acc_slot_i8(m, f0 + i * 32, hf, 16);
                    *(u32x4*)(q8 + idx) = s1;
                    *(u32x4*)(q8 + q8_plane + idx) = s2;
                }
            }
        }
    }
};

// linear_out (M:139) fused with the whole DDPM tail (M:235-256): x0 from the objective, clamp,
// posterior mean, noise, in-place x update, optional prefix in-painting (M:395-397), and the
// re-split of the new x into the embed GEMM's operand for the next step.
struct OutParams {
    const float* bias;     // [256] zero padded
    int mode;              // 0: raw model output -> out_raw; 1: posterior update of x; 2: DDIM update of x
    float* x;              // [B][T][D] in/out (modes 1, 2; single step — loops take it from the step state)
    float* out_raw;        // [B][T][D] (mode 0)
    __bf16* xall;          // fragment-tiled [Mp][KE]
    size_t xall_plane;
    int KE16;
    const float* sched;    // [S][8]: c1, c2, sigma, sqrt_recip, sqrt_recipm1, abar, -, -
    const int* t_idx;      // [B]
    const float* noise;    // [B][T][D] or nullptr (single step)
    int noise_mode;        // EGOEGO_NOISE_*
    uint64_t seed;         // (single step)
    int64_t window_offset; // (single step)
    int clip;
    int objective;         // 1 = pred_x0
    const float* prefix;   // [B][prefix_len][D] or nullptr (single step: never set)
    int prefix_len;        // > 0: in-paint the first frames of every window from the step state's prefix after the update
    // DDIM (mode 2): x <- sqrt(abar_prev) * x0 + dir * eps + sig * z, per-step coefficients indexed by the step index
    const float* ddim_tab;  // [n][4]: sqrt(abar_prev), dir = sqrt(1 - abar_prev - sig^2), sig, -
    // multi-step loops: everything that changes from call to call or from step to step lives in device memory (the
    // step state, pointwise.h), so that ONE captured step replays for every timestep, every chain and every caller
    // buffer: state->out_step - 1 is the index of the running step (selects the timestep, the injected-noise slice
    // and the DDIM table row); x / noise / prefix / seed / window_offset come from it too.  nullptr: single step.
    StepState* state;
    const int* ts;         // explicit timestep list or nullptr
    size_t step_elems;     // B*T*D: stride between the injected-noise slices of consecutive steps
    int Lp, T, B, D, DP;
};

// what a launch of the out kernel reads from the step state (or, for a single step, from OutParams)
struct OutDyn {
    float* x;
    const float* noise;
    const float* prefix;
    uint64_t seed;
    int64_t window_offset;
    float a_prev, dir, sig;  // DDIM coefficients of this step
};

// Explicit address spaces for EpiOut::run_block: hipcc does not infer LDS through its offsets, and x comes out of the step state
// (a pointer loaded from memory is generic to it).  Native vectors: HIP's float2 / float4 classes cannot carry an address space.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) f32x2 LdsF2;
typedef __attribute__((address_space(3))) f32x4 LdsF4;
typedef __attribute__((address_space(1))) f32x2 GlbF2;
typedef __attribute__((address_space(1))) f32x4 GlbF4;
template <int NP>
struct EpiOut {
    OutParams p;
    // Kernels whose workgroup owns ALL the (padded) output features of its token rows may stage the x rows through LDS (run_block)
    static constexpr bool BLOCK_ROWS = true;
    // one group = 4 consecutive features f..f+3 of one frame.  xs (run_block): this frame's row of x in LDS — read x_t from it and
    // leave the new x there instead of in global memory (same values either way)
    template <bool STAGED>
    EG_D void group(const float (&o)[4], int f, int m, int b, int frame, size_t row, int t, float c1, float c2,
                          float sigma, float srec, float srecm1, float abar, const OutDyn& d, float* xs) const {
        if (p.mode == 0) {
#pragma unroll
            for (int c = 0; c < 4; c += 2)
                if (f + c < p.D) *(float2*)(p.out_raw + row + f + c) = make_float2(o[c], o[c + 1]);
            return;
        }
        float xt[4] = {0.f, 0.f, 0.f, 0.f}, nz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; c += 2)
            if (f + c < p.D) {
                if (STAGED) {
                    const f32x2 v = *(const LdsF2*)(xs + f + c);
                    xt[c] = v.x;
                    xt[c + 1] = v.y;
                } else {
                    const float2 v = *(const float2*)(d.x + row + f + c);
                    xt[c] = v.x;
                    xt[c + 1] = v.y;
                }
            }
        if (sigma != 0.f) {  // the step's noise scale: ancestral sigma_t (mode 1) or the DDIM sig (mode 2, eta > 0)
            if (p.noise_mode == 0) {
#pragma unroll
                for (int c = 0; c < 4; c += 2)
                    if (f + c < p.D) {
                        const float2 v = *(const float2*)(d.noise + row + f + c);
                        nz[c] = v.x;
                        nz[c + 1] = v.y;
                    }
            } else if (p.noise_mode == 1) {
                philox_normal4(d.seed, (uint32_t)(f >> 2), (uint32_t)frame, (uint32_t)(d.window_offset + b), (uint32_t)t, nz);
            }
        }
        float xn[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x0 = (p.objective == 1) ? o[c] : (srec * xt[c] - srecm1 * o[c]);
            if (p.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            if (p.mode == 1) {
                const float mean = c1 * x0 + c2 * xt[c];
                xn[c] = mean + sigma * nz[c];
            } else {  // DDIM (Song et al. 2021, eq. 12): predicted noise from x0, then the eta-weighted step
                const float eps = (xt[c] - sqrtf(abar) * x0) / sqrtf(fmaxf(1.0f - abar, 1e-20f));
                xn[c] = d.a_prev * x0 + d.dir * eps + sigma * nz[c];
            }
            if (f + c >= p.D) xn[c] = 0.f;
        }
        if (d.prefix && frame < p.prefix_len) {
            const float* pr = d.prefix + ((size_t)b * p.prefix_len + frame) * p.D;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (f + c < p.D) xn[c] = pr[f + c];
        }
#pragma unroll
        for (int c = 0; c < 4; c += 2)
            if (f + c < p.D) {
                if (STAGED) *(LdsF2*)(xs + f + c) = f32x2{xn[c], xn[c + 1]};
                else *(float2*)(d.x + row + f + c) = make_float2(xn[c], xn[c + 1]);
            }
        uint2 hi, lo;
        split4(xn, hi, lo);
        const size_t idx = tiled_index(m, f, p.KE16);
        *(uint2*)(p.xall + idx) = hi;
        if (NP == 2) *(uint2*)(p.xall + p.xall_plane + idx) = lo;
    }

    // what this launch reads from the step state (or from OutParams for a single step)
    EG_D OutDyn dynamic(int& step, int& t_loop) const {
        OutDyn d{p.x, p.noise, p.prefix, p.seed, p.window_offset, 1.0f, 0.f, 0.f};
        step = 0;
        t_loop = 0;
        if (p.state) {
            step = p.state->out_step - 1;
            t_loop = p.ts ? p.ts[step] : p.state->t_start - step;
            d.x = p.state->x;
            d.noise = p.state->noise ? p.state->noise + (size_t)step * p.step_elems : nullptr;
            d.prefix = p.prefix_len ? p.state->prefix : nullptr;
            d.seed = p.state->seed;
            d.window_offset = p.state->window_offset;
            // no block of THIS launch reads embed_step; the next step's embed kernel does
            if (blockIdx.x == 0 && threadIdx.x == 0) p.state->embed_step = step + 1;
        }
        if (p.ddim_tab) {
            const float4 dd = *(const float4*)(p.ddim_tab + 4 * step);
            d.a_prev = dd.x; d.dir = dd.y; d.sig = dd.z;
        }
        return d;
    }

    // xs0: LDS copy of the x rows [r_first, ...) of the workgroup's tokens (run_block) or nullptr (x read and written in place)
    template <int FT, int TT, bool STAGED>
    EG_D void tiles(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, const OutDyn& d, int t_loop, float* xs0, int r_first) const {
        const int hf = lane >> 5, col = lane & 31;
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int m = t0 + j * 32 + col;
            const int b = m / p.Lp, lw = m % p.Lp;
            const bool valid = (b < p.B) && (lw >= 1) && (lw <= p.T);
            const int frame = lw - 1;
            const size_t row = valid ? ((size_t)b * p.T + frame) * p.D : 0;
            float* xs = STAGED ? xs0 + (valid ? (b * p.T + frame - r_first) * p.D : 0) : nullptr;
            float c1 = 0.f, c2 = 0.f, sigma = 0.f, srec = 0.f, srecm1 = 0.f, abar = 0.f;
            int t = 0;
            if (valid && p.mode != 0) {
                t = p.state ? t_loop : p.t_idx[b];
                const float* s = p.sched + (size_t)t * 8;
                c1 = s[0]; c2 = s[1]; sigma = p.mode == 2 ? d.sig : s[2]; srec = s[3]; srecm1 = s[4]; abar = s[5];
            }
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = f0 + i * 32 + 8 * g + 4 * hf;
                    if (valid && f < p.DP) {
                        const float4 b4 = *(const float4*)(p.bias + f);
                        const float o[4] = {acc[i][j][4 * g + 0] + b4.x, acc[i][j][4 * g + 1] + b4.y,
                                            acc[i][j][4 * g + 2] + b4.z, acc[i][j][4 * g + 3] + b4.w};
                        group<STAGED>(o, f, m, b, frame, row, t, c1, c2, sigma, srec, srecm1, abar, d, xs);
                    }
                }
        }
    }

    template <int FT, int TT>
    __device__ void run(f32x16 (&acc)[FT][TT], int f0, int t0, int lane, int, int, char*) const {
        int step, t_loop;
        const OutDyn d = dynamic(step, t_loop);
        tiles<FT, TT, false>(acc, f0, t0, lane, d, t_loop, nullptr, 0);
    }

    // The same for a workgroup of NT threads that owns all DP features of the BT token rows from blk_t0 on (after its main loop's
    // closing barrier; `smem` holds BT * D floats + 16 bytes).  A lane owns a token, so `run` reads and writes x 8 bytes at a time
    // in 32 different rows per instruction; but the x rows of a block's valid tokens are ONE contiguous range of the [B][T][D]
    // tensor (the padding rows between two windows have no x, and a window's last frame is followed by the next one's first):
    // it is copied into LDS 16 bytes per lane, updated there, and copied back.  Same values as `run`.
    template <int FT, int TT, int BT, int NT>
    EG_D void run_block(f32x16 (&acc)[FT][TT], int f0, int t0, int blk_t0, int lane, char* smem) const {
        int step, t_loop;
        const OutDyn d = dynamic(step, t_loop);
#ifndef EGOEGO_OUT_DIRECT
#define EGOEGO_OUT_DIRECT 0  // (A/B knob of variant builds: 1 = x read and written in place by every kernel, round 4's form)
#endif
        if (EGOEGO_OUT_DIRECT || p.mode == 0 || (p.D & 1)) {  // (raw model output: nothing to read; an odd row length: rows are not 8-byte aligned)
            tiles<FT, TT, false>(acc, f0, t0, lane, d, t_loop, nullptr, 0);
            return;
        }
        // first and last valid frame of the block, as rows of the [B * T][D] tensor
        const int m1 = blk_t0 + BT - 1;
        const int b0 = blk_t0 / p.Lp, lw0 = blk_t0 % p.Lp, b1 = m1 / p.Lp, lw1 = m1 % p.Lp;
        const int r_first = lw0 > p.T ? (b0 + 1) * p.T : b0 * p.T + max(lw0 - 1, 0);
        const int r_last = min(lw1 == 0 ? b1 * p.T - 1 : b1 * p.T + min(lw1, p.T) - 1, p.B * p.T - 1);
        const int n = max(r_last - r_first + 1, 0) * p.D;  // floats (even)
        float* const gx = d.x + (size_t)r_first * p.D;
        const int mis = (int)(((size_t)r_first * p.D) & 3);  // 0 or 2: LDS index = global index + mis, so 16-byte pieces line up on both sides
        float* const xs0 = (float*)smem + mis;
        const int head = min((4 - mis) & 3, n), body = (n - head) >> 2, tail = n - head - 4 * body;
        __syncthreads();  // (every wave is done with the operand buffers this overlays)
        if ((int)threadIdx.x < head / 2) *(LdsF2*)(xs0 + 2 * threadIdx.x) = *(const GlbF2*)(gx + 2 * threadIdx.x);
        for (int i = threadIdx.x; i < body; i += NT) *(LdsF4*)(xs0 + head + 4 * i) = *(const GlbF4*)(gx + head + 4 * i);
        if ((int)threadIdx.x < tail / 2) *(LdsF2*)(xs0 + head + 4 * body + 2 * threadIdx.x) = *(const GlbF2*)(gx + head + 4 * body + 2 * threadIdx.x);
        __syncthreads();
        tiles<FT, TT, true>(acc, f0, t0, lane, d, t_loop, xs0, r_first);
        __syncthreads();
        if ((int)threadIdx.x < head / 2) *(GlbF2*)(gx + 2 * threadIdx.x) = *(const LdsF2*)(xs0 + 2 * threadIdx.x);
        for (int i = threadIdx.x; i < body; i += NT) *(GlbF4*)(gx + head + 4 * i) = *(const LdsF4*)(xs0 + head + 4 * i);
        if ((int)threadIdx.x < tail / 2) *(GlbF2*)(gx + head + 4 * body + 2 * threadIdx.x) = *(const LdsF2*)(xs0 + head + 4 * body + 2 * threadIdx.x);
    }
};
