// tail_fused.h — one decoder layer's tail (TM:92-93, 107-116, 135, 139) in ONE kernel for batches that do not fill the
// chip with the 64-token workgroups of layer_tail_kernel (fewer than ~200 windows per GPU: every shard of the 8-GPU
// split of BASELINE configs[2], configs[1]'s B = 64, the reference's own sample_bs = 1):
//     fc (HD -> 512) + residual + LayerNorm  ->  FFN w_1 + ReLU  ->  FFN w_2 + residual + LayerNorm (+ padding mask)
//
// One 4-wave workgroup per block of 32*TT tokens (TT = 1, 2), one workgroup per CU: each wave owns a SIMD and 128 of the
// 512 features for ALL tokens of the block, so every GEMM is one pass and a token's whole row is in the workgroup for
// the LayerNorms.  Small batches are latency-bound (a lone workgroup's k-step through the LDS ring of gemm.h costs
// 0.8 us whatever the occupancy, and a step is 11 dependent launches), so the operand fetch is built to run free:
//   * WEIGHTS never touch LDS.  A wave's feature tiles are needed by no other wave, so its weight fragments go
//     global -> VGPR through a buffer resource (SGPR base + SGPR offset + one VGPR holding 16*lane: the scalar unit
//     does the address arithmetic), prefetched 3 k-steps ahead through a register ring: no LDS write, no LDS read,
//     no workgroup barrier per k-step.
//   * ACTIVATIONS stream global -> LDS (LDS-DMA) in chunks of 8 k-steps (128 columns), double-buffered: the next
//     chunk is requested at the start of the current one and lands during its MFMAs; one barrier per chunk.  The
//     three GEMMs read O (the attention output), the LayerNorm-1 output and the FFN hidden activations, the latter
//     two written by this very workgroup moments earlier (they come back from L2).
// The arithmetic is that of the large-batch kernels to the bit: the same MFMA order per output element (k ascending;
// lo*hi, hi*lo, hi*hi) and the very same epilogue code (gemm.h EpiResLN / EpiTiled), so a window's numbers do not depend
// on the batch it is sampled in (tests: batch-size and shard invariance).
#pragma once
#include <type_traits>

#include "gemm.h"

// perf-debug ablations of the k-loop (compile-time, results become wrong): 1 = no weight loads, 2 = no LDS fragment
// reads, 4 = no LDS-DMA, 8 = no MFMAs
#ifndef TAIL_ABLATE
#define TAIL_ABLATE 0
#endif
// weight-ring depth (k-steps of register prefetch + 1) of the one-workgroup-per-CU builds: the all-int8 tail / embed and linear_out.
// Measured at B=32 / 64 (round 3): depth 8 (with the FFNs in two feature passes to make room) is 4 % SLOWER for the tail (0.407 vs
// 0.391 ms per step at B=32) and changes nothing for embed / linear_out — a lone workgroup per CU is not short of bytes in flight.
#ifndef TAIL_RING1
#define TAIL_RING1 4
#endif
#ifndef TAIL_RING_IO
#define TAIL_RING_IO 4
#endif
// weight-ring depth of the FFN contractions of the eight-wave tail (its fc keeps 4: the fp32 running sums need the registers)
#ifndef TAIL8_RING
#define TAIL8_RING 4
#endif
#ifndef TAIL8_RING_FC
#define TAIL8_RING_FC 4
#endif

struct TailArgs {
    // fc + residual + LayerNorm (TM:92-93, 135)
    const __bf16* o;         // attention output [Mp][HD], split-bf16 fragment-tiled (accumulator order along K)
    size_t o_plane;          // elements between the hi and lo planes
    int HD16;                // HD / 16
    const __bf16* wfc;       // [512][HD] split-bf16 fragment-tiled, K permuted like o
    size_t wfc_plane;
    EpiResLN<2, 4, 0> ln1;   // bias, residual (layer input), gamma/beta, row mask, output hb (BT is a compile-time detail)
    // FFN (TM:107-116, 139)
    const __bf16* w1;
    size_t w1_plane;
    EpiTiled<true, 2> relu;  // bias, output f
    const __bf16* w2;
    size_t w2_plane;
    EpiResLN<2, 4, 0> ln2;   // bias, residual hb, gamma/beta, row mask, output (+ optional int8 copy)
    // i8x3 fc (fc8 != 0; windows of more than 64 tokens, whose attention kernels write O as int8 rows): the attention output as
    // int8 slices with one scale per row and head, w_fc as int8 slices with one scale per output row; one exact integer
    // chain per head (8 k-blocks = one chunk), folded into an fp32 running sum (gemm.h i8_fold)
    int fc8, H;
    const int8_t* o8;        // [Mp][HD] two slices, o8_plane BYTES apart
    size_t o8_plane;
    const float* o_scale;    // [Mp][H]
    const int8_t* wfc8;      // [512][HD] two slices, wfc8_plane BYTES apart
    size_t wfc8_plane;
    const float* s_wfc;      // [512]
    const int8_t* wfc8_3;    // EGOEGO_FLAG_FC24: [third slice | zeros] of w_fc, wfc8_plane BYTES apart, or nullptr
    // i8x3 FFN (ffn8 != 0): w_1 / w_2 as int8 slices with one scale per output row; LayerNorm-1 also emits int8 rows (ln1.q8),
    // FFN-1 writes the ReLU output as int8 rows (relu8) and FFN-2 reads them
    int ffn8;
    const int8_t* w1_8;      // [512][512] two slices, w8_plane BYTES apart
    const int8_t* w2_8;
    size_t w8_plane;
    const float* s_w1;       // [512] weight row scales
    const float* s_w2;
    EpiReluQ8<4, 0> relu8;   // bias, int8 output rows + scales
    int stop;                // debug taps: 1 = return after LayerNorm-1, 2 = after FFN-1, 0 = run everything
    unsigned* outlier;       // outlier monitor (StepState::ln_max + 2 * layer: LayerNorm-1's site, LayerNorm-2's right behind) or nullptr
    int outlier_rows;        // rows below this index are recorded
    EG_DBG(unsigned long long* trace;)  // perf-debug build: [grid][32] phase timestamps or nullptr
};

static constexpr int tail_smem_bytes(int TT, int NWV = 4) { return TT * 32 * 1024 + 3 * TT * NWV * 32 * TT * 4; }  // chunk double buffer + epilogue scratch
static constexpr int TAIL_PAR_BYTES = 10 * 512 * 4;  // the all-int8 build also stages its ten per-feature parameter vectors in LDS

// Weight / activation fragments are fetched through buffer resources: address = SGPR base + SGPR offset + one VGPR
// (plain global loads make hipcc build a 64-bit VGPR address per fragment and k-step: hundreds of registers of them).
using tail_rsrc = __amdgpu_buffer_rsrc_t;
// `bytes`: extent of the tensor — loads past it return zeros (the weight prefetch of a short last chunk runs one k-block over)
EG_D tail_rsrc tail_make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu), 0x00020000);
}
EG_D i32x4 tail_load(tail_rsrc r, int voff, unsigned soff) {
    return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// ---- one chunk of 8 k-steps -----------------------------------------------------------------------------------
// acc += W[FT feature tiles][8 k-blocks at wcur] x act[TT token tiles][8 k-blocks of the LDS chunk buffer at `act`]
// (layout [plane][t-tile][8 k-blocks][1 KiB], lo plane at +act_plane).  wq is the weight ring (RING slots, prefetch
// distance PD = RING - 1): on entry the fragments of this chunk's k-steps 0 .. PD-1 are in it or in flight, with
// nothing but younger weight loads issued after them; every k-step loads the fragments PD steps ahead — the next
// chunk's first ones from wnext — unless LAST.  DP > 0: this wave's DP 1-KiB pieces of the next chunk are issued in
// k-step 0 through dma(piece) (loads return in order: issued first, they have the whole chunk to land).
// Activation fragments: the hi plane is double-buffered (read a k-step ahead), the lo plane is re-read right after the
// one MFMA group that uses it; each fragment is consumed >= FT*TT MFMAs after its read was issued.
// I8 = true: the operands are int8 slices (k-blocks of 32; "hi" plane = slice 1, "lo" plane = slice 2) and the sums go to
// an I8Acc pair — s2*s1 and s1*s2 into .m, s1*s1 into .h, the groups in the order of the split-bf16 parts.
template <int FT, int TT, int RING, bool LAST, int DP, int NKS = 8, bool I8 = false, bool WLO0 = false>
struct TailChunk {
    using AccT = typename std::conditional<I8, I8Acc, f32x16>::type;
    static constexpr int PD = RING - 1, NW = FT * 2;  // NKS < 8: the short last chunk of a contraction whose k-blocks are no multiple of 8
    // WLO0: the weights' second slice is all zero (the third-slice pass of EGOEGO_FLAG_FC24): it is neither loaded nor multiplied
    static constexpr int NWL = WLO0 ? FT : NW;  // weight loads per k-step
    static_assert(!WLO0 || I8, "int8 slices only");
    static_assert(NKS == 8 || (LAST && DP == 0), "only the last chunk may be short");
    static_assert(RING == 2 || RING == 4 || RING == 8, "ring slots must divide the 8 k-steps of a chunk");
    static constexpr int n_ops(int s) { return ((s + PD < NKS || !LAST) ? NWL : 0) + (s == 0 ? DP : 0); }
    // vector-memory operations issued after the DMA pieces: what may stay in flight when the next chunk must have landed
    static constexpr int after_dma() {
        int n = 0;
        for (int s = 1; s < 8; ++s) n += n_ops(s);
        return n < 63 ? n : 63;
    }
    // vector-memory operations issued after the weight loads of k-step ks: what may still be in flight when they are needed
    static constexpr int allowed(int ks) {
        int n = 0;
        if (ks < PD) {
            n += (PD - 1 - ks) * NWL;
            for (int s = 0; s < ks; ++s) n += n_ops(s);
        } else {
            n += ks - PD == 0 ? DP : 0;
            for (int s = ks - PD + 1; s < ks; ++s) n += n_ops(s);
        }
        return n < 63 ? n : 63;
    }
    // group 0: lo * hi, group 1: hi * lo, group 2: hi * hi
    template <int G>
    static EG_D void mma(i32x4 w, i32x4 a, f32x16& c) {
        if (TAIL_ABLATE & 8) return;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    }
    template <int G>
    static EG_D void mma(i32x4 w, i32x4 a, I8Acc& c) {
        if (TAIL_ABLATE & 8) return;
        i32x16& d = G == 2 ? c.h : c.m;
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, d, 0, 0, 0);
    }

    template <int KS, class Dma>
    static EG_D void step(AccT (&acc)[FT][TT], i32x4 (&wq)[RING][NW], i32x4 (&ah)[2][TT], i32x4 (&al)[TT], tail_rsrc wr,
                          const unsigned (&wcur)[NW], const unsigned (&wnext)[NW], const char* act, int act_plane, int lane, Dma& dma) {
        constexpr int cur = KS & 1;
        // the weights and the hi-plane activations of this k-step have landed (counted: younger loads stay in flight)
        asm volatile("" ::: "memory");
        wait_counts<allowed(KS), (KS == 0 ? 0 : TT)>();
        __builtin_amdgcn_sched_barrier(0);
        if (KS + PD < NKS || !LAST) {
            constexpr int kn = KS + PD;
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (!(TAIL_ABLATE & 1) && !(WLO0 && (q & 1))) wq[kn % RING][q] = tail_load(wr, lane * 16, kn < 8 ? wcur[q] + (kn << 10) : wnext[q] + ((kn & 7) << 10));
        }
        if (KS == 0 && DP) {
#pragma unroll
            for (int d = 0; d < DP; ++d)
                if (!(TAIL_ABLATE & 4)) dma(d);
        }
        if (KS < NKS - 1) {
#pragma unroll
            for (int j = 0; j < TT; ++j)
                if (!(TAIL_ABLATE & 2)) ah[cur ^ 1][j] = *(const i32x4*)(act + ((j * 8 + KS + 1) << 10) + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const i32x4(&w)[NW] = wq[KS % RING];
        // part-major (two MFMAs on one accumulator are never back to back), in the order of gemm.h's mma_part:
        // lo*hi, hi*lo, hi*hi.  w[2i] = hi plane of feature tile i, w[2i+1] = lo plane.
        if (!WLO0) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) mma<0>(w[2 * i + 1], ah[cur][j], acc[i][j]);
        }
        asm volatile("" ::: "memory");
        wait_counts<63, (KS < NKS - 1 ? TT : 0)>();  // the lo-plane activations of this k-step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) mma<1>(w[2 * i], al[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (KS < NKS - 1) {
#pragma unroll
            for (int j = 0; j < TT; ++j)
                if (!(TAIL_ABLATE & 2)) al[j] = *(const i32x4*)(act + act_plane + ((j * 8 + KS + 1) << 10) + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) mma<2>(w[2 * i], ah[cur][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    }

    template <class Dma>
    static EG_D void run(AccT (&acc)[FT][TT], i32x4 (&wq)[RING][NW], tail_rsrc wr, const unsigned (&wcur)[NW],
                         const unsigned (&wnext)[NW], const char* act, int act_plane, int lane, Dma dma) {
        i32x4 ah[2][TT], al[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            ah[0][j] = *(const i32x4*)(act + ((j * 8) << 10) + lane * 16);
            al[j] = *(const i32x4*)(act + act_plane + ((j * 8) << 10) + lane * 16);
        }
        step<0>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 1) step<1>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 2) step<2>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 3) step<3>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 4) step<4>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 5) step<5>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 6) step<6>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (NKS > 7) step<7>(acc, wq, ah, al, wr, wcur, wnext, act, act_plane, lane, dma);
        if (DP) {  // the next chunk has landed; the weight prefetches issued after its pieces stay in flight
            asm volatile("" ::: "memory");
            wait_counts<after_dma(), 15>();
        }
    }
};

// ---- the GEMM: acc = in[32 TT tokens of this workgroup][K] x W[this wave's FT feature tiles][K]^T ------------------
// in / w: split-bf16 fragment-tiled tensors with K16 = K / 16 k-blocks per row tile: a multiple of 8, or (REM2) a multiple of
// 8 plus 2 — the embed operand's 26 k-blocks.  `act` is the
// chunk double buffer in LDS (TT * 32 KiB).  Feature tile i of wave `wave` is tile wave * FT + i of W.
// I8: int8-slice operands ([R/32][K/32][2][32][16] byte planes, the two slices `*_plane` bf16-element units = bytes / 2
// apart; K16 then counts 32-wide k-blocks) accumulated into I8Acc pairs.
// NWAVES: waves of the workgroup (4, or the 8 of the eight-wave all-int8 tail): they share the LDS-DMA pieces of a chunk.
template <int FT, int TT, int RING, bool REM2 = false, bool I8 = false, int NWAVES = 4, bool WLO0 = false>
struct DirectGemm {
    using AccT = typename std::conditional<I8, I8Acc, f32x16>::type;
    static constexpr int NW = 2 * FT, PD = RING - 1;
    static constexpr int CH_PLANE = TT * 8 * 1024, CH_BYTES = 2 * CH_PLANE;  // chunk buffer: [plane][t-tile][8 k-blocks][1 KiB]
    static constexpr int DMA_PIECES = 16 * TT / NWAVES;                      // 1-KiB pieces of a chunk per wave
    static constexpr int SMEM_BYTES = 2 * CH_BYTES;

    struct NoPost {
        EG_D void operator()(int) const {}
    };
    // post(q): called after the MFMAs of chunk q (8 k-blocks), e.g. to fold a chunk's integer sums away (int8 fc)
    struct NoPre {
        EG_D void operator()() const {}
    };
    // pre(): called once in the prologue, AFTER the first chunk's pieces and the first weight fragments have been requested and
    // before they are waited for — work that needs a memory round trip of its own (the tail's parameter staging) then shares theirs
    template <class Mark, class Post = NoPost, class Pre = NoPre>
    static EG_D void run(AccT (&acc)[FT][TT], const __bf16* in, size_t in_plane, int K16, const __bf16* w, size_t w_plane, char* act,
                         int tt0, int wave, int lane, Mark mark, int wtile0 = -1, Post post = Post{}, Pre pre = Pre{}) {
        const int wt0 = wtile0 >= 0 ? wtile0 : wave * FT;  // first weight row tile of this wave (default: FT consecutive tiles per wave)
        i32x4 wq[RING][NW];
        // The activation resource starts at THIS workgroup's first row tile (64-bit address arithmetic on the scalar unit), so the
        // 32-bit offsets below stay small whatever the batch: with the base at row 0 the second slice of the int8 attention output
        // of the largest accepted call (2^20 rows x 1024 bytes) ends exactly at 2^31, one byte beyond the largest num_records a
        // buffer resource can hold — its last lanes read zeros (round 4: test_large_batches...[8192-120-9]).
        const tail_rsrc wr = tail_make_rsrc(w, w_plane * 4);
        const tail_rsrc ir = tail_make_rsrc((const char*)in + (((size_t)tt0 * (size_t)K16) << 10), 0x7fffffffu);
        const unsigned wpb = (unsigned)(w_plane * 2), ipb = (unsigned)(in_plane * 2);
        const int NQ = REM2 ? K16 / 8 + 1 : K16 / 8;  // chunks, the last one 2 k-steps long with REM2
        auto w_offsets = [&](int kb, unsigned (&out)[NW]) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) out[2 * i + s] = s * wpb + (unsigned)(((wt0 + i) * K16 + kb) << 10);
        };
        auto dma_piece = [&](int q, int piece) {
            const int x = wave * DMA_PIECES + piece;  // flat index over [plane][t-tile][8 k-blocks]
            const int s = x / (8 * TT), j = (x >> 3) % TT;
            int kb = x & 7;
            if (REM2 && q == NQ - 1) kb &= 1;  // the short chunk has 2 k-blocks: the other pieces re-load them (the counted waits assume a fixed piece count)
            const unsigned src = s * ipb + (unsigned)((j * K16 + 8 * q + kb) << 10);
            char* dst = act + (q & 1) * CH_BYTES + s * CH_PLANE + ((j * 8 + kb) << 10);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ir, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, src, 0, 0);
        };
        unsigned wcur[NW], wnext[NW];
#pragma unroll
        for (int pc = 0; pc < DMA_PIECES; ++pc) dma_piece(0, pc);
        w_offsets(0, wcur);
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (!(WLO0 && (q & 1))) wq[k][q] = tail_load(wr, lane * 16, wcur[q] + (k << 10));
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) acc_zero(acc[i][j]);
        pre();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mark();
        for (int q = 0; q + 1 < NQ; ++q) {
            w_offsets(8 * q, wcur);
            w_offsets(8 * q + 8, wnext);
            TailChunk<FT, TT, RING, false, DMA_PIECES, 8, I8, WLO0>::run(acc, wq, wr, wcur, wnext, act + (q & 1) * CH_BYTES, CH_PLANE, lane,
                                                             [&](int piece) { dma_piece(q + 1, piece); });
            post(q);
            // the next chunk has landed (counted wait inside run) and everyone is done with this one
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        w_offsets(8 * (NQ - 1), wcur);
        TailChunk<FT, TT, RING, true, 0, (REM2 ? 2 : 8), I8, WLO0>::run(acc, wq, wr, wcur, wcur, act + ((NQ - 1) & 1) * CH_BYTES, CH_PLANE, lane, [](int) {});
        post(NQ - 1);
        __syncthreads();  // every wave is done with the chunk buffers before the next GEMM's first DMA
    }

    // The same contraction with the WHOLE activation operand already in LDS: K = 16 k-blocks of 32 (the 512-wide int8 rows of
    // the FFN contractions) are exactly the two chunk buffers, written there by the previous epilogue (gemm.h lds_chunk_slot)
    // instead of going to memory and coming back by LDS-DMA.  No DMA, no barrier inside: the caller's barrier after the
    // epilogue's LDS writes opens the loop, and the caller's next barrier closes it (before anything overwrites the buffers).
    // Same k order, same MFMA order: same integers.  w_primed: the first PD k-steps of weights are already in the ring (issued
    // by prime() before the epilogue, whose latency then hides their fetch).
    static EG_D void prime(i32x4 (&wq)[RING][NW], const __bf16* w, size_t w_plane, int K16, int wt0, int lane) {
        const tail_rsrc wr = tail_make_rsrc(w, w_plane * 4);
        const unsigned wpb = (unsigned)(w_plane * 2);
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) wq[k][2 * i + sl] = tail_load(wr, lane * 16, sl * wpb + (unsigned)(((wt0 + i) * K16 + k) << 10));
    }
    template <class Mark>
    static EG_D void run_resident(AccT (&acc)[FT][TT], i32x4 (&wq)[RING][NW], const __bf16* w, size_t w_plane, const char* act, int lane, Mark mark,
                                  int wt0) {
        static_assert(!REM2, "two whole chunks");
        constexpr int K16 = 16;
        const tail_rsrc wr = tail_make_rsrc(w, w_plane * 4);
        const unsigned wpb = (unsigned)(w_plane * 2);
        auto w_offsets = [&](int kb, unsigned (&out)[NW]) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) out[2 * i + sl] = sl * wpb + (unsigned)(((wt0 + i) * K16 + kb) << 10);
        };
        unsigned wcur[NW], wnext[NW];
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) acc_zero(acc[i][j]);
        mark();
        w_offsets(0, wcur);
        w_offsets(8, wnext);
        TailChunk<FT, TT, RING, false, 0, 8, I8>::run(acc, wq, wr, wcur, wnext, act, CH_PLANE, lane, [](int) {});
        TailChunk<FT, TT, RING, true, 0, 8, I8>::run(acc, wq, wr, wnext, wnext, act + CH_BYTES, CH_PLANE, lane, [](int) {});
    }
};

// (Measured and NOT kept for B = 256: the same kernel with 2 ring slots, 256 registers and two co-resident 64-token workgroups per
// CU — 278 us per launch against the 222 us of layer_tail_kernel, and no better with the second half of the grid started 20-80 us
// late: a LayerNorm epilogue that takes 13 us alone takes 60 us next to a wave that saturates the matrix pipe of the same SIMD.)
// FFN8: the two FFN contractions on int8 slices (TailArgs::ffn8 ...; the i8x3 precision); FC8: fc too (TailArgs::fc8 ...) —
// separate instantiations, so that the split-bf16 kernel's register allocation is untouched.
// W2 (int8 fc + FFN only): a 256-register build, TWO workgroups per CU — for grids of more than one workgroup per CU, where a
// lone workgroup's weight stream is latency-bound (bytes in flight / L2 latency): a second resident workgroup doubles the bytes
// in flight and fills the other's epilogues.  Every contraction then runs its 4 feature tiles per wave in two passes of 2
// (I8Acc pairs of 2 tiles + a 4-slot weight ring = 128 registers); integer sums and float operations are unchanged: same bits.
// NWV = 8 (int8 fc + FFN only): ONE eight-wave workgroup per CU, two 256-register waves per SIMD, a wave owning 64 features — for grids
// of at most one workgroup per CU (every shard of the 8- and 4-GPU splits of BASELINE configs[2]), where a workgroup is a serial
// chain and the chip is not full: a SIMD's second wave issues its weight loads and MFMAs in the first one's waits, and the three
// epilogues (load-latency chains over a wave's feature tiles) are split over twice the waves.  Same integers, same float operations
// per value, LayerNorm partials combined in the order every tiling uses (gemm.h EpiResLN): same bits.
// RES (int8 fc + FFN only; the product path of precision 9, whose inter-kernel activations are int8 rows only): the FFN operands stay
// on the CU.  LayerNorm-1 writes its int8 rows into the two LDS chunk buffers — for K = 512 they ARE the whole operand of FFN-1
// — and keeps them in registers as LayerNorm-2's residual; FFN-1's epilogue writes the hidden rows over them for FFN-2.  Neither
// tensor goes to memory (2 x 33 MB written and read back per launch at B=256 otherwise), no s_waitcnt vmcnt(0) + LDS-DMA round trip
// stands between a LayerNorm and the contraction behind it, and the next contraction's first weight fragments are fetched
// before the epilogue that precedes it.  Same integers, same float operations: same bits as the memory round trip.
static constexpr int TAIL_RES_BYTES = 272;  // two [32] row-scale vectors in LDS + the parked LayerNorm-1 outlier maximum
template <int TT, bool FFN8, bool FC8 = false, bool W2 = false, int NWV = 4, bool RES = false>
__global__ __launch_bounds__(64 * NWV, ((W2 || NWV == 8) ? 2 : 1)) void tail_kernel(TailArgs a) {
    static_assert(!RES || (FFN8 && FC8 && TT == 1), "LDS-resident FFN operands: the all-int8 32-token tail");
    static_assert(!W2 || (FFN8 && FC8 && TT == 1), "the two-workgroups-per-CU build exists for the all-int8 32-token tail");
    static_assert(NWV == 4 || (NWV == 8 && TT == 1 && !W2 && ((FFN8 && FC8) || (!FFN8 && !FC8 && !RES))),
                  "the eight-wave build exists for the 32-token tails: all int8, or (round 5) all split-bf16");
    constexpr int TOK = 32 * TT, FT = 16 / NWV;
    using G = DirectGemm<FT, TT, 4, false, false, NWV>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const act = smem;
    char* const red = smem + G::SMEM_BYTES;  // the LayerNorm epilogues' cross-wave reduction scratch
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63;
    const int tok0 = (int)blockIdx.x * TOK;
    const int tt0 = (int)blockIdx.x * TT;
    EG_DBG(unsigned long long* const tr = a.trace ? a.trace + 180224 + (size_t)blockIdx.x * 32 : nullptr;)
    auto mark = [&](int i) {
        EG_DBG(if (tr && threadIdx.x == 0) {
            tr[i] = wall_clock64();
            tr[16 + i] = __builtin_readcyclecounter();  // shader clock ticks: clock = d(ticks) / d(wall)
        })
        (void)i;
    };
    mark(0);
    f32x16 acc[FT][TT];
    // epilogue structs carry the large-batch kernels' code; only the tokens-per-block template argument differs
    auto as_ln = [&](const EpiResLN<2, 4, 0>& e, int which = 0) {
        return EpiResLN<2, NWV, TOK>{e.bias, e.res, e.res_plane, e.gamma, e.beta, e.row_mask, e.out, e.out_plane, e.eps, e.q8, e.q8_plane, e.q8_scale, e.res8, e.res8_plane, e.res8_scale, nullptr, nullptr,
                                     a.outlier ? a.outlier + which : nullptr, nullptr, a.outlier_rows};
    };
    // All-int8 build: the ten per-feature parameter vectors of the three epilogues (weight row scales, biases, LayerNorm gains
    // and shifts) are staged in LDS once per workgroup — the epilogues of a 32-token workgroup are load-latency chains, and an
    // LDS read costs a tenth of an L2 round trip.  Visible after the first GEMM's prologue barrier.
    // (Round 4: the copy is issued from INSIDE the first GEMM's prologue, behind its first chunk's pieces and first weight fragments —
    // placed in front of them, as it was, its loads were waited for before those were even requested: two dependent memory round
    // trips at the start of every workgroup, 16-21 us of a workgroup's 66 at B=256 in the phase trace.)
    const float *p_swfc = a.s_wfc, *p_sw1 = a.s_w1, *p_sw2 = a.s_w2;
    const float* const par_src[10] = {a.s_wfc, a.ln1.bias, a.ln1.gamma, a.ln1.beta, a.s_w1, a.relu8.bias, a.s_w2, a.ln2.bias, a.ln2.gamma, a.ln2.beta};
    auto stage_params = [&] {
        float* par = (float*)(smem + tail_smem_bytes(TT, NWV));
        if (threadIdx.x < 128) {
#pragma unroll
            for (int v = 0; v < 10; ++v) *(float4*)(par + v * 512 + 4 * threadIdx.x) = *(const float4*)(par_src[v] + 4 * threadIdx.x);
        }
    };
    if constexpr (FFN8 && FC8) {
        float* par = (float*)(smem + tail_smem_bytes(TT, NWV));
        p_swfc = par;
        a.ln1.bias = par + 512; a.ln1.gamma = par + 1024; a.ln1.beta = par + 1536;
        p_sw1 = par + 2048; a.relu8.bias = par + 2560;
        p_sw2 = par + 3072; a.ln2.bias = par + 3584; a.ln2.gamma = par + 4096; a.ln2.beta = par + 4608;
    }

    // =============================================================== 1. fc + residual + LayerNorm (TM:92-93, 135)
    if constexpr (FC8) {
        // One exact integer chain per head (8 k-blocks = one chunk), folded into the fp32 running sum `acc` between chains.
        // A wave's 4 feature tiles go in FP passes of 4 / FP tiles: with 64 tokens per workgroup (TT = 2) the I8Acc pairs of
        // all 8 tiles next to their 8 running-sum tiles would exceed the register file, so the two feature halves run one
        // after the other (the activation chunks are streamed twice, the weights once either way).
        constexpr int FP = NWV == 8 ? 1 : ((W2 || TAIL_RING1 == 8) ? 2 : TT), FTP = FT / FP;
        using GF = DirectGemm<FTP, TT, (NWV == 8 ? TAIL8_RING_FC : (W2 ? 4 : TAIL_RING1)), false, true, NWV>;
        const int col = lane & 31;
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int fp = 0; fp < FP; ++fp) {
            I8Acc q[FTP][TT];
            GF::run(q, (const __bf16*)a.o8, a.o8_plane / 2, a.HD16 / 2, (const __bf16*)a.wfc8, a.wfc8_plane / 2, act, tt0, wave, lane, [&] { mark(7); },
                    wave * FT + fp * FTP, [&](int h) {  // head h's chain is complete: fold it into the running sum, start the next from zero
                        float so[TT];
#pragma unroll
                        for (int j = 0; j < TT; ++j) so[j] = a.o_scale[(size_t)(tok0 + j * 32 + col) * a.H + h] * 256.0f;
#pragma unroll
                        for (int i = 0; i < FTP; ++i)
#pragma unroll
                            for (int j = 0; j < TT; ++j) {
                                i8_fold(q[i][j], acc[fp * FTP + i][j], so[j]);
                                acc_zero(q[i][j]);
                            }
                    }, [&] { if (fp == 0) stage_params(); });
        }
        // EGOEGO_FLAG_FC24: a second contraction per feature pass with the weights' THIRD slice (w ~ scale (q16 + w3 / 256)): the same
        // chain with [w3 | 0] as the weight slices gives 256 sum(w3 a1) + sum(w3 a2), worth 1 / 65536 of the first pass's units.
        // (Round 4: on the trained-like checkpoint fc on the 16-bit weight grid is what separates precision 9 from 8 at the end of a
        // whole chain — 7.0e-4 against 4.3e-4 with this pass, 3.1e-4 in precision 8.  The all-zero second weight slice is neither loaded
        // nor multiplied: DirectGemm's WLO0 form, two MFMAs per product.)
        if (a.wfc8_3) {
#pragma unroll
            for (int fp = 0; fp < FP; ++fp) {
                I8Acc q[FTP][TT];
                using GF3 = DirectGemm<FTP, TT, (NWV == 8 ? TAIL8_RING_FC : (W2 ? 4 : TAIL_RING1)), false, true, NWV, true>;  // second weight slice: zero, skipped
                GF3::run(q, (const __bf16*)a.o8, a.o8_plane / 2, a.HD16 / 2, (const __bf16*)a.wfc8_3, a.wfc8_plane / 2, act, tt0, wave, lane, [&] {},
                        wave * FT + fp * FTP, [&](int h) {
                            float so[TT];
#pragma unroll
                            for (int j = 0; j < TT; ++j) so[j] = a.o_scale[(size_t)(tok0 + j * 32 + col) * a.H + h] * (1.0f / 256.0f);
#pragma unroll
                            for (int i = 0; i < FTP; ++i)
#pragma unroll
                                for (int j = 0; j < TT; ++j) {
                                    i8_fold(q[i][j], acc[fp * FTP + i][j], so[j]);
                                    acc_zero(q[i][j]);
                                }
                        });
            }
        }
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) i8_fold_finish(acc[i][j], p_swfc + wave * FT * 32 + i * 32 + 4 * (lane >> 5));
    } else {
        G::run(acc, a.o, a.o_plane, a.HD16, a.wfc, a.wfc_plane, act, tt0, wave, lane, [&] { mark(7); });
    }
    mark(1);
    if constexpr (RES) {
        constexpr int FP8 = W2 ? 2 : 1, FTP8 = FT / FP8, RING8 = NWV == 8 ? TAIL8_RING : 4;
        using G8 = DirectGemm<FTP8, TT, RING8, false, true, NWV>;
        float* const ls1 = (float*)(smem + tail_smem_bytes(TT, NWV) + TAIL_PAR_BYTES);  // row scales of the LayerNorm-1 rows, of the hidden rows
        float* const ls2 = ls1 + 32;
        i32x4 wq[RING8][2 * FTP8];
        Rows8<FT> h1;
        // the contraction of all FP8 feature passes over the LDS-resident operand; pass 0's first weights were primed by the caller
        auto ffn_resident = [&](I8Acc (&q)[FT][TT], const int8_t* w8, int mk) {
#pragma unroll
            for (int fp = 0; fp < FP8; ++fp) {
                if (fp) G8::prime(wq, (const __bf16*)w8, a.w8_plane / 2, 16, wave * FT + fp * FTP8, lane);
                G8::run_resident(*(I8Acc(*)[FTP8][TT]) & q[fp * FTP8], wq, (const __bf16*)w8, a.w8_plane / 2, act, lane, [&] { mark(mk); }, wave * FT + fp * FTP8);
            }
        };
        G8::prime(wq, (const __bf16*)a.w1_8, a.w8_plane / 2, 16, wave * FT, lane);
        {
            auto e = as_ln(a.ln1);
            if (a.stop != 1) { e.q8 = nullptr; e.q8_scale = nullptr; }  // (debug tap 1: the LayerNorm-1 rows also go to memory, a.ln1.q8)
            e.lds_q8 = act; e.lds_scale = ls1;
            if (e.outlier) { e.outlier = nullptr; e.outlier_park = ls2 + 32; }  // flushed at the end of the kernel (thread 0 wrote it)
            e.template run<FT, TT, NoRows, Rows8<FT>>(acc, wave * FT * 32, tok0, lane, wave, 0, red, nullptr, &h1);
        }
        if (a.stop == 1) return;
        __syncthreads();  // the rows and their scales are in LDS
        mark(2);
        I8Acc q[FT][TT];
        ffn_resident(q, a.w1_8, 8);
        mark(3);
        G8::prime(wq, (const __bf16*)a.w2_8, a.w8_plane / 2, 16, wave * FT, lane);
        {
            // (the barrier inside the epilogue, between its row-maximum exchange and its stores, is also what lets the stores
            // overwrite the LayerNorm-1 rows: every wave has left its k-loop by then)
            // (debug tap 2: the hidden rows also go to memory, a.relu8.q8 — the epilogue then needs the block's first token for them)
            const bool tap = a.stop == 2;
            const EpiReluQ8<NWV, TOK> e{a.relu8.bias, tap ? a.relu8.q8 : nullptr, a.relu8.q8_plane, tap ? a.relu8.q8_scale : nullptr, act, ls2, tap ? tok0 : 0};
            e.template run<false, I8Acc, FT, TT>(q, p_sw1, ls1, wave * FT * 32, 0, lane, wave, 0, red);
        }
        if (a.stop == 2) return;
        __syncthreads();
        mark(4);
        ffn_resident(q, a.w2_8, 9);
        mark(5);
        i8_dequant_tile<false>(q, acc, p_sw2, ls2, wave * FT * 32, 0, lane);
        {
            auto e = as_ln(a.ln2, 1);
            e.res8 = nullptr;
            e.template run<FT, TT, Rows8<FT>, NoRows>(acc, wave * FT * 32, tok0, lane, wave, 0, red, &h1, nullptr);
        }
        if (a.outlier && threadIdx.x == 0) atomicMax(a.outlier, __builtin_bit_cast(unsigned, ls2[32]));  // LayerNorm-1's parked maximum
        EG_DBG(if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); })
        mark(6);
        return;
    }
    as_ln(a.ln1).template run<FT, TT>(acc, wave * FT * 32, tok0, lane, wave, 0, red);
    // this workgroup's LayerNorm-1 rows must have reached L2 before its LDS-DMAs of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mark(2);
    if (a.stop == 1) return;

    // =============================================================== 2. FFN w_1 + ReLU (TM:111)
    if constexpr (FFN8) {
        // int8 slices, one pass into I8Acc pairs (a lone wave per SIMD has the registers): the integer sums — and so every
        // bit downstream — equal the two-pass one-accumulator form of the large-batch kernel (layer_tail_i8_kernel)
        constexpr int FP8 = NWV == 8 ? 1 : ((W2 || (TT == 1 && TAIL_RING1 == 8)) ? 2 : 1), FTP8 = FT / FP8;  // feature passes of the FFN contractions (two of 2 tiles where the ring is deep or the registers few)
        using G8 = DirectGemm<FTP8, TT, (TT == 2 ? 2 : (W2 ? 4 : (NWV == 8 ? TAIL8_RING : TAIL_RING1))), false, true, NWV>;  // TT = 2: 256 accumulator registers, so a 2-slot weight ring
        const EpiReluQ8<NWV, TOK> e8{a.relu8.bias, a.relu8.q8, a.relu8.q8_plane, a.relu8.q8_scale};
        auto ffn_gemm = [&](I8Acc (&q)[FT][TT], const int8_t* in8, size_t in_plane_bytes, const int8_t* w8, int mk) {
#pragma unroll
            for (int fp = 0; fp < FP8; ++fp)
                G8::run(*(I8Acc(*)[FTP8][TT]) & q[fp * FTP8], (const __bf16*)in8, in_plane_bytes / 2, 16, (const __bf16*)w8, a.w8_plane / 2, act, tt0, wave,
                        lane, [&] { mark(mk); }, wave * FT + fp * FTP8);
        };
        {
            I8Acc q[FT][TT];
            ffn_gemm(q, a.ln1.q8, a.ln1.q8_plane, a.w1_8, 8);
            mark(3);
            e8.template run<false, I8Acc, FT, TT>(q, p_sw1, a.ln1.q8_scale, wave * FT * 32, tok0, lane, wave, 0, red);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mark(4);
        if (a.stop == 2) return;
        // =========================================================== 3. FFN w_2 + residual + LayerNorm (TM:111-114, 139)
        {
            I8Acc q[FT][TT];
            ffn_gemm(q, a.relu8.q8, a.relu8.q8_plane, a.w2_8, 9);
            mark(5);
            i8_dequant_tile<false>(q, acc, p_sw2, a.relu8.q8_scale, wave * FT * 32, tok0, lane);
        }
        as_ln(a.ln2, 1).template run<FT, TT>(acc, wave * FT * 32, tok0, lane, wave, 0, red);
        EG_DBG(if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); })
        mark(6);
        return;
    }
    if constexpr (!FFN8) {
    G::run(acc, a.ln1.out, a.ln1.out_plane, 32, a.w1, a.w1_plane, act, tt0, wave, lane, [&] { mark(8); });
    mark(3);
    a.relu.template run<FT, TT>(acc, wave * FT * 32, tok0, lane, wave, 0, red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mark(4);
    if (a.stop == 2) return;

    // =============================================================== 3. FFN w_2 + residual + LayerNorm (TM:111-114, 139)
    G::run(acc, a.relu.out, a.relu.out_plane, 32, a.w2, a.w2_plane, act, tt0, wave, lane, [&] { mark(9); });
    mark(5);
    as_ln(a.ln2, 1).template run<FT, TT>(acc, wave * FT * 32, tok0, lane, wave, 0, red);
    EG_DBG(if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); })
    mark(6);
    }
}

// ---- embed and linear_out on the same operand path -----------------------------------------------------------------
// One 4-wave workgroup per 32 TT tokens, one per CU.  embed (TM:199-216): 512 features, a wave owns 128 of them (FT = 4),
// epilogue EpiEmbed (bias, position embedding, time token, optional int8 rows).  linear_out + DDPM posterior (M:139,
// 235-256): 256 (= padded 198) features, a wave owns 64 (FT = 2), epilogue EpiOut.  Same MFMA order per output element
// and same epilogue code as the LDS-ring kernels they replace: same bits.
struct EmbedArgs {
    const __bf16* x;   // embed operand [Mp][KE], split-bf16 fragment-tiled
    size_t x_plane;
    int K16;
    const __bf16* w;   // [512][KE]
    size_t w_plane;
    EpiEmbed<2, 4, 0> epi;
};
// NWV = 8: eight 256-register waves, 64 features each (like the eight-wave tail: for grids of at most one workgroup per CU)
template <int TT, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, (NWV == 8 ? 2 : 1)) void embed_kernel(EmbedArgs a) {
    constexpr int FT = 16 / NWV;
    using G = DirectGemm<FT, TT, TAIL_RING_IO, true, false, NWV>;  // the embed operand has 26 k-blocks (2 x 208 columns)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = wave_id_uniform(), lane = threadIdx.x & 63;
    f32x16 acc[FT][TT];
    G::run(acc, a.x, a.x_plane, a.K16, a.w, a.w_plane, smem, (int)blockIdx.x * TT, wave, lane, [] {});
    const EpiEmbed<2, NWV, 32 * TT> e{a.epi.bias, a.epi.pe, a.epi.tt_table, a.epi.t_idx, a.epi.out, a.epi.out_plane, a.epi.Lp, a.epi.T, a.epi.B,
                                      a.epi.q8, a.epi.q8_plane, a.epi.q8_scale, a.epi.state, a.epi.ts};
    e.template run<FT, TT>(acc, wave * FT * 32, (int)blockIdx.x * 32 * TT, lane, wave, 0, smem);
}

struct OutArgs {
    const __bf16* h;   // last layer's output [Mp][512]
    size_t h_plane;
    const __bf16* w;   // [256][512] (rows >= d_feats zero)
    size_t w_plane;
    EpiOut<2> epi;
    // I8 builds (precision 9's product path: the last layer hands its output over as int8 rows only): the same contraction on int8
    // slices — the rows with one scale each, linear_out's weights with one scale per output row
    const int8_t* h8;
    size_t h8_plane;   // bytes between the slices
    const float* h_scale;
    const int8_t* w8;  // [256][512] two slices (rows >= d_feats zero, scale 0)
    size_t w8_plane;
    const float* s_w;  // [256]
};
// FS = 2: grids of at most 128 token blocks (half the CUs) split the 256 output features over two workgroups per token block
// (a wave then owns 32 features): the posterior epilogue is elementwise, so nothing crosses the split.
template <int TT, int FS, bool I8 = false>
__global__ __launch_bounds__(256, 1) void out_kernel(OutArgs a) {
    constexpr int FT = 2 / FS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = wave_id_uniform(), lane = threadIdx.x & 63;
    const int tb = (int)blockIdx.x / FS, fh = (int)blockIdx.x % FS;
    f32x16 acc[FT][TT];
    if constexpr (I8) {
        using G = DirectGemm<FT, TT, TAIL_RING_IO, false, true>;
        I8Acc q[FT][TT];
        G::run(q, (const __bf16*)a.h8, a.h8_plane / 2, 16, (const __bf16*)a.w8, a.w8_plane / 2, smem, tb * TT, wave, lane, [] {}, (fh * 4 + wave) * FT);
        i8_dequant_tile<false>(q, acc, a.s_w, a.h_scale, (fh * 4 + wave) * FT * 32, tb * 32 * TT, lane);
    } else {
        using G = DirectGemm<FT, TT, TAIL_RING_IO>;
        G::run(acc, a.h, a.h_plane, 32, a.w, a.w_plane, smem, tb * TT, wave, lane, [] {}, (fh * 4 + wave) * FT);
    }
    if constexpr (FS == 1) {  // the workgroup owns whole rows: x staged through LDS (EpiOut::run_block; 32 TT rows of <= 256 floats fit the chunk buffers)
        a.epi.template run_block<FT, TT, 32 * TT, 256>(acc, wave * FT * 32, tb * 32 * TT, tb * 32 * TT, lane, smem);
    } else {
        a.epi.template run<FT, TT>(acc, (fh * 4 + wave) * FT * 32, tb * 32 * TT, lane, wave, 0, smem);
    }
}
