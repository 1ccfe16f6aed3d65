// egoego_hip.hip — context, workspace carving, launch sequence and the C ABI (include/egoego_hip.h)
// of the MI355X stage-2 diffusion sampling step.  gfx950 only.
#include "../../include/egoego_hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "attention.h"
#include "attn_layer_i8.h"
#include "attn_layer_i8w.h"
#include "attn_layer_i8h.h"
#include "attn_split_i8.h"
#include "attn_core_i8.h"
#include "attn_core_i8w.h"
#include "common.h"
#include "gemm.h"
#include "pointwise.h"
#include "tail_fused.h"

// hipcc drops the implicit instantiation of this kernel template when its only use sits in a nested branch of run_chunk_np
// (the host object then references an undefined kernel handle): instantiate it explicitly.
template __global__ void qkv_i8_kernel<EpiQK<2>, EpiV<2>>(QkvI8Args, EpiQK<2>, EpiV<2>);

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(EGOEGO_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                         \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------ context
struct LayerDev {
    __bf16 *w_qkv, *w_fc, *w_1, *w_2;  // fragment-tiled, 2 planes each
    int8_t* w_qkv8n;                   // i8x3 copy of w_qkv for attn_layer_i8w_kernel: two slices, K in acc32 order
    float* s_qkv;                      // its row scales [3*HD]
    int8_t *w_1_8, *w_2_8;             // i8x3 copies of the FFN weights: two slices of [512][512], K in acc32 order
    float *s_1, *s_2;                  // their row scales [512]
    int8_t* w_fc_8;                    // i8x3 copy of w_fc: two slices of [512][HD]
    int8_t* w_fc_3;                    // EGOEGO_FLAG_FC24: [third slice | zeros] of w_fc, or nullptr
    float* s_fc;                       // its row scales [512]
    float *b_qkv, *b_fc, *ln1_g, *ln1_b, *b_1, *b_2, *ln2_g, *ln2_b;
};

// One captured diffusion step (embed + layers + out).  Everything a kernel of the step receives by value is part of
// the key; what changes from step to step (the step index that selects the timestep, the injected-noise slice or the
// DDIM table row) and from call to call (the caller's x / noise / prefix buffers, the Philox key) lives in device
// memory (Workspace::state, a StepState), so ONE graph serves every step of every chain of that shape on that workspace.
struct StepKey {
    int B, T, mode, noise_mode, prefix_len, clip, ddim, has_mask;
    const void* ws;
    bool operator==(const StepKey& o) const {
        return B == o.B && T == o.T && mode == o.mode && noise_mode == o.noise_mode && prefix_len == o.prefix_len && clip == o.clip &&
               ddim == o.ddim && has_mask == o.has_mask && ws == o.ws;
    }
};
struct StepGraph {
    StepKey key;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

struct egoego_ctx {
    egoego_config cfg;
    int device;
    int D, DP, KE, H, HD, NOUT, S;
    std::vector<void*> allocs;
    __bf16 *w_embed, *w_out;
    int8_t* w_out_8;  // linear_out as int8 slices [NOUT][512] + row scales (precision 9's product path)
    float* s_out;
    float *b_embed, *b_out, *pe, *tt_table, *sched;
    std::vector<LayerDev> layers;
    bool have_weights, have_sched;
    std::vector<float> abar_host;
    // hipGraph replay of the per-step launch sequence (egoego_sample_loop / egoego_ddim_loop)
    hipStream_t cap_stream;
    std::vector<StepGraph> graphs;
    // pinned staging slots for the strided sampler's per-step tables (egoego_ddim_loop): a slot is reused only after the
    // copy that last read it has completed (its event), so the upload never makes the call wait for the stream
    static const int N_STAGE = 4;
    void* stage[N_STAGE];
    hipEvent_t stage_ev[N_STAGE];
    int stage_next;
    // profiling
    int prof_id;
    std::vector<hipEvent_t> prof_events;
    const char* last_kernel[EGOEGO_K_COUNT];  // the kernel variant each launch site of a step last dispatched to (egoego_last_kernel_name)
};

static const int N_MODEL = 512;
#ifdef EGOEGO_PERFDEBUG
// perf-debug build only (tools/*_trace.py): per-block timestamps and stage ablation, set through egoego_debug_*
static unsigned long long* g_trace = nullptr;
static int g_ablate = 0;
#endif

// the int8-slice precisions: EGOEGO_PREC_I8X3 (attention layer + FFN) and EGOEGO_PREC_I8X3_FC (the same + fc)
static inline bool prec_i8(const egoego_ctx* c) { return c->cfg.precision == EGOEGO_PREC_I8X3 || c->cfg.precision == EGOEGO_PREC_I8X3_FC; }

struct Geometry {
    int B, T, L, KT, Lp, Lr, Mp, Mvalid;  // Lp = 32 KT: tokens of a window's attention images; Lr: token ROWS per window (row stride)
};

// aligned: every window on 32-row tile boundaries (Lr = Lp) — what the split-bf16 attention kernels and the Q/K/V debug stops need.
// Otherwise long windows of the int8 precisions are packed on 16-row boundaries: 208 rows instead of 224 for 129..208 tokens
// (BASELINE configs[3]: T + 1 = 197), 7 % fewer rows through every row-parallel kernel; only the attention images keep 32-key tiles.
static int make_geometry(const egoego_ctx* c, int B, int T, Geometry& g, bool aligned = false) {
    if (B < 1 || T < 1) return fail(EGOEGO_E_INVALID, "B and T must be positive (B=%d, T=%d)", B, T);
    if (T > c->cfg.max_timesteps - 1)
        return fail(EGOEGO_E_INVALID, "window length T=%d exceeds max_timesteps-1=%d (position table rows, TM:180-182)", T,
                    c->cfg.max_timesteps - 1);
    g.B = B; g.T = T; g.L = T + 1;
    if (g.L <= 32) g.KT = 1;
    else if (g.L <= 64) g.KT = 2;
    else if (g.L <= 128) g.KT = 4;
    else if (g.L <= 224) g.KT = 7;
    else return fail(EGOEGO_E_INVALID, "window length T=%d not supported (T+1 must be <= 224)", T);
    g.Lp = 32 * g.KT;
    g.Lr = (!aligned && prec_i8(c) && g.KT == 7 && g.L <= 208) ? 208 : g.Lp;
    // 32-bit byte offsets into one operand plane (buffer-resource addressing, rows x 2 KiB at most): 2^20 padded rows per call
    if ((size_t)B * g.Lp > (size_t)1 << 20)
        return fail(EGOEGO_E_INVALID, "B=%d windows of %d padded rows exceed 1048576 rows per call: split the batch", B, g.Lp);
    g.Mvalid = B * g.Lr;
    g.Mp = (int)align_up((size_t)g.Mvalid, 256);
    return 0;
}

struct Workspace {
    StepState* state; // device-resident step state (common.h): a captured step replays for any t and any caller buffer
    int* step_ts;     // [S] explicit timestep list (DDIM)
    float* step_tab;  // [S][4] per-step DDIM coefficients: sqrt(abar_prev), dir, sig, -
    int* t_idx;
    float* row_mask;
    __bf16 *xall, *hA, *hB, *F, *Q, *K, *V, *O;
    size_t xall_plane, h_plane, qkv_plane, o_plane;
    int8_t* hA8;      // int8 slices of hA (i8x3 consumers), slice stride h_plane bytes
    float* hA_scale;  // [Mp]
    int8_t *hB8, *F8;           // i8x3 FFN: LayerNorm-1 output and ReLU output as int8 slices (same layout as hA8)
    float *hB_scale, *F_scale;  // [Mp] their row scales
    int8_t* O8;                 // i8x3 fc: the attention output as int8 slices, slice stride o_plane bytes
    float* O_scale;             // [Mp][H] one scale per row and head
    int8_t* att_img;         // image region of the split attention forms (see carve) or nullptr
    float *sq8, *sk8, *sv8;  // [B*H][Lp] row scales of the int8 Q / K / V images (attn_core_i8.h; the images alias Q, K, V)
    size_t total;
};

static void carve(const egoego_ctx* c, const Geometry& g, char* base, Workspace& w) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.state = (StepState*)take(sizeof(StepState));
    w.step_ts = (int*)take(sizeof(int) * c->S);
    w.step_tab = (float*)take(sizeof(float) * 4 * c->S);
    w.t_idx = (int*)take(sizeof(int) * g.B);
    w.row_mask = (float*)take(sizeof(float) * g.Mp);
    w.xall_plane = (size_t)g.Mp * c->KE;
    w.h_plane = (size_t)g.Mp * N_MODEL;
    w.qkv_plane = (size_t)g.B * c->H * g.Lp * 256;
    w.o_plane = (size_t)g.Mp * c->HD;
    w.xall = (__bf16*)take(2 * w.xall_plane * 2);
    w.hA = (__bf16*)take(2 * w.h_plane * 2);
    w.hB = (__bf16*)take(2 * w.h_plane * 2);
    w.F = (__bf16*)take(2 * w.h_plane * 2);
    w.Q = (__bf16*)take(2 * w.qkv_plane * 2);
    w.K = (__bf16*)take(2 * w.qkv_plane * 2);
    w.V = (__bf16*)take(2 * w.qkv_plane * 2);
    // The split attention forms (attn_split_i8.h; Lp = 128 only) pass their int8 images through the Q and K buffers as ONE region of
    // [B*H][3][64 KiB] — Q and K are carved back to back and hold 256 KiB per window x head between them — and V's column scales
    // ([B*H][256] floats) through the V buffer.  nullptr if that ever stops being true: the dispatch then keeps to the one-kernel forms.
    w.att_img = ((char*)w.K == (char*)w.Q + 2 * w.qkv_plane * 2 && g.Lp == 128) ? (int8_t*)w.Q : nullptr;
    w.O = (__bf16*)take(2 * w.o_plane * 2);
    w.hA8 = (int8_t*)take(2 * w.h_plane);
    w.hA_scale = (float*)take(sizeof(float) * g.Mp);
    w.hB8 = (int8_t*)take(2 * w.h_plane);
    w.hB_scale = (float*)take(sizeof(float) * g.Mp);
    w.F8 = (int8_t*)take(2 * w.h_plane);
    w.F_scale = (float*)take(sizeof(float) * g.Mp);
    w.O8 = (int8_t*)take(2 * w.o_plane);
    w.O_scale = (float*)take(sizeof(float) * (size_t)g.Mp * c->H);
    w.sq8 = (float*)take(sizeof(float) * (size_t)g.B * c->H * g.Lp);
    w.sk8 = (float*)take(sizeof(float) * (size_t)g.B * c->H * g.Lp);
    w.sv8 = (float*)take(sizeof(float) * (size_t)g.B * c->H * g.Lp);
    w.total = off;
}

// ------------------------------------------------------------------------------------ fused QKV + attention
// One workgroup per (window, head): the three 256-feature projection blocks Q_h, K_h, V_h of that window
// (same main loop and epilogues as qkv_kernel), then attention over them.  The operands attention reads
// were written microseconds earlier by the same CU, so they come back from L2 / Infinity Cache instead of
// HBM, the attention launch disappears, and — since the two workgroups resident on a CU drift apart — one
// workgroup's HBM/L2-bound attention phases overlap the other's MFMA-bound projections.
// Valid when a window is exactly one token block of the QKV tile (Lp == 128).
template <class CQK, class EQK, class CV, class EV, class CQ, int KT, int NP>
__global__ __launch_bounds__(CQK::NT, CQK::MINW) void qkv_attn_kernel(GemmOperands g, EQK eqk, EV ev, AttnArgs a, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the H heads of a window share an XCD (and its L2)
    const int bh = lid + a.bh0;
    const int b = bh / H, h = bh - b * H;
    EG_DBG(unsigned long long* tr = g.trace ? g.trace + 131072 + (size_t)blockIdx.x * 8 : nullptr;
           if (tr && threadIdx.x == 0) tr[0] = wall_clock64();)
    GemmBody<CQK, EQK>::run(g, eqk, H + h, b, smem);     // K_h -> global (accumulator order along d_k)
    EG_DBG(if (tr && threadIdx.x == 0) tr[1] = wall_clock64();)
    GemmBody<CV, EV>::run(g, ev, 2 * H + h, b, smem);    // V_h -> global (transposed, key-permuted)
    EG_DBG(if (tr && threadIdx.x == 0) tr[2] = wall_clock64();)
    // Q_h last, with waves laid out 1(f) x 4(t): wave w ends up holding all 256 d_k of its 32 queries.
    // It never goes to memory: bias, 1/sqrt(d_k), split-bf16 — and the accumulator registers 8jj..8jj+7 of
    // feature tile i ARE the B-operand fragment of k-step 2i+jj in the order K was stored in.
    static_assert(CQ::FT == 8 && CQ::TT == 1 && CQ::NWF == 1 && CQ::NWT == 4, "Q layout: 256 features x 32 queries per wave");
    bf16x8 qh[16], ql[16];
    {
        f32x16 acc[CQ::FT][CQ::TT];
        GemmBody<CQ, EQK>::mainloop(g, h, b, smem, acc);
        const int lane = threadIdx.x & 63, hf = lane >> 5;
        const float* bias = eqk.bias + h * 256;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4 b0 = *(const float4*)(bias + 32 * i + 16 * jj + 4 * hf);
                const float4 b1 = *(const float4*)(bias + 32 * i + 16 * jj + 4 * hf + 8);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (acc[i][0][8 * jj + e] + bb[e]) * eqk.qscale;
                    const __bf16 x = (__bf16)v;
                    qh[2 * i + jj][e] = x;
                    ql[2 * i + jj][e] = (__bf16)(v - (float)x);
                }
            }
    }
    // this workgroup's own K/V stores must have reached L2 before its LDS-DMAs of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    EG_DBG(if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
           if (g.ablate & 4) return;)
    attn_body<KT, NP, true>(a, bh, 0, smem, qh, ql);
    EG_DBG(if (tr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) tr[4] = wall_clock64();
    })
}

// ------------------------------------------------------------------------------------ fused layer tail
// One workgroup takes a block of tokens through fc+residual+LayerNorm -> FFN-1+ReLU -> FFN-2+residual+LayerNorm.  Two tilings of
// the same arithmetic (same bits): 64-token blocks with 4 waves and two workgroups per CU, so that one's epilogues and barrier waits
// overlap the other's main loops; from one workgroup per CU on (round 5), 128-token blocks with 8 waves, which stream the weights half
// as often (the dispatch site has the measurements).
// Rows are independent, so the tile a phase reads is exactly the tile the previous phase of the SAME workgroup
// wrote: it comes back from L2 instead of HBM, and two kernel boundaries per layer disappear.
template <class C, class ELN, class ETI>
__global__ __launch_bounds__(C::NT, C::MINW) void layer_tail_kernel(GemmOperands g_fc, ELN e_fc, GemmOperands g_1, ETI e_1,
                                                                      GemmOperands g_2, ELN e_2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tblk = (int)blockIdx.x + g_fc.tblk0;
    GemmBody<C, ELN>::run(g_fc, e_fc, 0, tblk, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GemmBody<C, ETI>::run(g_1, e_1, 0, tblk, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GemmBody<C, ELN>::run(g_2, e_2, 0, tblk, smem);
}

// The same with the two FFN contractions on int8 slices ("i8x3", the default precision).  A 256-register wave cannot
// hold the I8Acc pair of its 128-feature x 64-token tile, so each contraction runs as TWO passes over K into ONE int32
// accumulator (gemm.h I8One): high slices only (half the operand stream), shift by 8, then the two cross terms on top —
// the exact integer the one-pass form of tail_kernel produces, hence the same bits downstream.  Against split-bf16 the
// FFN main loops issue half the MFMAs and stream 3/4 of the operand bytes; LayerNorm-1 also emits its rows as int8
// slices (the operand of FFN-1) and FFN-1's epilogue quantises the ReLU output per row (the operand of FFN-2), so the
// hidden activations cross memory as 2 instead of 4 bytes per value.
// stop (debug taps): 1 = return after LayerNorm-1, 2 = after FFN-1.
template <class C, class C8a, class C8b, class ELN, class ERQ>
__global__ __launch_bounds__(C::NT, C::MINW) void layer_tail_i8_kernel(GemmOperands g_fc, ELN e_fc, GemmOperands g_1, const float* sw1, ERQ e_1,
                                                                         GemmOperands g_2, const float* sw2, ELN e_2, int stop) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tblk = (int)blockIdx.x + g_fc.tblk0;
    // The per-feature parameters of the int8 epilogues (weight row scales of both convs, the first conv's bias) are staged
    // behind the ring once: the epilogues read them tile by tile between fences, and an LDS read costs them a tenth of the
    // L2 round trip.  Visible after the fc main loop's first barrier.
    float* par = (float*)(smem + C::SMEM_BYTES);  // [3][512]: sw1, bias1, sw2
    for (int i = threadIdx.x; i < 384; i += C::NT) {
        const float* src = i < 128 ? sw1 + 4 * i : (i < 256 ? e_1.bias + 4 * (i - 128) : sw2 + 4 * (i - 256));
        *(float4*)(par + 4 * i) = *(const float4*)src;
    }
    GemmBody<C, ELN>::run(g_fc, e_fc, 0, tblk, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stop == 1) return;
    ERQ e_1l = e_1;
    e_1l.bias = par + 512;
    const int wave = wave_id_uniform(), lane = threadIdx.x & 63;
    const int wf = wave % C::NWF, wt = wave / C::NWF;
    const int f0 = wf * C::FT * 32, t0 = (tblk * C::AT + wt * C::TT) * 32;
    auto i8_gemm = [&](const GemmOperands& g8, I8One (&q)[C::FT][C::TT]) {
        GemmBody<C8a, ELN>::template mainloop<I8One, true>(g8, 0, tblk, smem, q);
#pragma unroll
        for (int i = 0; i < C::FT; ++i)
#pragma unroll
            for (int j = 0; j < C::TT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) q[i][j].v[r] <<= 8;
        GemmBody<C8b, ELN>::template mainloop<I8One, false>(g8, 0, tblk, smem, q);
    };
    {
        I8One q[C::FT][C::TT];
        i8_gemm(g_1, q);
        e_1l.template run<true, I8One, C::FT, C::TT>(q, par, e_fc.q8_scale, f0, t0, lane, wf, wt, smem);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stop == 2) return;
    {
        // The integer sums are dequantised IN PLACE, tile by tile (an fp32 value takes its int32's register), and the LayerNorm epilogue reads
        // the same storage as floats: a separate fp32 tile next to the integer one is 2 x 128 registers of a 256-register wave (round 4:
        // 83 spilled registers, 264 bytes of scratch — and a launch slower than the all-split-bf16 tail's).
        I8One q[C::FT][C::TT];
        i8_gemm(g_2, q);
        {
            const int hf = lane >> 5, col = lane & 31;
            float sa[C::TT];
#pragma unroll
            for (int j = 0; j < C::TT; ++j) sa[j] = e_1.q8_scale[t0 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < C::FT; ++i) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < C::TT; ++j) {
                    f32x16 o;
                    i8_dequant(q[i][j], o, par + 1024 + f0 + i * 32 + 4 * hf, sa[j]);
                    q[i][j].v = __builtin_bit_cast(i32x16, o);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        e_2.template run<C::FT, C::TT>(reinterpret_cast<f32x16(&)[C::FT][C::TT]>(q), f0, t0, lane, wf, wt, smem);
    }
}
// the int8 FFN passes of the 512f x 64t, 4-wave tile: pass 1 stages the high slices of two k-blocks per ring stage, pass 2 both
// slices of one — 36 KiB per stage either way, the ring of the split-bf16 phase
using CfgT8a = GemmCfg<4, 2, 4, 1, 2, 1, false, 2, 2>;
using CfgT8b = GemmCfg<4, 2, 4, 1, 1, 2, false, 2, 2>;

// ------------------------------------------------------------------------------------ launch helpers
struct ProfScope {
    egoego_ctx* c;
    hipStream_t s;
    bool on;
    ProfScope(egoego_ctx* c_, int id, hipStream_t s_) : c(c_), s(s_), on(c_->prof_id == id) {
        if (on) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            (void)hipEventRecord(e, s);
            c->prof_events.push_back(e);
        }
    }
    ~ProfScope() {
        if (on) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            (void)hipEventRecord(e, s);
            c->prof_events.push_back(e);
        }
    }
};

template <class K>
static hipError_t allow_smem(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
// hipFuncSetAttribute applies to the CURRENT device's copy of the kernel, so the dynamic-LDS opt-in of a launch site is done once
// per device a context lives on (the header allows one context per device), not once per process.  Every entry point has called
// hipSetDevice(ctx->device) by the time a launch site runs.
struct DevOnce {
    unsigned long long mask = 0;  // devices (hipGetDevice ordinal, < 64) this launch site has opted in on
    int dev = 0;
    bool pending() {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        return !(mask & (1ull << (dev & 63)));
    }
    void done() { mask |= 1ull << (dev & 63); }
};

// GEMM tile configurations (features x tokens per block), chosen by measurement on MI355X
// (tools/kernel_times.py, tools/block_trace.py, EGOEGO_ABLATE):
//   A: 256 x 128, 4 waves 2(f) x 2(t), 3-stage LDS ring (72 KiB), two workgroups per CU so one block's
//      store-heavy epilogue overlaps the other's main loop              — embed, QKV, FFN-1
//   B: 512 x 128, 8 waves 4(f) x 2(t), 3-stage ring (120 KiB)            — fc / FFN-2 + residual + LayerNorm
//   C: 256 x 128, 8 waves of 64f x 64t, 2-stage ring                      — linear_out + DDPM tail
// Every wave owns a 128f x 64t (C: 64f x 64t) accumulator tile.
template <int NP> using CfgA = GemmCfg<4, 2, 2, 2, 1, NP, false, 2, 3>;
template <int NP> using CfgAV = GemmCfg<4, 2, 2, 2, 1, NP, true, 2, 3>;
template <int NP> using CfgB = GemmCfg<4, 2, 4, 2, (NP == 2 ? 1 : 2), NP, false, 1, 3>;
template <int NP> using CfgC = GemmCfg<2, 2, 4, 2, 2, NP, false>;
// B for small batches: 512 x 64, 4 waves, 2-stage ring, two workgroups per CU — twice the blocks of B
template <int NP> using CfgBs = GemmCfg<4, 2, 4, 1, (NP == 2 ? 1 : 2), NP, false, 2, 2>;
// Small grids (fewer workgroups than CUs at the tiles above): the same contractions on smaller tiles, so that more CUs
// take part and each workgroup's k-loop carries less work per step.  Same k order per output element -> same numbers.
template <int NP> using CfgBt = GemmCfg<4, 1, 4, 1, (NP == 2 ? 2 : 4), NP, false, 1, 2>;  // 512f x 32t, 4 waves
template <int NP> using CfgAh = GemmCfg<2, 2, 2, 2, 1, NP, false, 2, 3>;                   // 128f x 128t, 4 waves
template <int NP> using CfgC2 = GemmCfg<2, 2, 4, 1, 2, NP, false, 2, 2>;                   // 256f x 64t, 4 waves
static const int SMALL_GRID = 160;  // workgroups
template <int NP> using CfgQ = GemmCfg<8, 1, 1, 4, 1, NP, false, 2, 3>;  // fused kernel's Q projection: 256f x 32t per wave
static const int BLK_A_F = 256, BLK_A_T = 128, BLK_B_T = 128;

template <class C, class Epi>
static int launch_gemm(const GemmOperands& g, const Epi& epi, hipStream_t s) {
    auto kern = gemm_kernel<C, Epi>;
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem(kern, C::SMEM_BYTES));
        once.done();
    }
    kern<<<dim3(g.nfb * g.ntb), dim3(C::NT), C::SMEM_BYTES, s>>>(g, epi);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int KT, int NP>
static int launch_attn_kt(const AttnArgs& a, int BH, hipStream_t s) {
    auto kern = attn_kernel<KT, NP>;
    constexpr int smem = attn_smem<KT, NP>();
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem(kern, smem));
        once.done();
    }
    kern<<<dim3((KT + 3) / 4, BH), dim3(256), smem, s>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the long window in split-bf16: one eight-wave workgroup per (window, head) with a four-slot ring (attention.h, round 5);
// EGOEGO_ATTN_WG4=1 (variant builds) keeps the two four-wave workgroups for A/B runs
#ifndef EGOEGO_ATTN_WG4
#define EGOEGO_ATTN_WG4 0
#endif
template <int KT>
static int launch_attn8_kt(const AttnArgs& a, int BH, hipStream_t s) {
    auto kern = attn8_kernel<KT, 2>;
    constexpr int smem = 4 * KT * 2 * 2 * 1024;
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem(kern, smem));
        once.done();
    }
    kern<<<dim3(BH), dim3(512), smem, s>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NP>
static int launch_attn(const AttnArgs& a, int KT, int BH, hipStream_t s) {
    if constexpr (NP == 2) {
        // (same bits either way; below one workgroup per CU the two four-wave workgroups per (window, head) spread over twice the CUs:
        // B=1 0.447 against 0.453 ms per step)
        if (KT == 7 && !EGOEGO_ATTN_WG4 && BH >= 256) return launch_attn8_kt<7>(a, BH, s);
    }
    switch (KT) {
        case 1: return launch_attn_kt<1, NP>(a, BH, s);
        case 2: return launch_attn_kt<2, NP>(a, BH, s);
        case 4: return launch_attn_kt<4, NP>(a, BH, s);
        case 7: return launch_attn_kt<7, NP>(a, BH, s);
    }
    return fail(EGOEGO_E_INVALID, "unsupported key-tile count %d", KT);
}

// Fused small-batch tail (tail_fused.h): 64-token workgroups when they fill the CUs exactly once, 32-token ones below.
// -DTAIL8_BF16=0 (variant builds): the split-bf16 tail of small grids on four 512-register waves, round 4's form
#ifndef TAIL8_BF16
#define TAIL8_BF16 1
#endif
#ifndef TAIL8_BF16_MAX_BLOCKS
#define TAIL8_BF16_MAX_BLOCKS 160  // 32-token blocks up to which the split-bf16 tail runs as eight-wave workgroups (well under one per CU)
#endif
template <bool FFN8>
static int launch_tail_f(egoego_ctx* c, const TailArgs& ta, int rows, hipStream_t s) {
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem((tail_kernel<2, FFN8, false>), tail_smem_bytes(2)));
        HIP_TRY(allow_smem((tail_kernel<1, FFN8, false>), tail_smem_bytes(1)));
        HIP_TRY(allow_smem((tail_kernel<1, false, false, false, 8>), tail_smem_bytes(1, 8)));
        once.done();
    }
    if (rows / 64 >= 256) {
        c->last_kernel[EGOEGO_K_FC_LN] = FFN8 ? "tail_kernel<2,true,false>" : "tail_kernel<2,false,false>";
        tail_kernel<2, FFN8, false><<<dim3(rows / 64), dim3(256), tail_smem_bytes(2), s>>>(ta);
    } else if (!FFN8 && TAIL8_BF16 && rows / 32 <= TAIL8_BF16_MAX_BLOCKS) {
        // well under one workgroup per CU: the eight-wave build (two 256-register waves per SIMD, 64 features each), like the all-int8
        // tail's — a workgroup is a serial chain there, and a SIMD's second wave issues its weight loads and MFMAs in the first one's waits
        // (round 5, ms per step in split-bf16 at B = 1 / 8 / 32: 0.408 / 0.413 / 0.522 against 0.429 / 0.431 / 0.542 on four waves; a tie at 64 windows)
        c->last_kernel[EGOEGO_K_FC_LN] = "tail_kernel<1,false,false,false,8>";
        tail_kernel<1, false, false, false, 8><<<dim3(rows / 32), dim3(512), tail_smem_bytes(1, 8), s>>>(ta);
    } else {
        c->last_kernel[EGOEGO_K_FC_LN] = FFN8 ? "tail_kernel<1,true,false>" : "tail_kernel<1,false,false>";
        tail_kernel<1, FFN8, false><<<dim3(rows / 32), dim3(256), tail_smem_bytes(1), s>>>(ta);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
// two half-query workgroups per (window, head) while they fill at most three quarters of the CUs (measured, ms per step at
// B = 1 / 16 / 20 / 24 / 32: 0.269 / 0.268 / 0.272 / 0.287 / 0.303 with them, 0.292 / 0.295 / 0.294 / 0.296 / 0.297 without: with every CU
// busy the redundant K / V projections cost more than the shorter chain saves)
// the projections as three workgroups per (window, head) + a core launch (attn_split_i8.h) while those fit the chip at once
// (measured, ms per step at B = 1 / 8 / 12 / 16 / 20 / 21: 0.232 / 0.235 / 0.237 / 0.247 / 0.265 / 0.273 against the half-query
// kernel's 0.266 / 0.265 / 0.265 / 0.266 / 0.270 / 0.273; in two rounds — B = 24 / 32 / 40 — it loses: 0.304 / 0.333 / 0.358 against
// 0.283 / 0.297 / 0.312)
#ifndef ATTN_SPLIT_MAX_BLOCKS
#define ATTN_SPLIT_MAX_BLOCKS 256
#endif
// ... and as SIX four-wave workgroups per (window, head) while each has a CU of its own (up to 10 windows; measured at B = 1 / 8 /
// 16 / 20: 0.222 / 0.225 / 0.256 / 0.266 ms against the three-workgroup form's 0.232 / 0.235 / 0.247 / 0.265; at 22 windows and
// more, two per CU in 1.5+ rounds, it loses to the one-kernel forms: 0.285 / 0.312 / 0.350 against 0.270 / 0.297 / 0.309 at
// B = 22 / 32 / 40)
#ifndef ATTN_SPLIT6_MAX_BLOCKS
#define ATTN_SPLIT6_MAX_BLOCKS 256
#endif
// ... and as TWO eight-wave workgroups per (window, head), 1.5 projections each, + the core launch for 22..32 windows (one round of
// the chip where the three-workgroup form takes 1.5 and the one-kernel forms leave half of it idle)
// (measured, round 4, ms per step at B = 22 / 24 / 28 / 32: 0.272 / 0.281 / 0.305 / 0.321 with it, 0.279 / 0.286 / 0.302 / 0.307 without:
// with every CU busy the images' L2 round trip and the fuller chip's clock cost more than the shorter chain saves — it replaces the
// half-query form for 22..24 windows and stops there)
#ifndef ATTN_SPLIT2_MAX_BLOCKS
#define ATTN_SPLIT2_MAX_BLOCKS 192
#endif
// (round 4 also built the one-kernel layer with the projections' weights global -> VGPR and 8 (f) x 1 (t) waves — bit-identical, and
// SLOWER: 1.445 against 1.368 ms per step at B=256, 0.331 against 0.304 at B=32; git history "attn_layer_i8x.h", HISTORY.md R4)
#ifndef ATTN_HALF_MAX_BLOCKS
#define ATTN_HALF_MAX_BLOCKS 192
#endif
#ifndef EMBED8_MAX_BLOCKS
#define EMBED8_MAX_BLOCKS 256
#endif
#ifndef TAIL8_MAX_BLOCKS
#define TAIL8_MAX_BLOCKS 256
#endif
static int launch_tail(egoego_ctx* c, const TailArgs& ta, int rows, hipStream_t s, bool resident = false) {
    if (ta.fc8) {
        // int8 fc: 32-token workgroups at every batch size.  (64-token ones — tail_kernel<2, true, true>, fc in two feature
        // passes, one workgroup per CU — measured slower: 250 against 235 us per launch at B=256, 0.94 against 0.80 ms per step at
        // B=128: what a large grid needs is more bytes in flight per CU, not fewer bytes per token.)
        static DevOnce once;
        if (once.pending()) {
            HIP_TRY(allow_smem((tail_kernel<1, true, true, true>), tail_smem_bytes(1) + TAIL_PAR_BYTES));
            HIP_TRY(allow_smem((tail_kernel<1, true, true, false, 8>), tail_smem_bytes(1, 8) + TAIL_PAR_BYTES));
            HIP_TRY(allow_smem((tail_kernel<1, true, true, true, 4, true>), tail_smem_bytes(1) + TAIL_PAR_BYTES + TAIL_RES_BYTES));
            HIP_TRY(allow_smem((tail_kernel<1, true, true, false, 8, true>), tail_smem_bytes(1, 8) + TAIL_PAR_BYTES + TAIL_RES_BYTES));
            once.done();
        }
        if (resident) {  // precision 9's product path: the FFN operands never leave the CU (tail_fused.h RES)
            if (rows / 32 > TAIL8_MAX_BLOCKS) {
                c->last_kernel[EGOEGO_K_FC_LN] = "tail_kernel<1,true,true,true,4,true>";
                tail_kernel<1, true, true, true, 4, true><<<dim3(rows / 32), dim3(256), tail_smem_bytes(1) + TAIL_PAR_BYTES + TAIL_RES_BYTES, s>>>(ta);
            } else {
                c->last_kernel[EGOEGO_K_FC_LN] = "tail_kernel<1,true,true,false,8,true>";
                tail_kernel<1, true, true, false, 8, true><<<dim3(rows / 32), dim3(512), tail_smem_bytes(1, 8) + TAIL_PAR_BYTES + TAIL_RES_BYTES, s>>>(ta);
            }
            HIP_TRY(hipGetLastError());
            return 0;
        }
        // more than one workgroup per CU: the 256-register four-wave build, two workgroups per CU (measured at B=256: 190 against
        // 235 us per launch of the 512-register four-wave build).  At most one workgroup per CU: the eight-wave build, the same
        // 256-register waves as ONE workgroup (measured against the four-wave builds, which tie there: 31.5 against 39.5 us per
        // launch at B=32, 0.316 against 0.344 ms per step; 0.387 against 0.415 at B=64; 0.304 against 0.332 at B=1)
        if (rows / 32 > TAIL8_MAX_BLOCKS) {
            c->last_kernel[EGOEGO_K_FC_LN] = "tail_kernel<1,true,true,true>";
            tail_kernel<1, true, true, true><<<dim3(rows / 32), dim3(256), tail_smem_bytes(1) + TAIL_PAR_BYTES, s>>>(ta);
        } else {
            c->last_kernel[EGOEGO_K_FC_LN] = "tail_kernel<1,true,true,false,8>";
            tail_kernel<1, true, true, false, 8><<<dim3(rows / 32), dim3(512), tail_smem_bytes(1, 8) + TAIL_PAR_BYTES, s>>>(ta);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    return ta.ffn8 ? launch_tail_f<true>(c, ta, rows, s) : launch_tail_f<false>(c, ta, rows, s);
}

// The direct-operand embed / linear_out kernels run 32-token workgroups (TT = 1) at every size they are used for
// (64-token ones measured slower at B=128: 46 / 49 us against 40 / 38).

template <int TT>
static int launch_embed_tt(const EmbedArgs& ea, int rows, hipStream_t s) {
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem((embed_kernel<TT, 4>), TT * 32 * 1024));
        HIP_TRY(allow_smem((embed_kernel<TT, 8>), TT * 32 * 1024));
        once.done();
    }
    if (rows / (32 * TT) <= EMBED8_MAX_BLOCKS)  // at most one workgroup per CU: eight waves
        embed_kernel<TT, 8><<<dim3(rows / (32 * TT)), dim3(512), TT * 32 * 1024, s>>>(ea);
    else
        embed_kernel<TT, 4><<<dim3(rows / (32 * TT)), dim3(256), TT * 32 * 1024, s>>>(ea);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <int TT, bool I8>
static int launch_out_tt(const OutArgs& oa, int rows, hipStream_t s) {
    static DevOnce once;
    if (once.pending()) {
        HIP_TRY(allow_smem((out_kernel<TT, 1, I8>), TT * 32 * 1024));
        HIP_TRY(allow_smem((out_kernel<TT, 2, I8>), TT * 32 * 1024));
        once.done();
    }
    const int nb = rows / (32 * TT);
    if (nb <= 128)  // fewer token blocks than half the CUs: two workgroups per token block, 128 features each
        out_kernel<TT, 2, I8><<<dim3(2 * nb), dim3(256), TT * 32 * 1024, s>>>(oa);
    else
        out_kernel<TT, 1, I8><<<dim3(nb), dim3(256), TT * 32 * 1024, s>>>(oa);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int KT, bool O8>
static int launch_attn_core8_kt(AttnCore8Args a, int BH, hipStream_t s) {
    auto kern = attn_core_i8_kernel<KT, O8>;
    constexpr int smem = 2 * (2 * KT * 4 * 1024);  // two buffers of half an image (both slices); the key scales are static LDS
    static DevOnce once;
    static int n_cu[64];  // compute units per device: the persistent grid (one workgroup per CU)
    if (once.pending()) {
        HIP_TRY(allow_smem(kern, smem));
        int cu = 0;
        HIP_TRY(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, once.dev));
        n_cu[once.dev & 63] = cu;
        once.done();
    }
    const int dev = once.dev;
    a.BH = BH;
    const int items = BH * ((KT + 3) / 4);
    kern<<<dim3(std::min(items, std::max(1, n_cu[dev & 63]))), dim3(256), smem, s>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}
// the eight-wave form (attn_core_i8w.h, round 5); EGOEGO_CORE4=1 (variant builds) keeps the four-wave form for A/B runs
#ifndef EGOEGO_CORE4
#define EGOEGO_CORE4 0
#endif
template <int KT, bool O8>
static int launch_attn_core8w_kt(AttnCore8Args a, int BH, hipStream_t s) {
    auto kern = attn_core_i8w_kernel<KT, O8>;
    constexpr int smem = attn_core8w_smem_bytes<KT>();
    static DevOnce once;
    static int n_cu[64];  // compute units per device: the persistent grid (one workgroup per CU)
    if (once.pending()) {
        HIP_TRY(allow_smem(kern, smem));
        int cu = 0;
        HIP_TRY(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, once.dev));
        n_cu[once.dev & 63] = cu;
        once.done();
    }
    const int dev = once.dev;
    a.BH = BH;
    const int items = BH * ((KT + 3) / 4);
    kern<<<dim3(std::min(items, std::max(1, n_cu[dev & 63]))), dim3(512), smem, s>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int launch_attn_core8(const AttnCore8Args& a, int KT, int BH, hipStream_t s) {
    // (only windows of 129..224 tokens take this path: seven key tiles; shorter ones run the one-kernel layer or the split-bf16 core)
    if (KT != 7) return fail(EGOEGO_E_INVALID, "unsupported key-tile count %d", KT);
    if (EGOEGO_CORE4) return a.o8 ? launch_attn_core8_kt<7, true>(a, BH, s) : launch_attn_core8_kt<7, false>(a, BH, s);
    return a.o8 ? launch_attn_core8w_kt<7, true>(a, BH, s) : launch_attn_core8w_kt<7, false>(a, BH, s);
}

// ------------------------------------------------------------------------------------ the step
struct StepIO {
    StepState* state;       // multi-step loops: the device-resident step state (nullptr: single step, timesteps from t_idx)
    const int* ts;          // explicit timestep list of a strided sampler or nullptr
    const float* row_mask;  // packed [Mp] or nullptr
    int stop_layer, stop_stage;  // debug early exit (-1: run everything)
    bool run_out;
    OutParams out;
};

// One denoiser pass over windows [w0, w0 + nw).  Token rows of a chunk are contiguous, so a chunk is just
// a token-block offset for the GEMMs and a (batch, head) offset for attention.
template <int NP>
static int run_chunk_np(egoego_ctx* c, const Geometry& g, const Workspace& w, const StepIO& io, hipStream_t s, int w0,
                        int nw) {
    const int H = c->H, HD = c->HD;
    const int row0 = w0 * g.Lr;
    int rows = nw * g.Lr;
    if (w0 + nw >= g.B) rows = g.Mp - row0;  // the last chunk also carries the rows that pad Mp to the block size
    const int tb_a = rows / BLK_A_T, tb_b = rows / BLK_B_T, tb_c = rows / CfgC<NP>::BT;
    const int t0_a = row0 / BLK_A_T, t0_b = row0 / BLK_B_T, t0_c = row0 / CfgC<NP>::BT;
    const bool small_ln = tb_b < 200;  // below ~400 64-token blocks three separate kernels beat the fused tail (split-bf16 precision; measured B = 32..192)
    const bool direct_io = rows / 64 <= 256;  // embed / linear_out on the direct-operand kernels up to 128 windows of 128 rows (measured: B=128 40 / 38 us against 50 / 55 on the ring kernels, B=256 80 / 72 against 67 / 69)
    // embed_kernel's chunking (DirectGemm REM2: chunks of 8 k-blocks + a last chunk of 2) fits d_feats = 198's 26 k-blocks;
    // any other operand width stays on the ring kernels, which take every K
    const bool direct_embed = direct_io && (c->KE / 16) % 8 == 2;
    // --- embed: start_conv + time token + position embedding (TM:199-216)
    // i8x3: windows of 65..128 tokens go through the int8-slice attention-layer kernel at any batch size, and its first
    // layer reads the embed output as int8 rows
    const bool i8_path = NP == 2 && prec_i8(c);  // every layer input also as int8 rows
    // precision 9, windows of more than 64 tokens: a layer's inter-kernel activations exist as int8 rows ONLY (residuals are
    // rebuilt from them); the split-bf16 copies are written just for the debug stops and, after the last layer, for linear_out
    // (the Q/K/V debug stops run the split-bf16 projections, which read the split-bf16 embed rows; every other stop taps the int8 rows
    // of the product path itself: egoego_debug_stage)
    const bool dbg_qkv_any = io.stop_stage == EGOEGO_DBG_Q || io.stop_stage == EGOEGO_DBG_K || io.stop_stage == EGOEGO_DBG_V;
    const bool act8_only = i8_path && c->cfg.precision == EGOEGO_PREC_I8X3_FC && (g.KT == 4 || g.KT == 7) && !dbg_qkv_any;
    __bf16* const embed_out = act8_only ? nullptr : w.hA;
    {
        ProfScope ps(c, EGOEGO_K_EMBED, s);
        if (NP == 2 && direct_embed) {
            // small grids: weights streamed into registers, activations by LDS-DMA chunks (tail_fused.h), same arithmetic
            EmbedArgs ea{w.xall, w.xall_plane, c->KE / 16, c->w_embed, (size_t)N_MODEL * c->KE,
                         EpiEmbed<2, 4, 0>{c->b_embed, c->pe, c->tt_table, w.t_idx, embed_out, w.h_plane, g.Lr, g.T, g.B, i8_path ? w.hA8 : nullptr,
                                           w.h_plane, i8_path ? w.hA_scale : nullptr, io.state, io.ts}};
            c->last_kernel[EGOEGO_K_EMBED] = "embed_kernel";
            if (int r = launch_embed_tt<1>(ea, rows, s)) return r;
        } else if (i8_path) {
            // 512-feature x 128-token blocks, eight waves, one workgroup per CU (the epilogue sees whole rows and also writes them as
            // int8 slices): the 16 weight tiles of a k-step are staged once for 4 token tiles instead of once per 2 (64-token
            // blocks, two per CU: 53.7 us per launch at B=256 against 51.8)
            GemmOperands go{c->w_embed, (size_t)N_MODEL * c->KE, w.xall, w.xall_plane, c->KE / 16, 1, rows / 128, row0 / 128 EG_DBG(, g_ablate, g_trace)};
            EpiEmbed<NP, 4, 128> e{c->b_embed, c->pe, c->tt_table, w.t_idx, embed_out, w.h_plane, g.Lr, g.T, g.B, w.hA8, w.h_plane, w.hA_scale, io.state, io.ts};
            c->last_kernel[EGOEGO_K_EMBED] = "gemm_kernel:EpiEmbed";
            if (int r = launch_gemm<CfgB<NP>>(go, e, s)) return r;
        } else {
            c->last_kernel[EGOEGO_K_EMBED] = "gemm_kernel:EpiEmbed";
            GemmOperands go{c->w_embed, (size_t)N_MODEL * c->KE, w.xall, w.xall_plane, c->KE / 16, N_MODEL / BLK_A_F, tb_a, t0_a EG_DBG(, g_ablate, g_trace)};
            EpiEmbed<NP> e{c->b_embed, c->pe, c->tt_table, w.t_idx, w.hA, w.h_plane, g.Lr, g.T, g.B, nullptr, 0, nullptr, io.state, io.ts};
            if (int r = launch_gemm<CfgA<NP>>(go, e, s)) return r;
        }
    }
    if (io.stop_stage == EGOEGO_DBG_EMBED) return 0;
    for (int li = 0; li < c->cfg.n_dec_layers; ++li) {
        const LayerDev& L = c->layers[li];
        const bool last_dbg = (li == io.stop_layer);
        const bool dbg_qkv = last_dbg && (io.stop_stage == EGOEGO_DBG_Q || io.stop_stage == EGOEGO_DBG_K || io.stop_stage == EGOEGO_DBG_V);
        // (the split-bf16 projection epilogues place whole 32-row tiles: they only ever run on aligned geometries, Lr == Lp)
        EpiQK<NP> eqk{L.b_qkv, w.Q, w.K, w.qkv_plane, 1.0f / sqrtf((float)c->cfg.d_k), g.Lp, H, HD, g.Mvalid};
        EpiV<NP> ev{L.b_qkv, w.V, w.qkv_plane, g.Lp, H, HD, g.Mvalid};
        AttnArgs aa{w.Q, w.K, w.V, w.qkv_plane, w.O, w.o_plane, HD / 16, H, g.L, w0 * H};
        // the fused kernel has one workgroup per (window, head): below ~one workgroup per CU the unfused pair
        // (12 projection blocks per window) spreads the same work over more CUs
        const bool i8 = NP == 2 && prec_i8(c);
        const bool ffn8 = i8 && !(c->cfg.flags & EGOEGO_FLAG_FFN16);  // i8x3: the FFN contractions run on int8 slices too (every batch size: same integers, same bits) unless EGOEGO_FLAG_FFN16 keeps them on split-bf16
        // EGOEGO_PREC_I8X3_FC: fc as well, where the attention kernel can hand O over as int8 rows (the two int8 attention back
        // ends: windows of more than 64 tokens); the Q/K/V debug stops run the split-bf16 projections and never reach fc
        const bool fc8 = i8 && c->cfg.precision == EGOEGO_PREC_I8X3_FC && (g.KT == 4 || g.KT == 7);
        const bool attn_geom = g.KT == 4 && g.Lp == BLK_A_T && (i8 || nw * H >= 192);
        const bool fused_attn = attn_geom && !dbg_qkv;
        // the layer's output also as int8 slices: the next layer's projections consume them
        const bool q8_out = i8 && (li + 1 < c->cfg.n_dec_layers || act8_only);  // (precision 9's product path: linear_out reads int8 rows too)
        int8_t* const q8p = q8_out ? w.hA8 : nullptr;
        // outlier monitor of the row-quantising LayerNorm epilogues (StepState::ln_max; read through egoego_outlier_stats)
        unsigned* const om1 = li < OUTLIER_SITES / 2 ? &w.state->ln_max[2 * li] : nullptr;
        unsigned* const om2 = li < OUTLIER_SITES / 2 ? &w.state->ln_max[2 * li + 1] : nullptr;
        if (fused_attn && i8) {
            ProfScope ps(c, EGOEGO_K_QKV, s);
            AttnLayerArgs al{L.w_qkv8n, (size_t)3 * HD * N_MODEL, L.s_qkv, L.b_qkv, w.hA8, w.h_plane, w.hA_scale, w.O, w.o_plane, HD / 16,
                             1.0f / sqrtf((float)c->cfg.d_k), H, g.L, w0 * H EG_DBG(, g_trace)};
            if (fc8) {
                al.o8 = w.O8; al.o8_plane = w.o_plane; al.o_scale = w.O_scale;
            }
            static DevOnce once;
            if (once.pending()) {
                HIP_TRY(allow_smem(attn_layer_i8w_kernel, AW_DYN_SMEM_BYTES));
                HIP_TRY(allow_smem(attn_layer_i8h_kernel, AL_SMEM_BYTES));
                HIP_TRY(allow_smem(attn_proj_i8_kernel, ATTN_PROJ_SMEM));
                HIP_TRY(allow_smem(attn_core_s_kernel, ATTN_CORE_S_SMEM));
                HIP_TRY(allow_smem(attn_proj6_i8_kernel, ATTN_PROJ6_SMEM));
                HIP_TRY(allow_smem(attn_proj2_i8_kernel, ATTN_PROJ_SMEM));
                once.done();
            }
            // the projections as three workgroups per (window, head) + a core launch while they fit the chip at once (attn_split_i8.h);
            // the images go through the Q / K buffers (carved back to back: 256 KiB per window x head), V's column scales through V's
            if (w.att_img && nw * H * 6 <= ATTN_SPLIT6_MAX_BLOCKS) {  // ... six four-wave workgroups while each of those still gets a CU of its own
                const AttnSplitBufs sb{w.att_img, w.sq8, w.sk8, (float*)w.V};
                c->last_kernel[EGOEGO_K_QKV] = "attn_proj6_i8_kernel";
                attn_proj6_i8_kernel<<<dim3(nw * H * 6), dim3(256), ATTN_PROJ6_SMEM, s>>>(al, sb);
                HIP_TRY(hipGetLastError());
                attn_core_s_kernel<<<dim3(nw * H), dim3(512), ATTN_CORE_S_SMEM, s>>>(al, sb);
            } else if (w.att_img && nw * H * 3 <= ATTN_SPLIT_MAX_BLOCKS) {
                const AttnSplitBufs sb{w.att_img, w.sq8, w.sk8, (float*)w.V};
                c->last_kernel[EGOEGO_K_QKV] = "attn_proj_i8_kernel";
                attn_proj_i8_kernel<<<dim3(nw * H * 3), dim3(512), ATTN_PROJ_SMEM, s>>>(al, sb);
                HIP_TRY(hipGetLastError());
                attn_core_s_kernel<<<dim3(nw * H), dim3(512), ATTN_CORE_S_SMEM, s>>>(al, sb);
            } else if (w.att_img && nw * H * 2 <= ATTN_SPLIT2_MAX_BLOCKS) {
                const AttnSplitBufs sb{w.att_img, w.sq8, w.sk8, (float*)w.V};
                c->last_kernel[EGOEGO_K_QKV] = "attn_proj2_i8_kernel";
                attn_proj2_i8_kernel<<<dim3(nw * H * 2), dim3(512), ATTN_PROJ_SMEM, s>>>(al, sb);
                HIP_TRY(hipGetLastError());
                attn_core_s_kernel<<<dim3(nw * H), dim3(512), ATTN_CORE_S_SMEM, s>>>(al, sb);
            } else
            // up to 24 windows x 4 heads: two workgroups per (window, head), half the queries each (attn_layer_i8h.h)
            if (nw * H * 2 <= ATTN_HALF_MAX_BLOCKS) {
                c->last_kernel[EGOEGO_K_QKV] = "attn_layer_i8h_kernel";
                attn_layer_i8h_kernel<<<dim3(nw * H * 2), dim3(512), AL_SMEM_BYTES, s>>>(al);
            } else {
                c->last_kernel[EGOEGO_K_QKV] = "attn_layer_i8w_kernel";
                attn_layer_i8w_kernel<<<dim3(nw * H), dim3(512), AW_DYN_SMEM_BYTES, s>>>(al);
            }
            HIP_TRY(hipGetLastError());
        } else if (fused_attn) {
            // --- fused: Q/K/V projections of one (window, head) + its attention (TM:71-88)
            ProfScope ps(c, EGOEGO_K_QKV, s);
            GemmOperands go{L.w_qkv, (size_t)3 * HD * N_MODEL, w.hA, w.h_plane, N_MODEL / 16, 3 * HD / BLK_A_F, tb_a, 0 EG_DBG(, g_ablate, g_trace)};
            auto kern = qkv_attn_kernel<CfgA<NP>, EpiQK<NP>, CfgAV<NP>, EpiV<NP>, CfgQ<NP>, 4, NP>;
            constexpr int smem = CfgA<NP>::SMEM_BYTES > 2 * 4 * NP * 4096 ? CfgA<NP>::SMEM_BYTES : 2 * 4 * NP * 4096;
            static DevOnce once;
            if (once.pending()) {
                HIP_TRY(allow_smem(kern, smem));
                once.done();
            }
            c->last_kernel[EGOEGO_K_QKV] = "qkv_attn_kernel";
            kern<<<dim3(nw * H), dim3(CfgA<NP>::NT), smem, s>>>(go, eqk, ev, aa, H);
            HIP_TRY(hipGetLastError());
        } else {
            // --- Q, K, V projections (TM:71-73)
            const bool core8 = i8 && !dbg_qkv && g.KT == 7;  // long windows: int8 operand images + int8 attention core
            if (core8) {
                // int8-slice projections written as int8 operand images + the int8 attention core (attn_core_i8.h): windows
                // outside the one-kernel form's range
                {
                    ProfScope ps(c, EGOEGO_K_QKV, s);
                    // 128-token blocks (eight waves, one workgroup per CU) once they fill the chip, else 64-token blocks (attn_core_i8.h); same bits
                    const bool wide = rows % 128 == 0 && (rows / 128) * (3 * HD / BLK_A_F) >= 256;
                    const int bt = wide ? 128 : 64;
                    QkvI8Args qa{L.w_qkv8n, (size_t)3 * HD * N_MODEL, L.s_qkv, w.hA8, w.h_plane, w.hA_scale, rows / bt, 2 * HD / BLK_A_F};
                    Qkv8Out qo{(int8_t*)w.Q, (int8_t*)w.K, (int8_t*)w.V, w.qkv_plane, w.sq8, w.sk8, w.sv8, L.b_qkv,
                               1.0f / sqrtf((float)c->cfg.d_k), g.Lp, g.KT, H, HD, g.Mvalid, g.Lr EG_DBG(, g_trace)};
                    static DevOnce once;
                    if (once.pending()) {
                        HIP_TRY(allow_smem(qkv_i8q_kernel<1>, Q8K<1>::SMEM_BYTES + 4096));
                        HIP_TRY(allow_smem(qkv_i8q_kernel<2>, Q8K<2>::SMEM_BYTES + 4096));
                        once.done();
                    }
                    c->last_kernel[EGOEGO_K_QKV] = "qkv_i8q_kernel";
                    if (wide)
                        qkv_i8q_kernel<2><<<dim3((3 * HD / BLK_A_F) * (rows / 128)), dim3(512), Q8K<2>::SMEM_BYTES + 4096, s>>>(qa, qo);
                    else
                        qkv_i8q_kernel<1><<<dim3((3 * HD / BLK_A_F) * (rows / 64)), dim3(256), Q8K<1>::SMEM_BYTES + 4096, s>>>(qa, qo);
                    HIP_TRY(hipGetLastError());
                }
                {
                    ProfScope ps(c, EGOEGO_K_ATTN, s);
                    AttnCore8Args ca{(const int8_t*)w.Q, (const int8_t*)w.K, (const int8_t*)w.V, w.qkv_plane, w.sq8, w.sk8, w.sv8, w.O, w.o_plane,
                                     HD / 16, H, g.L, g.Lp, g.Lr};
                    EG_DBG(ca.trace = g_trace ? g_trace + 90112 : nullptr;)
                    if (fc8) {
                        ca.o8 = w.O8; ca.o8_plane = w.o_plane; ca.o_scale = w.O_scale;
                    }
                    c->last_kernel[EGOEGO_K_ATTN] = EGOEGO_CORE4 ? "attn_core_i8_kernel" : "attn_core_i8w_kernel";
                    if (int r = launch_attn_core8(ca, g.KT, g.B * H, s)) return r;
                }
            } else if (i8 && !dbg_qkv) {
                // short windows: int8-slice projections feeding the split-bf16 attention core
                ProfScope ps(c, EGOEGO_K_QKV, s);
                QkvI8Args qa{L.w_qkv8n, (size_t)3 * HD * N_MODEL, L.s_qkv, w.hA8, w.h_plane, w.hA_scale, tb_a, 2 * HD / BLK_A_F};
                auto kern = qkv_i8_kernel<EpiQK<NP>, EpiV<NP>>;
                static DevOnce once;
                if (once.pending()) {
                    HIP_TRY(allow_smem(kern, AL8K::SMEM_BYTES));
                    once.done();
                }
                c->last_kernel[EGOEGO_K_QKV] = "qkv_i8_kernel";
                kern<<<dim3((3 * HD / BLK_A_F) * tb_a), dim3(256), AL8K::SMEM_BYTES, s>>>(qa, eqk, ev);
                HIP_TRY(hipGetLastError());
            } else {
                ProfScope ps(c, EGOEGO_K_QKV, s);
                GemmOperands go{L.w_qkv, (size_t)3 * HD * N_MODEL, w.hA, w.h_plane, N_MODEL / 16, 3 * HD / BLK_A_F, tb_a, t0_a EG_DBG(, g_ablate, g_trace)};
                auto kern = qkv_kernel<CfgA<NP>, EpiQK<NP>, CfgAV<NP>, EpiV<NP>>;
                static DevOnce once;
                if (once.pending()) {
                    HIP_TRY(allow_smem(kern, CfgA<NP>::SMEM_BYTES));
                    once.done();
                }
                c->last_kernel[EGOEGO_K_QKV] = "qkv_kernel";
                kern<<<dim3(go.nfb * go.ntb), dim3(CfgA<NP>::NT), CfgA<NP>::SMEM_BYTES, s>>>(go, eqk, ev, 2 * HD / BLK_A_F);
                HIP_TRY(hipGetLastError());
            }
            if (dbg_qkv) return 0;
            // --- softmax(QK^T / sqrt(dk)) V, heads merged (TM:75-88)
            if (!core8) {
                ProfScope ps(c, EGOEGO_K_ATTN, s);
                c->last_kernel[EGOEGO_K_ATTN] = (NP == 2 && g.KT == 7 && !EGOEGO_ATTN_WG4 && nw * H >= 256) ? "attn8_kernel" : "attn_kernel";
                if (int r = launch_attn<NP>(aa, g.KT, nw * H, s)) return r;
            }
        }
        if (last_dbg && io.stop_stage == EGOEGO_DBG_ATTN_OUT) return 0;
        if constexpr (NP == 2) {
            // int8 fc: the 512-register direct-operand kernel at EVERY batch size (its per-head fp32 running sums next to the
            // integer accumulators do not fit a 256-register wave)
            if (tb_b <= 128 || fc8) {
                // --- small batches (at most one 64-token workgroup per CU): fc+LN -> FFN-1 -> FFN-2+LN in one latency-optimised kernel (tail_fused.h), same arithmetic
                ProfScope ps(c, EGOEGO_K_FC_LN, s);
                TailArgs ta{};
                ta.o = w.O; ta.o_plane = w.o_plane; ta.HD16 = HD / 16;
                ta.wfc = L.w_fc; ta.wfc_plane = (size_t)N_MODEL * HD;
                ta.ln1 = EpiResLN<2, 4, 0>{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f, nullptr, 0, nullptr};
                if (fc8) {
                    ta.fc8 = 1; ta.H = H;
                    ta.o8 = w.O8; ta.o8_plane = w.o_plane; ta.o_scale = w.O_scale;
                    ta.wfc8 = L.w_fc_8; ta.wfc8_plane = (size_t)N_MODEL * HD; ta.s_wfc = L.s_fc;
                    ta.wfc8_3 = L.w_fc_3;
                }
                if (ffn8) {  // FFN on int8 slices: LayerNorm-1 also emits int8 rows, FFN-1 writes int8 rows
                    ta.ffn8 = 1;
                    ta.ln1.q8 = w.hB8; ta.ln1.q8_plane = w.h_plane; ta.ln1.q8_scale = w.hB_scale;
                    ta.w1_8 = L.w_1_8; ta.w2_8 = L.w_2_8; ta.w8_plane = (size_t)N_MODEL * N_MODEL;
                    ta.s_w1 = L.s_1; ta.s_w2 = L.s_2;
                    ta.relu8 = EpiReluQ8<4, 0>{L.b_1, w.F8, w.h_plane, w.F_scale};
                }
                ta.w1 = L.w_1; ta.w1_plane = (size_t)N_MODEL * N_MODEL;
                ta.relu = EpiTiled<true, 2>{L.b_1, w.F, w.h_plane, N_MODEL / 16};
                ta.w2 = L.w_2; ta.w2_plane = (size_t)N_MODEL * N_MODEL;
                ta.ln2 = EpiResLN<2, 4, 0>{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
                ta.outlier = om1; ta.outlier_rows = g.Mvalid;  // (om2 = om1 + 1)
                if (act8_only) {
                    // residuals from the int8 rows; no split-bf16 rows at all (linear_out reads the last layer's int8 rows)
                    ta.ln1.res8 = w.hA8; ta.ln1.res8_plane = w.h_plane; ta.ln1.res8_scale = w.hA_scale; ta.ln1.out = nullptr;
                    ta.ln2.res8 = w.hB8; ta.ln2.res8_plane = w.h_plane; ta.ln2.res8_scale = w.hB_scale;
                    ta.ln2.out = nullptr;
                }
                ta.stop = !last_dbg ? 0 : (io.stop_stage == EGOEGO_DBG_ATTN_LN ? 1 : (io.stop_stage == EGOEGO_DBG_FFN_HIDDEN ? 2 : 0));
                EG_DBG(ta.trace = g_trace;)
                if (int r = launch_tail(c, ta, rows, s, act8_only)) return r;
                if (last_dbg) return 0;
                continue;
            }
        }
        if (ffn8) {
            // --- fused layer tail with the FFN on int8 slices (every batch size above the small-batch kernel's; debug taps through `stop`)
            ProfScope ps(c, EGOEGO_K_FC_LN, s);
            const int nb = rows / 64, b0 = row0 / 64;
            const size_t wp8 = (size_t)N_MODEL * N_MODEL / 2;  // slice stride in the main loop's 2-byte units
            GemmOperands g1{L.w_fc, (size_t)N_MODEL * HD, w.O, w.o_plane, HD / 16, 1, nb, b0 EG_DBG(, g_ablate, g_trace)};
            EpiResLN<NP, 4, 64> e1{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f, w.hB8, w.h_plane, w.hB_scale};
            e1.outlier = om1; e1.outlier_rows = g.Mvalid;
            GemmOperands g2{(const __bf16*)L.w_1_8, wp8, (const __bf16*)w.hB8, w.h_plane / 2, N_MODEL / 32, 1, nb, b0 EG_DBG(, 0, nullptr)};
            EpiReluQ8<4, 64> e2{L.b_1, w.F8, w.h_plane, w.F_scale};
            GemmOperands g3{(const __bf16*)L.w_2_8, wp8, (const __bf16*)w.F8, w.h_plane / 2, N_MODEL / 32, 1, nb, b0 EG_DBG(, 0, nullptr)};
            EpiResLN<NP, 4, 64> e3{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
            e3.outlier = om2; e3.outlier_rows = g.Mvalid;
            auto kern = layer_tail_i8_kernel<CfgBs<NP>, CfgT8a, CfgT8b, EpiResLN<NP, 4, 64>, EpiReluQ8<4, 64>>;
            static_assert(CfgT8a::SMEM_BYTES <= CfgBs<2>::SMEM_BYTES && CfgT8b::SMEM_BYTES <= CfgBs<2>::SMEM_BYTES, "the int8 passes reuse the split-bf16 ring");
            static DevOnce once;
            if (once.pending()) {
                HIP_TRY(allow_smem(kern, CfgBs<NP>::SMEM_BYTES + 6144));
                once.done();
            }
            const int stop = !last_dbg ? 0 : (io.stop_stage == EGOEGO_DBG_ATTN_LN ? 1 : (io.stop_stage == EGOEGO_DBG_FFN_HIDDEN ? 2 : 0));
            c->last_kernel[EGOEGO_K_FC_LN] = "layer_tail_i8_kernel";
            kern<<<dim3(nb), dim3(CfgBs<NP>::NT), CfgBs<NP>::SMEM_BYTES + 6144, s>>>(g1, e1, g2, L.s_1, e2, g3, L.s_2, e3, stop);
            HIP_TRY(hipGetLastError());
            if (last_dbg) return 0;
            continue;
        }
        if (!small_ln && !last_dbg) {
            // --- fused layer tail: fc+LN -> FFN-1 -> FFN-2+LN per 64-token block, two workgroups per CU (TM:92-93, 111-114, 135, 139)
            ProfScope ps(c, EGOEGO_K_FC_LN, s);
            const int nb = rows / 64, b0 = row0 / 64;
            GemmOperands g1{L.w_fc, (size_t)N_MODEL * HD, w.O, w.o_plane, HD / 16, 1, nb, b0 EG_DBG(, g_ablate, g_trace)};
            EpiResLN<NP, 4, 64> e1{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f};
            // perf-debug: the three phases stamp disjoint parts of the trace buffer
            GemmOperands g2{L.w_1, (size_t)N_MODEL * N_MODEL, w.hB, w.h_plane, N_MODEL / 16, 1, nb, b0 EG_DBG(, g_ablate, g_trace ? g_trace + 4096 : nullptr)};
            EpiTiled<true, NP> e2{L.b_1, w.F, w.h_plane, N_MODEL / 16};
            GemmOperands g3{L.w_2, (size_t)N_MODEL * N_MODEL, w.F, w.h_plane, N_MODEL / 16, 1, nb, b0 EG_DBG(, g_ablate, g_trace ? g_trace + 8192 : nullptr)};
            EpiResLN<NP, 4, 64> e3{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
            e3.outlier = om2; e3.outlier_rows = g.Mvalid;
            // From one workgroup per CU on: ONE eight-wave workgroup per 128 tokens and CU instead of two four-wave ones per 64 — the weight
            // tiles (16 of a stage's 18 fragment blocks) cross L2 -> LDS once per 128 tokens; its epilogues have no second workgroup to hide
            // behind, and it still wins since the operand stream is the largest item of this kernel (round 5's ablations: 233 us as
            // shipped, 148 without the stream): 215-218 against 231-236 us per launch at B=256 x T=120, 413-421 against 426-429 at T=196,
            // same bits.  (Round 2 had measured this form 4 % SLOWER, on a two-stage ring.)  EGOEGO_TAIL128_BF16=0: variant builds, A/B.
#ifndef EGOEGO_TAIL128_BF16
#define EGOEGO_TAIL128_BF16 1
#endif
            if (EGOEGO_TAIL128_BF16 && rows % 128 == 0 && row0 % 128 == 0 && rows / 128 >= 256) {
                g1.ntb = g2.ntb = g3.ntb = rows / 128;
                g1.tblk0 = g2.tblk0 = g3.tblk0 = row0 / 128;
                EpiResLN<NP, 4, 128> f1{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f};
                EpiResLN<NP, 4, 128> f3{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
                f3.outlier = om2; f3.outlier_rows = g.Mvalid;
                auto kern = layer_tail_kernel<CfgB<NP>, EpiResLN<NP, 4, 128>, EpiTiled<true, NP>>;
                static DevOnce once;
                if (once.pending()) {
                    HIP_TRY(allow_smem(kern, CfgB<NP>::SMEM_BYTES));
                    once.done();
                }
                c->last_kernel[EGOEGO_K_FC_LN] = "layer_tail_kernel:128";
                kern<<<dim3(rows / 128), dim3(CfgB<NP>::NT), CfgB<NP>::SMEM_BYTES, s>>>(g1, f1, g2, e2, g3, f3);
                HIP_TRY(hipGetLastError());
                continue;
            }
            auto kern = layer_tail_kernel<CfgBs<NP>, EpiResLN<NP, 4, 64>, EpiTiled<true, NP>>;
            static DevOnce once;
            if (once.pending()) {
                HIP_TRY(allow_smem(kern, CfgBs<NP>::SMEM_BYTES));
                once.done();
            }
            c->last_kernel[EGOEGO_K_FC_LN] = "layer_tail_kernel";
            kern<<<dim3(nb), dim3(CfgBs<NP>::NT), CfgBs<NP>::SMEM_BYTES, s>>>(g1, e1, g2, e2, g3, e3);
            HIP_TRY(hipGetLastError());
            continue;
        }
        // --- fc + residual + LayerNorm (+ padding mask) (TM:92-93, 135)
        {
            ProfScope ps(c, EGOEGO_K_FC_LN, s);
            c->last_kernel[EGOEGO_K_FC_LN] = "gemm_kernel:EpiResLN";
            if (small_ln) {
                GemmOperands go{L.w_fc, (size_t)N_MODEL * HD, w.O, w.o_plane, HD / 16, 1, rows / 64, row0 / 64 EG_DBG(, g_ablate, g_trace)};
                if (go.ntb <= SMALL_GRID) {
                    GemmOperands gt{L.w_fc, (size_t)N_MODEL * HD, w.O, w.o_plane, HD / 16, 1, rows / 32, row0 / 32 EG_DBG(, g_ablate, g_trace)};
                    EpiResLN<NP, 4, 32> e{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f};
                    if (int r = launch_gemm<CfgBt<NP>>(gt, e, s)) return r;
                } else {
                    EpiResLN<NP, 4, 64> e{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f};
                    if (int r = launch_gemm<CfgBs<NP>>(go, e, s)) return r;
                }
            } else {
                GemmOperands go{L.w_fc, (size_t)N_MODEL * HD, w.O, w.o_plane, HD / 16, 1, tb_b, t0_b EG_DBG(, g_ablate, g_trace)};
                EpiResLN<NP, 4, 128> e{L.b_fc, w.hA, w.h_plane, L.ln1_g, L.ln1_b, io.row_mask, w.hB, w.h_plane, 1e-5f};
                if (int r = launch_gemm<CfgB<NP>>(go, e, s)) return r;
            }
        }
        if (last_dbg && io.stop_stage == EGOEGO_DBG_ATTN_LN) return 0;
        // --- FFN conv 1 + ReLU (TM:111)
        {
            ProfScope ps(c, EGOEGO_K_FFN1, s);
            GemmOperands go{L.w_1, (size_t)N_MODEL * N_MODEL, w.hB, w.h_plane, N_MODEL / 16, N_MODEL / BLK_A_F, tb_a, t0_a EG_DBG(, g_ablate, g_trace)};
            EpiTiled<true, NP> e{L.b_1, w.F, w.h_plane, N_MODEL / 16};
            if (go.nfb * go.ntb <= SMALL_GRID) {
                go.nfb = N_MODEL / 128;
                if (int r = launch_gemm<CfgAh<NP>>(go, e, s)) return r;
            } else if (int r = launch_gemm<CfgA<NP>>(go, e, s)) return r;
        }
        if (last_dbg && io.stop_stage == EGOEGO_DBG_FFN_HIDDEN) return 0;
        // --- FFN conv 2 + residual + LayerNorm (+ padding mask) (TM:111-114, 139)
        {
            ProfScope ps(c, EGOEGO_K_FFN2_LN, s);
            if (small_ln) {
                GemmOperands go{L.w_2, (size_t)N_MODEL * N_MODEL, w.F, w.h_plane, N_MODEL / 16, 1, rows / 64, row0 / 64 EG_DBG(, g_ablate, g_trace)};
                if (go.ntb <= SMALL_GRID) {
                    GemmOperands gt{L.w_2, (size_t)N_MODEL * N_MODEL, w.F, w.h_plane, N_MODEL / 16, 1, rows / 32, row0 / 32 EG_DBG(, g_ablate, g_trace)};
                    EpiResLN<NP, 4, 32> e{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
                    e.outlier = om2; e.outlier_rows = g.Mvalid;
                    if (int r = launch_gemm<CfgBt<NP>>(gt, e, s)) return r;
                } else {
                    EpiResLN<NP, 4, 64> e{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
                    e.outlier = om2; e.outlier_rows = g.Mvalid;
                    if (int r = launch_gemm<CfgBs<NP>>(go, e, s)) return r;
                }
            } else {
                GemmOperands go{L.w_2, (size_t)N_MODEL * N_MODEL, w.F, w.h_plane, N_MODEL / 16, 1, tb_b, t0_b EG_DBG(, g_ablate, g_trace)};
                EpiResLN<NP, 4, 128> e{L.b_2, w.hB, w.h_plane, L.ln2_g, L.ln2_b, io.row_mask, w.hA, w.h_plane, 1e-5f, q8p, w.h_plane, w.hA_scale};
                    e.outlier = om2; e.outlier_rows = g.Mvalid;
                if (int r = launch_gemm<CfgB<NP>>(go, e, s)) return r;
            }
        }
        if (last_dbg && io.stop_stage == EGOEGO_DBG_LAYER_OUT) return 0;
    }
    if (io.run_out) {
        ProfScope ps(c, EGOEGO_K_OUT, s);
        if (NP == 2 && direct_io) {
            OutArgs oa{w.hA, w.h_plane, c->w_out, (size_t)c->NOUT * N_MODEL, EpiOut<2>{io.out},
                       w.hA8, w.h_plane, w.hA_scale, c->w_out_8, (size_t)c->NOUT * N_MODEL, c->s_out};
            c->last_kernel[EGOEGO_K_OUT] = "out_kernel";
            if (int r = act8_only ? launch_out_tt<1, true>(oa, rows, s) : launch_out_tt<1, false>(oa, rows, s)) return r;
        } else if (NP == 2 && act8_only) {
            // the last layer's output exists as int8 rows only: linear_out on int8 slices, 256 features x 128 tokens per eight-wave workgroup
            GemmOperands go{(const __bf16*)c->w_out_8, (size_t)c->NOUT * N_MODEL / 2, (const __bf16*)w.hA8, w.h_plane / 2, N_MODEL / 32, 1, rows / 128, row0 / 128 EG_DBG(, g_ablate, g_trace)};
            auto kern = gemm_i8_kernel<AW8K, EpiOut<2>>;
            // (after the main loop the ring's LDS holds the block's x rows: EpiOut::run_block)
            const int out_smem = std::max((int)AW8K::SMEM_BYTES, AW8K::BT * c->cfg.d_feats * 4 + 16);
            static DevOnce once;
            if (once.pending()) {
                HIP_TRY(allow_smem(kern, 160 * 1024));
                once.done();
            }
            if (out_smem > 160 * 1024) return fail(EGOEGO_E_INVALID, "d_feats too large for the linear_out kernel's LDS staging");
            c->last_kernel[EGOEGO_K_OUT] = "gemm_i8_kernel:EpiOut";
            kern<<<dim3(go.ntb), dim3(AW8K::NT), out_smem, s>>>(go, c->s_out, w.hA_scale, EpiOut<2>{io.out});
            HIP_TRY(hipGetLastError());
        } else {
            GemmOperands go{c->w_out, (size_t)c->NOUT * N_MODEL, w.hA, w.h_plane, N_MODEL / 16, 1, tb_c, t0_c EG_DBG(, g_ablate, g_trace)};
            EpiOut<NP> e{io.out};
            c->last_kernel[EGOEGO_K_OUT] = "gemm_kernel:EpiOut";
            if (tb_c <= SMALL_GRID) {
                GemmOperands gs{c->w_out, (size_t)c->NOUT * N_MODEL, w.hA, w.h_plane, N_MODEL / 16, 1, rows / 64, row0 / 64 EG_DBG(, g_ablate, g_trace)};
                if (int r = launch_gemm<CfgC2<NP>>(gs, e, s)) return r;
            } else if (int r = launch_gemm<CfgC<NP>>(go, e, s)) return r;
        }
    }
    return 0;
}

template <int NP>
static int run_denoiser_np(egoego_ctx* c, const Geometry& g, const Workspace& w, const StepIO& io, hipStream_t s) {
    return run_chunk_np<NP>(c, g, w, io, s, 0, g.B);
}

static int run_denoiser(egoego_ctx* c, const Geometry& g, const Workspace& w, const StepIO& io, hipStream_t s) {
    return c->cfg.precision != EGOEGO_PREC_BF16X1 ? run_denoiser_np<2>(c, g, w, io, s) : run_denoiser_np<1>(c, g, w, io, s);
}

static int check_ready(const egoego_ctx* c, bool need_sched) {
    if (!c) return fail(EGOEGO_E_INVALID, "null context");
    if (!c->have_weights) return fail(EGOEGO_E_STATE, "egoego_load_weights has not been called");
    if (need_sched && !c->have_sched) return fail(EGOEGO_E_STATE, "egoego_load_schedule has not been called");
    return 0;
}

static int prepare(egoego_ctx* c, int B, int T, void* d_ws, size_t ws_bytes, Geometry& g, Workspace& w, bool aligned = false) {
    if (int r = make_geometry(c, B, T, g, aligned)) return r;
    if (!d_ws || ((uintptr_t)d_ws & 255)) return fail(EGOEGO_E_WORKSPACE, "workspace must be a 256-byte aligned device pointer");
    carve(c, g, (char*)d_ws, w);
    if (ws_bytes < w.total)
        return fail(EGOEGO_E_WORKSPACE, "workspace too small: %zu bytes given, %zu needed for B=%d T=%d", ws_bytes, w.total, B, T);
    return 0;
}

static int pack_inputs(egoego_ctx* c, const Geometry& g, const Workspace& w, const float* d_x, const float* d_xc,
                       const float* d_row_mask, const float** packed_mask, hipStream_t s) {
    const size_t n = (size_t)g.Mp * (c->KE / 2);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    k_pack_pose<<<blocks, 256, 0, s>>>(d_x, d_xc, w.xall, w.xall_plane, g.Mp, c->KE, g.Lr, g.T, g.B, c->D, c->DP, 1);
    if (g.Lr != g.Lp)  // keys Lr .. Lp-1 of every window exist in the attention images only: their V scales must be finite (they meet probability 0)
        HIP_TRY(hipMemsetAsync(w.sv8, 0, sizeof(float) * (size_t)g.B * c->H * g.Lp, s));
    HIP_TRY(hipGetLastError());
    *packed_mask = nullptr;
    if (d_row_mask) {
        k_pack_row_mask<<<(g.Mp + 255) / 256, 256, 0, s>>>(d_row_mask, w.row_mask, g.Mp, g.Lr, g.T, g.B);
        HIP_TRY(hipGetLastError());
        *packed_mask = w.row_mask;
    }
    return 0;
}

static void base_out_params(const egoego_ctx* c, const Geometry& g, const Workspace& w, OutParams& o) {
    memset(&o, 0, sizeof o);
    o.bias = c->b_out;
    o.xall = w.xall;
    o.xall_plane = w.xall_plane;
    o.KE16 = c->KE / 16;
    o.sched = c->sched;
    o.t_idx = w.t_idx;
    o.objective = c->cfg.objective;
    o.clip = 1;
    o.Lp = g.Lr; o.T = g.T; o.B = g.B; o.D = c->D; o.DP = c->DP;
}

// ==================================================================================== C ABI
extern "C" {

int egoego_abi_version(void) { return EGOEGO_ABI_VERSION; }
const char* egoego_last_error(void) { return g_err.c_str(); }

int egoego_ctx_create(const egoego_config* cfg, int device, egoego_ctx** out) {
    if (!cfg || !out) return fail(EGOEGO_E_INVALID, "null argument");
    if (cfg->d_model != 512) return fail(EGOEGO_E_INVALID, "d_model=%d unsupported (LayerNorm-fused GEMM tiles are built for 512)", cfg->d_model);
    if (cfg->d_k != 256 || cfg->d_v != 256) return fail(EGOEGO_E_INVALID, "d_k=%d d_v=%d unsupported (attention tiles are built for 256)", cfg->d_k, cfg->d_v);
    if (cfg->n_head < 1 || cfg->n_head > 16) return fail(EGOEGO_E_INVALID, "n_head=%d unsupported", cfg->n_head);
    if (cfg->n_dec_layers < 1) return fail(EGOEGO_E_INVALID, "n_dec_layers must be >= 1");
    if (cfg->d_feats < 2 || (cfg->d_feats & 1) || cfg->d_feats > 248)
        return fail(EGOEGO_E_INVALID, "d_feats=%d unsupported (must be even and <= 248)", cfg->d_feats);
    if (cfg->max_timesteps < 2 || cfg->num_timesteps < 1) return fail(EGOEGO_E_INVALID, "bad max_timesteps/num_timesteps");
    if (cfg->objective != EGOEGO_PRED_X0 && cfg->objective != EGOEGO_PRED_NOISE)
        return fail(EGOEGO_E_INVALID, "unknown objective %d", cfg->objective);
    if (cfg->precision != EGOEGO_PREC_BF16X3 && cfg->precision != EGOEGO_PREC_BF16X1 && cfg->precision != EGOEGO_PREC_I8X3 &&
        cfg->precision != EGOEGO_PREC_I8X3_FC)
        return fail(EGOEGO_E_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->flags & ~(EGOEGO_FLAG_NO_GRAPH | EGOEGO_FLAG_FC24 | EGOEGO_FLAG_FFN16)) return fail(EGOEGO_E_INVALID, "unknown flags 0x%x", cfg->flags);
    if ((cfg->flags & EGOEGO_FLAG_FFN16) && cfg->precision != EGOEGO_PREC_I8X3)
        return fail(EGOEGO_E_INVALID, "EGOEGO_FLAG_FFN16 needs precision %d", EGOEGO_PREC_I8X3);
    if ((cfg->flags & EGOEGO_FLAG_FC24) && cfg->precision != EGOEGO_PREC_I8X3_FC)
        return fail(EGOEGO_E_INVALID, "EGOEGO_FLAG_FC24 needs precision %d", EGOEGO_PREC_I8X3_FC);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(EGOEGO_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    egoego_ctx* c = new egoego_ctx();
    c->cfg = *cfg;
    c->device = device;
    c->D = cfg->d_feats;
    c->DP = (int)align_up(c->D, 8);
    c->KE = (int)align_up(2 * c->DP, 32);
    c->H = cfg->n_head;
    c->HD = c->H * 256;
    c->NOUT = 256;
    c->S = cfg->num_timesteps;
    c->have_weights = c->have_sched = false;
    c->prof_id = -1;
    for (int i = 0; i < EGOEGO_K_COUNT; ++i) c->last_kernel[i] = "";
    c->w_embed = c->w_out = nullptr;
    c->cap_stream = nullptr;
    c->stage_next = 0;
    for (int i = 0; i < egoego_ctx::N_STAGE; ++i) {
        c->stage[i] = nullptr;
        c->stage_ev[i] = nullptr;
    }
    *out = c;
    return 0;
}

static void drop_graphs(egoego_ctx* c) {
    for (StepGraph& g : c->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

void egoego_ctx_destroy(egoego_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drop_graphs(c);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    for (int i = 0; i < egoego_ctx::N_STAGE; ++i) {
        if (c->stage_ev[i]) {
            (void)hipEventSynchronize(c->stage_ev[i]);
            (void)hipEventDestroy(c->stage_ev[i]);
        }
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
    }
    for (void* p : c->allocs) (void)hipFree(p);
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    delete c;
}

static int dev_alloc(egoego_ctx* c, void** p, size_t bytes, bool zero, hipStream_t s) {
    HIP_TRY(hipMalloc(p, bytes));
    c->allocs.push_back(*p);
    if (zero) HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return 0;
}

static int pack_weight(const float* src, int R, int ncols, int ld, int c0, __bf16* dst, size_t plane, int K16, int r0,
                       int k0, hipStream_t s, int acc_order = 1) {
    const size_t n = (size_t)R * (ncols / 2);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    k_pack_rows<<<blocks, 256, 0, s>>>(src, R, ncols, ld, c0, dst, plane, K16, r0, k0, 1, acc_order);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int copy_vec(egoego_ctx* c, float** dst, const float* src, int n, int n_alloc, hipStream_t s) {
    if (int r = dev_alloc(c, (void**)dst, sizeof(float) * n_alloc, n_alloc != n, s)) return r;
    HIP_TRY(hipMemcpyAsync(*dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    return 0;
}

int egoego_load_weights(egoego_ctx* c, const egoego_weights* wt, void* stream) {
    if (!c || !wt || !wt->layers) return fail(EGOEGO_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    // (re)loading frees the previous copies after the stream drained; captured steps hold the old weight pointers and may
    // still be executing on it
    if (c->have_weights || !c->graphs.empty()) HIP_TRY(hipStreamSynchronize(s));
    drop_graphs(c);
    if (c->have_weights) {
        for (void* p : c->allocs) (void)hipFree(p);
        c->allocs.clear();
        c->layers.clear();
        c->have_weights = c->have_sched = false;
    }
    const int HD = c->HD, D = c->D, DP = c->DP, KE = c->KE;
    int r;
    // embed: start_conv.weight (512, 2D, 1): input columns [0,D) -> k [0,D), [D,2D) -> k [DP, DP+D)
    if ((r = dev_alloc(c, (void**)&c->w_embed, (size_t)2 * N_MODEL * KE * 2, true, s))) return r;
    if ((r = pack_weight(wt->start_conv_w, N_MODEL, D, 2 * D, 0, c->w_embed, (size_t)N_MODEL * KE, KE / 16, 0, 0, s, 0))) return r;
    if ((r = pack_weight(wt->start_conv_w, N_MODEL, D, 2 * D, D, c->w_embed, (size_t)N_MODEL * KE, KE / 16, 0, DP, s, 0))) return r;
    if ((r = copy_vec(c, &c->b_embed, wt->start_conv_b, N_MODEL, N_MODEL, s))) return r;
    const int pe_rows = c->cfg.max_timesteps + 1;
    if ((r = copy_vec(c, &c->pe, wt->position_vec, pe_rows * N_MODEL, pe_rows * N_MODEL, s))) return r;
    // linear_out, rows padded to NOUT
    if ((r = dev_alloc(c, (void**)&c->w_out, (size_t)2 * c->NOUT * N_MODEL * 2, true, s))) return r;
    if ((r = pack_weight(wt->linear_out_w, D, N_MODEL, N_MODEL, 0, c->w_out, (size_t)c->NOUT * N_MODEL, N_MODEL / 16, 0, 0, s))) return r;
    if ((r = copy_vec(c, &c->b_out, wt->linear_out_b, D, c->NOUT, s))) return r;
    if ((r = dev_alloc(c, (void**)&c->w_out_8, (size_t)2 * c->NOUT * N_MODEL, true, s))) return r;
    if ((r = dev_alloc(c, (void**)&c->s_out, sizeof(float) * c->NOUT, true, s))) return r;
    k_pack_rows_i8<<<D, 256, 0, s>>>(wt->linear_out_w, N_MODEL, N_MODEL, c->w_out_8, (size_t)c->NOUT * N_MODEL, c->s_out, 0);
    HIP_TRY(hipGetLastError());
    c->layers.resize(c->cfg.n_dec_layers);
    for (int li = 0; li < c->cfg.n_dec_layers; ++li) {
        const egoego_layer_weights& lw = wt->layers[li];
        LayerDev& L = c->layers[li];
        const size_t qkv_plane = (size_t)3 * HD * N_MODEL;
        if ((r = dev_alloc(c, (void**)&L.w_qkv, 2 * qkv_plane * 2, false, s))) return r;
        if ((r = pack_weight(lw.w_q, HD, N_MODEL, N_MODEL, 0, L.w_qkv, qkv_plane, N_MODEL / 16, 0, 0, s))) return r;
        if ((r = pack_weight(lw.w_k, HD, N_MODEL, N_MODEL, 0, L.w_qkv, qkv_plane, N_MODEL / 16, HD, 0, s))) return r;
        if ((r = pack_weight(lw.w_v, HD, N_MODEL, N_MODEL, 0, L.w_qkv, qkv_plane, N_MODEL / 16, 2 * HD, 0, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.s_qkv, sizeof(float) * 3 * HD, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_qkv8n, 2 * qkv_plane, false, s))) return r;
        k_pack_rows_i8<<<HD, 256, 0, s>>>(lw.w_q, N_MODEL, N_MODEL, L.w_qkv8n, qkv_plane, L.s_qkv, 0);
        k_pack_rows_i8<<<HD, 256, 0, s>>>(lw.w_k, N_MODEL, N_MODEL, L.w_qkv8n, qkv_plane, L.s_qkv, HD);
        k_pack_rows_i8<<<HD, 256, 0, s>>>(lw.w_v, N_MODEL, N_MODEL, L.w_qkv8n, qkv_plane, L.s_qkv, 2 * HD);
        HIP_TRY(hipGetLastError());
        if ((r = dev_alloc(c, (void**)&L.b_qkv, sizeof(float) * 3 * HD, false, s))) return r;
        HIP_TRY(hipMemcpyAsync(L.b_qkv, lw.b_q, sizeof(float) * HD, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(L.b_qkv + HD, lw.b_k, sizeof(float) * HD, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(L.b_qkv + 2 * HD, lw.b_v, sizeof(float) * HD, hipMemcpyDeviceToDevice, s));
        if ((r = dev_alloc(c, (void**)&L.w_fc, (size_t)2 * N_MODEL * HD * 2, false, s))) return r;
        if ((r = pack_weight(lw.w_fc, N_MODEL, HD, HD, 0, L.w_fc, (size_t)N_MODEL * HD, HD / 16, 0, 0, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_1, (size_t)2 * N_MODEL * N_MODEL * 2, false, s))) return r;
        if ((r = pack_weight(lw.w_1, N_MODEL, N_MODEL, N_MODEL, 0, L.w_1, (size_t)N_MODEL * N_MODEL, N_MODEL / 16, 0, 0, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_2, (size_t)2 * N_MODEL * N_MODEL * 2, false, s))) return r;
        if ((r = pack_weight(lw.w_2, N_MODEL, N_MODEL, N_MODEL, 0, L.w_2, (size_t)N_MODEL * N_MODEL, N_MODEL / 16, 0, 0, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_1_8, (size_t)2 * N_MODEL * N_MODEL, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_2_8, (size_t)2 * N_MODEL * N_MODEL, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.s_1, sizeof(float) * N_MODEL, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.s_2, sizeof(float) * N_MODEL, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.w_fc_8, (size_t)2 * N_MODEL * HD, false, s))) return r;
        if ((r = dev_alloc(c, (void**)&L.s_fc, sizeof(float) * N_MODEL, false, s))) return r;
        k_pack_rows_i8<<<N_MODEL, 256, 0, s>>>(lw.w_fc, HD, HD, L.w_fc_8, (size_t)N_MODEL * HD, L.s_fc, 0);
        L.w_fc_3 = nullptr;
        if (c->cfg.flags & EGOEGO_FLAG_FC24) {
            if ((r = dev_alloc(c, (void**)&L.w_fc_3, (size_t)2 * N_MODEL * HD, true, s))) return r;
            k_pack_rows_i8_third<<<N_MODEL, 256, 0, s>>>(lw.w_fc, HD, HD, L.w_fc_3, (size_t)N_MODEL * HD, 0);
        }
        k_pack_rows_i8<<<N_MODEL, 256, 0, s>>>(lw.w_1, N_MODEL, N_MODEL, L.w_1_8, (size_t)N_MODEL * N_MODEL, L.s_1, 0);
        k_pack_rows_i8<<<N_MODEL, 256, 0, s>>>(lw.w_2, N_MODEL, N_MODEL, L.w_2_8, (size_t)N_MODEL * N_MODEL, L.s_2, 0);
        HIP_TRY(hipGetLastError());
        if ((r = copy_vec(c, &L.b_fc, lw.b_fc, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.ln1_g, lw.ln1_g, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.ln1_b, lw.ln1_b, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.b_1, lw.b_1, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.b_2, lw.b_2, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.ln2_g, lw.ln2_g, N_MODEL, N_MODEL, s))) return r;
        if ((r = copy_vec(c, &L.ln2_b, lw.ln2_b, N_MODEL, N_MODEL, s))) return r;
    }
    // time-token table.  Frequencies follow M:69-70: c = log(10000)/(32-1) in double, then
    // exp(float(k) * float(-c)) in fp32.
    {
        float freqs[32];
        const float cneg = (float)(-(log(10000.0) / 31.0));
        for (int k = 0; k < 32; ++k) freqs[k] = (float)exp((double)((float)k * cneg));
        float* d_freqs;
        if ((r = dev_alloc(c, (void**)&d_freqs, sizeof freqs, false, s))) return r;
        HIP_TRY(hipMemcpyAsync(d_freqs, freqs, sizeof freqs, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));  // freqs lives on this stack frame
        if ((r = dev_alloc(c, (void**)&c->tt_table, sizeof(float) * (size_t)c->S * N_MODEL, false, s))) return r;
        k_time_table<<<c->S, 256, 0, s>>>(d_freqs, wt->time_mlp1_w, wt->time_mlp1_b, wt->time_mlp3_w, wt->time_mlp3_b,
                                          c->pe + N_MODEL, c->tt_table);
        HIP_TRY(hipGetLastError());
    }
    c->have_weights = true;
    return 0;
}

int egoego_load_schedule(egoego_ctx* c, const egoego_schedule* sc, void* stream) {
    if (!c || !sc) return fail(EGOEGO_E_INVALID, "null argument");
    if (!c->have_weights) return fail(EGOEGO_E_STATE, "load weights before the schedule");
    if (!sc->posterior_mean_coef1 || !sc->posterior_mean_coef2 || !sc->posterior_log_variance_clipped ||
        !sc->sqrt_recip_alphas_cumprod || !sc->sqrt_recipm1_alphas_cumprod || !sc->alphas_cumprod)
        return fail(EGOEGO_E_INVALID, "null schedule array");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float> tab((size_t)c->S * 8, 0.f);
    c->abar_host.assign(sc->alphas_cumprod, sc->alphas_cumprod + c->S);
    for (int t = 0; t < c->S; ++t) {
        float* r = &tab[(size_t)t * 8];
        r[0] = sc->posterior_mean_coef1[t];
        r[1] = sc->posterior_mean_coef2[t];
        // M:255-256: nonzero_mask * exp(0.5 * log_variance); no noise at t == 0
        r[2] = t == 0 ? 0.f : expf(0.5f * sc->posterior_log_variance_clipped[t]);
        r[3] = sc->sqrt_recip_alphas_cumprod[t];
        r[4] = sc->sqrt_recipm1_alphas_cumprod[t];
        r[5] = sc->alphas_cumprod[t];
    }
    if (!c->have_sched) {
        if (int r = dev_alloc(c, (void**)&c->sched, tab.size() * sizeof(float), false, s)) return r;
    }
    HIP_TRY(hipMemcpyAsync(c->sched, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->have_sched = true;
    return 0;
}

size_t egoego_workspace_bytes(const egoego_ctx* c, int B, int T) {
    Geometry g;
    if (!c || make_geometry(c, B, T, g, true)) return 0;  // the aligned geometry is the larger one: it covers both
    Workspace w;
    carve(c, g, nullptr, w);
    return w.total;
}

int egoego_denoise(egoego_ctx* c, const float* d_x, const float* d_xc, const int64_t* d_t, const float* d_row_mask,
                   float* d_out, int B, int T, void* d_ws, size_t ws_bytes, void* stream) {
    if (int r = check_ready(c, false)) return r;
    if (!d_x || !d_xc || !d_t || !d_out) return fail(EGOEGO_E_INVALID, "null tensor pointer");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w)) return r;
    StepIO io{};
    if (int r = pack_inputs(c, g, w, d_x, d_xc, d_row_mask, &io.row_mask, s)) return r;
    k_convert_t<<<(B + 255) / 256, 256, 0, s>>>(d_t, w.t_idx, B, c->S);
    HIP_TRY(hipGetLastError());
    io.stop_layer = io.stop_stage = -1;
    io.run_out = true;
    base_out_params(c, g, w, io.out);
    io.out.mode = 0;
    io.out.out_raw = d_out;
    return run_denoiser(c, g, w, io, s);
}

int egoego_p_sample(egoego_ctx* c, float* d_x, const float* d_xc, const int64_t* d_t, const float* d_row_mask,
                    const float* d_noise, int noise_mode, uint64_t seed, int64_t window_offset, int clip_denoised,
                    int B, int T, void* d_ws, size_t ws_bytes, void* stream) {
    if (int r = check_ready(c, true)) return r;
    if (!d_x || !d_xc || !d_t) return fail(EGOEGO_E_INVALID, "null tensor pointer");
    if (noise_mode == EGOEGO_NOISE_INJECTED && !d_noise) return fail(EGOEGO_E_INVALID, "noise_mode INJECTED needs d_noise");
    if (noise_mode < 0 || noise_mode > 2) return fail(EGOEGO_E_INVALID, "unknown noise_mode %d", noise_mode);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w)) return r;
    StepIO io{};
    if (int r = pack_inputs(c, g, w, d_x, d_xc, d_row_mask, &io.row_mask, s)) return r;
    k_convert_t<<<(B + 255) / 256, 256, 0, s>>>(d_t, w.t_idx, B, c->S);
    HIP_TRY(hipGetLastError());
    io.stop_layer = io.stop_stage = -1;
    io.run_out = true;
    base_out_params(c, g, w, io.out);
    io.out.mode = 1;
    io.out.x = d_x;
    io.out.noise = d_noise;
    io.out.noise_mode = noise_mode;
    io.out.seed = seed;
    io.out.window_offset = window_offset;
    io.out.clip = clip_denoised ? 1 : 0;
    return run_denoiser(c, g, w, io, s);
}

// n_steps diffusion steps on `s`.  The first step of a configuration the context has not seen is launched directly
// (this also sets the kernels' shared-memory attributes); one step is then captured into a hipGraph on the context's
// own capture stream (nothing executes there) and the remaining steps are replays of that graph: 10 launches per
// step become one hipGraphLaunch.  The caller's buffers and the Philox key reach the kernels through the device-resident
// StepState, so the captured step serves every later call of the same shape on the same workspace.
struct StepCall {
    int t_start;
    float* x;
    const float* noise;
    const float* prefix;
    uint64_t seed;
    int64_t window_offset;
};

static int run_steps(egoego_ctx* c, const Geometry& g, const Workspace& w, StepIO& io, const StepKey& key, const StepCall& call,
                     int n_steps, hipStream_t s) {
    k_state_init<<<1, 1, 0, s>>>(w.state, call.t_start, call.x, call.noise, call.prefix, call.seed, call.window_offset);
    HIP_TRY(hipGetLastError());
    io.state = w.state;
    io.ts = key.ddim ? w.step_ts : nullptr;
    io.out.state = w.state;
    io.out.ts = io.ts;
    auto one_step = [&](hipStream_t st) -> int { return run_denoiser(c, g, w, io, st); };
    const bool graphs = !(c->cfg.flags & EGOEGO_FLAG_NO_GRAPH) && c->prof_id < 0;
    int i = 0;
    if (graphs && n_steps >= 2) {
        StepGraph* sg = nullptr;
        for (StepGraph& e : c->graphs)
            if (e.key == key) sg = &e;
        if (!sg) {
            if (int r = one_step(s)) return r;
            i = 1;
            if (!c->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
            StepGraph e{};
            e.key = key;
            HIP_TRY(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeRelaxed));
            const int r = one_step(c->cap_stream);
            const hipError_t ce = hipStreamEndCapture(c->cap_stream, &e.graph);
            if (r || ce != hipSuccess) {
                if (ce == hipSuccess && e.graph) (void)hipGraphDestroy(e.graph);
                return r ? r : fail(EGOEGO_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            }
            const hipError_t ie = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
            if (ie != hipSuccess) {
                (void)hipGraphDestroy(e.graph);
                return fail(EGOEGO_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
            }
            if (c->graphs.size() >= 8) {
                // the evicted graph may still be executing (graphs are shared across calls, and a caller may have moved to
                // another stream since): drain the device before destroying it (rare: the ninth distinct step shape of a context)
                HIP_TRY(hipDeviceSynchronize());
                (void)hipGraphExecDestroy(c->graphs.front().exec);
                (void)hipGraphDestroy(c->graphs.front().graph);
                c->graphs.erase(c->graphs.begin());
            }
            c->graphs.push_back(e);
            sg = &c->graphs.back();
        }
        for (; i < n_steps; ++i) HIP_TRY(hipGraphLaunch(sg->exec, s));
        return 0;
    }
    for (; i < n_steps; ++i)
        if (int r = one_step(s)) return r;
    return 0;
}

int egoego_sample_loop(egoego_ctx* c, float* d_x, const float* d_xc, int t_start, int n_steps, const float* d_noise,
                       int noise_mode, uint64_t seed, int64_t window_offset, const float* d_prefix, int prefix_len,
                       const float* d_row_mask, int B, int T, void* d_ws, size_t ws_bytes, void* stream) {
    if (int r = check_ready(c, true)) return r;
    if (!d_x || !d_xc) return fail(EGOEGO_E_INVALID, "null tensor pointer");
    if (t_start < 0 || t_start >= c->S || n_steps < 0 || n_steps > t_start + 1)
        return fail(EGOEGO_E_INVALID, "bad step range: t_start=%d n_steps=%d (num_timesteps=%d)", t_start, n_steps, c->S);
    if (noise_mode == EGOEGO_NOISE_INJECTED && !d_noise) return fail(EGOEGO_E_INVALID, "noise_mode INJECTED needs d_noise");
    if (noise_mode < 0 || noise_mode > 2) return fail(EGOEGO_E_INVALID, "unknown noise_mode %d", noise_mode);
    if (d_prefix && (prefix_len < 1 || prefix_len > T)) return fail(EGOEGO_E_INVALID, "bad prefix_len %d", prefix_len);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w)) return r;
    if (n_steps == 0) return 0;
    StepIO io{};
    // x and x_cond are split/packed ONCE; afterwards the posterior epilogue keeps the embed operand current.
    if (int r = pack_inputs(c, g, w, d_x, d_xc, d_row_mask, &io.row_mask, s)) return r;
    io.stop_layer = io.stop_stage = -1;
    io.run_out = true;
    base_out_params(c, g, w, io.out);
    io.out.mode = 1;
    io.out.noise_mode = noise_mode;
    io.out.prefix_len = d_prefix ? prefix_len : 0;
    io.out.step_elems = (size_t)B * T * c->D;
    const StepKey key{B, T, 1, noise_mode, io.out.prefix_len, 1, 0, d_row_mask ? 1 : 0, d_ws};
    const StepCall call{t_start, d_x, noise_mode == EGOEGO_NOISE_INJECTED ? d_noise : nullptr, d_prefix, seed, window_offset};
    return run_steps(c, g, w, io, key, call, n_steps, s);
}

int egoego_ddim_loop(egoego_ctx* c, float* d_x, const float* d_xc, const int32_t* ts, int n, float eta, const float* d_noise,
                     int noise_mode, uint64_t seed, int64_t window_offset, int B, int T, void* d_ws, size_t ws_bytes, void* stream) {
    if (int r = check_ready(c, true)) return r;
    if (!d_x || !d_xc || !ts || n < 1 || n > c->S) return fail(EGOEGO_E_INVALID, "bad argument");
    if (!(eta >= 0.f && eta <= 1.f)) return fail(EGOEGO_E_INVALID, "eta must be in [0, 1], got %g", (double)eta);
    if (noise_mode < 0 || noise_mode > 2) return fail(EGOEGO_E_INVALID, "unknown noise_mode %d", noise_mode);
    if (noise_mode == EGOEGO_NOISE_INJECTED && !d_noise) return fail(EGOEGO_E_INVALID, "noise_mode INJECTED needs d_noise");
    if (eta > 0.f && noise_mode == EGOEGO_NOISE_NONE) return fail(EGOEGO_E_INVALID, "eta > 0 needs a noise source (INJECTED or PHILOX)");
    for (int i = 0; i < n; ++i)
        if (ts[i] < 0 || ts[i] >= c->S || (i && ts[i] >= ts[i - 1]))
            return fail(EGOEGO_E_INVALID, "DDIM timesteps must be strictly descending in [0, %d)", c->S);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w)) return r;
    StepIO io{};
    if (int r = pack_inputs(c, g, w, d_x, d_xc, nullptr, &io.row_mask, s)) return r;
    // The timestep list and each step's coefficients (Song et al. 2021, eq. 12 / 16), read by the step kernels through
    // the step index:  sig = eta sqrt((1 - abar_prev) / (1 - abar_t)) sqrt(1 - abar_t / abar_prev),
    // x <- sqrt(abar_prev) x0 + sqrt(1 - abar_prev - sig^2) eps + sig z;  abar_prev = 1 after the last entry.
    // Staged in a pinned slot that is only reused once the copy that read it has finished: no stream synchronisation.
    {
        const int slot = c->stage_next;
        c->stage_next = (slot + 1) % egoego_ctx::N_STAGE;
        const size_t bytes = (size_t)c->S * (sizeof(int) + 4 * sizeof(float));
        if (!c->stage[slot]) {
            HIP_TRY(hipHostMalloc(&c->stage[slot], bytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&c->stage_ev[slot], hipEventDisableTiming));
        } else {
            HIP_TRY(hipEventSynchronize(c->stage_ev[slot]));
        }
        int* h_ts = (int*)c->stage[slot];
        float* h_tab = (float*)(h_ts + c->S);
        for (int i = 0; i < n; ++i) {
            const double at = c->abar_host[ts[i]], ap = (i + 1 < n) ? (double)c->abar_host[ts[i + 1]] : 1.0;
            double sig = 0.0;
            if (eta > 0.f && ap < 1.0 && at < 1.0) sig = (double)eta * sqrt((1.0 - ap) / (1.0 - at)) * sqrt(fmax(1.0 - at / ap, 0.0));
            h_ts[i] = ts[i];
            h_tab[4 * i + 0] = (float)sqrt(ap);
            h_tab[4 * i + 1] = (float)sqrt(fmax(1.0 - ap - sig * sig, 0.0));
            h_tab[4 * i + 2] = (float)sig;
            h_tab[4 * i + 3] = 0.f;
        }
        HIP_TRY(hipMemcpyAsync(w.step_ts, h_ts, sizeof(int) * n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(w.step_tab, h_tab, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(c->stage_ev[slot], s));
    }
    io.stop_layer = io.stop_stage = -1;
    io.run_out = true;
    base_out_params(c, g, w, io.out);
    io.out.mode = 2;
    io.out.noise_mode = eta > 0.f ? noise_mode : EGOEGO_NOISE_NONE;
    io.out.ddim_tab = w.step_tab;
    io.out.step_elems = (size_t)B * T * c->D;
    const StepKey key{B, T, 2, io.out.noise_mode, 0, 1, 1, 0, d_ws};
    const StepCall call{0, d_x, io.out.noise_mode == EGOEGO_NOISE_INJECTED ? d_noise : nullptr, nullptr, seed, window_offset};
    return run_steps(c, g, w, io, key, call, n, s);
}

int egoego_rot6d_to_matrix(const float* d_in, float* d_out, int64_t n, void* stream) {
    if (n < 0 || (n && (!d_in || !d_out))) return fail(EGOEGO_E_INVALID, "bad argument");
    if (n == 0) return 0;
    const int64_t blocks = (n + 255) / 256;
    k_rot6d_to_matrix<<<(int)(blocks < 8192 ? blocks : 8192), 256, 0, (hipStream_t)stream>>>(d_in, d_out, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int egoego_convert_model_res(const float* d_x, const float* d_rec_quat, const float* d_jpos_min, const float* d_jpos_max,
                             const int32_t* parents_host, int head_idx, int B, int T, float* d_aa, float* d_root, float* d_head,
                             void* stream) {
    if (!d_x || !d_rec_quat || !d_jpos_min || !d_jpos_max || !parents_host || !d_aa || !d_root || !d_head)
        return fail(EGOEGO_E_INVALID, "null argument");
    if (B < 1 || T < 1 || head_idx < 0 || head_idx >= 22) return fail(EGOEGO_E_INVALID, "bad shape (B=%d, T=%d, head_idx=%d)", B, T, head_idx);
    ConvertArgs a{d_x, d_rec_quat, d_jpos_min, d_jpos_max, d_aa, d_root, d_head, {}, head_idx, B, T};
    for (int j = 0; j < 22; ++j) {
        a.parents[j] = parents_host[j];
        if (j > 0 && (a.parents[j] < 0 || a.parents[j] >= j)) return fail(EGOEGO_E_INVALID, "parents[%d] = %d is not an earlier joint", j, a.parents[j]);
    }
    const int64_t n = (int64_t)B * T * 22;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    k_convert_model_res<<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int egoego_window_condition(const float* d_head_jpos, const float* d_head_jquat, const float* d_jpos_min, const float* d_jpos_max,
                            int head_idx, int B, int Tw, float* d_x_start, float* d_recover_quat, void* stream) {
    if (!d_head_jpos || !d_head_jquat || !d_jpos_min || !d_jpos_max || !d_x_start || !d_recover_quat) return fail(EGOEGO_E_INVALID, "null argument");
    if (B < 1 || Tw < 1 || head_idx < 0 || head_idx >= 22) return fail(EGOEGO_E_INVALID, "bad shape (B=%d, Tw=%d, head_idx=%d)", B, Tw, head_idx);
    CondArgs a{d_head_jpos, d_head_jquat, d_jpos_min, d_jpos_max, d_x_start, d_recover_quat, head_idx, B, Tw};
    const int n = B * Tw;
    k_window_condition<<<(n + 127) / 128, 128, 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int egoego_window_prefix(const float* d_aa, const float* d_root, const float* d_rest_offsets, const float* d_jpos_min,
                         const float* d_jpos_max, const int32_t* parents_host, int head_idx, int B, int Tw, int n_last,
                         float* d_prefix, void* stream) {
    if (!d_aa || !d_root || !d_rest_offsets || !d_jpos_min || !d_jpos_max || !parents_host || !d_prefix)
        return fail(EGOEGO_E_INVALID, "null argument");
    if (B < 1 || Tw < 1 || n_last < 1 || n_last > Tw || head_idx < 0 || head_idx >= 22)
        return fail(EGOEGO_E_INVALID, "bad shape (B=%d, Tw=%d, n_last=%d, head_idx=%d)", B, Tw, n_last, head_idx);
    PrefixArgs a{d_aa, d_root, d_rest_offsets, d_jpos_min, d_jpos_max, d_prefix, {}, head_idx, B, Tw, n_last};
    for (int j = 0; j < 22; ++j) {
        a.parents[j] = j ? parents_host[j] : 0;
        if (j > 0 && (a.parents[j] < 0 || a.parents[j] >= j)) return fail(EGOEGO_E_INVALID, "parents[%d] = %d is not an earlier joint", j, a.parents[j]);
    }
    const int n = B * n_last;
    k_window_prefix<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef EGOEGO_PERFDEBUG
/* perf-debug build only (not in the public header): per-block timestamps of every GEMM launch go to `buf`
 * ([grid][4] u64, overwritten by each launch), nullptr disables; ablate bit 1 skips GEMM epilogues, bit 2 attention. */
int egoego_debug_trace_buffer(unsigned long long* buf) {
    g_trace = buf;
    return 0;
}
int egoego_debug_ablate(int bits) {
    g_ablate = bits;
    return 0;
}
#endif

int egoego_outlier_stats(egoego_ctx* c, int B, int T, void* d_ws, size_t ws_bytes, float* host_out, int n_out, int reset, void* stream) {
    if (!c) return fail(EGOEGO_E_INVALID, "null context");
    if (n_out < 0 || n_out > OUTLIER_SITES || (n_out && !host_out)) return fail(EGOEGO_E_INVALID, "bad output array (at most %d sites)", OUTLIER_SITES);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w)) return r;
    if (n_out) {
        unsigned bits[OUTLIER_SITES];
        HIP_TRY(hipMemcpyAsync(bits, w.state->ln_max, sizeof(unsigned) * n_out, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < n_out; ++i) memcpy(&host_out[i], &bits[i], sizeof(float));
    }
    if (reset) HIP_TRY(hipMemsetAsync(w.state->ln_max, 0, sizeof(unsigned) * OUTLIER_SITES, s));
    return 0;
}

const char* egoego_last_kernel_name(const egoego_ctx* c, int kernel_id) {
    if (!c || kernel_id < 0 || kernel_id >= EGOEGO_K_COUNT) return "";
    return c->last_kernel[kernel_id];
}

int egoego_profile_begin(egoego_ctx* c, int kernel_id) {
    if (!c || kernel_id < 0 || kernel_id >= EGOEGO_K_COUNT) return fail(EGOEGO_E_INVALID, "bad kernel id");
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    c->prof_events.clear();
    c->prof_id = kernel_id;
    return 0;
}

int egoego_profile_end(egoego_ctx* c, double* mean_us, int* launches) {
    if (!c || !mean_us || !launches) return fail(EGOEGO_E_INVALID, "null argument");
    c->prof_id = -1;
    const size_t n = c->prof_events.size() / 2;
    double total = 0;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(c->prof_events[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
        total += ms;
    }
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    c->prof_events.clear();
    *launches = (int)n;
    *mean_us = n ? total * 1000.0 / (double)n : 0.0;
    return 0;
}

int egoego_debug_stage(egoego_ctx* c, const float* d_x, const float* d_xc, const int64_t* d_t, const float* d_row_mask,
                       int layer, int stage, float* d_out, int B, int T, void* d_ws, size_t ws_bytes, void* stream) {
    if (int r = check_ready(c, false)) return r;
    if (!d_x || !d_xc || !d_t || !d_out) return fail(EGOEGO_E_INVALID, "null tensor pointer");
    if (stage < EGOEGO_DBG_EMBED || stage > EGOEGO_DBG_LAYER_OUT || layer < 0 || layer >= c->cfg.n_dec_layers)
        return fail(EGOEGO_E_INVALID, "bad layer/stage");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    Geometry g;
    Workspace w;
    if (int r = prepare(c, B, T, d_ws, ws_bytes, g, w, true)) return r;  // the Q/K/V stops run the tile-aligned split-bf16 projections
    StepIO io{};
    if (int r = pack_inputs(c, g, w, d_x, d_xc, d_row_mask, &io.row_mask, s)) return r;
    k_convert_t<<<(B + 255) / 256, 256, 0, s>>>(d_t, w.t_idx, B, c->S);
    HIP_TRY(hipGetLastError());
    io.stop_layer = layer;
    io.stop_stage = stage;
    io.run_out = false;
    if (int r = run_denoiser(c, g, w, io, s)) return r;
    const int lo = c->cfg.precision != EGOEGO_PREC_BF16X1;
    const int L = g.L;
    // precision 9, windows of more than 64 tokens: the stops run the PRODUCT kernels (run_chunk_np: act8_only), whose inter-kernel
    // activations exist as int8 rows only — the taps read those
    const bool rows8 = c->cfg.precision == EGOEGO_PREC_I8X3_FC && (g.KT == 4 || g.KT == 7);
    switch (stage) {
        case EGOEGO_DBG_EMBED:
        case EGOEGO_DBG_LAYER_OUT:
            if (rows8)
                k_unpack_rows_i8<<<2048, 256, 0, s>>>(w.hA8, w.h_plane, w.hA_scale, N_MODEL, g.Lp, L, B, d_out, 1);
            else
                k_unpack_tiled<<<2048, 256, 0, s>>>(w.hA, w.h_plane, N_MODEL, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_ATTN_LN:
            if (rows8)
                k_unpack_rows_i8<<<2048, 256, 0, s>>>(w.hB8, w.h_plane, w.hB_scale, N_MODEL, g.Lp, L, B, d_out, 1);
            else
                k_unpack_tiled<<<2048, 256, 0, s>>>(w.hB, w.h_plane, N_MODEL, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_FFN_HIDDEN:
            if (prec_i8(c) && !(c->cfg.flags & EGOEGO_FLAG_FFN16))  // the hidden activations exist as int8 rows only
                k_unpack_rows_i8<<<2048, 256, 0, s>>>(w.F8, w.h_plane, w.F_scale, N_MODEL, g.Lp, L, B, d_out, 1);
            else
                k_unpack_tiled<<<2048, 256, 0, s>>>(w.F, w.h_plane, N_MODEL, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_ATTN_OUT:
            if (rows8)  // int8 rows, one scale per row and head
                k_unpack_rows_i8<<<2048, 256, 0, s>>>(w.O8, w.o_plane, w.O_scale, c->HD, g.Lp, L, B, d_out, c->H);
            else
                k_unpack_tiled<<<2048, 256, 0, s>>>(w.O, w.o_plane, c->HD, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_Q:
            k_unpack_qk<<<2048, 256, 0, s>>>(w.Q, w.qkv_plane, c->H, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_K:
            k_unpack_qk<<<2048, 256, 0, s>>>(w.K, w.qkv_plane, c->H, g.Lp, L, B, d_out, lo);
            break;
        case EGOEGO_DBG_V:
            k_unpack_v<<<2048, 256, 0, s>>>(w.V, w.qkv_plane, c->H, g.Lp, L, B, d_out, lo);
            break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
