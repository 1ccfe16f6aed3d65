// attn_layer_i8.h — one decoder layer's attention front end (TM:71-88) in ONE kernel, nothing but h in and O out.
//
// One 4-wave workgroup per (window, head), one workgroup per CU (each wave owns a SIMD and its 512 registers):
//   1. K_h = h W_k^T + b      int8-slice GEMM (gemm.h "i8x3"), quantised per key row -> LDS (64 KiB)
//   2. Q_h = (h W_q^T + b)/sqrt(d_k)   waves laid 1(f) x 4(t): a lane ends up with all 256 d_k of ONE query;
//                                      row maximum in-lane, quantised into registers (64 VGPRs)
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

__global__ __launch_bounds__(256, 1) void attn_layer_i8_kernel(AttnLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                               // K image, later V^T image: [slice][tile][k32 block][1 KiB]
    float* sk = (float*)(smem + AL_KV_BYTES);      // [128] key row scales
    float* sv = sk + 128;                          // [256] V column scales
    float* red = sv + 256;                         // [512] cross-wave maxima
    // this (window, head)'s epilogue parameters, staged once: a lone wave per SIMD cannot hide their L2 latency
    float* p_ws = red + 512;                       // [3][256] weight row scales of Q_h, K_h, V_h
    float* p_b = p_ws + 768;                       // [3][256] biases
    float* p_hs = p_b + 768;                       // [128] row scales of the window's int8 input rows
    char* ring = smem + AL_KV_BYTES + AL_MISC_BYTES;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // the H heads of a window share an XCD (and its L2)
    const int bh = lid + a.bh0;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    // operand strides are counted in bf16 elements (2 bytes) by the main loop; K16 counts 32-wide k blocks here
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    EG_DBG(unsigned long long* tr = a.trace ? a.trace + 131072 + (size_t)blockIdx.x * 16 : nullptr;)
    auto mark = [&](int i) {
        EG_DBG(if (tr && threadIdx.x == 0) {
            tr[i] = wall_clock64();
            if (i < 2) tr[12 + i] = __builtin_readcyclecounter();
        })
        (void)i;
    };
    mark(0);
    {
        const int HD = a.H * 256;
        for (int i = threadIdx.x; i < 768; i += 256) {
            const int src = (i >> 8) * HD + h * 256 + (i & 255);
            p_ws[i] = a.w_scale[src];
            p_b[i] = a.bias[src];
        }
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    }  // visible after the first barrier of the K projection's main loop

    // ---- 1. K_h -> LDS ------------------------------------------------------------------------------
    {
        I8Acc q[4][2];
        GemmBody<AL8K, NoEpi>::mainloop(g, a.H + h, b, ring, q);
        mark(1);
        const int wf = wave & 1, wt = wave >> 1;
        const int f0 = 256 + wf * 128;  // index into the staged parameters
        const int t0 = wt * 64;
        f32x16 v[4][2];
        float amax[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sa = p_hs[t0 + j * 32 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                i8_dequant(q[i][j], v[i][j], p_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *(const float4*)(p_b + f0 + i * 32 + 8 * gq + 4 * hf);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[i][j][4 * gq + c] += bb[c];
                        amax[j] = fmaxf(amax[j], fabsf(v[i][j][4 * gq + c]));
                    }
                }
            }
            amax[j] = fmaxf(amax[j], __shfl_xor(amax[j], 32));
            if (hf == 0) red[wf * 128 + wt * 64 + j * 32 + col] = amax[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = wt * 64 + j * 32 + col;
            const float rmax = fmaxf(red[key], red[128 + key]);
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) sk[key] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                char* dst = kv + (((wt * 2 + j) * 8 + wf * 4 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    }

    mark(2);
    // ---- 2. Q_h -> registers ------------------------------------------------------------------------
    i32x4 qs1[8], qs2[8];
    float sq;
    {
        I8Acc q[8][1];
        GemmBody<AL8Q, NoEpi>::mainloop(g, h, b, ring, q);
        mark(3);
        const int f0 = 0;  // index into the staged parameters
        const float sa = p_hs[wave * 32 + col];
        f32x16 v[8];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            i8_dequant(q[i][0], v[i], p_ws + f0 + i * 32 + 4 * hf, sa);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 b4 = *(const float4*)(p_b + f0 + i * 32 + 8 * gq + 4 * hf);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[i][4 * gq + c] = (v[i][4 * gq + c] + bb[c]) * a.qscale;
                    amax = fmaxf(amax, fabsf(v[i][4 * gq + c]));
                }
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
        sq = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = v[i][r];
            u32x4 s1, s2;
            quant16(t, inv, s1, s2);
            qs1[i] = __builtin_bit_cast(i32x4, s1);
            qs2[i] = __builtin_bit_cast(i32x4, s2);
        }
    }

    mark(4);
    // ---- 3. S^T = K Q^T, softmax over keys (TM:76-82) -------------------------------------------------
    i32x4 ps1[4], ps2[4];
    float rsum;
    {
        I8Acc s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc_zero(s[kt]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            i32x4 k1[4], k2[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const char* src = kv + ((kt * 8 + i) << 10) + lane * 16;
                k1[kt] = lds_frag(src);
                k2[kt] = lds_frag(src + AL_SLICE);
            }
            // part-major: two MFMAs on one accumulator are never back to back
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k2[kt], qs1[i], s[kt].m, 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1[kt], qs2[i], s[kt].m, 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1[kt], qs1[i], s[kt].h, 0, 0, 0);
        }
        float p[4][16];
        float mx = -INFINITY;
        // logits in units of log2(e): softmax through v_exp_f32 (2^x), one instruction per probability
        const float sq256 = sq * 256.0f * 1.44269504088896f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
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
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[kt][r] = __builtin_amdgcn_exp2f(p[kt][r] - mx);
                sum += p[kt][r];
            }
        sum += __shfl_xor(sum, 32);
        rsum = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 s1, s2;
            quant16(p[kt], I8_QMAX, s1, s2);
            ps1[kt] = __builtin_bit_cast(i32x4, s1);
            ps2[kt] = __builtin_bit_cast(i32x4, s2);
        }
    }

    mark(5);
    // ---- 4. V_h -> LDS (transposed, over the K image) ---------------------------------------------------
    {
        I8Acc q[4][2];
        GemmBody<AL8V, NoEpi>::mainloop(g, 2 * a.H + h, b, ring, q);
        mark(6);
        const int wf = wave & 1, wt = wave >> 1;
        const int f0 = 512 + wf * 128;  // index into the staged parameters
        const int t0 = wt * 64;
        f32x16 v[4][2];
        float amax[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
            if (hf == 0) red[wt * 256 + wf * 128 + i * 32 + col] = amax[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dv = wf * 128 + i * 32 + col;
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
                char* dst = kv + (((wf * 4 + i) * 4 + wt * 2 + j) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
        __syncthreads();
    }

    mark(7);
    // ---- 5. O^T = V^T P (TM:83-88), heads merged on store ------------------------------------------------
    const int m = b * 128 + wave * 32 + col;
    const float oscale = rsum * (256.0f / I8_QMAX);
    auto pv_half = [&](int dvh, I8Acc (&o)[4]) {
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
            for (int dt = 0; dt < 4; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v2[dt], ps1[kb], o[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt].m = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], ps2[kb], o[dt].m, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt].h = __builtin_amdgcn_mfma_i32_32x32x32_i8(v1[dt], ps1[kb], o[dt].h, 0, 0, 0);
        }
    };
    // fp32 value of accumulator register r of d_v tile `tile`
    auto o_val = [&](const I8Acc& o, int tile, int g, float (&t)[4]) {
        const float4 s4 = *(const float4*)(sv + tile * 32 + 8 * g + 4 * hf);
        const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = (float)i8_combine(o.h[4 * g + c], o.m[4 * g + c]) * (ss[c] * oscale);
    };
    if (a.o8) {
        // int8 rows for the int8 fc: a lane holds all 256 d_v of its query over the two halves, so the row maximum of this
        // head is in-lane + one cross-half shuffle; the first half's values wait in registers for it
        float t[8][16];
        float amax = 0.f;
#pragma unroll
        for (int dvh = 0; dvh < 2; ++dvh) {
            I8Acc o[4];
            pv_half(dvh, o);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    o_val(o[dt], dvh * 4 + dt, g, v);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        t[dvh * 4 + dt][4 * g + c] = v[c];
                        amax = fmaxf(amax, fabsf(v[c]));
                    }
                }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
        if (hf == 0) a.o_scale[(size_t)m * a.H + h] = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            u32x4 s1, s2;
            quant16(t[tile], inv, s1, s2);
            const size_t idx = acc_slot_i8(m, h * 256 + tile * 32, hf, a.HD16 / 2);
            *(u32x4*)(a.o8 + idx) = s1;
            *(u32x4*)(a.o8 + a.o8_plane + idx) = s2;
        }
    } else {
#pragma unroll 1
        for (int dvh = 0; dvh < 2; ++dvh) {
            I8Acc o[4];
            pv_half(dvh, o);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int tile = dvh * 4 + dt;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float t[8];
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        float v[4];
                        o_val(o[dt], tile, 2 * jj + g2, v);
#pragma unroll
                        for (int c = 0; c < 4; ++c) t[4 * g2 + c] = v[c];
                    }
                    u32x4 hi, lo;
                    split8(t, hi, lo);
                    const size_t idx = acc_slot(m, h * 256 + tile * 32, jj, hf, a.HD16);
                    *(u32x4*)(a.o + idx) = hi;
                    *(u32x4*)(a.o + a.o_plane + idx) = lo;
                }
            }
        }
    }
    EG_DBG(if (tr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(8);
    })
}

// ---- Q/K/V projections on int8 slices for windows the fused kernel above does not cover (T + 1 <= 64 or > 128) ----------
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
