// attn_layer_i8h.h — the eight-wave attention layer (attn_layer_i8w.h, TM:71-88) for small grids: up to 24 windows x 4 heads per
// call (its 192 workgroups then fill three quarters of the CUs) — the reference's own sample_bs = 1 (BASELINE configs[0]), small
// interactive batches.
//
// A (window, head) workgroup is a serial chain of ~38 us whatever the batch, and with 64 of them three quarters of the chip idle.
// Here TWO workgroups share a (window, head): each projects K and V for ALL keys (redundantly — that work cannot be split without an
// exchange between workgroups) but Q, the logits, the softmax and PV for only its HALF of the queries:
//   * K_h, V_h: exactly the phases of attn_layer_i8w_kernel (same configuration, same epilogues);
//   * Q_h for 64 queries: waves 4 (features) x 2 (query tiles), 64f x 32t per wave;
//   * S^T, softmax: wave (query tile, key tile): one 32 x 32 tile of logits each; the row maximum and the row sum of a query cross
//     the four key-tile waves through LDS;
//   * PV: wave (query tile, d_v quarter); the int8 output's row maximum crosses the four quarter waves through LDS.
// Per workgroup: K 8.9 + 2.3, Q 4.5 + 1.5, S 1.5, V 7.8 + 2.7, PV 2 us instead of 38 (tools/attn_layer_trace.py).
// Same integers; the same float operations per value in the same order (the row sum of the probabilities is (t0 + t1) + (t2 + t3) over
// the four key tiles in both kernels): a window's bits do not depend on which of the two kernels computed it.
#pragma once
#include "attn_layer_i8w.h"

using AH8Q = GemmCfg<2, 1, 4, 2, 1, 2, false, 2, 3>;  // 256 features x 64 queries; its 20 one-KiB blocks per stage go unevenly over 8 waves (gemm.h)
static_assert(AH8Q::SMEM_BYTES <= AW8K::SMEM_BYTES, "the Q half fits the ring of the K / V projections");

__global__ __launch_bounds__(512, 2) void attn_layer_i8h_kernel(AttnLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kv = smem;                               // K image, later V^T image: [slice][tile][k32 block][1 KiB]
    float* sk = (float*)(smem + AL_KV_BYTES);      // [128] key row scales
    float* sv = sk + 128;                          // [256] V column scales
    float* red = sv + 256;                         // [512] cross-wave maxima
    float* p_ws = red + 512;                       // [3][256] weight row scales of Q_h, K_h, V_h
    float* p_b = p_ws + 768;                       // [3][256] biases
    float* p_hs = p_b + 768;                       // [128] row scales of the window's int8 input rows
    float* sqv = p_hs + 128;                       // [64] query row scales (this half)
    float* psum = sqv + 128;                       // [4][64] per-key-tile row sums of the probabilities
    char* ring = smem + AL_KV_BYTES + AL_MISC_BYTES;  // operand ring; between main loops: the Q image, then the P image
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // both halves and all H heads of a window share an XCD
    const int bh = (lid >> 1) + a.bh0, qh = lid & 1;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = wave_id_uniform();
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int wf = wave & 3, wt = wave >> 2;       // projections: feature quarter, token half (K, V) / query tile (Q)
    const GemmOperands g{(const __bf16*)a.w8, a.w_plane / 2, (const __bf16*)a.h8, a.h_plane / 2, 16, 0, 0, 0 EG_DBG(, 0, nullptr)};
    {
        const int HD = a.H * 256;
        for (int i = threadIdx.x; i < 768; i += 512) {
            const int src = (i >> 8) * HD + h * 256 + (i & 255);
            p_ws[i] = a.w_scale[src];
            p_b[i] = a.bias[src];
        }
        if (threadIdx.x < 128) p_hs[threadIdx.x] = a.h_scale[b * 128 + threadIdx.x];
    }  // visible after the first barrier of the K projection's main loop

    // ---- 1. K_h -> LDS (all 128 keys; attn_layer_i8w.h phase 1) -------------------------------------------
    {
        I8Acc q[2][2];
        GemmBody<AW8K, NoEpi>::mainloop(g, a.H + h, b, ring, q);
        const int f0 = 256 + wf * 64, t0 = wt * 64;
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
            if (wf == 0 && hf == 0) sk[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][j][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                char* dst = kv + (((wt * 2 + j) * 8 + wf * 2 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + AL_SLICE) = s2;
            }
        }
    }
    // ---- 2. Q_h of this half's 64 queries -> LDS image over the ring -> B fragments of the query-tile waves --------------
    const int qt = wave & 1, kt = wave >> 1;  // phases 3 and 5: query tile of the half; key tile (3) / d_v quarter (5)
    i32x4 qs1[8], qs2[8];
    float sq = 0.f;
    {
        I8Acc q[2][1];
        GemmBody<AH8Q, NoEpi>::mainloop(g, h, b * 2 + qh, ring, q);  // (ends with a barrier: the ring is idle, every wave is past the K image writes)
        const int f0 = wf * 64;
        f32x16 v[2];
        float amax = 0.f;
        const float sa = p_hs[qh * 64 + wt * 32 + col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
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
        if (hf == 0) red[wf * 64 + wt * 32 + col] = amax;
        __syncthreads();
        {
            const int tok = wt * 32 + col;
            const float rmax = fmaxf(fmaxf(red[tok], red[64 + tok]), fmaxf(red[128 + tok], red[192 + tok]));
            const float inv = rmax > 0.f ? I8_QMAX / rmax : 0.f;
            if (wf == 0 && hf == 0) sqv[tok] = rmax > 0.f ? rmax / I8_QMAX : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = v[i][r];
                u32x4 s1, s2;
                quant16(t, inv, s1, s2);
                char* dst = ring + ((wt * 8 + wf * 2 + i) << 10) + lane * 16;
                *(u32x4*)dst = s1;
                *(u32x4*)(dst + 16384) = s2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* src = ring + ((qt * 8 + i) << 10) + lane * 16;
            qs1[i] = lds_frag(src);
            qs2[i] = lds_frag(src + 16384);
        }
        sq = sqv[qt * 32 + col];
        __syncthreads();  // the Q image is in registers: the ring may be refilled (V projection)
    }
    // ---- 3. S^T = K Q^T, softmax over keys (TM:76-82): wave (query tile qt, key tile kt) ---------------------------------
    i32x4 ps1, ps2, ps3;  // this wave's key block of the probabilities, three slices (attn_layer_i8.h quant_p)
    {
        I8Acc s;
        acc_zero(s);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* src = kv + ((kt * 8 + i) << 10) + lane * 16;
            const i32x4 k1 = lds_frag(src), k2 = lds_frag(src + AL_SLICE);
            s.m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k2, qs1[i], s.m, 0, 0, 0);
            s.m = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1, qs2[i], s.m, 0, 0, 0);
            s.h = __builtin_amdgcn_mfma_i32_32x32x32_i8(k1, qs1[i], s.h, 0, 0, 0);
        }
        float p[16];
        float mx = -INFINITY;
        const float sq256 = sq * 256.0f * 1.44269504088896f;  // logits in units of log2(e): softmax through v_exp_f32
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 k4 = *(const float4*)(sk + kt * 32 + 8 * gq + 4 * hf);
            const float ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = 4 * gq + c;
                float val = (float)i8_combine(s.h[r], s.m[r]) * (sq256 * ks[c]);
                if (kt * 32 + 8 * gq + 4 * hf + c >= a.L) val = -INFINITY;
                p[r] = val;
                mx = fmaxf(mx, val);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (hf == 0) red[kt * 64 + qt * 32 + col] = mx;
        __syncthreads();
        {
            const int qi = qt * 32 + col;
            mx = fmaxf(fmaxf(red[qi], red[64 + qi]), fmaxf(red[128 + qi], red[192 + qi]));  // (key 0 always exists: finite)
        }
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(p[r] - mx);
            s1 += p[r];
        }
        s1 += __shfl_xor(s1, 32);
        if (hf == 0) psum[kt * 64 + qt * 32 + col] = s1;  // read in phase 5, behind the V projection's barriers
        u32x4 u1, u2, u3;
        quant_p(p, P_QMAX, u1, u2, u3);
        ps1 = __builtin_bit_cast(i32x4, u1);
        ps2 = __builtin_bit_cast(i32x4, u2);
        ps3 = __builtin_bit_cast(i32x4, u3);
    }
    // ---- 4. V_h -> LDS (transposed, over the K image; all 128 keys; attn_layer_i8w.h phase 4) -------------------------------
    {
        I8Acc q[2][2];
        GemmBody<AW8V, NoEpi>::mainloop(g, 2 * a.H + h, b, ring, q);  // its first barrier: every wave is past S^T (the K image is dead)
        {
            // the probabilities of the two query tiles -> LDS (over the idle ring) for the d_v-quarter waves of phase 5
            char* dst = ring + ((qt * 4 + kt) << 10) + lane * 16;
            *(i32x4*)dst = ps1;
            *(i32x4*)(dst + 8192) = ps2;
            *(i32x4*)(dst + 16384) = ps3;
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
    // ---- 5. O^T = V^T P (TM:83-88): wave (query tile qt, d_v quarter dvq), heads merged on store ---------------------------
    const int dvq = kt;
    const int m = b * 128 + qh * 64 + qt * 32 + col;
    i32x4 pa1[4], pa2[4], pa3[4];  // all four key blocks of this wave's query tile, three slices
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const char* src = ring + ((qt * 4 + kb) << 10) + lane * 16;
        pa1[kb] = lds_frag(src);
        pa2[kb] = lds_frag(src + 8192);
        pa3[kb] = lds_frag(src + 16384);
    }
    const int qi = qt * 32 + col;
    const float oscale = (1.0f / ((psum[qi] + psum[64 + qi]) + (psum[128 + qi] + psum[192 + qi]))) * (256.0f / P_QMAX);
    PVAcc o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) acc_zero(o[dt]);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        i32x4 v1[2], v2[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const char* src = kv + (((dvq * 2 + dt) * 4 + kb) << 10) + lane * 16;
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
    float t[2][16];
    float amax = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 s4 = *(const float4*)(sv + (dvq * 2 + dt) * 32 + 8 * gq + 4 * hf);
            const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float val = pv_value(o[dt].h[4 * gq + c], o[dt].m[4 * gq + c], o[dt].l[4 * gq + c]) * (ss[c] * oscale);
                t[dt][4 * gq + c] = val;
                amax = fmaxf(amax, fabsf(val));
            }
        }
    if (a.o8) {
        // int8 rows for the int8 fc: one scale per row and head = the maximum over the four d_v quarters (the other waves' through LDS)
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if (hf == 0) red[dvq * 64 + qi] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[qi], red[64 + qi]), fmaxf(red[128 + qi], red[192 + qi]));
        const float inv = amax > 0.f ? I8_QMAX / amax : 0.f;
        if (dvq == 0 && hf == 0) a.o_scale[(size_t)m * a.H + h] = amax > 0.f ? amax / I8_QMAX : 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            u32x4 s1, s2;
            quant16(t[dt], inv, s1, s2);
            const size_t idx = acc_slot_i8(m, h * 256 + (dvq * 2 + dt) * 32, hf, a.HD16 / 2);
            *(u32x4*)(a.o8 + idx) = s1;
            *(u32x4*)(a.o8 + a.o8_plane + idx) = s2;
        }
    } else {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                u32x4 hi, lo;
                split8(t[dt] + 8 * jj, hi, lo);
                const size_t idx = acc_slot(m, h * 256 + (dvq * 2 + dt) * 32, jj, hf, a.HD16);
                *(u32x4*)(a.o + idx) = hi;
                *(u32x4*)(a.o + a.o_plane + idx) = lo;
            }
    }
}
