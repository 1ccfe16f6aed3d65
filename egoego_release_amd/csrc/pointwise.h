// pointwise.h — HBM-bound helpers around the GEMM/attention chain: operand packing, the
// time-token table, timestep plumbing, 6D->rotation-matrix, and debug unpacking.
#pragma once
#include "common.h"

// fp32 row-major [R][ld] (columns c0 .. c0+ncols) -> fragment-tiled split-bf16 planes, written at
// (row r0 + r, column k0 + c) and scaled by `scale`.  The destination is pre-zeroed by the caller,
// so padding rows/columns stay zero.  One thread per PAIR of columns (ncols must be even).
__global__ void k_pack_rows(const float* __restrict__ src, int R, int ncols, int ld, int c0, __bf16* dst,
                            size_t dst_plane, int K16, int r0, int k0, int write_lo, int acc_order) {
    const int half = ncols >> 1;
    const size_t n = (size_t)R * half;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / half), c = (int)(i % half) * 2;
        const float2 v = *(const float2*)(src + (size_t)r * ld + c0 + c);
        __bf16 h0, l0, h1, l1;
        split_bf16(v.x, h0, l0);
        split_bf16(v.y, h1, l1);
        // acc_order: the K axis of these weights contracts with an activation kept in accumulator order
        // (common.h swap23); the pair (c, c+1) stays adjacent under that permutation
        const size_t idx = acc_order ? tiled_index_acc(r0 + r, k0 + c, K16) : tiled_index(r0 + r, k0 + c, K16);
        bf16x2 hh = {h0, h1}, ll = {l0, l1};
        *(bf16x2*)(dst + idx) = hh;
        if (write_lo) *(bf16x2*)(dst + dst_plane + idx) = ll;
    }
}

// fp32 weights [R][K] -> int8 slices (common.h "i8x3"), one scale per output row.  One block per row (row r0 + blockIdx.x
// of the destination).  The K axis is stored in accumulator order (acc32), matching the activation planes.
__global__ __launch_bounds__(256) void k_pack_rows_i8(const float* __restrict__ src, int K, int ld, int8_t* dst, size_t plane,
                                                      float* __restrict__ scales, int r0) {
    __shared__ float red[256];
    const int r = blockIdx.x + r0;
    const float* row = src + (size_t)blockIdx.x * ld;
    float mx = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    mx = red[0];
    const float inv = mx > 0.f ? I8_QMAX / mx : 0.f;
    if (threadIdx.x == 0) scales[r] = mx > 0.f ? mx / I8_QMAX : 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int q = (int)rintf(row[k] * inv);
        const int a1 = (q + 128) >> 8, a2 = q - (a1 << 8);
        const size_t idx = tiled_index_i8(r, acc32(k), K >> 5);
        dst[idx] = (int8_t)a1;
        dst[plane + idx] = (int8_t)a2;
    }
}

// The THIRD slice of a weight row on the grid of k_pack_rows_i8 (EGOEGO_FLAG_FC24) — w ~ scale * (q16 + w3 / 256), w3 = rint(256 * (w / scale - q16)),
// stored like a first slice at dst, with an all-zero plane behind it (the "second slice" of the pass that contracts it).
__global__ __launch_bounds__(256) void k_pack_rows_i8_third(const float* __restrict__ src, int K, int ld, int8_t* dst, size_t plane, int r0) {
    __shared__ float red[256];
    const int r = blockIdx.x + r0;
    const float* row = src + (size_t)blockIdx.x * ld;
    float mx = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    mx = red[0];
    const float inv = mx > 0.f ? I8_QMAX / mx : 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = row[k] * inv;
        const int q = (int)rintf(v);
        int w3 = (int)rintf((v - (float)q) * 256.0f);
        w3 = w3 > 127 ? 127 : (w3 < -127 ? -127 : w3);
        const size_t idx = tiled_index_i8(r, acc32(k), K >> 5);
        dst[idx] = (int8_t)w3;
        dst[plane + idx] = 0;
    }
}

// Concatenate x and x_cond (M:232) into the embed GEMM's operand: window b occupies rows
// b*Lp .. b*Lp+Lp-1; row 0 is the (input-less) time-token slot, rows 1..T the frames, the rest
// padding.  Columns [0, D) = x, [DP, DP + D) = x_cond, everything else zero.  The WHOLE buffer is
// rewritten, so no memset is needed.  One thread per pair of columns.
__global__ void k_pack_pose(const float* __restrict__ x, const float* __restrict__ xc, __bf16* dst,
                            size_t dst_plane, int Mp, int KE, int Lp, int T, int B, int D, int DP, int write_lo) {
    const int half = KE >> 1;
    const size_t n = (size_t)Mp * half;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / half), k = (int)(i % half) * 2;
        const int b = m / Lp, lw = m % Lp;
        float2 v = make_float2(0.f, 0.f);
        if (b < B && lw >= 1 && lw <= T) {
            const size_t row = ((size_t)b * T + (lw - 1)) * D;
            if (k < D)
                v = *(const float2*)(x + row + k);
            else if (k >= DP && k < DP + D)
                v = *(const float2*)(xc + row + (k - DP));
        }
        __bf16 h0, l0, h1, l1;
        split_bf16(v.x, h0, l0);
        split_bf16(v.y, h1, l1);
        const size_t idx = tiled_index(m, k, KE >> 4);
        bf16x2 hh = {h0, h1}, ll = {l0, l1};
        *(bf16x2*)(dst + idx) = hh;
        if (write_lo) *(bf16x2*)(dst + dst_plane + idx) = ll;
    }
}

// Time-token table: row t = time_mlp(t) + position_vec[1]  (M:61-73, 111-116, 122-123; TM:202-216:
// the time token is sequence position 0 and receives position id 1).  All windows of a sampling
// step share t, and t only takes num_timesteps values, so the 64->256->512 MLP is evaluated once
// per t at weight-load time instead of B times per step.  One block (256 threads) per t.
__global__ __launch_bounds__(256) void k_time_table(const float* __restrict__ freqs, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ w3,
                                                    const float* __restrict__ b3, const float* __restrict__ pe1,
                                                    float* __restrict__ out) {
    __shared__ float e[64];
    __shared__ float h[256];
    const int t = blockIdx.x, j = threadIdx.x;
    if (j < 32) {
        const float ang = (float)t * freqs[j];
        e[j] = sinf(ang);
        e[j + 32] = cosf(ang);
    }
    __syncthreads();
    {
        float acc = 0.f;
        for (int k = 0; k < 64; ++k) acc = fmaf(e[k], w1[j * 64 + k], acc);
        acc += b1[j];
        h[j] = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752440f));  // exact-erf GELU (nn.GELU default)
    }
    __syncthreads();
    for (int o = j; o < 512; o += 256) {
        float acc = 0.f;
        for (int k = 0; k < 256; ++k) acc = fmaf(h[k], w3[o * 256 + k], acc);
        out[(size_t)t * 512 + o] = (acc + b3[o]) + pe1[o];
    }
}

// torch.long timesteps -> clamped int32 (the kernels index tables with it).
__global__ void k_convert_t(const int64_t* __restrict__ t, int* __restrict__ out, int B, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        long long v = t[i];
        v = v < 0 ? 0 : (v >= S ? S - 1 : v);
        out[i] = (int)v;
    }
}

// (Re)arm the step state (gemm.h StepState) at the start of a multi-step call: one thread.
__global__ void k_state_init(StepState* __restrict__ st, int t_start, float* x, const float* noise, const float* prefix, uint64_t seed,
                             int64_t window_offset) {
    st->embed_step = 0;
    st->t_start = t_start;
    st->out_step = 0;
    st->pad = 0;
    st->x = x;
    st->noise = noise;
    st->prefix = prefix;
    st->seed = seed;
    st->window_offset = window_offset;
}

// Padding mask [B][T+1] -> one multiplier per padded token row.
__global__ void k_pack_row_mask(const float* __restrict__ mask, float* __restrict__ out, int Mp, int Lp, int T, int B) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < Mp) {
        const int b = m / Lp, lw = m % Lp;
        out[m] = (b < B && lw <= T) ? mask[(size_t)b * (T + 1) + lw] : 0.f;
    }
}

// 6D -> rotation matrix (Zhou et al. 2019 as pytorch3d.transforms.rotation_6d_to_matrix defines it;
// call site M:493): b1 = a1/|a1|, b2 = (a2 - <b1,a2> b1)/|.|, b3 = b1 x b2, rows (b1, b2, b3).
// Norms are clamped at 1e-12 like F.normalize.
__global__ void k_rot6d_to_matrix(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float2* p = (const float2*)(in + i * 6);
        const float2 v0 = p[0], v1 = p[1], v2 = p[2];
        const float a1x = v0.x, a1y = v0.y, a1z = v1.x, a2x = v1.y, a2y = v2.x, a2z = v2.y;
        const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
        const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
        const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
        const float cx = a2x - dt * b1x, cy = a2y - dt * b1y, cz = a2z - dt * b1z;
        const float n2 = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
        const float b2x = cx / n2, b2y = cy / n2, b2z = cz / n2;
        float* o = out + i * 9;
        o[0] = b1x; o[1] = b1y; o[2] = b1z;
        o[3] = b2x; o[4] = b2y; o[5] = b2z;
        o[6] = b1y * b2z - b1z * b2y;
        o[7] = b1z * b2x - b1x * b2z;
        o[8] = b1x * b2y - b1y * b2x;
    }
}

// ---------------------------------------------------------------- post-loop conversion chain
// convert_model_res_to_data (M:469-525) + quat_ik (amass_diffusion_dataset.py:109-125) for one batch of windows, one
// thread per (window, frame, joint).  Every step follows the torch chain of harness.py / rotations.py (themselves
// restatements of the pytorch3d definitions the reference calls) so that the two agree to rounding:
//   6D -> matrix -> quaternion (M:493-494); un-canonicalise: q = standardize(rec * q) (M:496);
//   -> matrix (M:505); IK: local = standardize(inv(q_parent) * q_child) via the matrix -> quaternion round trip of
//   quat_ik; -> matrix -> quaternion -> axis-angle (M:507);  root / head positions de-normalised and rotated by rec (M:498-501).
struct Quat {
    float w, x, y, z;
};
EG_D Quat q_std(Quat q) { return q.w < 0.f ? Quat{-q.w, -q.x, -q.y, -q.z} : q; }
EG_D Quat q_mul(Quat a, Quat b) {
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
EG_D void rot6d_rows(const float* in6, float (&m)[9]) {
    const float a1x = in6[0], a1y = in6[1], a1z = in6[2], a2x = in6[3], a2y = in6[4], a2z = in6[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
    const float cx = a2x - dt * b1x, cy = a2y - dt * b1y, cz = a2z - dt * b1z;
    const float n2 = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
    const float b2x = cx / n2, b2y = cy / n2, b2z = cz / n2;
    m[0] = b1x; m[1] = b1y; m[2] = b1z;
    m[3] = b2x; m[4] = b2y; m[5] = b2z;
    m[6] = b1y * b2z - b1z * b2y;
    m[7] = b1z * b2x - b1x * b2z;
    m[8] = b1x * b2y - b1y * b2x;
}
// largest-of-(w, x, y, z) branch, denominators clamped at 0.1, result with w >= 0 (rotations.matrix_to_quaternion)
EG_D Quat mat_to_quat(const float (&m)[9]) {
    const float qa[4] = {sqrtf(fmaxf(1.0f + m[0] + m[4] + m[8], 0.f)), sqrtf(fmaxf(1.0f + m[0] - m[4] - m[8], 0.f)),
                         sqrtf(fmaxf(1.0f - m[0] + m[4] - m[8], 0.f)), sqrtf(fmaxf(1.0f - m[0] - m[4] + m[8], 0.f))};
    int best = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (qa[i] > qa[best]) best = i;
    const float d = 2.0f * fmaxf(qa[best], 0.1f);
    Quat q;
    if (best == 0) q = Quat{qa[0] * qa[0], m[7] - m[5], m[2] - m[6], m[3] - m[1]};
    else if (best == 1) q = Quat{m[7] - m[5], qa[1] * qa[1], m[3] + m[1], m[2] + m[6]};
    else if (best == 2) q = Quat{m[2] - m[6], m[3] + m[1], qa[2] * qa[2], m[5] + m[7]};
    else q = Quat{m[3] - m[1], m[6] + m[2], m[7] + m[5], qa[3] * qa[3]};
    return q_std(Quat{q.w / d, q.x / d, q.y / d, q.z / d});
}
EG_D void quat_to_mat(Quat q, float (&m)[9]) {
    const float two_s = 2.0f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    m[0] = 1 - two_s * (q.y * q.y + q.z * q.z); m[1] = two_s * (q.x * q.y - q.z * q.w); m[2] = two_s * (q.x * q.z + q.y * q.w);
    m[3] = two_s * (q.x * q.y + q.z * q.w); m[4] = 1 - two_s * (q.x * q.x + q.z * q.z); m[5] = two_s * (q.y * q.z - q.x * q.w);
    m[6] = two_s * (q.x * q.z - q.y * q.w); m[7] = two_s * (q.y * q.z + q.x * q.w); m[8] = 1 - two_s * (q.x * q.x + q.y * q.y);
}
// rotate p by q: vector part of q * (0, p) * conj(q)
EG_D void q_apply(Quat q, const float (&p)[3], float (&o)[3]) {
    const Quat t = q_mul(q_mul(q, Quat{0.f, p[0], p[1], p[2]}), Quat{q.w, -q.x, -q.y, -q.z});
    o[0] = t.x; o[1] = t.y; o[2] = t.z;
}
struct ConvertArgs {
    const float* x;     // [B][T][198]: 22 x 3 normalised joint positions, then 22 x 6D rotations
    const float* rec;   // [B][4] un-canonicalising rotation (w, x, y, z)
    const float* jmin;  // [66]
    const float* jmax;  // [66]
    float* aa;          // [B][T][22][3]
    float* root;        // [B][T][3]
    float* head;        // [B][T][3]
    int parents[22];
    int head_idx, B, T;
};
EG_D Quat global_quat(const float* frame, int j, Quat rec) {
    float m[9];
    rot6d_rows(frame + 66 + 6 * j, m);
    const Quat ori = q_std(q_mul(rec, mat_to_quat(m)));  // M:494-496
    quat_to_mat(ori, m);                                 // M:505
    return mat_to_quat(m);                               // quat_ik's own matrix -> quaternion
}
__global__ void k_convert_model_res(ConvertArgs a) {
    const int64_t n = (int64_t)a.B * a.T * 22;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % 22);
        const int64_t bt = i / 22;
        const int b = (int)(bt / a.T);
        const float* frame = a.x + bt * 198;
        const Quat rec{a.rec[4 * b], a.rec[4 * b + 1], a.rec[4 * b + 2], a.rec[4 * b + 3]};
        Quat loc = global_quat(frame, j, rec);
        if (j > 0) {
            const Quat gp = global_quat(frame, a.parents[j], rec);
            loc = q_std(q_mul(Quat{gp.w, -gp.x, -gp.y, -gp.z}, loc));
        }
        float m[9];
        quat_to_mat(loc, m);                      // quat_ik returns matrices ...
        const Quat q = mat_to_quat(m);            // ... and matrix_to_axis_angle goes back through the quaternion
        const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
        const float half = atan2f(nrm, q.w), ang = 2.0f * half;
        const float s = fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.0f : sinf(half) / ang;
        float* o = a.aa + i * 3;
        o[0] = q.x / s; o[1] = q.y / s; o[2] = q.z / s;
        if (j == 0 || j == a.head_idx) {
            float p[3], r[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float lo = a.jmin[3 * j + c], hi = a.jmax[3 * j + c];
                p[c] = (frame[3 * j + c] + 1.0f) * 0.5f * (hi - lo) + lo;
            }
            q_apply(rec, p, r);
            float* dst = (j == 0 ? a.root : a.head) + bt * 3;
            dst[0] = r[0]; dst[1] = r[1]; dst[2] = r[2];
        }
    }
}

// ---------------------------------------------------------------- next-window condition
// The tail of one sliding-window iteration (M:399-467): forward kinematics of the last n_last frames (fk_smpl,
// amass_diffusion_dataset.py:265-293), re-canonicalisation about their first frame's head heading (rotate_at_frame,
// lafan1/utils.py:111-137), min/max normalisation and the 6D rotation form -> the [B][n_last][198] rows that overwrite the
// first frames of the next window after every diffusion step.  One thread per (window, frame); same operation order as the
// torch chain in harness.py.
struct PrefixArgs {
    const float* aa;    // [B][Tw][22][3] local axis-angle
    const float* root;  // [B][Tw][3]
    const float* rest;  // [22][3] rest-pose offsets
    const float* jmin;  // [66]
    const float* jmax;  // [66]
    float* out;         // [B][n_last][198]
    int parents[22];
    int head_idx, B, Tw, n_last;
};
EG_D Quat aa_to_quat_via_matrix(const float* a3) {
    const float ang = sqrtf(a3[0] * a3[0] + a3[1] * a3[1] + a3[2] * a3[2]);
    const float half = 0.5f * ang;
    const float sc = fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.0f : sinf(half) / ang;
    float m[9];
    quat_to_mat(Quat{cosf(half), a3[0] * sc, a3[1] * sc, a3[2] * sc}, m);  // axis_angle_to_matrix ...
    return mat_to_quat(m);                                                   // ... then matrix_to_quaternion
}
// global rotations / positions of all 22 joints of one frame
EG_D void fk_frame(const PrefixArgs& a, const float* aa, const float* root, Quat (&gq)[22], float (&gp)[22][3]) {
    gq[0] = aa_to_quat_via_matrix(aa);
    gp[0][0] = a.rest[0]; gp[0][1] = a.rest[1]; gp[0][2] = a.rest[2];
    for (int j = 1; j < 22; ++j) {
        const int p = a.parents[j];
        const float off[3] = {a.rest[3 * j], a.rest[3 * j + 1], a.rest[3 * j + 2]};
        float r[3];
        q_apply(gq[p], off, r);
        gp[j][0] = r[0] + gp[p][0]; gp[j][1] = r[1] + gp[p][1]; gp[j][2] = r[2] + gp[p][2];
        gq[j] = q_std(q_mul(gq[p], aa_to_quat_via_matrix(aa + 3 * j)));
    }
    for (int j = 0; j < 22; ++j) {
        gp[j][0] += root[0]; gp[j][1] += root[1]; gp[j][2] += root[2];
    }
}
__global__ void k_window_prefix(PrefixArgs a) {
    const int n = a.B * a.n_last;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = i / a.n_last, f = i % a.n_last;
        const int fr0 = a.Tw - a.n_last, fr = fr0 + f;
        Quat gq[22];
        float gp[22][3];
        // heading of the slice's first frame: rotate +x by its head rotation, drop z, normalise; yrot takes +x onto it
        fk_frame(a, a.aa + ((size_t)b * a.Tw + fr0) * 66, a.root + ((size_t)b * a.Tw + fr0) * 3, gq, gp);
        const Quat key = gq[a.head_idx];
        const float tx = 0.f, ty = 2.0f * key.z, tz = -2.0f * key.y;  // t = 2 * cross(key.xyz, ex)
        float fx = 1.0f + key.w * tx + (key.y * tz - key.z * ty);
        float fy = key.w * ty + (key.z * tx - key.x * tz);
        const float fn = sqrtf(fx * fx + fy * fy) + 1e-8f;
        fx /= fn; fy /= fn;
        // (|x||f| + x.f, x cross f) with x = (1,0,0), f = (fx, fy, 0)
        Quat yrot{sqrtf(fx * fx + fy * fy) + fx, 0.f, 0.f, fy};
        const float yn = sqrtf(yrot.w * yrot.w + yrot.z * yrot.z) + 1e-8f;
        yrot.w /= yn; yrot.z /= yn;
        const Quat inv{yrot.w, -yrot.x, -yrot.y, -yrot.z};
        // t_move: the rotated head position of the first frame, z dropped (rotate_at_frame's own formula)
        const float hp[3] = {gp[a.head_idx][0], gp[a.head_idx][1], gp[a.head_idx][2]};
        const float ix = 2.0f * (inv.y * hp[2] - inv.z * hp[1]), iy = 2.0f * (inv.z * hp[0] - inv.x * hp[2]), iz = 2.0f * (inv.x * hp[1] - inv.y * hp[0]);
        const float mvx = hp[0] + inv.w * ix + (inv.y * iz - inv.z * iy);
        const float mvy = hp[1] + inv.w * iy + (inv.z * ix - inv.x * iz);
        if (f != 0) fk_frame(a, a.aa + ((size_t)b * a.Tw + fr) * 66, a.root + ((size_t)b * a.Tw + fr) * 3, gq, gp);
        float* o = a.out + (size_t)i * 198;
        for (int j = 0; j < 22; ++j) {
            float r[3];
            q_apply(inv, gp[j], r);
            r[0] -= mvx; r[1] -= mvy;
            for (int c = 0; c < 3; ++c) {
                const float lo = a.jmin[3 * j + c], hi = a.jmax[3 * j + c];
                o[3 * j + c] = (r[c] - lo) / (hi - lo) * 2.0f - 1.0f;
            }
            float m[9];
            quat_to_mat(q_std(q_mul(inv, gq[j])), m);
            for (int c = 0; c < 6; ++c) o[66 + 6 * j + c] = m[c];  // first two rows
        }
    }
}

// ---------------------------------------------------------------- window condition
// Head of one sliding-window iteration (M:355-378): the window's head trajectory canonicalised about its first frame's heading
// (rotate_at_frame, lafan1/utils.py:111-137; xy of the first frame moved to the origin), written into an otherwise zero
// [B][Tw][198] x_start (head position dims 3*head_idx.., head 6D rotation dims 66 + 6*head_idx..), joint positions
// min/max-normalised.  Also returns the un-canonicalising rotation per window.  One thread per (window, frame).
struct CondArgs {
    const float* jpos;   // [B][Tw][3] global head position
    const float* jquat;  // [B][Tw][4] global head rotation (w, x, y, z)
    const float* jmin;   // [66]
    const float* jmax;   // [66]
    float* x_start;      // [B][Tw][198]
    float* recover;      // [B][4]
    int head_idx, B, Tw;
};
EG_D void rotate_cross(Quat q, const float (&p)[3], float (&o)[3]) {  // p + w * t + cross(v, t), t = 2 cross(v, p)
    const float tx = 2.0f * (q.y * p[2] - q.z * p[1]), ty = 2.0f * (q.z * p[0] - q.x * p[2]), tz = 2.0f * (q.x * p[1] - q.y * p[0]);
    o[0] = p[0] + q.w * tx + (q.y * tz - q.z * ty);
    o[1] = p[1] + q.w * ty + (q.z * tx - q.x * tz);
    o[2] = p[2] + q.w * tz + (q.x * ty - q.y * tx);
}
EG_D Quat heading_quat(Quat key) {  // quaternion taking +x onto the xy-projection of key's forward direction
    const float ex[3] = {1.f, 0.f, 0.f};
    float f[3];
    rotate_cross(key, ex, f);
    const float fn = sqrtf(f[0] * f[0] + f[1] * f[1]) + 1e-8f;
    const float fx = f[0] / fn, fy = f[1] / fn;
    Quat y{sqrtf(fx * fx + fy * fy) + fx, 0.f, 0.f, fy};
    const float yn = sqrtf(y.w * y.w + y.z * y.z) + 1e-8f;
    y.w /= yn; y.z /= yn;
    return y;
}
__global__ void k_window_condition(CondArgs a) {
    const int n = a.B * a.Tw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = i / a.Tw, t = i % a.Tw;
        const float* q0 = a.jquat + (size_t)b * a.Tw * 4;
        const Quat yrot = heading_quat(Quat{q0[0], q0[1], q0[2], q0[3]});
        const Quat inv{yrot.w, -yrot.x, -yrot.y, -yrot.z};
        if (t == 0) {
            float* r = a.recover + 4 * b;
            r[0] = yrot.w; r[1] = yrot.x; r[2] = yrot.y; r[3] = yrot.z;
        }
        const float* p0 = a.jpos + (size_t)b * a.Tw * 3;
        const float first[3] = {p0[0], p0[1], p0[2]}, cur[3] = {a.jpos[(size_t)i * 3], a.jpos[(size_t)i * 3 + 1], a.jpos[(size_t)i * 3 + 2]};
        float m0[3], pos[3];
        rotate_cross(inv, first, m0);
        rotate_cross(inv, cur, pos);
        pos[0] -= m0[0]; pos[1] -= m0[1];
        const float* qq = a.jquat + (size_t)i * 4;
        float m[9];
        quat_to_mat(q_mul(inv, Quat{qq[0], qq[1], qq[2], qq[3]}), m);  // raw product, like rotate_at_frame
        float* o = a.x_start + (size_t)i * 198;
        for (int j = 0; j < 22; ++j)
            for (int c = 0; c < 3; ++c) {
                const float v = j == a.head_idx ? pos[c] : 0.f;
                const float lo = a.jmin[3 * j + c], hi = a.jmax[3 * j + c];
                o[3 * j + c] = (v - lo) / (hi - lo) * 2.0f - 1.0f;
            }
        for (int c = 66; c < 198; ++c) o[c] = 0.f;
        for (int c = 0; c < 6; ++c) o[66 + 6 * a.head_idx + c] = m[c];
    }
}

// ---------------------------------------------------------------- debug / test-only unpackers
// fragment-tiled [Mp][N] -> fp32 [B][L][N] (drops the padding rows).
__global__ void k_unpack_tiled(const __bf16* __restrict__ src, size_t plane, int N, int Lp, int L, int B,
                               float* __restrict__ out, int use_lo) {
    const size_t n = (size_t)B * L * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % N);
        const size_t bl = i / N;
        const int l = (int)(bl % L), b = (int)(bl / L);
        const size_t idx = tiled_index_acc(b * Lp + l, f, N >> 4);
        float v = (float)src[idx];
        if (use_lo) v += (float)src[plane + idx];
        out[i] = v;
    }
}

// int8-slice rows [Mp][N] (K in acc32 order; `groups` scales per row, each covering N / groups consecutive columns) -> fp32 [B][L][N].
__global__ void k_unpack_rows_i8(const int8_t* __restrict__ src, size_t plane, const float* __restrict__ scale, int N, int Lp, int L, int B,
                                 float* __restrict__ out, int groups) {
    const size_t n = (size_t)B * L * N;
    const int gw = N / groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % N);
        const size_t bl = i / N;
        const int l = (int)(bl % L), b = (int)(bl / L);
        const int m = b * Lp + l;
        const size_t idx = tiled_index_i8(m, acc32(f), N >> 5);
        out[i] = (float)((int)src[idx] * 256 + (int)src[plane + idx]) * scale[(size_t)m * groups + f / gw];
    }
}

// Q/K operand [b][h][Lp/32][16][2][32][8] -> fp32 [B][H][L][256].
__global__ void k_unpack_qk(const __bf16* __restrict__ src, size_t plane, int H, int Lp, int L, int B,
                            float* __restrict__ out, int use_lo) {
    const size_t n = (size_t)B * H * L * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i & 255);
        const size_t r = i >> 8;
        const int l = (int)(r % L);
        const size_t bh = r / L;
        const size_t idx = bh * (size_t)Lp * 256 + tiled_index_acc(l, d, 16);
        float v = (float)src[idx];
        if (use_lo) v += (float)src[plane + idx];
        out[i] = v;
    }
}

// V operand [b][h][8][Lp/16][2][32][8] (transposed, key-permuted) -> fp32 [B][H][L][256].
__global__ void k_unpack_v(const __bf16* __restrict__ src, size_t plane, int H, int Lp, int L, int B,
                           float* __restrict__ out, int use_lo) {
    const size_t n = (size_t)B * H * L * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i & 255);
        const size_t r = i >> 8;
        const int key = (int)(r % L);
        const size_t bh = r / L;
        const int k16 = key & 15, a = k16 >> 3, bb = (k16 >> 2) & 1, c = k16 & 3;
        const size_t idx = ((((bh * 8 + (d >> 5)) * (size_t)(Lp >> 4) + (key >> 4)) * 2 + bb) << 8) + (d & 31) * 8 + 4 * a + c;
        float v = (float)src[idx];
        if (use_lo) v += (float)src[plane + idx];
        out[i] = v;
    }
}
