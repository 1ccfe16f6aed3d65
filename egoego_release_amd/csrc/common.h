// common.h — types, the fragment-tiled split-bf16 and int8-slice layouts, and small device helpers.
//
// Data layout used for every intermediate tensor and every weight (gfx950-first design):
//
//   * split-bf16: an fp32 value v is stored as two bf16 numbers hi = rn(v), lo = rn(v - hi)
//     in two separate planes.  A contraction then runs as three bf16 MFMAs
//     (hi*hi + lo*hi + hi*lo, fp32 accumulate), which keeps ~16 mantissa bits per operand —
//     the reference's fp32 results are reproduced to ~1e-5 where plain bf16 gives ~1e-2
//     (SURVEY.md §7 "hard parts").  The split is done ONCE, in the epilogue of the kernel that
//     produces the value, never in a consumer's main loop.
//
//   * fragment-tiled: a [R][K] matrix (R % 32 == 0, K % 16 == 0) is stored as
//     [R/32][K/16][2][32][8] bf16, i.e. one contiguous 1 KiB block per (32 rows x 16 k) MFMA
//     operand fragment, ordered exactly as v_mfma_f32_32x32x16_bf16 wants it: lane l of a wave
//     reads its 8 bf16 (16 bytes) at block_base + 16*l.  Global->LDS staging is therefore a
//     straight copy of contiguous kilobytes, LDS fragment reads are lane-linear 16-byte reads
//     (bank-conflict free by construction, no padding, no swizzle), and epilogues write 8 or 16
//     contiguous bytes per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // one 16-byte chunk (native vector: stays in VGPRs)
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

#define EG_HD __host__ __device__ __forceinline__
#define EG_D __device__ __forceinline__

// Element index of (row r, column k) inside one fragment-tiled plane with K/16 == K16.
EG_HD size_t tiled_index(int r, int k, int K16) {
    return ((((size_t)(r >> 5) * (size_t)K16 + (size_t)(k >> 4)) * 2 + (size_t)((k >> 3) & 1)) << 8) +
           (size_t)((r & 31) << 3) + (size_t)(k & 7);
}

// "Accumulator order" of the k (feature) axis.  In the swapped MFMA accumulator a lane owns features
// 8g + 4*hf + c (g = 0..3, c = 0..3) of a 32-feature tile, i.e. two runs of 4 inside every group of 16.
// Storing feature 8a + 4b + c of each group of 16 at position 8b + 4a + c (bits 2 and 3 swapped — an
// involution) makes those two runs ONE run of 8: every epilogue store / residual load becomes a single
// 16-byte access per lane and a whole wave writes one contiguous KiB.  A contraction does not care about
// the order of k as long as both operands agree, so every activation is kept in this order and the weights
// that consume it are packed with the same permutation along K (Q and K share it along d_k).
EG_HD int swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

// Element index of (row r, feature f) in a fragment-tiled plane kept in accumulator order.
EG_HD size_t tiled_index_acc(int r, int f, int K16) { return tiled_index(r, swap23(f), K16); }

// Split eight fp32 values and pack them as 2 x (8 bf16 = 16 bytes).
EG_D void split8(const float v[8], u32x4& hi, u32x4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (__bf16)v[i];
        l[i] = (__bf16)(v[i] - (float)h[i]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

EG_D void unpack8(u32x4 hi, u32x4 lo, float v[8]) {
    bf16x8 h = __builtin_bit_cast(bf16x8, hi);
    bf16x8 l = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
}

EG_D void unpack8_hi(u32x4 hi, float v[8]) {
    bf16x8 h = __builtin_bit_cast(bf16x8, hi);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}

// Swapped-accumulator helper: for accumulator tile registers 8*jj .. 8*jj+7 of lane (col, hf), the element
// offset of the lane's 16-byte slot in an accumulator-ordered plane.  f32 = first feature of the 32-tile.
EG_D size_t acc_slot(int m, int f32, int jj, int hf, int K16) {
    return ((((size_t)(m >> 5) * (size_t)K16 + (size_t)((f32 >> 4) + jj)) * 2 + (size_t)hf) << 8) + (size_t)((m & 31) << 3);
}

// ---- int8-slice operands ("i8x3" mode) ---------------------------------------------------------------
// A row of an operand is quantised to 16-bit integers q = rint(v / s) with one scale s per row
// (|q| <= 32639) and stored as two signed int8 slices q = 256*a1 + a2, a2 in [-128, 127].  A product of two rows
// is then  s_a*s_w * (65536*sum(a1*w1) + 256*sum(a1*w2 + a2*w1))  (+ a dropped 2^-16 term): three int8 MFMAs
// (v_mfma_i32_32x32x32_i8, twice the bf16 rate, half the operand bytes) into two int32 accumulators.
// A fragment-tiled int8 plane is [R/32][K/32][2][32][16]: again one 1 KiB block per MFMA operand fragment,
// lane l reading 16 bytes at block + 16*l.
struct I8Acc {
    i32x16 h, m;
};
static constexpr float I8_QMAX = 32639.0f;

EG_HD size_t tiled_index_i8(int r, int k, int K32) {
    return ((((size_t)(r >> 5) * (size_t)K32 + (size_t)(k >> 5)) * 2 + (size_t)((k >> 4) & 1)) << 9) + (size_t)((r & 31) << 4) +
           (size_t)(k & 15);
}

// Accumulator order of a 32-feature group for int8 planes: feature 8g + 4hf + c sits at position 16hf + 4g + c,
// so the 16 accumulator registers of a lane are 16 consecutive bytes (one store per slice per tile).
EG_HD int acc32(int f) {
    const int r = f & 31;
    return (f & ~31) | (((r >> 2) & 1) << 4) | ((r >> 3) << 2) | (r & 3);
}

// Quantise 16 values with one scale into the two slices (16 bytes each).
// q = rint(v * inv_scale) comes out of the float adder: u = bits(v * inv_scale + 1.5 * 2^23) = 0x4B400000 + q
// for |q| < 2^22, so the low byte of u IS the low slice s2 = q - 256 * s1 (as a signed byte) and byte 1 of
// u + 128 is the high slice s1 = (q + 128) >> 8.  Four values are packed with three byte permutes per slice.
EG_D void quant16(const float v[16], float inv_scale, u32x4& s1, u32x4& s2) {
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[i], inv_scale, 12582912.0f));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t a = u[4 * w], b = u[4 * w + 1], c = u[4 * w + 2], d = u[4 * w + 3];
        // v_perm_b32(hi, lo, sel): selector bytes 0-3 pick from lo, 4-7 from hi
        const uint32_t lo01 = __builtin_amdgcn_perm(b, a, 0x0c0c0400u);  // {a.b0, b.b0, 0, 0}
        const uint32_t lo23 = __builtin_amdgcn_perm(d, c, 0x04000c0cu);  // {0, 0, c.b0, d.b0}
        s2[w] = lo01 | lo23;
        const uint32_t a1 = a + 128u, b1 = b + 128u, c1 = c + 128u, d1 = d + 128u;
        const uint32_t hi01 = __builtin_amdgcn_perm(b1, a1, 0x0c0c0501u);  // {a1.b1, b1.b1, 0, 0}
        const uint32_t hi23 = __builtin_amdgcn_perm(d1, c1, 0x05010c0cu);  // {0, 0, c1.b1, d1.b1}
        s1[w] = hi01 | hi23;
    }
}

EG_D void dequant16(u32x4 s1, u32x4 s2, float scale, float v[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int a1 = (int)(int8_t)((s1[i >> 2] >> (8 * (i & 3))) & 255);
        const int a2 = (int)(int8_t)((s2[i >> 2] >> (8 * (i & 3))) & 255);
        v[i] = (float)(a1 * 256 + a2) * scale;
    }
}

// 16-byte slot of lane (m & 31, hf) for feature tile f32 (first feature of the 32-tile) in an int8 plane
EG_D size_t acc_slot_i8(int m, int f32, int hf, int K32) {
    return ((((size_t)(m >> 5) * (size_t)K32 + (size_t)(f32 >> 5)) * 2 + (size_t)hf) << 9) + (size_t)((m & 31) << 4);
}

EG_D void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// Split four fp32 values and pack them as 2 x (4 bf16 = 8 bytes).
EG_D void split4(const float v[4], uint2& hi, uint2& lo) {
    bf16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 a, b;
        split_bf16(v[i], a, b);
        h[i] = a;
        l[i] = b;
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

EG_D void unpack4(uint2 hi, uint2 lo, float v[4]) {
    bf16x4 h = __builtin_bit_cast(bf16x4, hi);
    bf16x4 l = __builtin_bit_cast(bf16x4, lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)h[i] + (float)l[i];
}

EG_D void unpack4_hi(uint2 hi, float v[4]) {
    bf16x4 h = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)h[i];
}

// s_waitcnt through the builtin, not inline asm: hipcc's own wait-count insertion then KNOWS the counter was
// drained and does not add a second, conservative wait later (after an LDS-DMA every wait it inserts itself
// is lgkmcnt(0), which would also wait for ds_reads issued a moment ago).  gfx9 encoding:
// vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8].
template <int VM, int LGKM>
EG_D void wait_counts() {
    static_assert(VM >= 0 && VM <= 63 && LGKM >= 0 && LGKM <= 15, "counter range");
    __builtin_amdgcn_s_waitcnt((VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (LGKM << 8));
}
EG_D void wait_lds() { wait_counts<63, 0>(); }

// LDS-DMA requests as `buffer_load_dwordx4 ... lds` (round 6): one buffer resource per operand in SGPRs, a wave-uniform byte offset in an
// SGPR, the lane's 16 bytes as a constant VGPR offset — instead of `global_load_lds_dwordx4` on a per-lane 64-bit pointer (two VALU adds
// per request and an address VGPR pair per chunk).  Same bytes to the same place: bit-equal results (9 tensors over the three precisions,
// T = 30 / 120 / 196), and 2-3 % less time per step in every configuration measured (profiles/r06_buffer_dma_ab.txt: B=256 x T=120 1.414 ->
// 1.384 ms in precision 9, 2.408 -> 2.336 in split-bf16; B=32 0.306 -> 0.299 / 0.520 -> 0.505; T=196 3.055 -> 3.004 / 4.681 -> 4.525).
// EGOEGO_GEMM_BUFFER_DMA=0: the global form (variant builds, A/B).  Offsets are 32-bit: every resource is based so that they stay far below 2^31.
#ifndef EGOEGO_GEMM_BUFFER_DMA
#define EGOEGO_GEMM_BUFFER_DMA 1
#endif
EG_D __amdgpu_buffer_rsrc_t gemm_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
// one 1-KiB piece: 16 bytes per lane from resource `r` at byte offset `soff` (wave-uniform) + 16 * lane into LDS at `dst` + 16 * lane
EG_D void gemm_dma_piece(__amdgpu_buffer_rsrc_t r, char* dst, unsigned soff, int lane) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, soff, 0, 0);
}

EG_D int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Row of accumulator register r of v_mfma_f32_32x32x16_bf16 for a lane in half hf (= lane >> 5):
// row = (r & 3) + 8 * (r >> 2) + 4 * hf; the column is lane & 31.
EG_D int mfma32_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// Step state of a multi-step loop, kept in device memory so that ONE captured step (hipGraph) replays for every
// timestep of every chain on every caller buffer.  The step's first kernel (embed) reads embed_step and publishes
// out_step = embed_step + 1; its last kernel (linear_out + posterior) reads out_step - 1 and publishes
// embed_step = out_step for the next step: each counter is written by a launch whose blocks do not read it, so
// there is no intra-launch race, and the stream orders the launches.
struct StepState {
    int embed_step;  // index of the step the next embed launch runs
    int t_start;     // timestep of step 0 (ancestral chains: t = t_start - step)
    int out_step;    // 1 + index of the step the next out launch finishes
    int pad;
    float* x;              // [B][T][D] the chain's sample, updated in place
    const float* noise;    // [n_steps][B][T][D] injected draws or nullptr
    const float* prefix;   // [B][prefix_len][D] in-painting source or nullptr
    uint64_t seed;         // Philox key
    int64_t window_offset; // global index of window 0 (shard-invariant noise)
    // Outlier monitor of the int8-slice precisions (egoego_outlier_stats): site 2 * layer + (0: LayerNorm-1, 1: LayerNorm-2) holds
    // the bit pattern of max |value| over every row that epilogue has quantised (one scale per row) since the last reset —
    // positive floats order like their bit patterns, so the kernels use one atomicMax per workgroup.  Layers >= 8 are not recorded.
    unsigned ln_max[16];
};
static constexpr int OUTLIER_SITES = 16;

// --------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator + Box-Muller: four N(0,1) per call.
struct Philox4 {
    uint32_t c[4];
};

EG_D Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 r;
    r.c[0] = c0; r.c[1] = c1; r.c[2] = c2; r.c[3] = c3;
    return r;
}

EG_D void philox_normal4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, float out[4]) {
    Philox4 r = philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    // (0,1] uniforms from 32 random bits, then Box-Muller on two pairs.
    const float inv = 2.3283064365386963e-10f;  // 2^-32
    float u0 = ((float)r.c[0] + 1.0f) * inv, u1 = (float)r.c[1] * inv;
    float u2 = ((float)r.c[2] + 1.0f) * inv, u3 = (float)r.c[3] * inv;
    u0 = fminf(u0, 1.0f);
    u2 = fminf(u2, 1.0f);
    // v_log_f32 / v_sin_f32 / v_cos_f32 directly (the latter two take their argument in turns):
    // statistical quality only matters here, and the libm versions drag scratch memory in.
    const float ln2 = 0.6931471805599453f;
    float ra = sqrtf(-2.0f * ln2 * __builtin_amdgcn_logf(u0)), rb = sqrtf(-2.0f * ln2 * __builtin_amdgcn_logf(u2));
    const float s0 = __builtin_amdgcn_sinf(u1), c0f = __builtin_amdgcn_cosf(u1);
    const float s1 = __builtin_amdgcn_sinf(u3), c1f = __builtin_amdgcn_cosf(u3);
    out[0] = ra * c0f; out[1] = ra * s0; out[2] = rb * c1f; out[3] = rb * s1;
}

// Blocks that share an activation tile (same token block, different feature blocks) are made
// consecutive in the remapped id and land on ONE XCD (hardware places block b on XCD b % 8), so the
// tile is fetched into one L2 instead of eight.  Bijective for any grid size.
EG_D int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
