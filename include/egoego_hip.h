/*
 * egoego_hip.h — C ABI of libegoego_hip.so: the MI355X (gfx950) implementation of EgoEgo's
 * stage-2 conditional motion-diffusion sampling step.
 *
 * The reference (lijiaman/egoego_release) is pure Python/PyTorch and has no native interface;
 * each entry point below replaces the PyTorch op sequence of the cited reference function.
 * Paths are relative to the reference root:
 *   M  = egoego/model/transformer_cond_diffusion_model.py
 *   TM = egoego/model/transformer_module.py
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (PyTorch allocates all I/O
 *     tensors and the workspace); the library never frees or retains caller memory except the
 *     workspace during a call, and launches only on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream).
 *   - synchronisation: the SAMPLING entry points (egoego_denoise, _p_sample, _sample_loop, _ddim_loop,
 *     _rot6d_to_matrix, _convert_model_res, _window_condition, _window_prefix, _debug_stage) only enqueue
 *     work and return; they never wait for the stream.  (egoego_ddim_loop stages its step table in a pinned
 *     slot and waits, at most, for the copy of the call four calls earlier; egoego_sample_loop drains the DEVICE
 *     once if a context has seen more than eight distinct step shapes and must evict a captured graph.)
 *     The SETUP entry points egoego_load_weights and egoego_load_schedule DO call hipStreamSynchronize(stream)
 *     (host staging buffers; a re-load first waits for work that still reads the old weights), and
 *     egoego_profile_end waits for its events.
 *   - `stream` must NOT be in capture mode (torch.cuda.graph / hipStreamBeginCapture by the caller):
 *     the multi-step loops capture their own per-step hipGraph on a private stream (relaxed mode) and launch
 *     graphs on `stream`, and the setup calls synchronise.  egoego_denoise / egoego_p_sample are plain kernel
 *     launches and can be captured by the caller once the context has run that shape (first use sets kernel
 *     attributes).
 *   - pose tensors are fp32, contiguous, [B][T][d_feats]; timesteps are int64 [B] (torch.long).
 *   - return value: 0 = ok; negative = error (EGOEGO_E_*); egoego_last_error() describes the last
 *     failure on the calling thread.
 *   - any number of contexts may live on one device (the Python side holds up to three at a time: the plan's, its unshifted
 *     twin for padding-mask calls, and the split-bf16 reference of a pack-time measurement); a context is not thread-safe.
 *     A context and each workspace are SINGLE-STREAM objects: the
 *     captured step graphs are shared by every call of a shape, and the per-workspace step state (timestep counters, the
 *     caller's buffer pointers, the Philox key) is rewritten by every loop call — two streams driving one workspace or one
 *     context concurrently race silently.  Use one context + workspace per stream.
 */
#ifndef EGOEGO_HIP_H
#define EGOEGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGOEGO_ABI_VERSION 6 /* 5: EGOEGO_FLAG_FC24; 6: EGOEGO_FLAG_FFN16 */

enum {
    EGOEGO_OK = 0,
    EGOEGO_E_INVALID = -1,   /* bad argument / unsupported shape */
    EGOEGO_E_HIP = -2,       /* a HIP runtime call failed */
    EGOEGO_E_STATE = -3,     /* weights or schedule not loaded yet */
    EGOEGO_E_WORKSPACE = -4  /* workspace too small or misaligned */
};

enum { EGOEGO_PRED_NOISE = 0, EGOEGO_PRED_X0 = 1 };        /* M:235-240 */
enum { EGOEGO_NOISE_INJECTED = 0, EGOEGO_NOISE_PHILOX = 1, EGOEGO_NOISE_NONE = 2 };
/* operand precision of every contraction: 3 = split-bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate;
 * meets the 1e-3 parity bar), 1 = plain bf16 operands (fast, does NOT meet it; reported only),
 * 8 = the contractions whose input rows have one natural scale — the Q/K/V projections, QK^T, PV and the two FFN convs —
 * computed from two int8 slices per operand (three int8 MFMAs per product, int32 accumulate; same parity bar); embed, fc and
 * linear_out stay split-bf16.
 * 9 = 8 with the attention output projection (fc) and linear_out on int8 slices as well: the attention kernels hand O over as
 * int8 rows with one scale per row and head, fc runs one exact integer chain per head, and every activation that crosses memory
 * between the kernels of a step is an int8 row (the residual stream included).  Same parity bar, about twice the error of 8
 * (~3e-4 against ~1.3e-4 on one forward), about 30 % less time per step at every batch size (8: +40 %, 3: +80-90 % over 9),
 * and more sensitive to outlier-heavy checkpoints
 * (LayerNorm gains far above the rest: DESIGN.md 3c). */
enum { EGOEGO_PREC_BF16X3 = 3, EGOEGO_PREC_BF16X1 = 1, EGOEGO_PREC_I8X3 = 8, EGOEGO_PREC_I8X3_FC = 9 };

typedef struct egoego_ctx egoego_ctx;

/* Shapes of TransformerDiffusionModel (M:75-116) / CondGaussianDiffusion.__init__ (M:144-161). */
typedef struct {
    int32_t d_feats;        /* 198 */
    int32_t d_model;        /* 512 (only value supported) */
    int32_t n_head;         /* 4 */
    int32_t n_dec_layers;   /* 4 */
    int32_t d_k;            /* 256 (only value supported) */
    int32_t d_v;            /* 256 (only value supported) */
    int32_t max_timesteps;  /* window + 1; the position table has max_timesteps + 1 rows (TM:180-182) */
    int32_t num_timesteps;  /* diffusion steps S, 1000 */
    int32_t objective;      /* EGOEGO_PRED_X0 | EGOEGO_PRED_NOISE */
    int32_t precision;      /* EGOEGO_PREC_BF16X3 | EGOEGO_PREC_BF16X1 | EGOEGO_PREC_I8X3 | EGOEGO_PREC_I8X3_FC */
    int32_t flags;          /* EGOEGO_FLAG_* (0 = defaults) */
} egoego_config;

/* egoego_sample_loop / egoego_ddim_loop capture one diffusion step into a hipGraph and replay it for the rest of the
 * chain (the timestep lives in device memory); this flag launches every kernel of every step individually instead. */
enum { EGOEGO_FLAG_NO_GRAPH = 1 };
/* EGOEGO_PREC_I8X3_FC only: fc's weights (self_attn.fc, TM:55) as THREE int8 slices — w = scale * (q16 + w3 / 256): a second
 * contraction per feature pass adds the third slice's products (about +10 % time per step).  On a trained checkpoint fc on the
 * 16-bit grid is what separates precision 9 from 8 at the end of a 1000-step chain; the Python layer's "auto"
 * tries this form before falling back to precision 8.  Hand the fc weights over UNROUNDED when it is set. */
enum { EGOEGO_FLAG_FC24 = 2 };
/* EGOEGO_PREC_I8X3 only: the FFN contractions (pos_ffn.w_1 / w_2, TM:102-103) on split-bf16 like fc — int8 slices are then confined to
 * the attention layer (Q/K/V projections, QK^T, PV).  Fewer fixed-point sites at the time per step of precision 8 on large batches (its
 * int8-FFN tail is no faster than the split-bf16 one there); the Python layer's "auto" tries this form after "8 prepared".
 * Hand the FFN weights over UNROUNDED when it is set. */
enum { EGOEGO_FLAG_FFN16 = 4 };

/* fp32 device tensors in the reference checkpoint layout, contiguous. */
typedef struct {
    const float* w_q; const float* b_q;      /* self_attn.w_q  (H*dk, 512), (H*dk)   TM:45 */
    const float* w_k; const float* b_k;      /* self_attn.w_k                         TM:46 */
    const float* w_v; const float* b_v;      /* self_attn.w_v  (H*dv, 512)            TM:47 */
    const float* w_fc; const float* b_fc;    /* self_attn.fc   (512, H*dv)            TM:55 */
    const float* ln1_g; const float* ln1_b;  /* self_attn.layer_norm                  TM:57 */
    const float* w_1; const float* b_1;      /* pos_ffn.w_1    (512, 512, 1)          TM:102 */
    const float* w_2; const float* b_2;      /* pos_ffn.w_2    (512, 512, 1)          TM:103 */
    const float* ln2_g; const float* ln2_b;  /* pos_ffn.layer_norm                    TM:104 */
} egoego_layer_weights;

typedef struct {
    const float* start_conv_w; const float* start_conv_b;  /* (512, 2*d_feats, 1), (512)   TM:179 */
    const float* position_vec;                              /* (max_timesteps+1, 512)       TM:180 */
    const float* linear_out_w; const float* linear_out_b;  /* (d_feats, 512), (d_feats)    M:102 */
    const float* time_mlp1_w; const float* time_mlp1_b;    /* (256, 64), (256)             M:113 */
    const float* time_mlp3_w; const float* time_mlp3_b;    /* (512, 256), (512)            M:115 */
    const egoego_layer_weights* layers;                     /* HOST array of n_dec_layers entries */
} egoego_weights;

/* fp32 HOST arrays of num_timesteps entries: the buffers registered at M:191-211. */
typedef struct {
    const float* posterior_mean_coef1;
    const float* posterior_mean_coef2;
    const float* posterior_log_variance_clipped;
    const float* sqrt_recip_alphas_cumprod;
    const float* sqrt_recipm1_alphas_cumprod;
    const float* alphas_cumprod;   /* used by the DDIM sampler only */
} egoego_schedule;

int egoego_abi_version(void);
const char* egoego_last_error(void);

/* Replaces: module construction + .to(device) (M:144-214). */
int egoego_ctx_create(const egoego_config* cfg, int device, egoego_ctx** out);
void egoego_ctx_destroy(egoego_ctx* ctx);

/* Replaces: load_state_dict (trainer_amass_cond_motion_diffusion.py:116-122).  Packs the fp32
 * tensors into the library's split-bf16 fragment-tiled copies and precomputes the time-token
 * table (M:61-73, 111-116).  The caller keeps its originals and may free them after the stream
 * has drained. */
int egoego_load_weights(egoego_ctx* ctx, const egoego_weights* w, void* stream);
int egoego_load_schedule(egoego_ctx* ctx, const egoego_schedule* s, void* stream);

/* Bytes of scratch a call with batch B and window length T needs (256-byte aligned base).
 * Returns 0 (egoego_last_error() says why) for shapes no call accepts: T + 1 > 224 or > max_timesteps, or more than
 * 2^20 padded rows per call (B * 32 * ceil((T + 1) / 32); T + 1 in 129..224 pads to 224): B <= 8192 at T = 120,
 * B <= 4681 at T = 196 — split larger batches, windows are independent. */
size_t egoego_workspace_bytes(const egoego_ctx* ctx, int B, int T);

/* Replaces TransformerDiffusionModel.forward on cat(x, x_cond) (M:118-141, 232-233):
 * d_out[B][T][D] = denoiser(cat(x, x_cond), t).  d_row_mask: optional fp32 [B][T+1] padding mask
 * (1 keep / 0 zero the row after attention and after the FFN, TM:135,139), NULL = all ones. */
int egoego_denoise(egoego_ctx* ctx, const float* d_x, const float* d_x_cond, const int64_t* d_t,
                   const float* d_row_mask, float* d_out, int B, int T,
                   void* d_workspace, size_t workspace_bytes, void* stream);

/* Replaces CondGaussianDiffusion.p_sample (M:248-256): x <- posterior_mean(clamp(x0_pred), x, t)
 * + 1[t>0] * exp(0.5*logvar[t]) * noise, in place.  d_noise: fp32 [B][T][D] (the caller's
 * randn_like draw) or NULL with noise_mode PHILOX/NONE.  clip_denoised mirrors M:242-243. */
int egoego_p_sample(egoego_ctx* ctx, float* d_x, const float* d_x_cond, const int64_t* d_t,
                    const float* d_row_mask, const float* d_noise, int noise_mode,
                    uint64_t seed, int64_t window_offset, int clip_denoised, int B, int T,
                    void* d_workspace, size_t workspace_bytes, void* stream);

/* Replaces the body of p_sample_loop (M:267-268) and of the sliding-window loop (M:392-397):
 * for i = t_start .. t_start-n_steps+1: x <- p_sample(x, i, x_cond, padding_mask); optionally overwrite the first
 * prefix_len frames of every window with d_prefix[B][prefix_len][D] after every step (M:395-397).
 * d_row_mask: the padding mask p_sample_loop hands to every step (M:259,268), fp32 [B][T+1] as for
 * egoego_denoise, or NULL.
 * Noise per step: EGOEGO_NOISE_INJECTED reads d_noise[step][B][T][D] (step 0 = first executed
 * step); EGOEGO_NOISE_PHILOX draws N(0,1) in-kernel from Philox4x32-10 keyed by
 * (seed; window_offset + b, timestep, frame, feature) so results do not depend on how windows are
 * sharded over GPUs. */
int egoego_sample_loop(egoego_ctx* ctx, float* d_x, const float* d_x_cond, int t_start, int n_steps,
                       const float* d_noise, int noise_mode, uint64_t seed, int64_t window_offset,
                       const float* d_prefix, int prefix_len, const float* d_row_mask, int B, int T,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* DDIM sampler (Song et al. 2021) on a strided subsequence of timesteps.  NOT in the reference —
 * no oracle from the reference exists for it.  timesteps_host: HOST int32 array of n strictly descending timesteps.
 * eta in [0, 1]: 0 = deterministic; 1 on the FULL timestep list is the ancestral DDPM chain of egoego_sample_loop
 * (sigma_t^2 = posterior variance), which is how the sampler is tied to the reference's chain (tests).  Noise for
 * eta > 0 as for egoego_sample_loop (d_noise[step][B][T][D] or Philox keyed by (seed; window_offset + b, timestep, ..)). */
int egoego_ddim_loop(egoego_ctx* ctx, float* d_x, const float* d_x_cond, const int32_t* timesteps_host, int n,
                     float eta, const float* d_noise, int noise_mode, uint64_t seed, int64_t window_offset,
                     int B, int T, void* d_workspace, size_t workspace_bytes, void* stream);

/* Replaces pytorch3d.transforms.rotation_6d_to_matrix at M:493: d_in [n][6] -> d_out [n][3][3]. */
int egoego_rot6d_to_matrix(const float* d_in, float* d_out, int64_t n, void* stream);

/* Replaces the post-loop conversion chain of one batch of windows: convert_model_res_to_data (M:469-525) with
 * quat_ik (amass_diffusion_dataset.py:109-125) and the pytorch3d calls inside them (6D -> matrix -> quaternion,
 * un-canonicalise, global -> local rotations, -> axis-angle; de-normalise and rotate the root / head positions).
 *   d_x [B][T][198] normalised model output; d_rec_quat [B][4] (w,x,y,z) = recover_rot_quat (M:470);
 *   d_jpos_min / d_jpos_max [66] = ds.global_jpos_min/max (amass_diffusion_dataset.py:379-392);
 *   parents_host[22]: HOST array, parents_host[j] < j for j > 0 (the SMPL-H kintree, an input: SURVEY.md §8f #1);
 *   d_aa [B][T][22][3] local axis-angle, d_root [B][T][3], d_head [B][T][3] (joint head_idx). */
int egoego_convert_model_res(const float* d_x, const float* d_rec_quat, const float* d_jpos_min, const float* d_jpos_max,
                             const int32_t* parents_host, int head_idx, int B, int T, float* d_aa, float* d_root, float* d_head,
                             void* stream);

/* The head condition of one sliding window (M:355-378): d_head_jpos [B][Tw][3] and d_head_jquat [B][Tw][4] (w,x,y,z) are
 * canonicalised about the first frame's heading (rotate_at_frame, lafan1/utils.py:111-137; its xy moved to the origin) and
 * written into an otherwise zero d_x_start [B][Tw][198] (position dims 3*head_idx.., 6D dims 66 + 6*head_idx..), joint
 * positions min/max-normalised; d_recover_quat [B][4] receives the un-canonicalising rotation (recover_rot_quat of M:470). */
int egoego_window_condition(const float* d_head_jpos, const float* d_head_jquat, const float* d_jpos_min, const float* d_jpos_max,
                            int head_idx, int B, int Tw, float* d_x_start, float* d_recover_quat, void* stream);

/* The condition of the next sliding window (M:399-467) from the current window's converted output: fk_smpl
 * (amass_diffusion_dataset.py:265-293) over the last n_last frames, rotate_at_frame (lafan1/utils.py:111-137) about
 * their first frame's head heading, joint positions min/max-normalised (amass_diffusion_dataset.py:379-392), rotations
 * as 6D.  d_aa [B][Tw][22][3], d_root [B][Tw][3] (egoego_convert_model_res' outputs, root already shifted),
 * d_rest_offsets [22][3] = ds.rest_human_offsets, parents_host[22] as above -> d_prefix [B][n_last][198], the
 * `d_prefix` argument of egoego_sample_loop for the next window. */
int egoego_window_prefix(const float* d_aa, const float* d_root, const float* d_rest_offsets, const float* d_jpos_min,
                         const float* d_jpos_max, const int32_t* parents_host, int head_idx, int B, int Tw, int n_last,
                         float* d_prefix, void* stream);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
 * kernel_id: EGOEGO_K_*.  begin() arms event pairs around every launch of that kernel;
 * end() synchronises the stream's events and returns the mean duration and launch count. */
enum { EGOEGO_K_QKV = 0, EGOEGO_K_ATTN = 1, EGOEGO_K_FC_LN = 2, EGOEGO_K_FFN1 = 3, EGOEGO_K_FFN2_LN = 4,
       EGOEGO_K_EMBED = 5, EGOEGO_K_OUT = 6, EGOEGO_K_COUNT = 7 };
int egoego_profile_begin(egoego_ctx* ctx, int kernel_id);
int egoego_profile_end(egoego_ctx* ctx, double* mean_us, int* launches);
/* The kernel variant the launch site `kernel_id` of a step last dispatched to in this context (the library picks tile shapes and
 * fused / split forms by batch size, window length and precision): e.g. "attn_layer_i8w_kernel", "tail_kernel<1,true,true,true,4,true>".
 * "" before the first step.  What bench.py names its roofline kernels from (no dispatch logic outside the library). */
const char* egoego_last_kernel_name(const egoego_ctx* ctx, int kernel_id);

/* Outlier monitor of the int8-slice precisions (8, 9).  Those keep ONE scale per activation row (16-bit fixed point): a row whose
 * largest entry is far above the rest costs every other entry of the row that many bits.  Every LayerNorm epilogue that
 * quantises its rows records the largest |value| it has seen (one atomicMax per workgroup into the workspace's step state; nothing
 * is recorded in precisions 3 / 1, for layers >= 8, or for the padding rows of the last token block).
 * host_out[2 * layer + k] (k = 0: self_attn.layer_norm, k = 1: pos_ffn.layer_norm; n_out <= 16 entries, 0 = nothing recorded)
 * receives the maxima accumulated by every call on this workspace since the last reset; reset != 0 clears them afterwards.
 * Divide by the rms of the LayerNorm's (gain, bias) to get the row's crest factor — what model.py's runtime guard compares
 * with its measured limit.  SYNCHRONISES `stream` when n_out > 0.  A freshly allocated workspace holds
 * undefined values: call once with n_out = 0, reset = 1 (engine.py does) before relying on the maxima. */
int egoego_outlier_stats(egoego_ctx* ctx, int B, int T, void* d_workspace, size_t workspace_bytes, float* host_out, int n_out,
                         int reset, void* stream);

/* Test/debug only: run the denoiser up to and including `stage` of decoder layer `layer` and
 * return that intermediate as fp32 row-major.  Stages: EGOEGO_DBG_*.  Output shapes:
 *   EMBED/ATTN_LN/FFN_HIDDEN/LAYER_OUT: [B][T+1][512]; Q/K/V: [B][H][T+1][256]; ATTN_OUT: [B][T+1][H*256]. */
enum { EGOEGO_DBG_EMBED = 0, EGOEGO_DBG_Q = 1, EGOEGO_DBG_K = 2, EGOEGO_DBG_V = 3, EGOEGO_DBG_ATTN_OUT = 4,
       EGOEGO_DBG_ATTN_LN = 5, EGOEGO_DBG_FFN_HIDDEN = 6, EGOEGO_DBG_LAYER_OUT = 7 };
int egoego_debug_stage(egoego_ctx* ctx, const float* d_x, const float* d_x_cond, const int64_t* d_t,
                       const float* d_row_mask, int layer, int stage, float* d_out, int B, int T,
                       void* d_workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGOEGO_HIP_H */
