"""Python surface of the reference's stage-2 diffusion module, backed by the HIP library.

Mirrors the reference interface this path exposes (names, argument meaning, error behaviour):
  CondGaussianDiffusion            /root/reference/egoego/model/transformer_cond_diffusion_model.py:143
    .sample / .p_sample_loop / .p_sample / .p_mean_variance / .q_posterior / .predict_start_from_noise
  TransformerDiffusionModel        ...:75   (parameter container; `denoise_fn`)
  Decoder & friends                /root/reference/egoego/model/transformer_module.py:36-225
so that `trainer.ema.ema_model.sample(x_start, cond_mask)` keeps working and reference checkpoints
load key-for-key (state_dict layout in SURVEY.md §8b).

The nn.Modules below hold parameters only.  Every SAMPLING method runs on the HIP library and raises
if it is unavailable or the tensors are not on a ROCm device — there is no PyTorch/CPU fallback.
The plain-PyTorch `forward` of the parameter containers exists solely for the training loss
(`p_losses` / `forward`), which is outside the accelerated path.
"""
import math
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .engine import HipEngine
from . import plan as plan_mod
from .precision import PrecisionProbe, _engine_cfg
from .synthetic import sinusoid_position_table


# ------------------------------------------------------------------------- parameter containers
class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freqs = torch.exp(torch.arange(half, device=t.device) * -(math.log(10000) / (half - 1)))
        ang = t[:, None] * freqs[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class MultiHeadAttention(nn.Module):
    def __init__(self, n_head, d_model, d_k, d_v):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_q = nn.Linear(d_model, n_head * d_k)
        self.w_k = nn.Linear(d_model, n_head * d_k)
        self.w_v = nn.Linear(d_model, n_head * d_v)
        for lin, dd in ((self.w_q, d_k), (self.w_k, d_k), (self.w_v, d_v)):
            nn.init.normal_(lin.weight, mean=0, std=np.sqrt(2.0 / (d_model + dd)))
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.layer_norm = nn.LayerNorm(d_model)
        self.p_drop = 0.1

    def forward(self, h):
        B, L, _ = h.shape
        q = self.w_q(h).view(B, L, self.n_head, self.d_k).transpose(1, 2)
        k = self.w_k(h).view(B, L, self.n_head, self.d_k).transpose(1, 2)
        v = self.w_v(h).view(B, L, self.n_head, self.d_v).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(self.d_k), dim=-1)
        p = F.dropout(p, self.p_drop, self.training)
        o = (p @ v).transpose(1, 2).reshape(B, L, self.n_head * self.d_v)
        return self.layer_norm(F.dropout(self.fc(o), self.p_drop, self.training) + h)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_in, d_hid):
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, 1)
        self.w_2 = nn.Conv1d(d_hid, d_in, 1)
        self.layer_norm = nn.LayerNorm(d_in)
        self.p_drop = 0.1

    def forward(self, h):
        y = F.linear(F.relu(F.linear(h, self.w_1.weight[:, :, 0], self.w_1.bias)), self.w_2.weight[:, :, 0], self.w_2.bias)
        return self.layer_norm(F.dropout(y, self.p_drop, self.training) + h)


class DecoderLayer(nn.Module):
    def __init__(self, d_model, n_head, d_k, d_v):
        super().__init__()
        self.self_attn = MultiHeadAttention(n_head, d_model, d_k, d_v)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_model)

    def forward(self, h, keep):
        h = self.self_attn(h)
        if keep is not None:
            h = h * keep
        h = self.pos_ffn(h)
        return h if keep is None else h * keep


class Decoder(nn.Module):
    def __init__(self, d_feats, d_model, n_layers, n_head, d_k, d_v, max_timesteps):
        super().__init__()
        self.start_conv = nn.Conv1d(d_feats, d_model, 1)
        self.position_vec = nn.Embedding.from_pretrained(sinusoid_position_table(max_timesteps + 1, d_model), freeze=True)
        self.layer_stack = nn.ModuleList([DecoderLayer(d_model, n_head, d_k, d_v) for _ in range(n_layers)])

    def forward(self, x_all, time_token, keep):
        e = F.linear(x_all, self.start_conv.weight[:, :, 0], self.start_conv.bias)
        h = torch.cat((time_token[:, None, :], e), dim=1)
        h = h + self.position_vec.weight[1:h.shape[1] + 1][None]
        for layer in self.layer_stack:
            h = layer(h, keep)
        return h


class TransformerDiffusionModel(nn.Module):
    def __init__(self, d_feats, d_model, n_dec_layers, n_head, d_k, d_v, max_timesteps):
        super().__init__()
        self.d_feats, self.d_model, self.n_head, self.n_dec_layers = d_feats, d_model, n_head, n_dec_layers
        self.d_k, self.d_v, self.max_timesteps = d_k, d_v, max_timesteps
        self.motion_transformer = Decoder(d_feats * 2, d_model, n_dec_layers, n_head, d_k, d_v, max_timesteps)
        self.linear_out = nn.Linear(d_model, d_feats)
        self.time_mlp = nn.Sequential(SinusoidalPosEmb(64), nn.Linear(64, 256), nn.GELU(), nn.Linear(256, d_model))

    def forward(self, src, noise_t, padding_mask=None):
        """Plain-PyTorch forward for the TRAINING loss only (autograd); sampling never calls it."""
        keep = None if padding_mask is None else padding_mask.squeeze(1).unsqueeze(-1).float()
        h = self.motion_transformer(src, self.time_mlp(noise_t), keep)
        return self.linear_out(h[:, 1:])


# ------------------------------------------------------------------------- schedule
def _cosine_betas(timesteps, s=0.008):
    u = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    abar = torch.cos(((u / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)


def _linear_betas(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def _world(group):
    import torch.distributed as tdist
    return tdist.get_world_size(group) if tdist.is_available() and tdist.is_initialized() else 1


def _extract(a, t, x_shape):
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


class _EngineSlot:
    """Holds the (non-copyable) HIP context; deep copies (ema_pytorch.EMA) start empty."""

    def __init__(self):
        self.engine, self.key, self.fingerprint = None, None, None
        self.engine_masked = None  # the context for padding-mask calls when the main one stores mean-shifted rows
        self.plan = None       # what the context was packed from (plan.resolve)
        self.ramp = None       # fingerprint weights (one per packed element)
        self.noise_buf = None  # scratch of the torch-RNG chain, reused across sample() calls
        self.envelope = None   # per LayerNorm site: the largest row maximum the pack-time probe has validated (runtime guard)
        self.demoted = False   # the runtime guard measured the int8 precision outside the limit on live inputs: "auto" now means 3
        self.force_repack = False  # ... and asked for the re-pack that applies it (a re-pack for any other reason clears `demoted`)
        self.unprobed_work = 0     # steps sampled on a "small job" context (split-bf16, no probe): plan.is_small_job
        self.caller_checked = False  # the plan was measured once more on a caller's conditions (at most once per packed weights)

    def __deepcopy__(self, memo):
        return _EngineSlot()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()


class CondGaussianDiffusion(nn.Module):
    def __init__(self, d_feats, d_model, n_head, n_dec_layers, d_k, d_v, max_timesteps, out_dim, timesteps=1000,
                 loss_type="l1", objective="pred_noise", beta_schedule="cosine", p2_loss_weight_gamma=0.0,
                 p2_loss_weight_k=1, batch_size=None):
        super().__init__()
        self.denoise_fn = TransformerDiffusionModel(d_feats=d_feats, d_model=d_model, n_head=n_head, d_k=d_k, d_v=d_v,
                                                    n_dec_layers=n_dec_layers, max_timesteps=max_timesteps)
        self.objective = objective
        self.seq_len = max_timesteps - 1
        self.out_dim = out_dim
        if beta_schedule == "linear":
            betas = _linear_betas(timesteps)
        elif beta_schedule == "cosine":
            betas = _cosine_betas(timesteps)
        else:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        alphas = 1.0 - betas
        abar = torch.cumprod(alphas, dim=0)
        abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
        self.num_timesteps = int(betas.shape[0])
        self.loss_type = loss_type
        post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
        for name, val in (
            ("betas", betas), ("alphas_cumprod", abar), ("alphas_cumprod_prev", abar_prev),
            ("sqrt_alphas_cumprod", torch.sqrt(abar)), ("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - abar)),
            ("log_one_minus_alphas_cumprod", torch.log(1.0 - abar)), ("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / abar)),
            ("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / abar - 1)), ("posterior_variance", post_var),
            ("posterior_log_variance_clipped", torch.log(post_var.clamp(min=1e-20))),
            ("posterior_mean_coef1", betas * torch.sqrt(abar_prev) / (1.0 - abar)),
            ("posterior_mean_coef2", (1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar)),
            ("p2_loss_weight", (p2_loss_weight_k + abar / (1 - abar)) ** -p2_loss_weight_gamma),
        ):
            self.register_buffer(name, val.to(torch.float32))
        # MI355X-specific knobs (not in the reference): operand precision and the noise source.
        # Precisions: 9 (PREC_I8X3_FC) = every contraction of the layers on int8 slices, int8-only activations between the kernels of
        # a step — the fastest; 8 (PREC_I8X3) = fc, linear_out and the residuals on split-bf16; 3 (PREC_BF16X3) = split-bf16
        # everywhere (~2e-5 on one forward on every checkpoint measured, ~85 % more time per step than 9).  The int8 precisions are
        # 16-bit FIXED point per row: what they lose depends on the checkpoint, so "auto" (default) is decided by MEASUREMENT and
        # remembered (plan.py): the ladder 9 as is -> 9 prepared -> 9 prepared + fc24 -> 8 as is -> 8 prepared -> 8 prepared + ffn16 -> 3, each candidate
        # against split-bf16 on a probe batch (stage 1: the end of a chain + two forwards; stage 2: whole chains on 128 windows);
        # the verdict is cached on disk per checkpoint; a chain-level call shorter than the probe runs split-bf16 unprobed; under
        # torch.distributed the sharded entry points (dist.py) make all ranks pack rank 0's plan.  `hip_precision_used` /
        # `hip_precision_probe` tell what was picked, from where ("probe" / "cache" / "small job" / "group rank 0 (...)"), and what
        # was measured.  While an int8 precision runs, the LayerNorm epilogues record their largest row maxima; at the end of a
        # chain `_outlier_guard` compares them with what the probe saw and, beyond that envelope, re-measures on the chain's own
        # tensors and steps "auto" down to 3.
        self.hip_precision = "auto"
        self.hip_precision_used = None
        self.hip_precision_probe = None   # the plan's measurement: {"errors": {"9 as is": .., "9 as is, full chain": ..}, "limit", "chain_limit", "row_max", "form", "source", ..}
        self.hip_probe_at_pack = True     # False: no measurement at all (auto = 9, absolute envelope for the runtime guard)
        self.hip_probe_full_chain = True  # False: 'auto' trusts stage 1 of the probe (saves ~1 s per measured checkpoint at 1000 steps)
        self.hip_fc24 = True              # 'auto' may run precision 9 with fc's weights as three int8 slices (FLAG_FC24, ~+10 % per step) before falling back to 8
        self.hip_ffn16 = True             # 'auto' may run precision 8 with the FFN on split-bf16 (FLAG_FFN16: int8 slices in the attention layer only) before falling back to 3
        self.hip_int8_prep = "auto"       # pack-time preparation of int8 precisions (precision.py): "auto" = only when the plain packing fails the probe; "always"; "never"
        self.hip_plan_cache = True        # remember / reuse the verdict on disk ($EGOEGO_HIP_CACHE, default ~/.cache/egoego_hip; plan.py)
        self.hip_plan_override = None     # tools/tests: (precision, prepared, flags) — pack exactly this form (measured and reported, never rejected)
        self.hip_outlier_guard = True     # False: no read-back (and no stream sync) at the end of a chain
        self.hip_outlier_seen = None      # per LayerNorm site, the largest row maximum of the last guarded chain
        self.hip_graph = True        # replay one captured step per chain (hipGraph); False launches every kernel
        self.sampling_rng = "torch"  # "torch": reference RNG draw order; "philox": in-kernel, shard-invariant
        self.philox_seed = 0
        self._slot = _EngineSlot()

    # ------------------------------------------------------------------ HIP engine plumbing
    _SCHEDULE_BUFFERS = ("posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped",
                         "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "alphas_cumprod")

    def _packed_tensors(self):
        """Every tensor the HIP context holds a packed copy of: the denoiser's parameters, the frozen position table
        and the schedule buffers the step kernels read."""
        return list(self.denoise_fn.parameters()) + [getattr(self, n) for n in self._SCHEDULE_BUFFERS]

    def _engine_key(self):
        """Cheap (host-only) identity of the packed state.  In-place writes through `.data` (ema_pytorch's
        `ma_params.data.copy_` / `.data.lerp_`) bump neither `data_ptr` nor `_version`: those are caught by
        `_weights_fingerprint()` at chain-level entry points and by the `invalidate_engine()` hooks."""
        dev = self.betas.device
        return (str(dev), self.hip_precision, bool(self.hip_graph), self.objective, int(self.betas.shape[0]),
                self.hip_int8_prep, bool(self.hip_probe_at_pack), bool(self.hip_probe_full_chain), bool(self.hip_fc24), bool(self.hip_ffn16),
                None if self.hip_plan_override is None else tuple(self.hip_plan_override), bool(self.hip_plan_cache),
                tuple((p.data_ptr(), p._version) for p in self._packed_tensors()))

    @torch.no_grad()
    def _weights_fingerprint(self):
        """Device-side checksum of everything packed into the HIP context (one concatenation, two reductions, one host
        sync): a position-weighted signed sum (hashed weights in [1, 2), so sign flips, swaps and permutations move it)
        and the sum of squares, both accumulated in float64.  A checksum, not a proof: an update that happens to
        preserve both goes unnoticed — call invalidate_engine() when in doubt."""
        flat = torch.cat([t.detach().reshape(-1).float() for t in self._packed_tensors()])
        ramp = self._slot.ramp
        if ramp is None or ramp.shape != flat.shape or ramp.device != flat.device:
            # position weights in [1, 2) from an integer hash of the index (Knuth's multiplier, 24 bits kept: exact in fp32), so that
            # neighbours get unrelated weights whatever the element count (a linear fp32 ramp repeats values beyond 2^23 elements)
            idx = torch.arange(flat.numel(), device=flat.device, dtype=torch.int64)
            ramp = self._slot.ramp = 1.0 + ((idx * 2654435761) & 0xFFFFFF).to(torch.float32) / float(1 << 24)
        fp = torch.stack(((flat * ramp).sum(dtype=torch.float64), (flat * flat).sum(dtype=torch.float64)))
        return tuple(fp.tolist())

    def invalidate_engine(self):
        """Drop the packed copy of the weights: the next sampling call re-packs from the module's current tensors.
        Called by load_state_dict / .to() / .cuda() / .half() etc.; call it yourself after writing weights through
        `.data` if you then use the per-step API (`p_sample`, `denoise`), which does not checksum."""
        for e in (self._slot.engine, self._slot.engine_masked):
            if e is not None:
                e.close()
        self._slot.engine, self._slot.key, self._slot.fingerprint = None, None, None
        self._slot.engine_masked, self._slot.plan = None, None
        self._slot.noise_buf = None
        self._slot.envelope, self._slot.demoted, self._slot.force_repack = None, False, False
        self._slot.unprobed_work = 0
        self._slot.caller_checked = False

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_engine()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if "_slot" in self.__dict__:
            self.invalidate_engine()
        return out

    def hip_engine(self, verify=False, masked=False, job=None, group=None, conditions=None):
        """The HIP context for the module's current device/weights (packed lazily, re-packed when parameters change
        or the module moves).  verify=True (every chain-level entry point: sample, p_sample_loop, ddim_sample, the
        sliding-window harness) additionally compares a device-side checksum of the weights with the one taken when
        they were packed.  masked=True: the context for calls that carry a padding mask — the same precision, but packed without
        the mean shift of the LayerNorm rows when the main context has it (precision.py: a mask zeroes rows AFTER the shift).
        job=(windows, frames, steps): the chain-level call this context is for — under hip_precision="auto" a job shorter than the
        precision probe runs split-bf16 unprobed (plan.py).  group: a torch.distributed process group whose ranks must all sample
        from ONE plan (dist.py passes it): every rank reports whether its packed copy is stale, and if any is, group rank 0
        resolves the plan and broadcasts it (prepared tensors included) — call it on every rank of the group.
        conditions: the x_cond rows [n, T, D] of the chain-level call (T = seq_len): stage 2 of the measurement runs on them, and an int8 form
        accepted earlier on the probe's self-generated conditions is measured once more on these (plan.wants_caller_conditions)."""
        dev = self.betas.device
        if dev.type != "cuda":
            raise _lib.EgoEgoHipError(
                "CondGaussianDiffusion sampling runs on the MI355X HIP path only: move the module to a ROCm "
                f"device first (it is on {dev}); there is no CPU fallback")
        slot = self._slot
        key = self._engine_key()
        fp = self._weights_fingerprint() if (verify or group is not None) else None
        if conditions is not None and not (conditions.dim() == 3 and conditions.shape[0] >= 1 and int(conditions.shape[1]) == int(self.seq_len)):
            conditions = None  # (a shorter trailing window of the harness, an empty shard: the probe keeps its own)
        remeasure = (verify and slot.engine is not None and slot.key == key and not slot.force_repack and not slot.caller_checked and job is not None
                     and not plan_mod.is_small_job(self, job) and plan_mod.wants_caller_conditions(self, slot.plan, conditions))
        stale = (slot.engine is None or slot.key != key or (fp is not None and slot.fingerprint != fp) or slot.force_repack or remeasure
                 or (slot.plan is not None and slot.plan["source"] == "small job" and job is not None and not plan_mod.is_small_job(self, job)))
        synced = group is not None and _world(group) > 1
        if synced:
            # one all_reduce, always: is any rank's copy stale, and do all ranks hold the same weights?
            got = plan_mod.group_max([1.0 if stale else 0.0, fp[0], -fp[0], fp[1], -fp[1], 1.0 if remeasure else 0.0], group)
            remeasure = got[5] > 0
            if got[1] != -got[2] or got[3] != -got[4]:
                raise _lib.EgoEgoHipError("the ranks of the process group hold different weights (checksums differ): window-sharded "
                                          "sampling needs the same checkpoint on every rank")
            stale = got[0] > 0
        if stale:
            for e in (slot.engine, slot.engine_masked):
                if e is not None:
                    e.close()
            slot.engine = slot.engine_masked = None
            slot.caller_checked = bool(remeasure)
            if not slot.force_repack:
                slot.demoted = False  # new weights / device / settings: measured afresh
            slot.force_repack = False
            if self.objective not in ("pred_noise", "pred_x0"):
                raise ValueError(f"unknown objective {self.objective}")
            if synced:
                import torch.distributed as tdist
                plan, failure = None, None
                if tdist.get_rank(group) == 0:
                    try:
                        plan = plan_mod.resolve(self, job, fp, conditions, remeasure)
                    except Exception as e:  # (out of memory in the probe, ...: the peers wait in the broadcast and must hear about it)
                        failure, plan = e, {"error": repr(e)}
                plan = plan_mod.group_broadcast(plan, group)
                if "error" in plan:
                    if failure is not None:
                        raise failure
                    raise _lib.EgoEgoHipError(f"group rank 0 failed while resolving the precision plan: {plan['error']}")
                plan = dict(plan, source=plan["source"] if tdist.get_rank(group) == 0 else f"group rank 0 ({plan['source']})")
            else:
                plan = plan_mod.resolve(self, job, fp, conditions, remeasure)
            slot.plan = plan
            plan_mod.adopt(self, plan)
            slot.engine = HipEngine(_engine_cfg(self), plan["sd"] if plan["sd"] is not None else self.state_dict(), dev, plan["precision"],
                                    (0 if self.hip_graph else _lib.FLAG_NO_GRAPH) | plan["flags"], row_shift=plan["row_shift"])
            slot.key = key
            slot.fingerprint = fp if fp is not None else self._weights_fingerprint()
        if masked and slot.plan["row_shift"]:
            if slot.engine_masked is None:
                slot.engine_masked = HipEngine(_engine_cfg(self), plan_mod.masked_state(slot.plan, self.state_dict()), dev, slot.plan["precision"],
                                               (0 if self.hip_graph else _lib.FLAG_NO_GRAPH) | slot.plan["flags"])
            return slot.engine_masked
        return slot.engine

    # the limits of the pack-time measurement live in plan.py (kept here for callers that read them off the module)
    PROBE_LIMIT = plan_mod.PROBE_LIMIT
    PROBE_TAIL = plan_mod.PROBE_TAIL
    CHAIN_LIMIT = plan_mod.CHAIN_LIMIT
    ENVELOPE_MARGIN = 1.5    # the runtime guard re-measures when a LayerNorm row maximum exceeds this multiple of what the probe validated
    ENVELOPE_ABSOLUTE = 8.0  # ... or this value when no probe ran (rows of the reference's initialisation peak at 4-5)

    def _note_job(self, job):
        """A chain-level call ran on a context packed for a small job (split-bf16, unprobed): add its work up (plan.is_small_job)."""
        if self._slot.plan is not None and self._slot.plan["source"] == "small job":
            self._slot.unprobed_work += job[2]

    @torch.no_grad()
    def _outlier_guard(self, eng, x, x_cond, group=None):
        """End of a chain in an int8 precision: read the LayerNorm row maxima the chain produced (one stream sync).  Inside the
        envelope the pack-time probe validated (x ENVELOPE_MARGIN) nothing else happens.  Beyond it the probe is repeated on the
        chain's OWN tensors (its final x re-noised and walked down again): within PROBE_LIMIT the envelope grows to what was seen;
        outside it a RuntimeWarning says so and `hip_precision = "auto"` steps down to split-bf16 from the next call on (this
        chain's result stands: it is what was measured).
        group (dist.py): the decision is COLLECTIVE — the row maxima are all-reduced (MAX) over the ranks, group rank 0 (which holds
        the global batch's first windows, like a single rank would) re-measures, and its verdict is broadcast: every rank steps
        down, or none does.  Every rank of the group must call this, with an empty shard too."""
        synced = group is not None and _world(group) > 1
        active = bool(self.hip_outlier_guard) and self.hip_precision_used in (_lib.PREC_I8X3, _lib.PREC_I8X3_FC)
        if not synced and (not active or x.shape[0] == 0):
            return
        n_sites = min(_lib.OUTLIER_SITES, 2 * self.denoise_fn.n_dec_layers)
        seen = eng.outlier_stats(x.shape[0], x.shape[1], reset=True) if (active and x.shape[0]) else [0.0] * n_sites
        env = self._slot.envelope
        lim = [self.ENVELOPE_ABSOLUTE] * n_sites if env is None else [self.ENVELOPE_MARGIN * max(v, 1e-30) for v in env]
        if synced:
            # ONE all_reduce carries everything the decision depends on, so every rank takes the same branch whatever its local
            # state: the row maxima (MAX), the limits (MIN, as MAX of the negatives: a rank whose envelope grew in an unsharded
            # call in between does not leave alone), and whether any / every rank is running an int8 precision with the guard on
            got = plan_mod.group_max(list(seen) + [-v for v in lim] + [1.0 if active else 0.0, 0.0 if active else 1.0], group)
            seen, lim = got[:n_sites], [-v for v in got[n_sites:2 * n_sites]]
            if got[-2] > 0 and got[-1] > 0:
                raise _lib.EgoEgoHipError("the ranks of the process group sample in different precisions (some int8 with the runtime guard, some not): "
                                          "pack through the sharded entry points (dist.py) with verify=True so that all ranks hold one plan")
            if got[-2] == 0:
                return
        self.hip_outlier_seen = seen
        if all(s <= l for s, l in zip(seen, lim)):
            return
        prec, plan = self.hip_precision_used, self._slot.plan
        measure = True
        if synced:
            import torch.distributed as tdist
            measure = tdist.get_rank(group) == 0
        ok, err, failure = True, 0.0, None
        if measure and x.shape[0]:
            try:
                n = min(int(x.shape[0]), 8)
                probe = PrecisionProbe(self, probe=(x[:n], x_cond[:n]), tail=self.PROBE_TAIL)
                try:
                    psd = plan["sd"] if plan["sd"] is not None else probe.sd
                    err, _ = probe.error(psd, prec, plan["row_shift"], plan["flags"])
                    ok = err <= self.PROBE_LIMIT
                    if ok and self.hip_probe_full_chain:
                        err, _ = probe.chain_error(psd, prec, plan["row_shift"], plan["flags"])
                        ok = err <= self.CHAIN_LIMIT
                finally:
                    probe.close()
            except Exception as e:  # (the peers wait in the broadcast below: they must hear about it)
                if not synced:
                    raise
                failure = e
        if synced:
            ok, err, msg = plan_mod.group_broadcast((ok, err, None if failure is None else repr(failure)), group)
            if msg is not None:
                if failure is not None:
                    raise failure
                raise _lib.EgoEgoHipError(f"group rank 0 failed while re-measuring the precision: {msg}")
        if ok:
            self._slot.envelope = [max(a, b) for a, b in zip(seen, env)] if env is not None else list(seen)
            return
        worst = max(range(len(seen)), key=lambda i: seen[i] / lim[i])
        msg = (f"LayerNorm rows of this chain peak at {seen[worst]:.1f} (layer {worst // 2}, "
               f"{'self_attn' if worst % 2 == 0 else 'pos_ffn'}.layer_norm), beyond what the pack-time probe validated, and precision "
               f"{prec} differs from split-bf16 by {err:.1e} on this chain's own tensors (limits {self.PROBE_LIMIT:.0e} / {self.CHAIN_LIMIT:.1e})")
        if self.hip_precision == "auto":
            self._slot.demoted = self._slot.force_repack = True  # re-pack at the next call
            plan_mod.cache_drop(self._slot.plan.get("cache_key") if self._slot.plan else None)  # (a later process must not start from the int8 verdict again)
            warnings.warn(msg + ": hip_precision='auto' uses split-bf16 (3) from the next call on", RuntimeWarning, stacklevel=4)
        else:
            warnings.warn(msg + f": hip_precision={prec} was set explicitly and is kept", RuntimeWarning, stacklevel=4)

    def _check_t(self, t):
        """The reference indexes its schedule buffers with t (`extract`, M:36-39) and the time embedding accepts any
        value; an out-of-range timestep raises there (index error / device assert).  Same here, before the kernels
        (which clamp) see it."""
        lo, hi = int(t.min()), int(t.max())
        n = int(self.betas.shape[0])
        if lo < 0 or hi >= n:
            raise IndexError(f"timestep out of range: got [{lo}, {hi}], schedule has {n} steps")

    @staticmethod
    def _f32c(t):
        return t.to(torch.float32).contiguous()

    # ------------------------------------------------------------------ reference API (sampling)
    def predict_start_from_noise(self, x_t, t, noise):
        return (_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (_extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, _extract(self.posterior_variance, t, x_t.shape),
                _extract(self.posterior_log_variance_clipped, t, x_t.shape))

    @torch.no_grad()
    def denoise(self, x, t, x_cond, padding_mask=None):
        """denoise_fn(cat(x, x_cond), t) on the HIP path."""
        self._check_t(t)
        return self.hip_engine(masked=padding_mask is not None).denoise(self._f32c(x), self._f32c(x_cond), t.long().contiguous(), padding_mask)

    def p_mean_variance(self, x, t, x_cond, clip_denoised, padding_mask=None):
        out = self.denoise(x, t, x_cond, padding_mask)
        if self.objective == "pred_noise":
            x_start = self.predict_start_from_noise(x, t=t, noise=out)
        elif self.objective == "pred_x0":
            x_start = out
        else:
            raise ValueError(f"unknown objective {self.objective}")
        if clip_denoised:
            x_start.clamp_(-1.0, 1.0)
        return self.q_posterior(x_start=x_start, x_t=x, t=t)

    @torch.no_grad()
    def p_sample(self, x, t, x_cond, clip_denoised=True, padding_mask=None, noise=None):
        """One ancestral step (fused on the GPU).  `noise=None` draws torch.randn_like(x), exactly
        where the reference draws it."""
        eng = self.hip_engine(masked=padding_mask is not None)
        self._check_t(t)
        if noise is None:
            noise = torch.randn_like(x)
        out = self._f32c(x).clone()
        eng.p_sample_(out, self._f32c(x_cond), t.long().contiguous(), self._f32c(noise), padding_mask, clip_denoised)
        return out

    @torch.no_grad()
    def p_sample_loop(self, shape, x_start, cond_mask, padding_mask=None, noise=None, prefix=None):
        """x_T ~ N(0,I); x_cond = x_start*(1-m) + m*N(0,I); then num_timesteps ancestral steps.

        noise (optional): dict with 'x_T' [B,T,D], 'cond' [B,T,D] and 'steps' [S,B,T,D] (step 0 = first
        executed, i.e. t = S-1) to inject the reference's own draws; otherwise torch's generator is
        consumed in the reference's order (sampling_rng='torch') or the per-step noise is drawn in-kernel
        (sampling_rng='philox').
        """
        job = (int(shape[0]), int(shape[1]), int(self.num_timesteps))
        device = self.betas.device
        if device.type != "cuda":
            self.hip_engine()  # (raises: no CPU path)
        S = self.num_timesteps
        if noise is not None:
            x = self._f32c(noise["x_T"].to(device)).clone()
            cn = noise["cond"].to(device)
        else:
            x = torch.randn(shape, device=device)
            cn = torch.randn_like(x_start).to(x_start.device)
        x_cond = self._f32c(x_start * (1.0 - cond_mask) + cond_mask * cn)
        # (after the draws, which keeps the reference's RNG order whatever the measurement does; the caller's conditions shape its stage 2)
        eng = self.hip_engine(verify=True, masked=padding_mask is not None, job=job, conditions=x_cond)
        pfx = None if prefix is None else self._f32c(prefix)
        # padding_mask reaches every step's denoiser pass like in the reference (M:259, 268), inside the one HIP loop
        if noise is not None:
            steps = noise["steps"]
            chunk = max(1, min(S, (1 << 26) // max(1, x.numel())))
            for s0 in range(0, S, chunk):
                n = min(chunk, S - s0)
                eng.sample_loop_(x, x_cond, S - 1 - s0, n, noise=self._f32c(steps[s0:s0 + n].to(device)), prefix=pfx,
                                 row_mask=padding_mask)
        elif self.sampling_rng == "philox":
            eng.sample_loop_(x, x_cond, S - 1, S, noise_mode=_lib.NOISE_PHILOX, seed=self.philox_seed, prefix=pfx,
                             row_mask=padding_mask)
        elif self.sampling_rng == "torch":
            self._torch_rng_chain(eng, x, x_cond, S, pfx, padding_mask)
        else:
            raise ValueError(f"unknown sampling_rng {self.sampling_rng}")
        self._note_job(job)
        self._outlier_guard(eng, x, x_cond)
        return x

    def _torch_rng_chain(self, eng, x, x_cond, S, prefix=None, padding_mask=None):
        """The S ancestral steps with torch's generator consumed exactly as the reference consumes it — one
        `randn_like(x)` per step, t = S-1 .. 0 (M:253, 267-268) — but drawn a chunk of steps ahead into one buffer
        that a single call of the HIP loop then walks: the draws are the same `normal_` launches on the same shape
        in the same order, only no longer interleaved with the steps, so the embed operand is packed once per chunk
        instead of once per step and the steps of a chunk replay as one captured graph.  The buffer (at most 128 MiB:
        enough steps per chunk to amortise the per-call work) lives on the engine slot and is reused by later calls."""
        chunk = max(1, min(S, (1 << 25) // max(1, x.numel())))
        need = chunk * x.numel()
        buf = self._slot.noise_buf
        if buf is None or buf.numel() < need or buf.device != x.device:
            buf = self._slot.noise_buf = torch.empty(need, device=x.device, dtype=torch.float32)
        buf = buf[:need].view((chunk,) + tuple(x.shape))
        for s0 in range(0, S, chunk):
            n = min(chunk, S - s0)
            for j in range(n):
                buf[j].normal_()
            eng.sample_loop_(x, x_cond, S - 1 - s0, n, noise=buf[:n], prefix=prefix, row_mask=padding_mask)

    @torch.no_grad()
    def sample(self, x_start, cond_mask, padding_mask=None, noise=None):
        # like the reference (M:528-535): padding_mask is accepted and ignored; eval() then train()
        self.denoise_fn.eval()
        res = self.p_sample_loop(x_start.shape, x_start, cond_mask, noise=noise)
        self.denoise_fn.train()
        return res

    @torch.no_grad()
    def ddim_sample(self, x_start, cond_mask, n_steps=50, noise=None, eta=0.0):
        """DDIM on a uniform stride of the training timesteps (eta=0: deterministic; eta > 0 draws in-kernel Philox
        noise keyed by `philox_seed`; eta=1 with n_steps=num_timesteps is the ancestral chain).  Not part of the
        reference (it only has the full ancestral chain); provided for BASELINE config 4."""
        job = (int(x_start.shape[0]), int(x_start.shape[1]), int(n_steps))
        device = self.betas.device
        if device.type != "cuda":
            self.hip_engine()  # (raises: no CPU path)
        if noise is not None:
            x, cn = self._f32c(noise["x_T"].to(device)).clone(), noise["cond"].to(device)
        else:
            x, cn = torch.randn(x_start.shape, device=device), torch.randn_like(x_start)
        x_cond = self._f32c(x_start * (1.0 - cond_mask) + cond_mask * cn)
        eng = self.hip_engine(verify=True, job=job, conditions=x_cond)
        ts = sorted({int(round(v)) for v in np.linspace(0, self.num_timesteps - 1, n_steps)}, reverse=True)
        eng.ddim_loop_(x, x_cond, ts, eta=eta, seed=self.philox_seed)
        self._note_job(job)
        self._outlier_guard(eng, x, x_cond)
        return x

    # ------------------------------------------------------------------ sliding-window harness (harness.py)
    def convert_model_res_to_data(self, ds, all_res_list, recover_rot_quat, curr_global_head_jpos=None):
        from . import harness
        return harness.convert_model_res_to_data(ds, all_res_list, recover_rot_quat, curr_global_head_jpos)

    @torch.no_grad()
    def p_sample_loop_sliding_window_w_canonical(self, ds, shape, global_head_jpos, global_head_jquat, cond_mask, noise=None,
                                                 parents=None):
        from . import harness
        return harness.p_sample_loop_sliding_window_w_canonical(self, ds, shape, global_head_jpos, global_head_jquat,
                                                                cond_mask, noise=noise, parents=parents)

    @torch.no_grad()
    def sample_sliding_window_w_canonical(self, ds, global_head_jpos, global_head_jquat, x_start, cond_mask, noise=None,
                                          parents=None, window_offset=0):
        """`parents` (or `ds.parents`) overrides the SMPL-H kintree the conversion chain walks;
        `window_offset`: global index of sequence 0 (dist.harness_sharded)."""
        from . import harness
        return harness.sample_sliding_window_w_canonical(self, ds, global_head_jpos, global_head_jquat, x_start, cond_mask,
                                                         noise=noise, parents=parents, window_offset=window_offset)

    # ------------------------------------------------------------------ training half (plain PyTorch)
    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (_extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    @property
    def loss_fn(self):
        if self.loss_type == "l1":
            return F.l1_loss
        if self.loss_type == "l2":
            return F.mse_loss
        raise ValueError(f"invalid loss type {self.loss_type}")

    def p_losses(self, x_start, cond_mask, t, noise=None, padding_mask=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        x = self.q_sample(x_start=x_start, t=t, noise=noise)
        x_cond = x_start * (1.0 - cond_mask) + cond_mask * torch.randn_like(x_start)
        out = self.denoise_fn(torch.cat((x, x_cond), dim=-1), t, padding_mask)
        if self.objective == "pred_noise":
            target = noise
        elif self.objective == "pred_x0":
            target = x_start
        else:
            raise ValueError(f"unknown objective {self.objective}")
        loss = self.loss_fn(out, target, reduction="none")
        if padding_mask is not None:
            loss = loss * padding_mask[:, 0, 1:][:, :, None]
        loss = loss.flatten(1).mean(dim=1) * _extract(self.p2_loss_weight, t, (loss.shape[0],))
        return loss.mean()

    def forward(self, x_start, cond_mask, padding_mask=None):
        t = torch.randint(0, self.num_timesteps, (x_start.shape[0],), device=x_start.device).long()
        return self.p_losses(x_start, cond_mask, t, padding_mask=padding_mask)
