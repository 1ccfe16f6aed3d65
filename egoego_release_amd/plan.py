"""The "plan" a HIP context is packed from: which operand precision runs for a checkpoint — decided by MEASUREMENT, remembered on
disk, and agreed on by all ranks of a process group.

    resolve(model, job)         the ladder of model.hip_precision = "auto" (and the check of an explicit int8 precision):
                                  9 as is -> 9 prepared -> 9 prepared + fc24 -> 8 as is -> 8 prepared -> 8 prepared + ffn16 -> 3 (RuntimeWarning)
                                stage 1 per candidate (precision.PrecisionProbe.error: the end of a chain + two forwards against
                                split-bf16, PROBE_LIMIT), stage 2 for the candidate that passed (the WHOLE num_timesteps chain from
                                noise on CHAIN_WINDOWS windows against split-bf16: CHAIN_LIMIT, and AMPLIFICATION_LIMIT for what the chain makes
                                of one forward's error — a chain that amplifies operand rounding runs split-bf16 whatever the packing).
    cache                       the verdict (and the prepared weights) keyed by the weights' checksum, the module's shape and knobs,
                                the limits and the library's own hash, under $EGOEGO_HIP_CACHE (default ~/.cache/egoego_hip):
                                a second process packs the same checkpoint without measuring again.
    small jobs                  a chain-level call that is SHORTER than the measurement (B x steps below SMALL_JOB_WINDOW_STEPS, no
                                verdict cached) runs split-bf16 — always inside the bar, no probe, ~85 % more time per step on a job
                                of a fraction of a second (the reference's own use: run_egoego.py:146, sample_bs = 1, two windows).
                                The steps run that way are added up; beyond PROBE_AFTER_STEPS the probe runs after all.
    sync(model, ...)            under torch.distributed: ONE plan for all ranks.  Every rank reports whether its packed copy is
                                stale (one all_reduce, always), and if any is, rank 0 resolves and broadcasts the plan — precision,
                                form, flags AND the prepared tensors, so that every rank packs the same bits.

Limits (measured: tools/chain_tail_b256.py, profiles/r05_chain_tail_b256.txt).
"""
import hashlib
import json
import os
import tempfile
import warnings

import torch

from . import _lib
from .precision import PrecisionProbe, prepare_int8_state, _engine_cfg

PROBE_LIMIT = 5e-4       # stage 1 (cheap, every candidate): largest difference from split-bf16 on the end of a chain + two forwards:
                         # half the 1e-3 bar — a short chain on other data ran 1.5-1.7x its probe figure (trained-like checkpoint)
PROBE_TAIL = 50          # ancestral steps of stage 1's end-of-chain run
CHAIN_WINDOWS = 128      # stage 2: windows of the whole-chain probe, conditions shaped like the reference's use (precision.PrecisionProbe).  Round 4
                         # used 4 windows, round 5 first 32: two draws of 32 windows read the SAME packing 4.95e-4 and 7.06e-4, and one draw
                         # accepted a checkpoint whose real batch holds a window 9.9e-3 away (profiles/r05_chain_tail_probe128_*.txt)
CHAIN_WINDOWS_CALLER = 256  # ... and when the chain-level call hands its own conditions over (round 6), ALL of its windows up to this many
CHAIN_LIMIT = 6.0e-4     # stage 2: the worst of CHAIN_WINDOWS whole chains against split-bf16.  DERIVED (round 5, profiles/r05_chain_tail_*.txt): the bar is
                         # 1e-3 against the fp32 reference; split-bf16 itself ends whole chains <= 1.24e-4 from the fp32 oracle (16 windows), which
                         # leaves 8.5e-4 against split-bf16 for the worst window of a B = 256 batch; that worst window sat at <= 1.40x the 128-window
                         # probe's figure for the same packing wherever no amplifying window was involved (1.00-1.21x on the initialisation):
                         # 8.5e-4 / 1.40 = 6.07e-4, rounded down
AMPLIFICATION_LIMIT = 3.0  # stage 2: the whole chain's worst window over ONE forward's error of the same packing.  The reference's initialisation: ~1.0 (a
                         # chain ends where its last forward ends).  Trained-like checkpoints: 5-10 — their chains ACCUMULATE operand rounding,
                         # heavy-tailed over windows: in 2 of 6 (checkpoint, window length) pairs a batch or probe of 128-256 windows held a window
                         # whose last 50 steps multiply any difference along one direction by ~11 and end 1e-2 away in EVERY int8 form
                         # (outlier_window.py (round-4/5 experiment, removed; results: HISTORY.md)), while the other windows sat at 5-8e-4 — no sample of windows bounds that tail.  So a
                         # 16-bit fixed-point form is accepted only for a checkpoint whose chain does not amplify it; otherwise "auto" is split-bf16
GAIN_LIMIT = 1.0         # stage 2 (round 6): the chain's own response to a deliberate perturbation, in split-bf16 alone (precision.PrecisionProbe.chain_gain:
                         # 1e-4 x N(0, 1) added to x with 40 % of the chain to go; per window |difference of the final poses|max / 1e-4).  The worst window
                         # of the chain batch must not expand it at all ...
GAIN_TAIL_LIMIT = 2.5    # ... AND must answer with less than this multiple of the median window: the response must not be heavy-tailed over windows.
                         # Measured along training (profiles/r06_contraction_vs_training.txt, r06_amplification_vs_training.txt, r06_gate_on_caller_conditions.txt;
                         # 128-256 windows): the initialisation max 0.19-0.20 / median 0.14 (x1.4), 10 and 30 Adam steps 0.10 / 0.08 and 0.09 / 0.07 (x1.3), 70 steps
                         # 0.17 / 0.08 (x2.1) — every int8 form inside the bar on all 256 windows there; 50 steps 0.55-1.55 / 0.08 (x7-19), 100 steps 0.33-1.14 /
                         # 0.08-0.09 (x4-13, three seeds), 300 steps 1.1-27 / 0.2-0.35, 1000-3000 steps 1.5-3.5 / 0.6-0.7 — and there "9 as is" held windows 2.3e-3
                         # (50 steps: ONE of 256, where the chain / forward ratio on self-generated conditions read 1.1) to 7e-3 away.  An error-based figure sees
                         # only the windows it samples; the tail of this one moves from x1.4 to x4-19 before the bulk of the windows moves at all.  (A uniformly
                         # less contractive chain — the initialisation with six LayerNorm gains doubled: 0.62 / 0.37 — is left to the error-based limits.)
SMALL_JOB_WINDOW_STEPS = 16 * 1000       # "auto", chain-level calls: below this many window-steps the job is shorter than the probe
PROBE_AFTER_STEPS = 8 * 1000             # ... until one module has run this many STEPS unprobed: small jobs are launch-bound (0.43 ms per step in split-bf16
                                         # against 0.23 in precision 9 whatever the batch, round 5), so each unprobed step loses ~0.2 ms and 8 chains of 1000 steps
                                         # lose what the measurement costs (2-4 s)


def form_name(prepared, flags):
    return ("prepared" if prepared else "as is") + (" + fc24" if flags & _lib.FLAG_FC24 else "") + (" + ffn16" if flags & _lib.FLAG_FFN16 else "")


def plain_plan(precision, source, probe=None):
    return {"precision": precision, "sd": None, "row_shift": None, "prepared": False, "flags": 0, "form": "as is", "source": source,
            "probe": probe, "envelope": None, "warn": None}


def ladder(model):
    """[(precision, prepared, flags)] in the order tried."""
    if model.hip_plan_override is not None:
        p, prepared, flags = model.hip_plan_override
        return [(int(p), bool(prepared), int(flags))]
    want = model.hip_precision
    prep = model.hip_int8_prep
    if prep not in ("auto", "always", "never"):
        raise ValueError(f"unknown hip_int8_prep {prep!r}")
    out = []
    for p in ((_lib.PREC_I8X3_FC, _lib.PREC_I8X3) if want == "auto" else (want,)):
        if prep in ("auto", "never"):
            out.append((p, False, 0))
        if prep in ("auto", "always"):
            out.append((p, True, 0))
        if want == "auto" and p == _lib.PREC_I8X3_FC and prep == "auto" and model.hip_fc24:
            out.append((p, True, _lib.FLAG_FC24))
        if want == "auto" and p == _lib.PREC_I8X3 and prep == "auto" and model.hip_ffn16:
            out.append((p, True, _lib.FLAG_FFN16))
    return out


def masked_state(plan, sd):
    """The state dict of padding-mask calls when the plan's own stores mean-shifted LayerNorm rows (a mask zeroes rows AFTER the
    shift): the plan's rounded weights with the module's own biases and LayerNorm shifts (built on the first masked call)."""
    return {k: (plan["sd"][k] if k.endswith(".weight") else v) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------------------------ the ladder
def run_ladder(model, conditions=None):
    """conditions: [n, T, D] x_cond rows of the chain-level call the context is packed for (stage 2 then runs on them) or None."""
    want = model.hip_precision
    explicit = want != "auto" or model.hip_plan_override is not None
    full_chain = model.hip_probe_full_chain and (want == "auto" or model.hip_plan_override is not None)
    probe = PrecisionProbe(model, tail=PROBE_TAIL, chain_windows=CHAIN_WINDOWS, conditions=conditions, caller_windows_max=CHAIN_WINDOWS_CALLER)
    errors, calib, pick, best, rounded, amplifies, unstable = {}, None, None, None, {}, None, None
    try:
        sd = probe.sd
        for prec, prepared, flags in ladder(model):
            fname = form_name(prepared, flags)
            if prepared:
                calib = probe.calibration() if calib is None else calib
                # (`rounded`: the compensated rounding of a weight depends on the weight and its calibration rows only — shared by every form)
                sd_s, row_shift = prepare_int8_state(sd, calib, prec, shift=True, fc24=bool(flags & _lib.FLAG_FC24), ffn16=bool(flags & _lib.FLAG_FFN16),
                                                       cache=rounded)
            else:
                sd_s, row_shift = sd, None
            err, row_max = probe.error(sd_s, prec, row_shift, flags)
            errors[f"{prec} {fname}"] = err
            cand = {"precision": prec, "sd": sd_s if prepared else None, "row_shift": row_shift, "prepared": prepared, "flags": flags,
                    "form": fname, "envelope": row_max}
            if best is None or err < best[0]:
                best = (err, cand)
            if err <= PROBE_LIMIT or model.hip_plan_override is not None:  # (an override is measured in full, whatever stage 1 says)
                if full_chain:
                    fwd = max(probe.last_forward_error, 1e-7)
                    cerr, per_window = probe.chain_error(sd_s, prec, row_shift, flags)
                    errors[f"{prec} {fname}, full chain"] = cerr
                    errors[f"{prec} {fname}, amplification"] = cerr / fwd
                    cand["chain_per_window"] = per_window
                    gain = probe.chain_gain()
                    if gain is not None:
                        errors["chain gain, max"], errors["chain gain, median"] = gain
                    if not explicit:
                        if gain is not None and (gain[0] > GAIN_LIMIT or gain[0] > GAIN_TAIL_LIMIT * gain[1]):
                            unstable = gain  # a property of the checkpoint's chain (measured in split-bf16 alone): no int8 form is tried further
                            break
                        if cerr / fwd > AMPLIFICATION_LIMIT:
                            amplifies = (prec, fname, cerr / fwd)  # a property of the checkpoint's chain, not of this packing: no int8 form is tried further
                            break
                        if cerr > CHAIN_LIMIT:
                            continue
                pick = cand
                break
    finally:
        probe.close()
    shown = ", ".join(f"precision {k}: {e:.1e}" for k, e in errors.items())
    warn = None
    if pick is None and not explicit and unstable is not None:
        plan = plain_plan(_lib.PREC_BF16X3, "probe")
        warn = (f"hip_precision='auto': this checkpoint's sampling chain does not contract a perturbation evenly over windows (1e-4 x N(0,1) added with "
                f"{probe.t_gain} steps to go moves the final pose by up to {unstable[0]:.2f}x its size, median window {unstable[1]:.2f}x; limits {GAIN_LIMIT} and "
                f"{GAIN_TAIL_LIMIT}x the median, on {probe.conditions} conditions) — such a chain holds windows that multiply 16-bit fixed-point rounding beyond "
                f"the bar (DESIGN.md 3c), so split-bf16 (3) runs, ~85 % more time per step; measured: {shown}.  An explicit hip_precision = 8 / 9 overrides this")
    elif pick is None and not explicit and amplifies is not None:
        plan = plain_plan(_lib.PREC_BF16X3, "probe")
        warn = (f"hip_precision='auto': this checkpoint's sampling chain amplifies operand rounding {amplifies[2]:.1f}x (precision {amplifies[0]} {amplifies[1]}: the worst of "
                f"{CHAIN_WINDOWS} whole chains over one forward's error; limit {AMPLIFICATION_LIMIT:.0f}x) — on such a chain single windows of a large batch end "
                f"10x further from the reference than the rest in every 16-bit fixed-point form (DESIGN.md 3c), so split-bf16 (3) runs, ~85 % more time per "
                f"step; measured: {shown}.  An explicit hip_precision = 8 / 9 overrides this")
    elif pick is None and not explicit:
        plan = plain_plan(_lib.PREC_BF16X3, "probe")
        warn = (f"hip_precision='auto': the int8-slice precisions differ from split-bf16 by more than {PROBE_LIMIT:.0e} (end of a chain, "
                f"forwards) / {CHAIN_LIMIT:.1e} (whole chain) on the probe batch for this checkpoint ({shown}); falling back to split-bf16 "
                f"(3), ~85 % more time per step (tools/precision_compare.py measures each precision on it)")
    elif pick is None:
        plan = dict(best[1], source="probe")  # the explicit precision is kept, in the packing that measured best
        warn = (f"hip_precision={want} differs from split-bf16 by more than {PROBE_LIMIT:.0e} on the probe batch for this checkpoint "
                f"({shown}; the limit is half the 1e-3 bar): it may leave the bar; set model.hip_precision = 'auto' or {_lib.PREC_BF16X3}")
    else:
        plan = dict(pick, source="probe")
    chosen = pick if pick is not None else (best[1] if explicit else None)
    plan["probe"] = {"errors": errors, "limit": PROBE_LIMIT, "chain_limit": CHAIN_LIMIT, "chain_windows": CHAIN_WINDOWS, "amplification_limit": AMPLIFICATION_LIMIT,
                     "gain_limits": (GAIN_LIMIT, GAIN_TAIL_LIMIT), "conditions": probe.conditions,
                     "row_max": chosen["envelope"] if chosen else None, "prepared": bool(chosen and chosen["prepared"]),
                     "form": chosen["form"] if chosen else None,
                     "chain_per_window": chosen.get("chain_per_window") if chosen else None}
    plan.pop("chain_per_window", None)
    plan["warn"] = warn
    return plan


# ------------------------------------------------------------------------------------------------------------------ the cache
_LIB_HASH = None


def _lib_hash():
    global _LIB_HASH
    if _LIB_HASH is None:
        h = hashlib.sha256()
        with open(_lib.LIB_PATH, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        _LIB_HASH = h.hexdigest()[:16]
    return _LIB_HASH


def cache_dir():
    d = os.environ.get("EGOEGO_HIP_CACHE")
    if d is not None and d.strip().lower() in ("", "0", "off", "none"):
        return None
    return d or os.path.join(os.path.expanduser("~"), ".cache", "egoego_hip")


def cache_key(model, fingerprint):
    """Everything a verdict depends on: the weights (checksum), the module's shape, the knobs that shape the ladder, the limits,
    the device kind and the library itself."""
    what = [_lib.ABI_VERSION, _lib_hash(), [repr(v) for v in fingerprint], sorted(_engine_cfg(model).items()), str(model.hip_precision),
            model.hip_int8_prep, bool(model.hip_fc24), bool(model.hip_ffn16), bool(model.hip_probe_full_chain),
            list(model.hip_plan_override) if model.hip_plan_override is not None else None,
            PROBE_LIMIT, PROBE_TAIL, CHAIN_WINDOWS, CHAIN_WINDOWS_CALLER, CHAIN_LIMIT, AMPLIFICATION_LIMIT, GAIN_LIMIT, GAIN_TAIL_LIMIT,
            torch.cuda.get_device_name(model.betas.device) if model.betas.device.type == "cuda" else str(model.betas.device)]
    return hashlib.sha256(json.dumps(what, sort_keys=True, default=str).encode()).hexdigest()[:32]


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def cache_load(key):
    d = cache_dir()
    if d is None:
        return None
    path = os.path.join(d, f"plan_{key}.pt")
    if not os.path.exists(path):
        return None
    try:
        # weights_only: a plan holds dicts, tuples, lists, strings, numbers and tensors — nothing that needs the full unpickler
        # (a writable cache directory must not mean code execution at pack time)
        plan = torch.load(path, map_location="cpu", weights_only=True)
        assert isinstance(plan, dict) and "precision" in plan
        os.utime(path)  # (recently used: cache_store evicts the oldest)
    except Exception:
        return None  # unreadable / half-written by another process / written by something else: measure again
    return plan


CACHE_MAX_FILES = 16  # a plan of a prepared form carries a full state dict (~45 MB): e.g. periodic EMA evaluations during training would pile them up


def cache_drop(key):
    """Forget a verdict (the runtime guard stepped `auto` down on live tensors: the next process must measure again)."""
    d = cache_dir()
    if d is None or key is None:
        return
    try:
        os.remove(os.path.join(d, f"plan_{key}.pt"))
    except OSError:
        pass


def cache_store(key, plan):
    d = cache_dir()
    if d is None:
        return
    try:
        os.makedirs(d, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
        os.close(fd)
        torch.save(_to_cpu(plan), tmp)
        os.replace(tmp, os.path.join(d, f"plan_{key}.pt"))  # atomic: a reader sees the old file or the whole new one
        old = sorted((os.path.join(d, f) for f in os.listdir(d) if f.startswith("plan_") and f.endswith(".pt")), key=os.path.getmtime)
        for path in old[:-CACHE_MAX_FILES]:
            os.remove(path)
    except OSError:
        pass  # a read-only home: the verdict is simply not remembered


# ------------------------------------------------------------------------------------------------------------------ resolve
def is_small_job(model, job):
    if job is None or model.hip_precision != "auto" or model.hip_plan_override is not None:
        return False
    b, _, steps = job
    return b * steps < SMALL_JOB_WINDOW_STEPS and model._slot.unprobed_work < PROBE_AFTER_STEPS


def measured_on_caller(plan):
    return bool(plan and plan.get("probe") and str(plan["probe"].get("conditions", "")).startswith("caller"))


def wants_caller_conditions(model, plan, conditions):
    """An int8 form that `auto` accepted on the probe's SELF-GENERATED conditions is measured once more on the caller's own, the first time a
    chain-level call hands some over (model.hip_engine(conditions=...)): what a chain does to rounding depends on what it is conditioned on."""
    if conditions is None or plan is None or model.hip_precision != "auto" or model.hip_plan_override is not None:
        return False
    if plan["precision"] not in (_lib.PREC_I8X3, _lib.PREC_I8X3_FC) or plan["source"] not in ("probe", "cache") and not str(plan["source"]).startswith("group rank 0"):
        return False
    ok_shape = conditions.dim() == 3 and conditions.shape[0] >= 1 and int(conditions.shape[1]) == int(model.seq_len)
    return ok_shape and model.hip_probe_full_chain and not measured_on_caller(plan)


def resolve(model, job=None, fingerprint=None, conditions=None, remeasure=False):
    """-> plan dict {"precision", "sd" (None = the module's own state dict), "row_shift", "prepared", "flags", "form", "source",
    "probe", "envelope", "warn"}.  `job` = (windows, frames, steps) of the chain-level call that needs the context, or None.
    conditions: that call's x_cond rows (stage 2 runs on them); remeasure: skip the cache READ (the verdict is written)."""
    want = model.hip_precision
    if model.hip_plan_override is None and want != "auto" and want not in (_lib.PREC_I8X3, _lib.PREC_I8X3_FC):
        return plain_plan(want, "explicit")
    if want == "auto" and model._slot.demoted:
        return plain_plan(_lib.PREC_BF16X3, "demoted by the runtime guard")
    if not model.hip_probe_at_pack:
        return plain_plan(_lib.PREC_I8X3_FC if want == "auto" else want, "no probe")
    key = None
    if model.hip_plan_cache and cache_dir() is not None:
        key = cache_key(model, fingerprint if fingerprint is not None else model._weights_fingerprint())
        hit = None if remeasure else cache_load(key)
        if hit is not None:
            return dict(hit, source="cache", cache_key=key)
    if is_small_job(model, job):
        return plain_plan(_lib.PREC_BF16X3, "small job",
                          {"skipped": f"a job of {job[0]} windows x {job[2]} steps is shorter than the precision probe: split-bf16 (no verdict "
                                      f"cached for these weights; jobs of >= {SMALL_JOB_WINDOW_STEPS} window-steps, or model.hip_engine(), measure)"})
    plan = run_ladder(model, conditions)
    if key is not None:
        cache_store(key, plan)
        plan["cache_key"] = key
    return plan


def adopt(model, plan):
    """Make `plan` the module's: report fields, the runtime guard's envelope, the warning (re-issued for a cached verdict too)."""
    model.hip_precision_used = plan["precision"]
    model.hip_precision_probe = None if plan["probe"] is None else dict(plan["probe"], source=plan["source"])
    model._slot.envelope = plan["envelope"]
    if plan.get("warn"):
        warnings.warn(plan["warn"], RuntimeWarning, stacklevel=5)


# ------------------------------------------------------------------------------------------------------------------ one plan for all ranks
def _group_device(group):
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def group_max(values, group):
    """all_reduce(MAX) of a short list of floats over `group` (float64: checksums survive)."""
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_group_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.tolist()


def group_broadcast(obj, group, src=0):
    """Broadcast a picklable object (tensors on the CPU) from group rank `src`."""
    import torch.distributed as dist
    box = [_to_cpu(obj) if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, src) if group is not None else src, group=group,
                               device=_group_device(group))
    return box[0]
