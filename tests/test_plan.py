"""plan.py on the CPU: the ladder `auto` walks, the small-job rule, the on-disk verdict cache, and the two collectives that make all
ranks of a process group sample from ONE plan (gloo, world_size 2) — plus the 8-rank form of the path's one all_gather.
(What the ladder MEASURES needs the GPU: tests/test_gpu_trained_like.py, tests/test_gpu_parity.py.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egoego_release_amd import _lib, plan
from egoego_release_amd import dist as D
from egoego_release_amd import ModelConfig
from egoego_release_amd.model import CondGaussianDiffusion


def _model():
    return CondGaussianDiffusion(**ModelConfig(max_timesteps=31).ctor_kwargs())


def test_ladder_order_and_knobs():
    m = _model()
    F, N = _lib.FLAG_FC24, _lib.FLAG_FFN16
    assert plan.ladder(m) == [(9, False, 0), (9, True, 0), (9, True, F), (8, False, 0), (8, True, 0), (8, True, N)]
    m.hip_fc24 = m.hip_ffn16 = False
    assert plan.ladder(m) == [(9, False, 0), (9, True, 0), (8, False, 0), (8, True, 0)]
    m.hip_int8_prep = "never"
    assert plan.ladder(m) == [(9, False, 0), (8, False, 0)]
    m.hip_int8_prep, m.hip_precision = "always", 8
    assert plan.ladder(m) == [(8, True, 0)]
    m.hip_plan_override = (9, True, F)
    assert plan.ladder(m) == [(9, True, F)]
    assert plan.form_name(True, F) == "prepared + fc24" and plan.form_name(False, 0) == "as is" and plan.form_name(True, N) == "prepared + ffn16"


def test_small_job_rule():
    m = _model()
    assert plan.is_small_job(m, (2, 120, 1000))            # the reference's own use: two windows of one clip
    assert not plan.is_small_job(m, (256, 120, 1000)) and not plan.is_small_job(m, (64, 120, 1000)) and not plan.is_small_job(m, None)
    assert plan.is_small_job(m, (256, 120, 20))            # a 20-step slice of a big batch is short too
    m._slot.unprobed_work = plan.PROBE_AFTER_STEPS  # ... until enough of them have run unprobed
    assert not plan.is_small_job(m, (2, 120, 1000))
    m._slot.unprobed_work = 0
    m.hip_precision = 9                                    # an explicit precision is never replaced
    assert not plan.is_small_job(m, (2, 120, 1000))
    # resolve() without a probe never touches the GPU: explicit split-bf16, probe off, demoted
    m.hip_precision = _lib.PREC_BF16X3
    assert plan.resolve(m, None)["precision"] == 3 and plan.resolve(m, None)["source"] == "explicit"
    m.hip_precision, m.hip_probe_at_pack = "auto", False
    assert plan.resolve(m, (2, 120, 1000))["precision"] == 9 and plan.resolve(m, None)["source"] == "no probe"
    m.hip_probe_at_pack = True
    m._slot.demoted = True
    assert plan.resolve(m, None)["precision"] == 3


def test_verdict_cache_round_trip(tmp_path, monkeypatch):
    monkeypatch.setenv("EGOEGO_HIP_CACHE", str(tmp_path / "c"))
    m = _model()
    fp = (1.25, 3.5)
    k1 = plan.cache_key(m, fp)
    assert k1 == plan.cache_key(m, fp) and k1 != plan.cache_key(m, (1.25, 3.5000001))
    m.hip_fc24 = False
    assert plan.cache_key(m, fp) != k1  # a knob that shapes the ladder is part of the key
    m.hip_fc24 = True
    assert plan.cache_load(k1) is None
    p = dict(plan.plain_plan(9, "probe", {"errors": {"9 as is": 1e-4}}), sd={"w": torch.arange(6.0).reshape(2, 3)},
             row_shift={"embed": torch.ones(4), (0, "attn_ln"): torch.zeros(4)}, prepared=True, form="prepared", envelope=[4.0] * 8)
    plan.cache_store(k1, p)
    got = plan.cache_load(k1)
    assert got["precision"] == 9 and got["form"] == "prepared" and torch.equal(got["sd"]["w"], p["sd"]["w"])
    assert torch.equal(got["row_shift"][(0, "attn_ln")], torch.zeros(4)) and got["envelope"] == [4.0] * 8
    (tmp_path / "c" / f"plan_{k1}.pt").write_bytes(b"not a checkpoint")  # a damaged file is measured again, not trusted
    assert plan.cache_load(k1) is None
    monkeypatch.setenv("EGOEGO_HIP_CACHE", "off")
    assert plan.cache_dir() is None and plan.cache_load(k1) is None
    plan.cache_store(k1, p)  # (a no-op)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = D._group_of(None)
    # the stale flag and the checksums of hip_engine(group=...): one all_reduce(MAX)
    got = plan.group_max([1.0 if rank == 1 else 0.0, 7.125, -7.125], grp)
    # the plan itself: rank 0's object (tensors included) reaches every rank
    mine = dict(plan.plain_plan(9 if rank == 0 else 3, "probe"), sd={"w": torch.full((3,), float(rank))}) if rank == 0 else None
    p = plan.group_broadcast(mine, grp)
    ok, err = plan.group_broadcast((False, 7e-4) if rank == 0 else None, grp)
    torch.save({"max": got, "prec": p["precision"], "w": p["sd"]["w"], "ok": ok, "err": err}, out + str(rank))
    dist.destroy_process_group()


def test_group_helpers_two_gloo_ranks(tmp_path):
    out = str(tmp_path / "r")
    mp.spawn(_sync_worker, args=(2, _port(), out), nprocs=2, join=True)
    for r in range(2):
        d = torch.load(out + str(r))
        assert d["max"] == [1.0, 7.125, -7.125]                 # rank 1's stale copy is everyone's business
        assert d["prec"] == 9 and torch.equal(d["w"], torch.zeros(3))  # rank 0's plan, rank 0's tensors
        assert d["ok"] is False and d["err"] == 7e-4


def _fake_sampler(xs, cm, noise, window_offset):
    idx = torch.arange(window_offset, window_offset + xs.shape[0], dtype=torch.float32)[:, None, None]
    return noise["x_T"] * 0.5 + xs * (1 - cm) + idx


def _eight_worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(B, 5, 9, generator=g)
    cm = (torch.rand(B, 5, 9, generator=g) > 0.5).float()
    noise = {"x_T": torch.randn(B, 5, 9, generator=g), "cond": torch.randn(B, 5, 9, generator=g)}
    res = D.sample_sharded(_fake_sampler, xs, cm, noise)
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_eight_gloo_ranks_even_and_ragged(tmp_path):
    """`bench.py --gpus 8`'s data path with a stand-in sampler: B = 256 windows over 8 ranks (32 each: the gathered buffer IS the
    result), B = 250 (ragged: 32 x 2 + 31 x 6) and B = 5 (three ranks sample nothing) — one all_gather_into_tensor each."""
    for B in (256, 250, 5):
        out = str(tmp_path / f"e{B}.pt")
        mp.spawn(_eight_worker, args=(8, _port(), B, out), nprocs=8, join=True)
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(B, 5, 9, generator=g)
        cm = (torch.rand(B, 5, 9, generator=g) > 0.5).float()
        noise = {"x_T": torch.randn(B, 5, 9, generator=g), "cond": torch.randn(B, 5, 9, generator=g)}
        assert torch.equal(torch.load(out), _fake_sampler(xs, cm, noise, 0))
    assert [D.shard_bounds(250, r, 8) for r in range(8)][:3] == [(0, 32), (32, 64), (64, 95)]


# ------------------------------------------------------------------------------------------ the ladder's decisions, with a scripted probe
class _ScriptedProbe:
    """Stand-in for precision.PrecisionProbe: errors per (precision, form) come from a table (no GPU)."""
    table = {}
    gain = (0.2, 0.14)  # (max, median) of chain_gain: the initialisation's figures unless a case sets others

    def __init__(self, model, tail=0, chain_windows=0, conditions=None, caller_windows_max=None):
        self.sd = {"w": torch.zeros(2)}
        self.last_forward_error = 0.0
        self.calls = []
        self.t_gain = 400
        self.conditions = "self-generated" if conditions is None else f"caller ({conditions.shape[0]} windows)"

    def chain_gain(self):
        return self.gain

    def calibration(self):
        return {"rows": {}, "n_layers": 0}

    def error(self, sd, prec, row_shift=None, flags=0):
        stage1, fwd, _ = self.table[(prec, sd.get("form", "as is"))]
        self.last_forward_error = fwd
        return stage1, [4.0] * 8

    def chain_error(self, sd, prec, row_shift=None, flags=0):
        chain = self.table[(prec, sd.get("form", "as is"))][2]
        return chain, [chain] * 4

    def close(self):
        pass


def _scripted(monkeypatch, table, gain=(0.2, 0.14)):
    _ScriptedProbe.table = table
    _ScriptedProbe.gain = gain
    monkeypatch.setattr(plan, "PrecisionProbe", _ScriptedProbe)

    def fake_prepare(sd, calib, prec, shift=True, fc24=False, ffn16=False, cache=None, **kw):
        return {"w": torch.ones(2), "form": plan.form_name(True, (_lib.FLAG_FC24 if fc24 else 0) | (_lib.FLAG_FFN16 if ffn16 else 0))}, {"embed": torch.zeros(1)}
    monkeypatch.setattr(plan, "prepare_int8_state", fake_prepare)


def test_ladder_decisions_with_a_scripted_probe(monkeypatch):
    """run_ladder on tables of (stage-1 error, one forward's error, worst whole chain) per packing: which form is picked, what stops the
    ladder, which warning is issued — the decisions of DESIGN.md 3c without a GPU."""
    L, C, A = plan.PROBE_LIMIT, plan.CHAIN_LIMIT, plan.AMPLIFICATION_LIMIT
    m = _model()
    # 1. the initialisation's pattern: "9 as is" inside every limit, the chain ends where its last forward ends
    _scripted(monkeypatch, {(9, "as is"): (3.3e-4, 3.2e-4, 3.5e-4)})
    p = plan.run_ladder(m)
    assert (p["precision"], p["form"], p["warn"]) == (9, "as is", None) and p["sd"] is None
    assert p["probe"]["errors"]["9 as is, amplification"] == 3.5e-4 / 3.2e-4 and p["envelope"] == [4.0] * 8
    # 2. a trained-like pattern: as is fails stage 1, prepared passes it, but the chain makes 7.8x of one forward's error: split-bf16, and
    #    no further int8 form is even tried
    _scripted(monkeypatch, {(9, "as is"): (7.4e-4, 6e-4, 0), (9, "prepared"): (1.5e-4, 1.24e-4, 9.7e-4)})
    p = plan.run_ladder(m)
    assert p["precision"] == 3 and "amplifies operand rounding 7.8x" in p["warn"] and set(p["probe"]["errors"]) == {
        "9 as is", "9 prepared", "9 prepared, full chain", "9 prepared, amplification", "chain gain, max", "chain gain, median"}
    # 2b. (round 6) the pattern of a checkpoint 50 Adam steps from the initialisation: every error figure is the initialisation's — stage 1, whole
    #     chain, chain / forward 1.1 — but the chain's response to a perturbation is heavy-tailed over windows (max 1.5, median 0.08): split-bf16.
    #     And a uniform response that EXPANDS the perturbation (max 1.2 at a median of 0.9) is refused as well
    _scripted(monkeypatch, {(9, "as is"): (1.8e-4, 1.6e-4, 1.8e-4)}, gain=(1.55, 0.081))
    p = plan.run_ladder(m, conditions=torch.zeros(256, 30, 198))
    assert p["precision"] == 3 and "does not contract a perturbation evenly over windows" in p["warn"] and "caller (256 windows)" in p["warn"]
    assert p["probe"]["errors"]["chain gain, max"] == 1.55 and p["probe"]["conditions"] == "caller (256 windows)"
    _scripted(monkeypatch, {(9, "as is"): (1.8e-4, 1.6e-4, 1.8e-4)}, gain=(1.2, 0.9))
    assert plan.run_ladder(m)["precision"] == 3
    _scripted(monkeypatch, {(9, "as is"): (1.8e-4, 1.6e-4, 1.8e-4)}, gain=(0.24, 0.077))  # (x3.1: refused; the 70-step checkpoint's own 128 windows read 0.17 / 0.08 = x2.1 and pass)
    assert plan.run_ladder(m)["precision"] == 3
    _scripted(monkeypatch, {(9, "as is"): (1.8e-4, 1.6e-4, 1.8e-4)}, gain=(0.62, 0.37))  # (uniformly less contractive: left to the error limits)
    assert plan.run_ladder(m)["precision"] == 9
    _scripted(monkeypatch, {(9, "as is"): (1.8e-4, 1.6e-4, 1.8e-4)}, gain=(0.108, 0.082))  # (10 steps: x1.3)
    assert plan.run_ladder(m)["precision"] == 9
    # 3. the same forward errors on a chain that does NOT amplify but whose worst window is over the limit: the ladder walks on, and
    #    precision 8 with the FFN on split-bf16 is the first form inside everything
    tab = {(9, "as is"): (7e-4, 6e-4, 0), (9, "prepared"): (4e-4, 3.9e-4, C * 1.1), (9, "prepared + fc24"): (3.9e-4, 3.8e-4, C * 1.05),
           (8, "as is"): (6e-4, 5e-4, 0), (8, "prepared"): (3.5e-4, 3.4e-4, C * 1.01), (8, "prepared + ffn16"): (3e-4, 2.9e-4, C * 0.9)}
    _scripted(monkeypatch, tab)
    p = plan.run_ladder(m)
    assert (p["precision"], p["form"], p["flags"], p["warn"]) == (8, "prepared + ffn16", _lib.FLAG_FFN16, None) and p["prepared"] and p["sd"] is not None
    # 4. nothing passes stage 1: split-bf16 with the other warning, every form measured once
    _scripted(monkeypatch, {k: (2 * L, 2 * L, 0) for k in tab})
    p = plan.run_ladder(m)
    assert p["precision"] == 3 and "falling back to split-bf16" in p["warn"] and len(p["probe"]["errors"]) == 6
    # 5. an explicit precision is kept — in the packing that measured best — and warned about; the amplification gate does not apply to it
    m.hip_precision = 9
    _scripted(monkeypatch, {(9, "as is"): (9e-4, 8e-4, 0), (9, "prepared"): (7e-4, 6e-4, 0)})
    p = plan.run_ladder(m)
    assert (p["precision"], p["form"]) == (9, "prepared") and "differs from split-bf16" in p["warn"]
    _scripted(monkeypatch, {(9, "as is"): (2e-4, 1e-4, 1e-3)})
    p = plan.run_ladder(m)
    assert (p["precision"], p["form"], p["warn"]) == (9, "as is", None)
    # 6. an override (tools) is measured in full, whatever the figures, and never rejected
    m.hip_precision, m.hip_plan_override = "auto", (8, True, _lib.FLAG_FFN16)
    _scripted(monkeypatch, {(8, "prepared + ffn16"): (9e-4, 1e-4, 2e-3)})
    p = plan.run_ladder(m)
    assert (p["precision"], p["form"]) == (8, "prepared + ffn16") and p["probe"]["errors"]["8 prepared + ffn16, amplification"] == 2e-3 / 1e-4
    assert A == 3.0 and C == 6.0e-4 and plan.CHAIN_WINDOWS == 128  # (the figures DESIGN.md 3c derives)
    assert (plan.GAIN_LIMIT, plan.GAIN_TAIL_LIMIT) == (1.0, 2.5)


def test_an_int8_plan_is_measured_once_more_on_the_callers_conditions():
    """plan.wants_caller_conditions: an int8 form `auto` accepted on the probe's self-generated conditions is re-measured the first time a chain-level
    call hands its own x_cond rows over; a verdict measured on a caller's conditions, split-bf16, an explicit precision or a mismatching window
    length is left alone."""
    m = CondGaussianDiffusion(**ModelConfig(max_timesteps=31).ctor_kwargs())
    cond = torch.zeros(40, 30, 198)
    own = dict(plan.plain_plan(9, "probe"), probe={"conditions": "self-generated"})
    assert plan.wants_caller_conditions(m, own, cond)
    assert plan.wants_caller_conditions(m, dict(own, source="cache"), cond) and plan.wants_caller_conditions(m, dict(own, source="group rank 0 (probe)"), cond)
    assert not plan.wants_caller_conditions(m, dict(own, probe={"conditions": "caller (40 windows)"}), cond)
    assert not plan.wants_caller_conditions(m, dict(own, precision=3), cond) and not plan.wants_caller_conditions(m, own, None)
    assert not plan.wants_caller_conditions(m, own, torch.zeros(40, 12, 198))  # (a trailing window of the harness)
    assert not plan.wants_caller_conditions(m, dict(own, source="no probe"), cond)
    m.hip_precision = 9
    assert not plan.wants_caller_conditions(m, own, cond)


def test_harness_sizes_its_plan_by_the_global_job(tmp_path, monkeypatch):
    """under dist.harness_sharded the small-job rule must see the WHOLE call's (sequence, sample) pairs, not this rank's shard —
    else two ranks with 5 pairs each run split-bf16 unprobed where one rank with 10 measures and picks an int8 form.  Both sides of the
    threshold, empty cache: 10 pairs x 2 windows x 1000 steps is a measured job, a 5-pair call of its own is a small one."""
    import pytest
    from egoego_release_amd import harness
    monkeypatch.setenv("EGOEGO_HIP_CACHE", str(tmp_path))
    m = CondGaussianDiffusion(**ModelConfig(max_timesteps=121).ctor_kwargs())
    seen = {}

    class Stop(Exception):
        pass

    def fake_engine(verify=False, masked=False, job=None, group=None, conditions=None):
        seen["job"] = job
        raise Stop()
    m.hip_engine = fake_engine
    ds = harness.SkeletonStats(-torch.ones(66), torch.ones(66), torch.zeros(66))
    head = torch.zeros(5, 140, 7)
    head[..., 3] = 1.0
    with pytest.raises(Stop):
        harness.full_body_gen_cond_head_pose_sliding_window(m, ds, head, global_pairs=10)
    assert seen["job"] == (10 * 2, 120, 1000) and not plan.is_small_job(m, seen["job"])
    with pytest.raises(Stop):
        harness.full_body_gen_cond_head_pose_sliding_window(m, ds, head)
    assert seen["job"] == (5 * 2, 120, 1000) and plan.is_small_job(m, seen["job"])
    # dist.harness_sharded hands the global count over (one process: its shard is everything; sample_bs multiplies the pairs)
    with pytest.raises(Stop):
        D.harness_sharded(m, ds, head[:2], sample_bs=3)
    assert seen["job"] == (6 * 2, 120, 1000)


def test_cache_is_loaded_without_the_full_unpickler_and_evicted(tmp_path, monkeypatch):
    """plans are read with weights_only=True (a writable cache directory must not mean code execution), a file that needs
    the full unpickler reads as a miss, the oldest files beyond CACHE_MAX_FILES go, and cache_drop forgets a verdict."""
    monkeypatch.setenv("EGOEGO_HIP_CACHE", str(tmp_path))
    p = dict(plan.plain_plan(9, "probe"), sd={"w": torch.ones(2)}, row_shift={"embed": torch.zeros(2), (0, "attn_ln"): torch.ones(2)}, envelope=[1.0, 2.0])
    plan.cache_store("k0", p)
    got = plan.cache_load("k0")
    assert got["precision"] == 9 and torch.equal(got["row_shift"][(0, "attn_ln")], torch.ones(2)) and got["envelope"] == [1.0, 2.0]

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    torch.save({"precision": 9, "x": Evil()}, os.path.join(str(tmp_path), "plan_evil.pt"))
    assert plan.cache_load("evil") is None
    for i in range(plan.CACHE_MAX_FILES + 3):
        plan.cache_store(f"n{i}", p)
    assert len([f for f in os.listdir(str(tmp_path)) if f.startswith("plan_")]) <= plan.CACHE_MAX_FILES
    plan.cache_drop(f"n{plan.CACHE_MAX_FILES + 2}")
    assert plan.cache_load(f"n{plan.CACHE_MAX_FILES + 2}") is None
