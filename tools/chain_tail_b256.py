#!/usr/bin/env python3
"""The WHOLE 1000-step chain at the metric's own size: per-window error of what `auto` picks (and of the other int8 forms) at B = 256.

For a checkpoint and window length: the full Philox chain on B windows of in-distribution data (synthetic.make_motion_windows) in
every requested packing and in split-bf16 (precision 3) with the same seed; per-window max-abs against precision 3 — the maximum, the
99th percentile, the median, the same over the first 64 windows (BASELINE configs[1]) — next to what the pack-time probe measured
for that packing on ITS windows (plan.py stage 2), i.e. the ratio the probe's limit has to leave room for.  `--oracle N`: N of the
windows with the oracle's injected draws against the fp32 CPU oracle as well (~20 s of CPU per window on the GPU box).

    python tools/chain_tail_b256.py [--weights init,seed0,seed1,seed2] [--windows 120,196] [--batch 256] [--forms auto,9p,9pf,8p]
                                    [--oracle 16] [--out profiles/r05_chain_tail_b256.txt]
"""
import argparse
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import ModelConfig, make_weights, head_condition_mask, _lib  # noqa: E402
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402

FORMS = {"auto": None, "9": (9, False, 0), "9p": (9, True, 0), "9pf": (9, True, _lib.FLAG_FC24), "8": (8, False, 0), "8p": (8, True, 0), "8pn": (8, True, _lib.FLAG_FFN16)}
S = 1000


def build(sd, T, form, cache=True):
    cfg = ModelConfig(max_timesteps=T + 1)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_plan_cache = cache
    if form == "3":
        m.hip_precision = _lib.PREC_BF16X3
    elif FORMS[form] is not None:
        m.hip_plan_override = FORMS[form]
    return m.cuda()


def stats(d):
    d = d.double()
    q = torch.quantile(d, torch.tensor([0.5, 0.99], dtype=torch.float64))
    return {"max": float(d.max()), "p99": float(q[1]), "median": float(q[0]), "argmax": int(d.argmax())}


def philox_chain(m, x_T, x_cond, seed):
    eng = m.hip_engine(verify=True)
    x = x_T.clone()
    eng.sample_loop_(x, x_cond, S - 1, S, noise_mode=_lib.NOISE_PHILOX, seed=seed)
    torch.cuda.synchronize()
    return x


def chain_tail(sd, T, B=256, forms=("auto",), data_seed=31337, seed=11, n_oracle=0, cache=True, log=print):
    """-> {form: {"precision", "form", "probe", "vs3": stats over B windows, "vs3_first64", "ratio_to_probe", "per_window": tensor,
    ["vs_oracle": [...], "three_vs_oracle": [...]]}}"""
    data = make_motion_windows(B, T, seed=data_seed)
    mask = head_condition_mask(data.shape)
    g = torch.Generator().manual_seed(data_seed + 1)
    x_T = torch.randn(data.shape, generator=g).cuda()
    x_cond = (data * (1 - mask) + mask * torch.randn(data.shape, generator=g)).cuda()
    m3 = build(sd, T, "3")
    t0 = time.time()
    want = philox_chain(m3, x_T, x_cond, seed)
    log(f"  split-bf16 chain at B={B}: {time.time() - t0:.1f} s")
    out = {}
    models = {"3": m3}
    for form in forms:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            t0 = time.time()
            m = build(sd, T, form, cache)
            m.hip_engine(verify=True, job=(B, T, S), conditions=x_cond)  # (stage 2 of the plan's measurement on this batch's own conditions, like a chain-level call)
            t_pack = time.time() - t0
            t0 = time.time()
            got = philox_chain(m, x_T, x_cond, seed)
            t_chain = time.time() - t0
        models[form] = m
        d = (got - want).abs().amax((1, 2)).cpu()
        pr = m.hip_precision_probe or {}
        name = f"{m.hip_precision_used} {pr.get('form') or 'as is'}"
        pchain = (pr.get("errors") or {}).get(name + ", full chain")
        r = {"precision": m.hip_precision_used, "form": pr.get("form"), "probe": pr.get("errors"), "probe_chain": pchain,
             "probe_per_window": pr.get("chain_per_window"), "source": pr.get("source"),
             "vs3": stats(d), "vs3_first64": stats(d[:64]), "per_window": d, "pack_s": t_pack, "chain_s": t_chain,
             "ratio_to_probe": (float(d.max()) / pchain) if pchain else None, "warnings": [str(w.message)[:160] for w in rec]}
        out[form] = r
        log(f"  {form:5s} -> runs {name:22s} pack {t_pack:5.1f} s chain {t_chain:4.1f} s | B={B} vs split-bf16: max {r['vs3']['max']:.2e} "
            f"p99 {r['vs3']['p99']:.2e} median {r['vs3']['median']:.2e} (first 64: max {r['vs3_first64']['max']:.2e}) | probe's whole chains "
            f"({len(pr.get('chain_per_window') or [])} windows): {pchain if pchain is None else format(pchain, '.2e')} -> ratio "
            f"{r['ratio_to_probe'] if r['ratio_to_probe'] is None else format(r['ratio_to_probe'], '.2f')}")
        pw = pr.get("chain_per_window") or []
        if len(pw) > 32:
            log("        the probe's maximum over its first n windows: " + ", ".join(f"{n}: {max(pw[:n]):.2e} (x{float(d.max()) / max(pw[:n]):.2f})" for n in (32, 64, 128, 256) if n <= len(pw)))
        if rec:
            log("        warnings: " + " | ".join(r["warnings"]))
    if n_oracle:
        from oracle import egoego_oracle as O  # (perf-debug tool: the oracle is the checker here, as in tests/)
        n = n_oracle
        gi = torch.Generator().manual_seed(data_seed + 2)
        nz = {"x_T": torch.randn((n,) + tuple(data.shape[1:]), generator=gi), "cond": torch.randn((n,) + tuple(data.shape[1:]), generator=gi),
              "steps": torch.randn((S, n) + tuple(data.shape[1:]), generator=gi)}
        sched = O.make_schedule(S)
        x = nz["x_T"].clone()
        xc = data[:n] * (1 - mask[:n]) + mask[:n] * nz["cond"]
        t0 = time.time()
        nt0 = torch.get_num_threads()
        torch.set_num_threads(min(nt0, 16))  # a few windows on 128 threads run slower than on 16
        with torch.no_grad():
            for i, tv in enumerate(reversed(range(S))):
                x = O.p_sample(sd, sched, x, torch.full((n,), tv, dtype=torch.long), xc, nz["steps"][i])
        log(f"  fp32 oracle chain on {n} windows: {time.time() - t0:.0f} s of CPU ({torch.get_num_threads()} threads)")
        torch.set_num_threads(nt0)
        for form, m in models.items():
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = m.sample(data[:n].cuda(), mask[:n].cuda(), noise=nz).cpu()
            d = (got - x).abs().amax((1, 2))
            (out[form] if form in out else out.setdefault("3", {}))["vs_oracle"] = [float(v) for v in d]
            log(f"  {form:5s} vs the fp32 ORACLE, {n} windows, the oracle's draws: max {float(d.max()):.2e}  per window "
                + " ".join(f"{float(v):.1e}" for v in d))
    for m in models.values():
        m.invalidate_engine()
    return out


def switch_study(sd, T, B=256, forms=("9",), switch_at=(50, 100, 200, 400), data_seed=31337, seed=11, n_oracle=0, log=print):
    """A precision SCHEDULE along the chain (round 6): the int8 form for t >= t*, split-bf16 for t < t*.  The Philox draws are keyed by
    the timestep, so the two engines share one noise stream: the int8 chain runs once from t = S-1 down, its state is kept at every
    t*, and a split-bf16 engine finishes each of them.  Per-window max-abs of the final poses against the pure split-bf16 chain (t* = 0
    is the pure int8 chain); `n_oracle` windows with injected draws against the fp32 CPU oracle for every t*.
    -> {form: {t*: stats}}, {form: {t*: [per-window vs oracle]}}"""
    data = make_motion_windows(B, T, seed=data_seed)
    mask = head_condition_mask(data.shape)
    g = torch.Generator().manual_seed(data_seed + 1)
    x_T = torch.randn(data.shape, generator=g).cuda()
    x_cond = (data * (1 - mask) + mask * torch.randn(data.shape, generator=g)).cuda()
    m3 = build(sd, T, "3")
    eng3 = m3.hip_engine(verify=True)
    want = philox_chain(m3, x_T, x_cond, seed)
    marks = sorted({int(v) for v in switch_at if 0 < int(v) < S}, reverse=True)
    nz, xo, xco = None, None, None
    if n_oracle:
        from oracle import egoego_oracle as O  # (perf-debug tool: the oracle is the checker here, as in tests/)
        n = n_oracle
        gi = torch.Generator().manual_seed(data_seed + 2)
        nz = {"x_T": torch.randn((n,) + tuple(data.shape[1:]), generator=gi), "cond": torch.randn((n,) + tuple(data.shape[1:]), generator=gi),
              "steps": torch.randn((S, n) + tuple(data.shape[1:]), generator=gi)}
        sched = O.make_schedule(S)
        xo = nz["x_T"].clone()
        xco = data[:n] * (1 - mask[:n]) + mask[:n] * nz["cond"]
        t0 = time.time()
        nt0 = torch.get_num_threads()
        torch.set_num_threads(min(nt0, 16))
        with torch.no_grad():
            for i, tv in enumerate(reversed(range(S))):
                xo = O.p_sample(sd, sched, xo, torch.full((n,), tv, dtype=torch.long), xco, nz["steps"][i])
        torch.set_num_threads(nt0)
        log(f"  fp32 oracle chain on {n} windows: {time.time() - t0:.0f} s of CPU")
        steps_dev = nz["steps"].cuda()
        xco_dev = xco.cuda().contiguous()

    def walk(eng, x, xc, t_from, t_to, injected):
        """steps t = t_from-1 .. t_to, in place"""
        if t_from <= t_to:
            return
        if injected:
            eng.sample_loop_(x, xc, t_from - 1, t_from - t_to, noise=steps_dev[S - t_from:S - t_to].contiguous())
        else:
            eng.sample_loop_(x, xc, t_from - 1, t_from - t_to, noise_mode=_lib.NOISE_PHILOX, seed=seed)

    def schedule_runs(eng, x0, xc, injected):
        """{t*: final x} for t* in marks + [0]"""
        x, cur, kept = x0.clone(), S, {}
        for ts in marks:
            walk(eng, x, xc, cur, ts, injected)
            cur = ts
            kept[ts] = x.clone()
        walk(eng, x, xc, cur, 0, injected)
        out = {0: x}
        for ts, y in kept.items():
            walk(eng3, y, xc, ts, 0, injected)
            out[ts] = y
        torch.cuda.synchronize()
        return out
    res, res_or = {}, {}
    if n_oracle:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y3 = nz["x_T"].cuda().clone()
            walk(eng3, y3, xco_dev, S, 0, True)
            d3 = (y3.cpu() - xo).abs().amax((1, 2))
        log(f"  split-bf16 vs the fp32 ORACLE, {n_oracle} windows: max {float(d3.max()):.2e}")
        res_or["3"] = {0: [float(v) for v in d3]}
    for form in forms:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = build(sd, T, form, cache=False)
            m.hip_probe_full_chain = False  # (the override is packed whatever the probe says; stage 1 only: the preparation + one tail)
            eng = m.hip_engine(verify=True)
        runs = schedule_runs(eng, x_T, x_cond, False)
        res[form] = {}
        for ts in [0] + marks[::-1]:
            st = stats((runs[ts] - want).abs().amax((1, 2)).cpu())
            res[form][ts] = st
            log(f"  {form:4s} int8 for t >= {ts:3d}, split-bf16 below | B={B} vs split-bf16: max {st['max']:.2e} (window {st['argmax']}) p99 {st['p99']:.2e} median {st['median']:.2e}")
        if n_oracle:
            runs = schedule_runs(eng, nz["x_T"].cuda(), xco_dev, True)
            res_or[form] = {}
            for ts in [0] + marks[::-1]:
                d = (runs[ts].cpu() - xo).abs().amax((1, 2))
                res_or[form][ts] = [float(v) for v in d]
                log(f"  {form:4s} int8 for t >= {ts:3d} vs the fp32 ORACLE, {n_oracle} windows: max {float(d.max()):.2e}")
        m.invalidate_engine()
    m3.invalidate_engine()
    return res, res_or


def weights_for(kind, T):
    if kind == "init":
        return make_weights(ModelConfig(max_timesteps=T + 1), 0), {"kind": "initialisation (make_weights seed 0)"}
    from make_trained_like_checkpoint import train_like
    seed = int(kind.replace("seed", ""))
    sd, info = train_like(3000, seed, "cuda", T)
    return {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="init,seed0,seed1,seed2")
    ap.add_argument("--windows", default="120,196")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--forms", default="auto")
    ap.add_argument("--oracle", type=int, default=0, help="windows against the fp32 CPU oracle (for the first --oracle-configs configurations)")
    ap.add_argument("--oracle-configs", default="seed0:120")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--probe-windows", type=int, default=0, help="run the plan's whole-chain probe on this many windows (default: plan.CHAIN_WINDOWS) and also print its "
                                                                 "maximum over the first 32 / 64 / ... of them: what a probe of that size would have said")
    ap.add_argument("--switch-at", default=None, help="t* list: the precision SCHEDULE study (int8 form for t >= t*, split-bf16 below) instead of the tail study")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lines = []

    def log(sx):
        print(sx, flush=True)
        lines.append(sx)
    from egoego_release_amd import plan
    if args.probe_windows:
        plan.CHAIN_WINDOWS = args.probe_windows
    log(f"# tools/chain_tail_b256.py --weights {args.weights} --windows {args.windows} --batch {args.batch} --forms {args.forms} --oracle {args.oracle}")
    log(f"# limits in force: stage 1 {plan.PROBE_LIMIT:.1e}, whole chain {plan.CHAIN_LIMIT:.2e} on {plan.CHAIN_WINDOWS} probe windows; {torch.cuda.get_device_name(0)}")
    ratios, switch_rows = [], []
    for T in [int(v) for v in args.windows.split(",")]:
        for kind in args.weights.split(","):
            sd, info = weights_for(kind, T)
            log(f"== weights {kind}, T={T}: {info}")
            n_or = args.oracle if f"{kind}:{T}" in args.oracle_configs.split(",") else 0
            if args.switch_at:
                marks = [int(v) for v in args.switch_at.split(",")]
                res, _ = switch_study(sd, T, args.batch, args.forms.split(","), marks, n_oracle=n_or, log=log)
                for form, by_t in res.items():
                    switch_rows.append((kind, T, form, by_t))
                continue
            res = chain_tail(sd, T, args.batch, args.forms.split(","), n_oracle=n_or, cache=not args.no_cache, log=log)
            for form, r in res.items():
                if r.get("ratio_to_probe"):
                    ratios.append((kind, T, form, r["precision"], r["form"], r["vs3"]["max"], r["probe_chain"], r["ratio_to_probe"]))
    if switch_rows:
        log(f"== worst window of {args.batch} against split-bf16, per t* (int8 form for t >= t*, split-bf16 for t < t*; t* = 0: the int8 form's own chain)")
        for kind, T, form, by_t in switch_rows:
            log(f"   {kind:6s} T={T:<3d} {form:4s} " + "  ".join(f"t*={ts}: {st['max']:.2e}" for ts, st in by_t.items()))
        for form in args.forms.split(","):
            rows = [by_t for _, _, f, by_t in switch_rows if f == form]
            if rows:
                log(f"   worst over all configurations, {form:4s}: " + "  ".join(f"t*={ts}: {max(r[ts]['max'] for r in rows):.2e}" for ts in rows[0]))
    else:
        log("== max over the batch / the probe's own whole-chain figure, per configuration")
        for row in ratios:
            log("   %-6s T=%-3d %-5s runs %s %-16s  B-max %.2e  probe %.2e  ratio %.2f" % row)
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
