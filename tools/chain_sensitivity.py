#!/usr/bin/env python3
"""How much of a whole chain's "error" is the chain itself?  The 1000-step ancestral chain of a trained denoiser amplifies ANY
perturbation; this tool measures that amplification in split-bf16 alone, at the metric's size:

    chain A: split-bf16 (precision 3), B windows, Philox draws;
    chain B: the same engine, the same draws, x_T perturbed by `eps` * N(0, 1) (default 1e-6: a few ulp of an fp32 value of order 1 —
             what another BLAS, another summation order or another machine does to the reference's own arithmetic in ONE step).

Per window: |A - B|max.  A window where that exceeds the 1e-3 bar is one where the REFERENCE's own result is not reproducible to the bar
between two machines, whatever runs the contractions.  With --forms the int8 packings run on the same batch and their per-window
distance from chain A is printed next to the window's sensitivity (tools/chain_tail_b256.py's outliers can be looked up here).

    python tools/chain_sensitivity.py [--weights seed2] [--window 120] [--batch 256] [--eps 1e-6] [--forms 8p,9pf] [--out file]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import head_condition_mask  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402
from chain_tail_b256 import build, philox_chain, weights_for, stats  # noqa: E402


def sensitivity(sd, T, B=256, eps=1e-6, data_seed=31337, seed=11, forms=(), log=print):
    data = make_motion_windows(B, T, seed=data_seed)
    mask = head_condition_mask(data.shape)
    g = torch.Generator().manual_seed(data_seed + 1)
    x_T = torch.randn(data.shape, generator=g).cuda()
    x_cond = (data * (1 - mask) + mask * torch.randn(data.shape, generator=g)).cuda()
    m3 = build(sd, T, "3")
    a = philox_chain(m3, x_T, x_cond, seed)
    out = {"eps": {}}
    for e in ([eps] if isinstance(eps, float) else eps):
        pert = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(99)).cuda() * e
        b = philox_chain(m3, x_T + pert, x_cond, seed)
        d = (a - b).abs().amax((1, 2)).cpu()
        out["eps"][e] = d
        st = stats(d)
        log(f"  split-bf16 vs split-bf16 with x_T + {e:.0e} N(0,1): per-window |diff|max: max {st['max']:.2e} (window {st['argmax']}) p99 {st['p99']:.2e} "
            f"median {st['median']:.2e}; windows beyond 1e-3: {int((d > 1e-3).sum())}, beyond 1e-4: {int((d > 1e-4).sum())} of {B}")
    sens = out["eps"][eps if isinstance(eps, float) else eps[0]]
    for form in forms:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = build(sd, T, form)
            got = philox_chain(m, x_T, x_cond, seed)
        d = (got - a).abs().amax((1, 2)).cpu()
        out[form] = d
        worst = torch.topk(d, min(5, B)).indices.tolist()
        log(f"  {form}: distance from split-bf16: max {float(d.max()):.2e}; its five worst windows (distance / that window's own sensitivity): "
            + ", ".join(f"#{w}: {float(d[w]):.1e} / {float(sens[w]):.1e}" for w in worst))
        stable = sens <= 1e-4
        log(f"  {form}: over the {int(stable.sum())} windows whose split-bf16 chain moves <= 1e-4 under the perturbation: max {float(d[stable].max()):.2e}")
        m.invalidate_engine()
    m3.invalidate_engine()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="seed2")
    ap.add_argument("--window", type=int, default=120)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--eps", default="1e-6")
    ap.add_argument("--forms", default="8p,9pf")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []

    def log(sx):
        print(sx, flush=True)
        lines.append(sx)
    for kind in a.weights.split(","):
        sd, info = weights_for(kind, a.window)
        log(f"== weights {kind}, T={a.window}, B={a.batch}: {info}")
        sensitivity(sd, a.window, a.batch, [float(v) for v in a.eps.split(",")], forms=[f for f in a.forms.split(",") if f], log=log)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
