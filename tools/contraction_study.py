#!/usr/bin/env python3
"""Does the sampling chain CONTRACT or AMPLIFY a perturbation, per window?  In split-bf16 alone, at the metric's size: chain A (Philox draws) and
chain B = the same engine and draws with eps * N(0, 1) added to x when t_inj steps remain; gain of a window = |A - B|max / eps at the end.
For checkpoints of one recipe stopped after 0 / 10 / ... / 3000 Adam steps (tools/amplification_study.py shows which of them `auto` accepts an
int8 form for, and what the worst window of 256 then is): is there a figure that separates the chains on which 16-bit fixed point is safe?

    python tools/contraction_study.py [--steps 0,10,30,50,70,100,300,1000,3000] [--inject 1000,400,100,20] [--eps 1e-4] [--out file]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import ModelConfig, make_weights, head_condition_mask, _lib  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402
from chain_tail_b256 import build, stats  # noqa: E402
from make_trained_like_checkpoint import train_like  # noqa: E402

S = 1000


def gains(sd, T, B, injects, eps, data_seed=31337, seed=11):
    data = make_motion_windows(B, T, seed=data_seed)
    mask = head_condition_mask(data.shape)
    g = torch.Generator().manual_seed(data_seed + 1)
    x_T = torch.randn(data.shape, generator=g).cuda()
    x_cond = (data * (1 - mask) + mask * torch.randn(data.shape, generator=g)).cuda()
    m3 = build(sd, T, "3")
    eng = m3.hip_engine(verify=True)
    pert = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(99)).cuda() * eps
    marks = sorted({int(v) for v in injects}, reverse=True)
    x, cur, kept = x_T.clone(), S, {}
    for ti in marks:  # states of chain A with ti steps still to run
        if cur > ti:
            eng.sample_loop_(x, x_cond, cur - 1, cur - ti, noise_mode=_lib.NOISE_PHILOX, seed=seed)
            cur = ti
        kept[ti] = x.clone()
    if cur > 0:
        eng.sample_loop_(x, x_cond, cur - 1, cur, noise_mode=_lib.NOISE_PHILOX, seed=seed)
    out = {}
    for ti, y in kept.items():
        y = y + pert
        eng.sample_loop_(y, x_cond, ti - 1, ti, noise_mode=_lib.NOISE_PHILOX, seed=seed)
        out[ti] = ((y - x).abs().amax((1, 2)) / eps).cpu()
    torch.cuda.synchronize()
    m3.invalidate_engine()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", default="0,10,30,50,70,100,300,1000,3000")
    ap.add_argument("--inject", default="1000,400,100,20")
    ap.add_argument("--eps", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--window", type=int, default=120)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
    T = a.window
    log(f"# tools/contraction_study.py --steps {a.steps} --inject {a.inject} --eps {a.eps:g} --seed {a.seed} --window {T} --batch {a.batch}; {torch.cuda.get_device_name(0)}")
    log("# gain of a window = |split-bf16 chain - the same chain with eps N(0,1) added to x when t_inj steps remain|max / eps")
    for steps in [int(v) for v in a.steps.split(",")]:
        if steps == 0:
            sd = make_weights(ModelConfig(max_timesteps=T + 1), a.seed)
        else:
            sd, _ = train_like(steps, a.seed, "cuda", T)
            sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
        res = gains(sd, T, a.batch, a.inject.split(","), a.eps)
        for ti, d in res.items():
            st = stats(d)
            log(f"   {steps:6d} Adam steps, perturbed with {ti:4d} steps to go: gain max {st['max']:8.3f} (window {st['argmax']:3d})  p99 {st['p99']:8.3f}  median {st['median']:8.3f}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
