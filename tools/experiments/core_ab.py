#!/usr/bin/env python3
"""A/B of two builds of the library on the long window (T = 196): dump what this build computes, or compare two dumps.
    python tools/experiments/core_ab.py dump out.pt       (EGOEGO_PERFDEBUG_TAG=<tag> selects a variant build; AB_PRECS=3 the precisions, default 9,8; AB_WINDOWS=120 / AB_BATCHES=1,130 the shapes, default 196,150 / 1,5,67)
    python tools/experiments/core_ab.py cmp a.pt b.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        print(f"{k:28s} equal {torch.equal(a[k], b[k])}   max |diff| {float(d.max()):.3e}   |value|max {float(a[k].abs().max()):.3f}")
    sys.exit(0 if all(torch.equal(a[k], b[k]) for k in a) else 1)
from egoego_release_amd import ModelConfig, make_weights, _lib  # noqa: E402
if "EGOEGO_PERFDEBUG_TAG" in os.environ:
    _lib.use_perfdebug_build()
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402

out = {}
for T in [int(v) for v in os.environ.get("AB_WINDOWS", "196,150").split(",")]:
    cfg = ModelConfig(max_timesteps=T + 1)
    for prec in [int(v) for v in os.environ.get("AB_PRECS", "9,8").split(",")]:
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(make_weights(cfg, 0), strict=False)
        m.hip_precision, m.hip_probe_at_pack = prec, False
        m = m.cuda()
        eng = m.hip_engine()
        for B in [int(v) for v in os.environ.get("AB_BATCHES", "1,5,67").split(",")]:
            g = torch.Generator().manual_seed(B)
            x, xc = torch.randn(B, T, 198, generator=g).cuda(), torch.randn(B, T, 198, generator=g).cuda()
            t = torch.randint(0, 1000, (B,), generator=g).cuda()
            out[f"T{T} p{prec} B{B} attn_out L0"] = eng.debug_stage(x, xc, t, 0, "attn_out").cpu()
            out[f"T{T} p{prec} B{B} attn_out L3"] = eng.debug_stage(x, xc, t, 3, "attn_out").cpu()
            out[f"T{T} p{prec} B{B} denoise"] = eng.denoise(x, xc, t).cpu()
            y = x.clone()
            eng.sample_loop_(y, xc, 999, 7, noise_mode=_lib.NOISE_PHILOX, seed=3)
            out[f"T{T} p{prec} B{B} 7 steps"] = y.cpu()
torch.save(out, sys.argv[2])
print("dumped", len(out), "tensors to", sys.argv[2])
