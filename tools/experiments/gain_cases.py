#!/usr/bin/env python3
"""What `auto` does on the LayerNorm-gain cases of tests/test_gpu_parity.py (x2, x3, x25 on six features) and what the stage taps of
precision 9 measure against the oracle and against split-bf16 — the numbers the tests pin (round 5)."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from egoego_release_amd import ModelConfig, make_weights, _lib  # noqa: E402
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402
from oracle import egoego_oracle as O  # noqa: E402  (perf-debug tool: the oracle is the checker, as in tests/)

cfg = ModelConfig(max_timesteps=121)
sd = make_weights(cfg, 0)
g = torch.Generator().manual_seed(3)
x_all = torch.randn(2, 120, 396, generator=g)
t = torch.tensor([7, 900])
xa, xb = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda()
for gain in (2.0, 3.0, 5.0, 25.0):
    w = {k: v.clone() for k, v in sd.items()}
    for k in w:
        if k.endswith("layer_norm.weight"):
            w[k][:6] *= gain
    with torch.no_grad():
        want = O.denoise(w, x_all, t)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(w, strict=False)
    m.hip_plan_cache = False
    m = m.cuda()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = m.denoise(xa, t.cuda(), xb).cpu()
    pr = m.hip_precision_probe
    print(f"gain x{gain:g}: auto runs {m.hip_precision_used} {pr['form']}; errors {dict((k, float(f'{v:.2e}')) for k, v in pr['errors'].items())}; "
          f"|y|max {float(want.abs().max()):.1f}; error vs oracle {float((got - want).abs().max()):.2e}; warnings {[str(r.message)[:60] for r in rec]}", flush=True)
# stage taps of precision 9 (product kernels) and 3 against the oracle
B, T, H = 2, 120, 4
x_all = torch.randn(B, T, 396, generator=torch.Generator().manual_seed(1120))
tt = torch.tensor([3, 977])
taps = {}
with torch.no_grad():
    O.denoise(sd, x_all, tt, taps=taps)
xd, xcd, td = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda(), tt.cuda()
engs = {}
for p in (9, 3):
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision, m.hip_probe_at_pack = p, False
    engs[p] = m.cuda().hip_engine()
    engs[p]._m = m
for li in (0, 3):
    for st in (("embed",) if li == 0 else ()) + ("attn_out", "attn_ln", "ffn_hidden", "out"):
        want = taps["embed"] if st == "embed" else taps[f"layer{li}"][st]
        a9, a3 = engs[9].debug_stage(xd, xcd, td, li, st).cpu(), engs[3].debug_stage(xd, xcd, td, li, st).cpu()
        print(f"layer {li} {st:10s} |tap|max {float(want.abs().max()):6.2f}: 9 vs oracle {float((a9 - want).abs().max()):.2e}, 9 vs 3 {float((a9 - a3).abs().max()):.2e}, "
              f"3 vs oracle {float((a3 - want).abs().max()):.2e}; 9 vs oracle relative to the row maximum: {float(((a9 - want).abs().amax(-1) / want.abs().amax(-1)).max()):.2e}")
