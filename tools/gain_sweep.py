#!/usr/bin/env python3
"""Worst case of tools/hostile_weights_check.py's LayerNorm-gain stress: the SAME six features amplified in every LayerNorm (their
outliers compound through the residual stream).  Error of each precision relative to max(1, |y|max), fp32 PyTorch on the GPU as
the comparison."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.model import CondGaussianDiffusion

warnings.simplefilter("ignore")
cfg = ModelConfig(max_timesteps=121)
g = torch.Generator().manual_seed(3)
x_all = torch.randn(2, 120, 396, generator=g)
t = torch.tensor([7, 900])
xa, xb = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda()
for mult in (1, 2, 3, 4, 5, 6, 8, 10, 15, 25):
    sd = make_weights(cfg, 0)
    for k in sd:
        if k.endswith("layer_norm.weight"):
            sd[k] = sd[k].clone()
            sd[k][:6] *= mult
    ref = CondGaussianDiffusion(**cfg.ctor_kwargs())
    ref.load_state_dict(sd, strict=False)
    ref = ref.cuda()
    ref.denoise_fn.eval()  # no dropout
    with torch.no_grad():
        want = ref.denoise_fn(torch.cat((xa, xb), -1), t.cuda())
    row = []
    for prec in (_lib.PREC_I8X3_FC, _lib.PREC_I8X3, _lib.PREC_BF16X3):
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(sd, strict=False)
        m.hip_precision = prec
        m = m.cuda()
        got = m.denoise(xa, t.cuda(), xb)
        row.append((got - want).abs().max().item() / max(1.0, want.abs().max().item()))
    print(f"gains x{mult:<3d} on the same 6 features of all LayerNorms: |y|max {want.abs().max().item():6.2f}  relative error  9: {row[0]:.2e}  8: {row[1]:.2e}  3: {row[2]:.2e}", flush=True)
