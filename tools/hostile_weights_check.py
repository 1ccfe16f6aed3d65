#!/usr/bin/env python3
"""How far can a checkpoint's statistics drift from the reference's initialisation before the int8-slice precision (one scale
per row) leaves the 1e-3 bar?  Sweeps outlier LayerNorm gains and heavy-tailed projection weights; fp32 PyTorch on the GPU
is the comparison (no oracle import: this is a perf-debug tool, the parity tests live in tests/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
if "EGOEGO_PERFDEBUG_TAG" in os.environ:  # check a variant build (libegoego_hip_perfdebug_<tag>.so)
    _lib.use_perfdebug_build()
from egoego_release_amd.model import CondGaussianDiffusion

cfg = ModelConfig(max_timesteps=121)


def run(gain_mult, n_gain, tail_mult, tail_frac, beta_std):
    g = torch.Generator().manual_seed(5)
    sd = make_weights(cfg, 0)
    for k in list(sd):
        if "layer_norm.weight" in k and gain_mult != 1:
            idx = torch.randperm(512, generator=g)[:n_gain]
            sd[k] = sd[k].clone()
            sd[k][idx] *= gain_mult
        if "layer_norm.bias" in k and beta_std:
            sd[k] = sd[k].clone() + beta_std * torch.randn(512, generator=g)
        if tail_mult != 1 and any(s in k for s in ("w_q.weight", "w_k.weight", "w_v.weight")):
            w = sd[k].clone()
            w[torch.rand(w.shape, generator=g) < tail_frac] *= tail_mult
            sd[k] = w
    x_all = torch.randn(2, 120, 396, generator=g)
    t = torch.tensor([3, 977])
    out = {}
    for prec in (3, 8, 9):
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(sd, strict=False)
        m.hip_precision = prec
        m = m.cuda()
        if prec == 3:
            m.denoise_fn.eval()  # no dropout
            with torch.no_grad():
                ref = m.denoise_fn(x_all.cuda(), t.cuda())  # the module's plain-PyTorch fp32 forward (training path)
        out[prec] = m.denoise(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda())
        layers = getattr(m.hip_engine(), "i8_layers", lambda: None)()
    e3, e8, e9 = (float((out[k] - ref).abs().max()) for k in (3, 8, 9))
    print(f"LN gain x{gain_mult:<4} on {n_gain:2d} features, beta std {beta_std:<4}, QKV weight tails x{tail_mult:<3} ({tail_frac:.3f}): "
          f"|y|max {float(ref.abs().max()):5.2f}  bf16x3 {e3:.2e}  i8x3 {e8:.2e}  i8x3+fc {e9:.2e}" + (f"  i8 layers {layers:04b}" if layers is not None else ""))


for args in ((1, 0, 1, 0, 0), (3, 6, 1, 0, 0), (8, 6, 1, 0, 0), (25, 6, 1, 0, 0), (25, 1, 1, 0, 0), (1, 0, 1, 0, 0.5), (1, 0, 4, 0.002, 0),
             (1, 0, 12, 0.002, 0), (25, 6, 12, 0.002, 0.5)):
    run(*args)
