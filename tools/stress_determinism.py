#!/usr/bin/env python3
"""Race screen: long chains at full size must be bit-identical run to run (any LDS-DMA / barrier hazard shows
up as nondeterminism), for several shapes and both precisions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.model import CondGaussianDiffusion

bad = 0
for (B, T, prec, steps) in ((256, 120, 9, 150), (128, 120, 9, 150), (64, 120, 9, 150), (32, 120, 9, 200), (24, 120, 9, 200), (16, 120, 9, 200), (8, 120, 9, 200), (1, 120, 9, 300),
                            (300, 100, 9, 60), (5, 127, 9, 60), (256, 196, 9, 40), (32, 196, 9, 60), (48, 60, 9, 100),  # the default precision: every batch-size tier of its dispatch
                            (256, 120, 8, 150), (64, 120, 8, 150), (7, 120, 8, 100), (300, 100, 8, 60), (5, 127, 8, 60), (256, 196, 8, 40), (32, 120, 8, 200), (128, 120, 8, 150), (32, 196, 8, 60), (1, 120, 8, 200), (48, 60, 8, 100),
                            (256, 120, 3, 150), (64, 120, 3, 100), (96, 30, 3, 100), (256, 120, 1, 100),
                            (256, 196, 3, 60), (70, 150, 3, 80), (32, 196, 3, 80), (1, 196, 3, 100)):  # split-bf16 long windows: the eight-wave ring (counted vmcnt, untracked loads) and the four-wave pair
    cfg = ModelConfig(max_timesteps=T + 1)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(make_weights(cfg, 0), strict=False)
    m.hip_precision = prec
    m = m.cuda()
    eng = m.hip_engine()
    g = torch.Generator().manual_seed(B + T)
    x0 = torch.randn(B, T, 198, generator=g).cuda()
    xc = torch.randn(B, T, 198, generator=g).cuda()
    outs = []
    for rep in range(3):
        x = x0.clone()
        eng.sample_loop_(x, xc, 999, steps, noise_mode=_lib.NOISE_PHILOX, seed=5)
        torch.cuda.synchronize()
        outs.append(x)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    finite = bool(torch.isfinite(outs[0]).all())
    print(f"B={B} T={T} prec={prec} steps={steps}: bit-identical x3 = {same}, finite = {finite}", flush=True)
    bad += (not same) or (not finite)
    del m, eng
sys.exit(1 if bad else 0)
