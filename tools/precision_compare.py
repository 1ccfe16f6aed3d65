#!/usr/bin/env python3
"""Max-abs difference between operand precisions on the same inputs (GPU only, no oracle):
one denoiser forward and a short sampling chain at a batch size that takes the fused kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.model import CondGaussianDiffusion

B, T = int(os.environ.get("PC_B", 64)), int(os.environ.get("PC_T", 120))
steps = int(os.environ.get("PC_STEPS", 20))
cfg = ModelConfig(max_timesteps=T + 1)
out = {}
g = torch.Generator(device="cpu").manual_seed(3)
x0 = torch.randn(B, T, 198, generator=g).cuda()
xc0 = torch.randn(B, T, 198, generator=g).cuda()
t = torch.randint(0, 1000, (B,), generator=g).cuda()
for prec in (_lib.PREC_BF16X3, _lib.PREC_I8X3, _lib.PREC_I8X3_FC, _lib.PREC_BF16X1):
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(make_weights(cfg, 0), strict=False)
    m.hip_precision = prec
    m = m.cuda()
    eng = m.hip_engine()
    fwd = eng.denoise(x0, xc0, t)
    x = x0.clone()
    eng.sample_loop_(x, xc0, steps - 1, steps, noise_mode=_lib.NOISE_PHILOX, seed=11)
    torch.cuda.synchronize()
    out[prec] = (fwd.clone(), x.clone())
ref = out[_lib.PREC_BF16X3]
for prec, name in ((_lib.PREC_I8X3, "i8x3"), (_lib.PREC_I8X3_FC, "i8x3+fc"), (_lib.PREC_BF16X1, "bf16x1")):
    print(f"{name:7s} vs bf16x3: forward {float((out[prec][0] - ref[0]).abs().max()):.3e}   "
          f"{steps}-step chain {float((out[prec][1] - ref[1]).abs().max()):.3e}   (|x| max {float(ref[1].abs().max()):.2f})")
