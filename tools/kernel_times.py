#!/usr/bin/env python3
"""Per-kernel mean launch time (HIP events inside the library) for one configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.model import CondGaussianDiffusion

if "EGOEGO_PERFDEBUG_TAG" in os.environ:  # time an ablation build (build --perfdebug --tag=X -D...)
    _lib.use_perfdebug_build()
B, T = int(os.environ.get("KT_B", 256)), int(os.environ.get("KT_T", 120))
prec = int(os.environ.get("KT_PREC", 9))  # 9 = i8x3 + int8 fc (default), 8 = i8x3, 3 = split-bf16 everywhere, 1 = plain bf16
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m.hip_precision = prec
m = m.cuda()
eng = m.hip_engine()
if "KT_ABLATE" in os.environ:  # perf-debug build: 2 = the ring-GEMM kernels (embed / linear_out at large grids) skip their epilogues
    _lib.load().egoego_debug_ablate(int(os.environ["KT_ABLATE"]))
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
NM = int(os.environ.get("KT_NOISE", _lib.NOISE_PHILOX))  # 1 = in-kernel Philox (default), 2 = no noise (what the generator costs the out kernel)
eng.sample_loop_(x, xc, 999, 3, noise_mode=NM)
torch.cuda.synchronize()
tot = 0.0
for k in ("embed", "qkv", "attn", "fc_ln", "ffn1", "ffn2_ln", "out"):
    eng.profile_begin(k)
    eng.sample_loop_(x, xc, 900, 5, noise_mode=NM)
    us, n = eng.profile_end()
    per_step = us * n / 5
    tot += per_step
    print(f"{k:8s} {us:9.1f} us x {n // 5}/step = {per_step:9.1f} us/step")
print(f"sum      {tot:9.1f} us/step")
