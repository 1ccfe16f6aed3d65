#!/usr/bin/env python3
"""The WHOLE 1000-step ancestral chain on the trained-like checkpoint against the fp32 CPU oracle (B windows, the oracle's draws):
what `auto` runs, precision 8 prepared, split-bf16.  ~2 min of CPU for the oracle.   python tools/trained_like_full_chain.py [B] [T]"""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import ModelConfig, head_condition_mask, _lib  # noqa: E402
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402
from make_trained_like_checkpoint import train_like  # noqa: E402
from oracle import egoego_oracle as O  # noqa: E402  (perf-debug tool: the oracle is the checker here, as in tests/)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
S = 1000
warnings.simplefilter("ignore")
cfg = ModelConfig(max_timesteps=T + 1)
sd, info = train_like(3000, 0, "cuda", T)
sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
data = make_motion_windows(B, T, seed=777)
mask = head_condition_mask(data.shape)
g = torch.Generator().manual_seed(99)
nz = {"x_T": torch.randn(data.shape, generator=g), "cond": torch.randn(data.shape, generator=g), "steps": torch.randn(S, *data.shape, generator=g)}
outs = {}
for name, prec, prep in (("auto", "auto", "auto"), ("8 prepared", _lib.PREC_I8X3, "always"), ("3", _lib.PREC_BF16X3, "never")):
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision, m.hip_int8_prep = prec, prep
    m = m.cuda()
    outs[name] = m.sample(data.cuda(), mask.cuda(), noise=nz).cpu()
    print(name, "runs precision", m.hip_precision_used, "probe", m.hip_precision_probe, flush=True)
sched = O.make_schedule(S)
x = nz["x_T"].clone()
x_cond = data * (1 - mask) + mask * nz["cond"]
t0 = time.time()
with torch.no_grad():
    for i, tv in enumerate(reversed(range(S))):
        x = O.p_sample(sd, sched, x, torch.full((B,), tv, dtype=torch.long), x_cond, nz["steps"][i])
print(f"oracle chain: {time.time() - t0:.0f} s")
for name, o in outs.items():
    d = (o - x).abs()
    print(f"{S}-step chain, B={B}, T={T}: {name:10s} max|HIP - oracle| = {float(d.max()):.2e}   per window {[f'{float(v):.1e}' for v in d.amax((1, 2))]}")
