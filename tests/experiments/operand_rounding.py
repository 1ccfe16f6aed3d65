#!/usr/bin/env python3
"""CPU experiment (uses the oracle, hence under tests/): what does rounding GEMM operands to fp16 / bf16 cost at the end of a chain?

Runs the oracle's 1000-step chain on B windows with the same noise, once exactly (fp32) and once per variant with the
operands of selected GEMMs rounded, and prints the max-abs difference of the final poses.

    python tests/experiments/operand_rounding.py [B] [steps]
"""
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from egoego_release_amd import ModelConfig, make_weights, make_head_windows  # noqa: E402
from oracle import egoego_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
T = 120
cfg = ModelConfig(max_timesteps=T + 1)
sd = {k: v.float() for k, v in make_weights(cfg, 0).items()}
sched = O.make_schedule(1000)
xs, cm = make_head_windows(B, T, seed=1)

TAIL = ("fc.weight", "w_1.weight", "w_2.weight")
ALL = TAIL + ("w_q.weight", "w_k.weight", "w_v.weight", "start_conv.weight", "linear_out.weight")


class Shim:
    """stands in for torch.nn.functional inside the oracle: rounds the activations of the marked GEMMs"""
    def __init__(self, marked, act_dtype):
        self.marked, self.act_dtype = marked, act_dtype

    def __getattr__(self, n):
        return getattr(TF, n)

    def _r(self, x, w):
        return x.to(self.act_dtype).float() if (self.act_dtype is not None and id(w) in self.marked) else x

    def linear(self, x, w, b=None):
        return TF.linear(self._r(x, w), w, b)

    def conv1d(self, x, w, b=None):
        return TF.conv1d(self._r(x, w), w, b)


def run(names, w_dtype, act_dtype):
    sd2 = dict(sd)
    marked = set()
    for k in sd:
        if any(k.endswith(n) for n in names):
            sd2[k] = sd[k].to(w_dtype).float() if w_dtype is not None else sd[k].clone()
            marked.add(id(sd2[k]))
    O.F = Shim(marked, act_dtype)
    g = torch.Generator().manual_seed(123)
    out = O.p_sample_loop(sd2, sched, xs, cm, g, num_timesteps=S)
    O.F = TF
    return out


torch.set_num_threads(8)
ref = run((), None, None)
for label, names, wd, ad in (("tail weights fp16, activations exact (f16x2)", TAIL, torch.float16, None),
                             ("tail weights and activations fp16 (f16x1)", TAIL, torch.float16, torch.float16),
                             ("tail weights bf16, activations exact", TAIL, torch.bfloat16, None),
                             ("all GEMM weights fp16, activations exact", ALL, torch.float16, None),
                             ("all GEMM weights and activations fp16", ALL, torch.float16, torch.float16)):
    out = run(names, wd, ad)
    print(f"{label:55s} max|d| {float((out - ref).abs().max()):.3e}  mean|d| {float((out - ref).abs().mean()):.3e}", flush=True)
