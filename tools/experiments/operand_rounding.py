#!/usr/bin/env python3
"""CPU experiment (uses the oracle, hence under tests/): what does rounding GEMM operands to fp16 / bf16 cost at the end of a chain?

Runs the oracle's 1000-step chain on B windows with the same noise, once exactly (fp32) and once per variant with the
operands of selected GEMMs rounded, and prints the max-abs difference of the final poses.

    python tools/experiments/operand_rounding.py [B] [steps]
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
        if self.act_dtype is None or id(w) not in self.marked:
            return x
        if self.act_dtype == "i8x2":  # two int8 slices: 15-bit integers, one scale per row (per row AND head for the fc input)
            rows = x if x.shape[-1] in (512, 1024, 396) else x.transpose(1, 2)  # conv1d takes [B, C, L]
            shp = rows.shape
            grp = rows.reshape(shp[:-1] + (-1, 256)) if shp[-1] == 1024 else rows.reshape(shp[:-1] + (1, shp[-1]))
            q = quant15(grp, -1).reshape(shp)
            return q if rows is x else q.transpose(1, 2)
        return x.to(self.act_dtype).float()

    def linear(self, x, w, b=None):
        return TF.linear(self._r(x, w), w, b)

    def conv1d(self, x, w, b=None):
        return TF.conv1d(self._r(x, w), w, b)


QMAX = 32639.0  # 127 * 256 + 127


def quant15(x, dim):
    amax = x.abs().amax(dim=dim, keepdim=True)
    sc = torch.where(amax > 0, amax / QMAX, torch.ones_like(amax))
    return torch.round(x / sc) * sc


def run(names, w_dtype, act_dtype):
    sd2 = dict(sd)
    marked = set()
    for k in sd:
        if any(k.endswith(n) for n in names):
            if w_dtype == "i8x2":  # one scale per output feature (row of W)
                sd2[k] = quant15(sd[k].reshape(sd[k].shape[0], -1), 1).reshape(sd[k].shape)
            else:
                sd2[k] = sd[k].to(w_dtype).float() if w_dtype is not None else sd[k].clone()
            marked.add(id(sd2[k]))
    O.F = Shim(marked, act_dtype)
    g = torch.Generator().manual_seed(123)
    out = O.p_sample_loop(sd2, sched, xs, cm, g, num_timesteps=S)
    O.F = TF
    return out


torch.set_num_threads(8)
ref = run((), None, None)
QKV = ("w_q.weight", "w_k.weight", "w_v.weight")
for label, names, wd, ad in (("Q/K/V projections on two int8 slices (as shipped, projections only)", QKV, "i8x2", "i8x2"),
                             ("tail GEMMs on two int8 slices", TAIL, "i8x2", "i8x2"),
                             ("FFN GEMMs only on two int8 slices", TAIL[1:], "i8x2", "i8x2"),
                             ("Q/K/V + tail GEMMs on two int8 slices", QKV + TAIL, "i8x2", "i8x2"),
                             ("tail weights fp16, activations exact (f16x2)", TAIL, torch.float16, None),
                             ("tail weights and activations fp16 (f16x1)", TAIL, torch.float16, torch.float16),
                             ("tail weights bf16, activations exact", TAIL, torch.bfloat16, None),
                             ("all GEMM weights fp16, activations exact", ALL, torch.float16, None),
                             ("all GEMM weights and activations fp16", ALL, torch.float16, torch.float16)):
    out = run(names, wd, ad)
    print(f"{label:55s} max|d| {float((out - ref).abs().max()):.3e}  mean|d| {float((out - ref).abs().mean()):.3e}", flush=True)
