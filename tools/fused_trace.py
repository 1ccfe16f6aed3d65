#!/usr/bin/env python3
"""Phase timeline of the fused QKV+attention kernel (perf-debug): K proj / V proj / Q proj / attention per workgroup."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd import _lib as _eglib
_eglib.use_perfdebug_build()  # needs `python -m egoego_release_amd.build --perfdebug`
from egoego_release_amd.model import CondGaussianDiffusion

B, T = 256, 120
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m.hip_precision = _lib.PREC_BF16X3  # this tool traces the split-bf16 fused kernel (attn_layer_trace.py: the int8 one)
m = m.cuda()
eng = m.hip_engine()
lib = _lib.load()
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
t = torch.full((B,), 500, device="cuda")
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
buf = torch.zeros(131072 + 8 * 4096, dtype=torch.int64, device="cuda")
lib.egoego_debug_trace_buffer.argtypes = [C.c_void_p]
lib.egoego_debug_trace_buffer(C.c_void_p(buf.data_ptr()))
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
tr = buf.cpu()[131072:].view(-1, 8)[:1024, :5].double() / 100.0
t0 = tr[:, 0].min()
tr = tr - t0
d = tr[:, 1:] - tr[:, :-1]
print("kernel span %.1f us" % tr[:, 4].max())
for name, col in zip(("K proj+epi", "V proj+epi", "Q proj (regs)", "attention"), range(4)):
    print(f"{name:14s} mean {d[:, col].mean():7.2f} us  min {d[:, col].min():7.2f}  max {d[:, col].max():7.2f}")
first = tr[tr[:, 0] < 5]
second = tr[tr[:, 0] >= 5]
print("round 1: %d WGs, end mean %.1f; round 2: %d WGs, start mean %.1f end mean %.1f" % (
    len(first), first[:, 4].mean(), len(second), second[:, 0].mean() if len(second) else -1, second[:, 4].mean() if len(second) else -1))
