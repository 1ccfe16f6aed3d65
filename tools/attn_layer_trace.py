#!/usr/bin/env python3
"""Phase timeline of attn_layer_i8w_kernel (perf-debug): per-workgroup timestamps at the phase boundaries."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd import _lib as _eglib
_eglib.use_perfdebug_build()  # needs `python -m egoego_release_amd.build --perfdebug`
from egoego_release_amd.model import CondGaussianDiffusion

B, T = int(os.environ.get("TT_B", 256)), 120
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m.hip_precision = int(os.environ.get("KT_PREC", _lib.PREC_I8X3_FC))
m = m.cuda()
eng = m.hip_engine()
lib = _lib.load()
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
t = torch.full((B,), 500, device="cuda")
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
buf = torch.zeros(131072 + 16 * 4096, dtype=torch.int64, device="cuda")
lib.egoego_debug_trace_buffer.argtypes = [C.c_void_p]
lib.egoego_debug_trace_buffer(C.c_void_p(buf.data_ptr()))
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
raw = buf.cpu()[131072:].view(-1, 16)[:B * 4]
cyc = (raw[:, 13] - raw[:, 12]).double()
tr = raw[:, :9].double() / 100.0
print(raw[:3, 12:14], raw[:3, :2])
print("shader clock during the K main loop: %.0f MHz (cycle counter / wall clock)" % (cyc / (tr[:, 1] - tr[:, 0])).mean())
tr = tr - tr[:, 0].min()
d = tr[:, 1:] - tr[:, :-1]
print("kernel span %.1f us; per-WG total mean %.2f us" % (tr[:, 8].max(), (tr[:, 8] - tr[:, 0]).mean()))
names = ("K main loop", "K epilogue", "Q main loop", "Q epilogue", "S^T + softmax", "V main loop", "V epilogue", "PV + store")
for name, col in zip(names, range(8)):
    print(f"{name:14s} mean {d[:, col].mean():7.2f} us  min {d[:, col].min():7.2f}  max {d[:, col].max():7.2f}")
