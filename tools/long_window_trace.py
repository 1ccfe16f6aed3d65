#!/usr/bin/env python3
"""Phase timelines of the long-window pair (T = 196: qkv_i8q_kernel + attn_core_i8w_kernel<7>, or attn_core_i8_kernel<7> in an EGOEGO_CORE4 build), perf-debug build:
per-workgroup wall-clock stamps at the phase boundaries.   python -m egoego_release_amd.build --perfdebug first."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
_lib.use_perfdebug_build()
from egoego_release_amd.model import CondGaussianDiffusion

B, T = int(os.environ.get("TT_B", 256)), 196
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m.hip_precision = _lib.PREC_I8X3_FC
m.hip_int8_prep = "never"
m = m.cuda()
eng = m.hip_engine()
lib = _lib.load()
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
t = torch.full((B,), 500, device="cuda")
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
buf = torch.zeros(262144, dtype=torch.int64, device="cuda")
lib.egoego_debug_trace_buffer.argtypes = [C.c_void_p]
lib.egoego_debug_trace_buffer(C.c_void_p(buf.data_ptr()))
eng.debug_stage(x, xc, t, 0, "attn_out")
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
raw = buf.cpu()


def report(title, tr, names):
    tr = tr.double() / 100.0
    tr = tr - tr[:, 0].min()
    d = tr[:, 1:] - tr[:, :-1]
    n = len(names)
    print(f"{title}: {tr.shape[0]} workgroups, kernel span {tr[:, n].max():.1f} us, per-workgroup total mean {(tr[:, n] - tr[:, 0]).mean():.2f} us")
    for i, name in enumerate(names):
        print(f"  {name:28s} mean {d[:, i].mean():7.2f} us  min {d[:, i].min():7.2f}  max {d[:, i].max():7.2f}")


Lr = ((T + 1 + 15) // 16) * 16
nq = 12 * ((B * Lr + 127) // 128)
q = raw[131072:131072 + nq * 8].view(-1, 8)
q = q[q[:, 3] > 0]
report("qkv_i8q_kernel", q[:, :4], ("prologue + main loop", "dequantise + row maxima", "quantise + store"))
c = raw[90112 + 131072:90112 + 131072 + B * 4 * 2 * 8].view(-1, 8)
c = c[c[:, 5] > 0]
cyc = (c[:, 7] - c[:, 6]).double()
print("S^T phase: %.0f shader cycles per workgroup = %.0f MHz; 168 MFMAs per wave -> %.1f cycles per MFMA issued" % (cyc.mean(), (cyc / ((c[:, 2] - c[:, 1]).double() / 100.0)).mean(), cyc.mean() / 168))
report("attention core (7 key tiles)", c[:, :6], ("K half 0 + Q landed", "S^T (both halves)", "softmax + quantise P", "PV, both d_v halves", "row maxima + int8 store"))
