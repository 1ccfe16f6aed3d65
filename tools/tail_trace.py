#!/usr/bin/env python3
"""Phase timeline of the fused layer-tail kernel (perf-debug build): per workgroup, wall-clock marks at
 0 start | 1 fc k-loops done | 2 LayerNorm-1 done | 3 FFN-1 k-loops | 4 FFN-1 epilogue | 5 FFN-2 k-loops | 6 LayerNorm-2 done.
 (wall_clock64 ticks at 100 MHz)    TT_B=<batch> selects the batch size (and with it the token-block variant)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
_lib.use_perfdebug_build()  # needs `python -m egoego_release_amd.build --perfdebug`
from egoego_release_amd.model import CondGaussianDiffusion

B = int(os.environ.get("TT_B", 256))
T = 120
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m = m.cuda()
eng = m.hip_engine()
lib = _lib.load()
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
t = torch.full((B,), 500, device="cuda")
eng.denoise(x, xc, t)
torch.cuda.synchronize()
buf = torch.zeros(262144, dtype=torch.int64, device="cuda")
lib.egoego_debug_trace_buffer.argtypes = [C.c_void_p]
lib.egoego_debug_trace_buffer(C.c_void_p(buf.data_ptr()))
eng.denoise(x, xc, t)
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
nwg = {True: B * 128 // 128}.get(True)
tr = buf.cpu().numpy()[180224:180224 + 32768].reshape(-1, 32)
tr = tr[(tr[:, 0] > 0) & (tr[:, 6] > 0)]
print("workgroups traced:", len(tr))
t0 = tr[:, 0].min()
names = {1: "fc k-loops", 2: "LayerNorm-1", 3: "FFN-1 k-loops", 4: "FFN-1 epilogue", 5: "FFN-2 k-loops", 6: "LayerNorm-2"}
prev = 0
for i in range(1, 7):
    if (tr[:, i] == 0).all():
        continue
    d = (tr[:, i] - tr[:, prev]) / 100.0
    mhz = ((tr[:, 16 + i] - tr[:, 16 + prev]) / d).mean()
    print(f"{names[i]:18s} mean {d.mean():7.2f} us   min {d.min():7.2f}   max {d.max():7.2f}   shader clock {mhz:6.0f} MHz")
    prev = i
# (fc in the 256-register build runs its two feature halves as two passes and BOTH stamp mark 7: "fc prologue" below is the whole first
#  pass + the second pass's pipeline fill, "chunk loop" the second pass alone — read as a 15-us prologue in round 4 until a build with
#  the two passes as one pipeline showed a 2.6-us fill and the same 27 us of fc)
for nm, (pro, st, en) in {"fc (pass 1 + fill of pass 2 | pass 2)": (0, 7, 1), "FFN-1": (2, 8, 3), "FFN-2": (4, 9, 5)}.items():
    print(f"{nm:6s} prologue (first chunk + weights + residual in flight -> landed) {((tr[:, st] - tr[:, pro]) / 100.0).mean():6.2f} us,"
          f" chunk loop {((tr[:, en] - tr[:, st]) / 100.0).mean():6.2f} us at {((tr[:, 16 + en] - tr[:, 16 + st]) / ((tr[:, en] - tr[:, st]) / 100.0)).mean():5.0f} MHz")
start = (tr[:, 0] - t0) / 100.0
early = start < 5.0  # the workgroups resident from the launch on; the others start as these finish
for nm, sel in (("first round (start < 5 us)", early), ("later rounds", ~early)):
    if sel.any():
        print(f"{nm:28s} n={int(sel.sum()):4d}  fc pass 1 + fill {((tr[sel, 7] - tr[sel, 0]) / 100.0).mean():6.2f} us  fc pass 2 {((tr[sel, 1] - tr[sel, 7]) / 100.0).mean():6.2f}"
              f"  LN-1 {((tr[sel, 2] - tr[sel, 1]) / 100.0).mean():5.2f}  FFN-1 {((tr[sel, 4] - tr[sel, 2]) / 100.0).mean():5.2f}  FFN-2+LN-2 {((tr[sel, 6] - tr[sel, 4]) / 100.0).mean():5.2f}"
              f"  total {((tr[sel, 6] - tr[sel, 0]) / 100.0).mean():6.2f}")
tot = (tr[:, 6] - tr[:, 0]) / 100.0
print(f"workgroup total     mean {tot.mean():7.2f} us   min {tot.min():7.2f}   max {tot.max():7.2f}")
print(f"kernel span {(tr[:, 6].max() - t0) / 100.0:.2f} us; start spread {(tr[:, 0].max() - t0) / 100.0:.2f} us")
