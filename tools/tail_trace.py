#!/usr/bin/env python3
"""Phase timeline of the fused layer-tail kernel (perf-debug): main loop / epilogue of fc+LN, FFN-1, FFN-2+LN per workgroup."""
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
m.hip_precision = int(os.environ.get("TT_PREC", _lib.PREC_I8X3))
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
eng.debug_stage(x, xc, t, 1, "embed") if False else eng.denoise(x, xc, t)
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
raw = buf.cpu()
nb = B * 128 // 64
ph = [raw[p * 4096:p * 4096 + nb * 4].view(nb, 4)[:, :3].double() / 100.0 for p in range(3)]
t0 = ph[0][:, 0].min()
print("kernel span %.1f us" % (ph[2][:, 2].max() - t0))
for name, a in zip(("fc + LN", "FFN-1", "FFN-2 + LN"), ph):
    print(f"{name:11s} main loop {(a[:, 1] - a[:, 0]).mean():7.2f} us   epilogue {(a[:, 2] - a[:, 1]).mean():7.2f} us")
print("gap fc->FFN-1 %.2f us, FFN-1->FFN-2 %.2f us" % ((ph[1][:, 0] - ph[0][:, 2]).mean(), (ph[2][:, 0] - ph[1][:, 2]).mean()))
