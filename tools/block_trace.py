#!/usr/bin/env python3
"""Per-block timeline of one GEMM launch (perf-debug): who shares a CU, how phases overlap."""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd import _lib as _eglib
_eglib.use_perfdebug_build()  # needs `python -m egoego_release_amd.build --perfdebug`
from egoego_release_amd.model import CondGaussianDiffusion

which = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else _lib.PREC_BF16X3  # 9: `out` / `embed` trace the product kernels of precision 9
B, T = 256, 120
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs())
m.load_state_dict(make_weights(cfg, 0), strict=False)
m.hip_precision = prec
m.hip_probe_at_pack = False
m = m.cuda()
eng = m.hip_engine()
lib = _lib.load()
x = torch.randn(B, T, 198, device="cuda")
xc = torch.randn(B, T, 198, device="cuda")
t = torch.full((B,), 500, device="cuda")
eng.debug_stage(x, xc, t, 0, "out")
torch.cuda.synchronize()
buf = torch.zeros(65536 + 2 * 16384, dtype=torch.int64, device="cuda")
lib.egoego_debug_trace_buffer.argtypes = [C.c_void_p]
lib.egoego_debug_trace_buffer(C.c_void_p(buf.data_ptr()))
stage = {"ffn1": "ffn_hidden", "qkv": "q", "fc_ln": "attn_ln", "ffn2_ln": "out", "embed": "embed"}.get(which)
if which == "out":
    eng.denoise(x, xc, t)
else:
    eng.debug_stage(x, xc, t, 0, stage)   # the LAST gemm launched is the one named by `which`
torch.cuda.synchronize()
lib.egoego_debug_trace_buffer(None)
cyc = buf.cpu()[65536:].view(-1, 2)
tr = buf.cpu()[:65536].view(-1, 4)
n = {"ffn1": 512, "qkv": 3072, "fc_ln": 256, "ffn2_ln": 256, "embed": 512 if prec == _lib.PREC_BF16X3 else 256, "out": 256}[which]  # blocks of the LAST launch (earlier launches leave stale rows)
tr = tr[:n]
t0 = int(tr[:, 0].min())
print(f"{which}: {n} blocks; kernel span {(int(tr[:, 2].max()) - t0) / 100:.1f} us")
ml = (tr[:, 1] - tr[:, 0]).float() / 100
ep = (tr[:, 2] - tr[:, 1]).float() / 100
print(f"main loop us: mean {ml.mean():.2f} min {ml.min():.2f} max {ml.max():.2f};  epilogue us: mean {ep.mean():.2f} min {ep.min():.2f} max {ep.max():.2f}")
by_cu = defaultdict(list)
for b in range(n):
    hw = int(tr[b, 3]) & 0xFFFFFFFF
    xcc = (int(tr[b, 3]) >> 32) & 0xF
    cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    by_cu[cu].append((int(tr[b, 0]) - t0, int(tr[b, 1]) - t0, int(tr[b, 2]) - t0, b))
print("distinct (xcc,se,sh,cu):", len(by_cu))
for cu in list(sorted(by_cu))[:3]:
    print(cu, [(f"b{b}", s / 100, m_ / 100, e / 100) for s, m_, e, b in sorted(by_cu[cu])][:8])
# max concurrency per CU
conc = []
for cu, lst in by_cu.items():
    ev = sorted([(s, 1) for s, _, e, _ in lst] + [(e, -1) for s, _, e, _ in lst])
    c = mx = 0
    for _, d in ev:
        c += d
        mx = max(mx, c)
    conc.append(mx)
print("max concurrent blocks per CU: min", min(conc), "max", max(conc))
dc = (cyc[:n, 1] - cyc[:n, 0]).float()
dt = (tr[:, 1] - tr[:, 0]).float() / 100
print(f"shader clock during main loops: {float((dc / dt).mean()) / 1000:.3f} GHz (min {float((dc / dt).min()) / 1000:.3f}, max {float((dc / dt).max()) / 1000:.3f}); main loop cycles mean {dc.mean():.0f}")
