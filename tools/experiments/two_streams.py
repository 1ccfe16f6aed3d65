"""Experiment: a small batch as two half batches on two contexts / two streams / two host threads (do the kernels of the two chains share the chip?)."""
import os, sys, time, threading, torch
sys.path.insert(0, os.getcwd())
from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.engine import HipEngine
from egoego_release_amd.model import CondGaussianDiffusion
from egoego_release_amd.precision import _engine_cfg
T = 120
cfg = ModelConfig(max_timesteps=T + 1)
m = CondGaussianDiffusion(**cfg.ctor_kwargs()); m.load_state_dict(make_weights(cfg, 0), strict=False); m = m.cuda()
ecfg = _engine_cfg(m); dev = torch.device("cuda")
K = 200
def mk(): return HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3_FC, 0)
def run_one(eng, x, xc, stream):
    with torch.cuda.stream(stream):
        eng.sample_loop_(x, xc, 900, K, noise_mode=_lib.NOISE_PHILOX, seed=7, window_offset=0)
for B in (24, 32, 48, 64):
    e0, e1, e2 = mk(), mk(), mk()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.randn(B, T, 198, device="cuda"); xc = torch.randn(B, T, 198, device="cuda")
    h = B // 2
    xa, xb, xca, xcb = x[:h].contiguous(), x[h:].contiguous(), xc[:h].contiguous(), xc[h:].contiguous()
    res = {}
    for rep in range(3):
        xx = x.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.sample_loop_(xx, xc, 900, K, noise_mode=_lib.NOISE_PHILOX, seed=7, window_offset=0); torch.cuda.synchronize()
        res["one"] = 1e3 * (time.perf_counter() - t0) / K
        a, b = xa.clone(), xb.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=run_one, args=(e1, a, xca, s1)), threading.Thread(target=run_one, args=(e2, b, xcb, s2))]
        [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
        res["two"] = 1e3 * (time.perf_counter() - t0) / K
    print(f"B={B}: one context {res['one']:.4f} ms/step; two half batches on two streams {res['two']:.4f} ms/step", flush=True)
    for e in (e0, e1, e2): e.close()
