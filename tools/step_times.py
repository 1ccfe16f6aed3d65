#!/usr/bin/env python3
"""Per-step time of the HIP sampling loop over batch sizes / window lengths, in-kernel Philox noise and the
module-level default path (sampling_rng='torch').  Prints one JSON line per configuration.

    python tools/step_times.py [--steps 100] [--batches 1,16,32,64,128,256] [--windows 120,196] [--api]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights, make_head_windows, _lib  # noqa: E402
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402


def main():
    if "EGOEGO_PERFDEBUG_TAG" in os.environ:  # time a variant build (build --perfdebug --tag=X -D...)
        _lib.use_perfdebug_build()
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batches", default="1,16,32,64,128,256")
    ap.add_argument("--windows", default="120")
    ap.add_argument("--precision", type=int, default=9)
    ap.add_argument("--api", action="store_true", help="also time model.sample() with sampling_rng torch / philox")
    a = ap.parse_args()
    for T in [int(v) for v in a.windows.split(",")]:
        cfg = ModelConfig(max_timesteps=T + 1)
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(make_weights(cfg, 0), strict=False)
        m.hip_precision = a.precision
        m = m.cuda()
        eng = m.hip_engine()
        for B in [int(v) for v in a.batches.split(",")]:
            xs, cm = make_head_windows(B, T, seed=1)
            x = torch.randn(xs.shape, device="cuda")
            xc = (xs * (1 - cm) + cm * torch.randn(xs.shape)).cuda()
            for _ in range(2):  # twice: the first chain of a new window length also pays first-touch costs (r03: 0.55 ms read once for T=196, B=1)
                eng.sample_loop_(x, xc, 999, 100, noise_mode=_lib.NOISE_PHILOX, seed=1)  # also long enough for the clocks to ramp after the host-side setup
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sample_loop_(x, xc, max(900, a.steps - 1), a.steps, noise_mode=_lib.NOISE_PHILOX, seed=1)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / a.steps
            rec = {"T": T, "B": B, "precision": a.precision, "philox_ms_per_step": round(ms, 4),
                   "window_steps_per_s": round(B / ms * 1e3, 1)}
            if a.api:
                m.num_timesteps = a.steps
                for rng in ("torch", "philox"):
                    m.sampling_rng = rng
                    xg, cg = xs.cuda(), cm.cuda()
                    for _ in range(2):  # the first calls allocate the draw buffer and capture the step graph for these buffers
                        m.sample(xg, cg)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    m.sample(xg, cg)
                    torch.cuda.synchronize()
                    rec[f"sample_{rng}_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / a.steps, 4)
                m.num_timesteps = 1000
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
