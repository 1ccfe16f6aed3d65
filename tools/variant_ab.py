#!/usr/bin/env python3
"""A/B of a variant build (python -m egoego_release_amd.build --variant --tag=X -D...) against the product library: the SAME seeded Philox chains
through both — run this once per library (EGOEGO_PERFDEBUG_TAG=X selects the variant), then `--compare a.pt b.pt` says whether every
tensor is bit-equal — and ms per step over batch sizes / precisions.

    python tools/variant_ab.py --out a.pt ; EGOEGO_PERFDEBUG_TAG=X python tools/variant_ab.py --out b.pt ; python tools/variant_ab.py --compare a.pt b.pt
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2, default=None)
    ap.add_argument("--cases", default="9:120:130,9:120:32,9:120:8,3:120:130,3:120:8,8:120:130,9:196:66,3:196:66,9:30:40")
    ap.add_argument("--time", default="9:120:256,3:120:256,9:120:32,3:120:32,9:196:256,3:196:256")
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    if a.compare:
        x, y = torch.load(a.compare[0]), torch.load(a.compare[1])
        bad = [k for k in x if not torch.equal(x[k], y[k])]
        print(f"{len(x)} tensors compared, {len(bad)} differ" + (": " + ", ".join(f"{k} ({float((x[k] - y[k]).abs().max()):.2e})" for k in bad) if bad else " (bit-equal)"))
        return 1 if bad else 0
    if "EGOEGO_PERFDEBUG_TAG" in os.environ:
        _lib.use_perfdebug_build()
    out = {}
    models = {}

    def model(prec, T):
        if (prec, T) not in models:
            cfg = ModelConfig(max_timesteps=T + 1)
            m = CondGaussianDiffusion(**cfg.ctor_kwargs())
            m.load_state_dict(make_weights(cfg, 0), strict=False)
            m.hip_precision, m.hip_probe_at_pack, m.hip_outlier_guard = prec, False, False
            models[(prec, T)] = m.cuda()
        return models[(prec, T)]
    for case in a.cases.split(","):
        prec, T, B = (int(v) for v in case.split(":"))
        eng = model(prec, T).hip_engine()
        xs, cm = make_head_windows(B, T, seed=B)
        g = torch.Generator().manual_seed(1000 + B)
        x = torch.randn(xs.shape, generator=g).cuda()
        xc = (xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)).cuda()
        eng.sample_loop_(x, xc, 999, 6, noise_mode=_lib.NOISE_PHILOX, seed=5)
        eng.sample_loop_(x, xc, 5, 6, noise_mode=_lib.NOISE_PHILOX, seed=5)
        out[case] = x.cpu()
    for case in a.time.split(","):
        prec, T, B = (int(v) for v in case.split(":"))
        eng = model(prec, T).hip_engine()
        x = torch.randn(B, T, 198, device="cuda")
        xc = torch.randn(B, T, 198, device="cuda")
        eng.sample_loop_(x, xc, 999, 10, noise_mode=_lib.NOISE_PHILOX, seed=1)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sample_loop_(x, xc, 900, a.steps, noise_mode=_lib.NOISE_PHILOX, seed=1)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / a.steps
            best = ms if best is None else min(best, ms)
        kt = {}
        for k in ("embed", "qkv", "attn", "fc_ln", "out"):
            eng.profile_begin(k)
            eng.sample_loop_(x, xc, 900, 10, noise_mode=_lib.NOISE_PHILOX, seed=1)
            torch.cuda.synchronize()
            us, n = eng.profile_end()
            if n:
                kt[k] = round(us, 1)
        print(json.dumps({"lib": os.environ.get("EGOEGO_PERFDEBUG_TAG", "product"), "precision": prec, "T": T, "B": B, "ms_per_step": round(best, 4), "launch_us": kt}), flush=True)
    if a.out:
        torch.save(out, a.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
