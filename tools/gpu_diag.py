#!/usr/bin/env python3
"""Stage-by-stage diagnostic of the HIP denoiser against the CPU oracle (run on the GPU box)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import ModelConfig, make_weights
from egoego_release_amd.model import CondGaussianDiffusion
from egoego_release_amd import _lib
from oracle import egoego_oracle as O


def main():
    B, T = int(os.environ.get("DIAG_B", 2)), int(os.environ.get("DIAG_T", 120))
    prec = int(os.environ.get("DIAG_PREC", 8))
    cfg = ModelConfig(max_timesteps=T + 1)
    sd = make_weights(cfg, 0)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision = prec
    m = m.cuda()
    eng = m.hip_engine()
    g = torch.Generator().manual_seed(1000 + T)
    x_all = torch.randn(B, T, 396, generator=g)
    x, xc = x_all[..., :198].contiguous(), x_all[..., 198:].contiguous()
    t = torch.tensor([3, 977][:B] if B <= 2 else [500] * B)
    taps = {}
    with torch.no_grad():
        ref = O.denoise(sd, x_all, t, taps=taps)
    xd, xcd, td = x.cuda(), xc.cuda(), t.cuda()
    rep = {}

    def cmp(name, got, want):
        got = got.float().cpu()
        err = (got - want).abs()
        rep[name] = dict(max_abs=err.max().item(), mean_abs=err.mean().item(), ref_absmax=want.abs().max().item(),
                         nan=bool(torch.isnan(got).any()))
        print(f"{name:24s} max_abs={err.max().item():.3e} mean_abs={err.mean().item():.3e} "
              f"ref_absmax={want.abs().max().item():.3f} nan={rep[name]['nan']}", flush=True)
        if err.max().item() > 1e-2:
            idx = torch.nonzero(err == err.max())[0].tolist()
            print("    worst at", idx, "got", got[tuple(idx)].item(), "want", want[tuple(idx)].item())

    H, L = 4, T + 1
    cmp("embed", eng.debug_stage(xd, xcd, td, 0, "embed"), taps["embed"])
    for li in range(4):
        lt = taps[f"layer{li}"]
        q = lt["q"].view(H, B, L, 256).permute(1, 0, 2, 3) / 16.0
        k = lt["k"].view(H, B, L, 256).permute(1, 0, 2, 3)
        v = lt["v"].view(H, B, L, 256).permute(1, 0, 2, 3)
        cmp(f"L{li}.q", eng.debug_stage(xd, xcd, td, li, "q"), q)
        cmp(f"L{li}.k", eng.debug_stage(xd, xcd, td, li, "k"), k)
        cmp(f"L{li}.v", eng.debug_stage(xd, xcd, td, li, "v"), v)
        cmp(f"L{li}.attn_out", eng.debug_stage(xd, xcd, td, li, "attn_out"), lt["attn_out"])
        cmp(f"L{li}.attn_ln", eng.debug_stage(xd, xcd, td, li, "attn_ln"), lt["attn_ln"])
        cmp(f"L{li}.ffn_hidden", eng.debug_stage(xd, xcd, td, li, "ffn_hidden"), lt["ffn_hidden"])
        cmp(f"L{li}.out", eng.debug_stage(xd, xcd, td, li, "out"), lt["out"])
        if li == 0 and os.environ.get("DIAG_L0_ONLY"):
            break
    cmp("denoise", eng.denoise(xd, xcd, td), ref)
    # one posterior step with injected noise
    sched = O.make_schedule(1000)
    noise = torch.randn(B, T, 198, generator=g)
    want = O.p_sample(sd, sched, x, t, xc, noise)
    got = m.p_sample(xd, td, xcd, noise=noise.cuda())
    cmp("p_sample", got, want)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open(f"gpurun_out/diag_B{B}_T{T}_p{prec}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
