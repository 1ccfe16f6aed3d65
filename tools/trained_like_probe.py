#!/usr/bin/env python3
"""What the operand precisions lose on a trained-like checkpoint (tools/make_trained_like_checkpoint.py), next to the statistics
the precision guards look at.  GPU only; the comparison is the module's plain-PyTorch fp32 forward on the GPU (a perf-debug
tool: the parity tests against the CPU oracle live in tests/test_gpu_trained_like.py).

    python tools/trained_like_probe.py [--steps 2000] [--lr 1e-3] [--ckpt file.pt]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import ModelConfig, head_condition_mask, _lib  # noqa: E402
from egoego_release_amd.model import CondGaussianDiffusion  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402


def crest_report(model, x_all, t):
    """max over rows of |row|max / rms(row) at every tensor the int8 precisions quantise with one scale per row."""
    rep = {}

    def hook(name):
        def f(mod, inp, out):
            v = out.detach().float()
            if v.dim() == 3 and v.shape[1] == 512 and v.shape[2] != 512:  # Conv1d output [B, C, L]
                v = v.transpose(1, 2)
            rep[name] = float((v.abs().amax(-1) / v.pow(2).mean(-1).sqrt().clamp_min(1e-20)).max())
        return f

    hs = []
    tr = model.denoise_fn.motion_transformer
    for i, layer in enumerate(tr.layer_stack):
        hs.append(layer.self_attn.layer_norm.register_forward_hook(hook(f"L{i}.ln1")))
        hs.append(layer.pos_ffn.layer_norm.register_forward_hook(hook(f"L{i}.ln2")))
    with torch.no_grad():
        model.denoise_fn(x_all, t)
    for h in hs:
        h.remove()
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--chain", type=int, default=50)
    ap.add_argument("--window", type=int, default=120)
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    T, B = args.window, 4
    cfg = ModelConfig(max_timesteps=T + 1)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu")["model"]
        info = {"ckpt": args.ckpt}
    else:
        from make_trained_like_checkpoint import train_like
        sd, info = train_like(args.steps, args.seed, "cuda", T, lr=args.lr)
    print(info, flush=True)
    dev = torch.device("cuda")
    data = make_motion_windows(B, T, seed=991, device=dev)
    mask = head_condition_mask(data.shape, device=dev)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(data.shape, generator=g).to(dev)
    xc = data * (1 - mask) + mask * torch.randn(data.shape, generator=g).to(dev)
    models = {}
    for prec in ("auto", _lib.PREC_BF16X3, _lib.PREC_I8X3, _lib.PREC_I8X3_FC):
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(sd, strict=False)
        m.hip_precision = prec
        if prec != "auto":
            m.hip_int8_prep = "never"
        m = m.to(dev)
        m.denoise_fn.eval()
        models[prec] = m
    ref = models[_lib.PREC_BF16X3]
    for tv in (0, 500, 999):
        t = torch.full((B,), tv, device=dev, dtype=torch.long)
        x = ref.q_sample(data, t, eps)
        with torch.no_grad():
            want = ref.denoise_fn(torch.cat((x, xc), -1), t)
        row = {p: float((models[p].denoise(x, t, xc) - want).abs().max()) for p in models}
        print(f"t={tv:3d}: |y|max {float(want.abs().max()):.2f}  forward error vs fp32 torch  " +
              "  ".join(f"{p}: {e:.2e}" for p, e in row.items()) + f"   crest {crest_report(ref, torch.cat((x, xc), -1), t)}", flush=True)
    print("auto picked", models["auto"].hip_precision_used, getattr(models["auto"], "hip_precision_probe", None))
    S = args.chain
    nz = {"x_T": torch.randn(data.shape, generator=g), "cond": torch.randn(data.shape, generator=g),
          "steps": torch.randn(S, *data.shape, generator=g)}
    outs = {}
    for p, m in models.items():
        m.num_timesteps = S
        outs[p] = m.sample(data, mask, noise=nz)
    for p in outs:
        print(f"{S}-step chain, precision {p}: max|x - bf16x3| = {float((outs[p] - outs[_lib.PREC_BF16X3]).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
