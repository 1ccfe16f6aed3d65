#!/usr/bin/env python3
"""Which int8-row sites cost a trained-like checkpoint its precision?  (round 4; CPU emulation on the oracle, run on the GPU box
because the checkpoint is trained there: tools/make_trained_like_checkpoint.py.)

The int8-slice precisions store an activation row as 16-bit fixed point with ONE scale per row.  This script emulates exactly that
rounding inside the fp32 oracle, site by site — the LayerNorm outputs (precision 9 keeps them as int8 rows ONLY: the projections, FFN-1
and the residual all read the rounded row), the attention output (one scale per row and head), the ReLU hidden rows, the weights (one
scale per output row) — on a 50-step chain (t = 49..0, B = 4) and prints the max-abs difference of the final poses from the exact
chain, next to each site's worst row crest factor.  It answers whether a per-layer choice of precision would be enough.

    python tools/experiments/int8_site_study.py [--steps 3000] [--chain 50]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from egoego_release_amd import ModelConfig, head_condition_mask  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402
from oracle import egoego_oracle as O  # noqa: E402

QMAX = 32639.0


def quant15(x, dim=-1):
    amax = x.abs().amax(dim=dim, keepdim=True)
    sc = torch.where(amax > 0, amax / QMAX, torch.ones_like(amax))
    return torch.round(x / sc) * sc


class Shim:
    """Stands in for torch.nn.functional inside the oracle.  `sites`: set of (layer, name) with name in
    ln1 / ln2 (LayerNorm outputs, rounded in place), o (attention output, per row and head), hid (ReLU rows), or ('all', name)."""

    def __init__(self, sites, shift=None):
        self.sites = sites
        self.shift = shift
        self.layer = 0
        self.ln_count = 0
        self.crest = {}

    def __getattr__(self, n):
        return getattr(TF, n)

    def on(self, name):
        return (self.layer, name) in self.sites or ("all", name) in self.sites

    def layer_norm(self, x, shape, w, b, eps):
        y = TF.layer_norm(x, shape, w, b, eps)
        name = "ln1" if self.ln_count % 2 == 0 else "ln2"
        self.layer = self.ln_count // 2
        key = (self.layer, name)
        c = float((y.abs().amax(-1) / y.pow(2).mean(-1).sqrt()).max())
        self.crest[key] = max(self.crest.get(key, 0.0), c)
        if self.on(name):
            mu = self.shift.get(key) if self.shift else None
            # (mean shift: the row is stored minus a per-feature constant; every consumer is linear (+ bias) or a residual add next to a
            # per-feature bias, so the constant folds into biases / beta at pack time and costs nothing at run time)
            y = quant15(y) if mu is None else quant15(y - mu) + mu
        self.ln_count += 1
        if name == "ln2":
            self.layer += 1
        return y

    def linear(self, x, w, b=None):
        if x.shape[-1] == 1024 and self.on("o"):  # fc input: one scale per row and head
            shp = x.shape
            x = quant15(x.reshape(shp[:-1] + (4, 256))).reshape(shp)
        return TF.linear(x, w, b)

    def relu(self, x):
        y = TF.relu(x)
        if self.on("hid"):  # [B, C, L]: rows are tokens
            y = quant15(y, dim=1)
        return y

    def begin_forward(self):
        self.layer, self.ln_count = 0, 0

    # the attention core's own int8 images (the kernels quantise Q and K per row and head, V per feature column over the window's keys,
    # the un-normalised probabilities exp(s - rowmax) with the fixed scale 1 / 32639): torch.bmm is patched while a chain with any of these sites runs
    def bmm(self, a, b):
        if a.shape[-1] == 256 and b.shape[-2] == 256:  # Q [HB, L, 256] x K^T [HB, 256, L]
            if self.on("q"):
                a = quant15(a, dim=-1)
            if self.on("k"):
                b = quant15(b, dim=-2)
        elif b.shape[-1] == 256:                        # P [HB, L, L] x V [HB, L, 256]
            if self.on("p"):  # the kernels quantise exp(s - rowmax) in (0, 1] — the row's largest probability is exactly 1 there
                pm = a.amax(-1, keepdim=True)
                a = torch.round(a / pm * QMAX) / QMAX * pm
            if self.on("v"):
                b = quant15(b, dim=-2)
        return self._bmm(a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--chain", type=int, default=50)
    ap.add_argument("--ckpt", default=None)
    args = ap.parse_args()
    T, B, S = 120, 4, args.chain
    cfg = ModelConfig(max_timesteps=T + 1)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu")["model"]
    else:
        from make_trained_like_checkpoint import train_like
        sd, info = train_like(args.steps, 0, None, T)
        print(info, flush=True)
    sd = {k: v.float().cpu() for k, v in sd.items() if k.startswith("denoise_fn.")}
    # weights as int8 slices everywhere (one scale per output row): they are not what differs between the sites
    sdq = {k: (quant15(v.reshape(v.shape[0], -1), 1).reshape(v.shape) if (k.endswith(".weight") and v.dim() >= 2 and "position_vec" not in k
                                                                          and "time_mlp" not in k and "start_conv" not in k) else v) for k, v in sd.items()}
    sched = O.make_schedule(1000)
    data = make_motion_windows(B, T, seed=991)
    mask = head_condition_mask(data.shape)
    g = torch.Generator().manual_seed(5)
    nz = {"x_T": torch.randn(data.shape, generator=g), "cond": torch.randn(data.shape, generator=g), "steps": torch.randn(S, *data.shape, generator=g)}
    x_cond = data * (1 - mask) + mask * nz["cond"]
    torch.set_num_threads(min(32, os.cpu_count() or 8))

    def chain(weights, sites, shift=None):
        shim = Shim(sites, shift)
        O.F = shim
        shim._bmm = torch.bmm
        torch.bmm = shim.bmm
        real_denoise = O.denoise

        x = nz["x_T"].clone()
        with torch.no_grad():
            for i, tv in enumerate(reversed(range(S))):
                shim.begin_forward()
                x = O.p_sample(weights, sched, x, torch.full((B,), tv, dtype=torch.long), x_cond, nz["steps"][i])
        O.F = TF
        torch.bmm = shim._bmm
        return x, shim.crest

    ref, crest = chain(sd, set())
    print("worst row crest per LayerNorm site over the chain:", {f"L{k[0]}.{k[1]}": round(v, 1) for k, v in sorted(crest.items())})

    def report(label, weights, sites, shift=None):
        out, _ = chain(weights, sites, shift)
        print(f"{label:70s} max|d| {float((out - ref).abs().max()):.2e}   mean|d| {float((out - ref).abs().mean()):.2e}", flush=True)

    # ---- which weights?  and: does correcting the biases for the weights' rounding help?
    GROUPS = {"qkv": ("w_q.weight", "w_k.weight", "w_v.weight"), "fc": ("fc.weight",), "w_1": ("w_1.weight",), "w_2": ("w_2.weight",),
              "linear_out": ("linear_out.weight",)}

    def quantised(names, layers=None):
        out = dict(sd)
        for k, v in sd.items():
            if any(k.endswith(n) for n in names) and (layers is None or any(f"layer_stack.{li}." in k for li in layers) or "linear_out" in k):
                out[k] = sdq[k]
        return out
    for gname, names in GROUPS.items():
        report(f"only the {gname} weights as int8 slices", quantised(names), set())
    for li in range(4):
        report(f"only layer {li}'s weights as int8 slices", quantised(sum(GROUPS.values(), ()), [li]) | {"denoise_fn.linear_out.weight": sd["denoise_fn.linear_out.weight"]}, set())
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2 and ("layer_stack.3" in k or "linear_out" in k):
            w = v.reshape(v.shape[0], -1)
            rc = (w.abs().amax(1) / w.pow(2).mean(1).sqrt())
            cc = (w.abs().amax(0) / w.pow(2).mean().sqrt())
            print(f"   {k:70s} row crest median {float(rc.median()):5.1f} max {float(rc.max()):5.1f}   column |max| / rms: median {float(cc.median()):5.1f} max {float(cc.max()):5.1f}   rms {float(w.pow(2).mean().sqrt()):.4f}")

    # ---- error-compensating rounding (GPTQ, Frantar et al. 2022) on the SAME integer grid: column by column, the rounding error of a
    # weight column is pushed onto the not-yet-rounded columns along the inverse Hessian of the layer's inputs (H = X^T X over
    # calibration rows), so that the OUTPUT error (W - W_q) x is minimised on inputs like the calibration rows instead of the weight error
    def gptq(W, X, damp=0.01):
        Wf = W.reshape(W.shape[0], -1).double().clone()
        scale = (Wf.abs().amax(1, keepdim=True) / QMAX).clamp_min(1e-30)
        H = (X.double().T @ X.double()) / X.shape[0]
        H += damp * H.diagonal().mean() * torch.eye(H.shape[0], dtype=torch.float64)
        U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
        Q = torch.zeros_like(Wf)
        for i in range(Wf.shape[1]):
            w = Wf[:, i]
            q = (torch.round(w / scale[:, 0]).clamp(-QMAX, QMAX)) * scale[:, 0]
            Q[:, i] = q
            err = (w - q) / U[i, i]
            if i + 1 < Wf.shape[1]:
                Wf[:, i + 1:] -= err[:, None] * U[i, i + 1:][None, :]
        return Q.float().reshape(W.shape)

    # bias correction (Nagel et al. 2019): b' = b + (W - W_q) mu, mu = the mean input row of that GEMM over calibration forwards
    cal = make_motion_windows(B, T, seed=17)
    gc = torch.Generator().manual_seed(99)
    xc_cal = cal * (1 - mask) + mask * torch.randn(cal.shape, generator=gc)
    sums, count, rows = {}, 0, {}
    for tv in (0, 20, 100, 500, 999):
        xt = sched["sqrt_alphas_cumprod"][tv] * cal + sched["sqrt_one_minus_alphas_cumprod"][tv] * torch.randn(cal.shape, generator=gc)
        taps = {}
        with torch.no_grad():
            O.denoise(sd, torch.cat((xt, xc_cal), -1), torch.full((B,), tv, dtype=torch.long), taps=taps)
        ins = {}
        for li in range(4):
            lt = taps[f"layer{li}"]
            ins[(li, "qkv")] = taps["embed"] if li == 0 else taps[f"layer{li - 1}"]["out"]
            ins[(li, "fc")] = lt["attn_out"]
            ins[(li, "w_1")] = lt["attn_ln"]
            ins[(li, "w_2")] = lt["ffn_hidden"]
        ins[("out", "linear_out")] = taps["layer3"]["out"][:, 1:]
        for key, v in ins.items():
            sums[key] = sums.get(key, 0) + v.reshape(-1, v.shape[-1]).mean(0)
            rows.setdefault(key, []).append(v.reshape(-1, v.shape[-1]))
        count += 1
    mu = {k: v / count for k, v in sums.items()}
    for key in ((3, "qkv"), (3, "w_1"), ("out", "linear_out")):
        m_ = mu[key]
        print(f"   mean input row of {key}: |mean| max {float(m_.abs().max()):.2f}, rms {float(m_.pow(2).mean().sqrt()):.2f}")
    sdc = dict(sdq)
    TRP = "denoise_fn.motion_transformer."
    for li in range(4):
        for nm, grp in (("self_attn.w_q", "qkv"), ("self_attn.w_k", "qkv"), ("self_attn.w_v", "qkv"), ("self_attn.fc", "fc"), ("pos_ffn.w_1", "w_1"), ("pos_ffn.w_2", "w_2")):
            kw, kb = f"{TRP}layer_stack.{li}.{nm}.weight", f"{TRP}layer_stack.{li}.{nm}.bias"
            dw = (sd[kw] - sdq[kw]).reshape(sd[kw].shape[0], -1)
            sdc[kb] = sd[kb] + dw @ mu[(li, grp)]
    dw = sd["denoise_fn.linear_out.weight"] - sdq["denoise_fn.linear_out.weight"]
    sdc["denoise_fn.linear_out.bias"] = sd["denoise_fn.linear_out.bias"] + dw @ mu[("out", "linear_out")]
    report("weights as int8 slices + bias correction", sdc, set())
    sdg = dict(sd)
    for li in range(4):
        for nm, grp in (("self_attn.w_q", "qkv"), ("self_attn.w_k", "qkv"), ("self_attn.w_v", "qkv"), ("self_attn.fc", "fc"), ("pos_ffn.w_1", "w_1"), ("pos_ffn.w_2", "w_2")):
            kw = f"{TRP}layer_stack.{li}.{nm}.weight"
            sdg[kw] = gptq(sd[kw], torch.cat(rows[(li, grp)], 0))
    sdg["denoise_fn.linear_out.weight"] = gptq(sd["denoise_fn.linear_out.weight"], torch.cat(rows[("out", "linear_out")], 0))
    for k in sdg:
        if k.endswith(".weight") and sdg[k] is not sd[k]:
            assert float((sdg[k] - sd[k]).abs().max()) < 20 * float(sd[k].abs().max()) / QMAX, k  # still the same weights, a few steps apart at most
    report("weights on the int8 grid with error-compensating rounding (GPTQ)", sdg, set())
    sdgc = dict(sdg)
    for li in range(4):
        for nm, grp in (("self_attn.w_q", "qkv"), ("self_attn.w_k", "qkv"), ("self_attn.w_v", "qkv"), ("self_attn.fc", "fc"), ("pos_ffn.w_1", "w_1"), ("pos_ffn.w_2", "w_2")):
            kw, kb = f"{TRP}layer_stack.{li}.{nm}.weight", f"{TRP}layer_stack.{li}.{nm}.bias"
            sdgc[kb] = sd[kb] + (sd[kw] - sdg[kw]).reshape(sd[kw].shape[0], -1) @ mu[(li, grp)]
    sdgc["denoise_fn.linear_out.bias"] = sd["denoise_fn.linear_out.bias"] + (sd["denoise_fn.linear_out.weight"] - sdg["denoise_fn.linear_out.weight"]) @ mu[("out", "linear_out")]
    report("GPTQ weights + bias correction", sdgc, set())
    report("GPTQ weights + bias correction + every activation site", sdgc, {("all", "ln1"), ("all", "ln2"), ("all", "o"), ("all", "hid")})
    report("GPTQ weights + every activation site", sdg, {("all", "ln1"), ("all", "ln2"), ("all", "o"), ("all", "hid")})
    shift = {}
    for li in range(4):
        shift[(li, "ln1")] = mu[(li, "w_1")]
        shift[(li, "ln2")] = mu[(li + 1, "qkv")] if li < 3 else torch.cat(rows[("out", "linear_out")], 0).mean(0)
    allsites = {("all", "ln1"), ("all", "ln2"), ("all", "o"), ("all", "hid")}
    report("nearest weights + every activation site, LayerNorm rows mean-shifted", sdq, allsites, shift)
    report("GPTQ weights + every activation site, LayerNorm rows mean-shifted", sdg, allsites, shift)
    report("GPTQ weights + bias correction + every activation site, LayerNorm rows mean-shifted", sdgc, allsites, shift)
    core = {("all", "q"), ("all", "k"), ("all", "v"), ("all", "p")}
    report("GPTQ + shifted LayerNorm rows + every site + the attention core's Q, K, V, P images", sdg, allsites | core, shift)
    for nm in ("q", "k", "v", "p"):
        report(f"GPTQ + shifted LayerNorm rows + every site + only the {nm.upper()} image", sdg, allsites | {("all", nm)}, shift)
    report("exact weights and rows, only the attention core's Q, K, V, P images", sd, core)
    _, cr = chain(sd, set())
    print("LayerNorm row crest after the mean shift:", {f"L{k[0]}.{k[1]}": round(float(((torch.cat(rows[(k[0], 'w_1')] if k[1] == 'ln1' else (rows[(k[0] + 1, 'qkv')] if k[0] < 3 else rows[('out', 'linear_out')]), 0) - v).abs().amax(-1) / (torch.cat(rows[(k[0], 'w_1')] if k[1] == 'ln1' else (rows[(k[0] + 1, 'qkv')] if k[0] < 3 else rows[('out', 'linear_out')]), 0) - v).pow(2).mean(-1).sqrt()).max()), 1) for k, v in shift.items()})
    report("weights + bias correction + every activation site", sdc, {("all", "ln1"), ("all", "ln2"), ("all", "o"), ("all", "hid")})
    report("weights only (int8 slices, one scale per output row)", sdq, set())
    report("weights + every activation site (the precision-9 emulation)", sdq, {("all", "ln1"), ("all", "ln2"), ("all", "o"), ("all", "hid")})
    return
    report("weights + LayerNorm rows only", sdq, {("all", "ln1"), ("all", "ln2")})
    report("weights + attention output + hidden rows only", sdq, {("all", "o"), ("all", "hid")})
    for li in range(4):
        report(f"weights + layer {li}'s LayerNorm rows only", sdq, {(li, "ln1"), (li, "ln2")})
    for li in range(4):
        for nm in ("ln1", "ln2"):
            report(f"weights + L{li}.{nm} only", sdq, {(li, nm)})
    report("weights + everything EXCEPT layer 3's LayerNorm rows", sdq,
           {(li, nm) for li in range(3) for nm in ("ln1", "ln2")} | {("all", "o"), ("all", "hid")})
    report("weights + everything EXCEPT layers 2-3's LayerNorm rows", sdq,
           {(li, nm) for li in range(2) for nm in ("ln1", "ln2")} | {("all", "o"), ("all", "hid")})
    report("weights + everything EXCEPT L2.ln2, L3.ln1, L3.ln2", sdq,
           {(li, nm) for li in range(3) for nm in ("ln1", "ln2")} - {(2, "ln2")} | {("all", "o"), ("all", "hid")})


if __name__ == "__main__":
    main()
