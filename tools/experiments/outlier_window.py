#!/usr/bin/env python3
"""Where does ONE window's whole-chain error come from?  (round 5: window 63 of the seed-2 trained-like checkpoint ends 9.9e-3 from
split-bf16 in "8 prepared" and "9 prepared + fc24" while every other window of the 256 ends below 8e-4, and its split-bf16 chain is
NOT sensitive: 1.4e-5 under a 1e-6 perturbation of x_T — tools/chain_sensitivity.py.)

For that window alone (Philox keyed by its global index, so the draws are the batch's): the free-running distance between the int8
chain and the split-bf16 chain per step, the TEACHER-FORCED one-step error (both engines stepped from the split-bf16 state), and at the
worst teacher-forced step the stage taps of both engines side by side.

    python tools/experiments/outlier_window.py [--weights seed2] [--window 120] [--index 63] [--form 8p]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import head_condition_mask, _lib  # noqa: E402
from egoego_release_amd.synthetic import make_motion_windows  # noqa: E402
from chain_tail_b256 import build, weights_for  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--weights", default="seed2")
ap.add_argument("--window", type=int, default=120)
ap.add_argument("--index", type=int, default=63)
ap.add_argument("--form", default="8p")
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
warnings.simplefilter("ignore")
T, S, W = a.window, 1000, a.index
sd, info = weights_for(a.weights, T)
data = make_motion_windows(a.batch, T, seed=31337)
mask = head_condition_mask(data.shape)
g = torch.Generator().manual_seed(31338)
x_T = torch.randn(data.shape, generator=g)
x_cond = data * (1 - mask) + mask * torch.randn(data.shape, generator=g)
xT, xc = x_T[W:W + 1].cuda().contiguous(), x_cond[W:W + 1].cuda().contiguous()
m3, m8 = build(sd, T, "3"), build(sd, T, a.form)
e3, e8 = m3.hip_engine(), m8.hip_engine()
print(f"{a.weights} T={T} window {W}: {a.form} -> {m8.hip_precision_used} {m8.hip_precision_probe['form']}; probe {m8.hip_precision_probe['errors']}")
x3, x8 = xT.clone(), xT.clone()
free, forced = [], []
for t in range(S - 1, -1, -1):
    y = x3.clone()  # teacher forcing: the int8 engine stepped from the split-bf16 state
    e8.sample_loop_(y, xc, t, 1, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
    e3.sample_loop_(x3, xc, t, 1, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
    e8.sample_loop_(x8, xc, t, 1, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
    forced.append(float((y - x3).abs().max()))
    free.append(float((x8 - x3).abs().max()))
forced_t = torch.tensor(forced)
print("free-running |x8 - x3|max every 50 steps (t = 999 ..):", " ".join(f"{free[i]:.1e}" for i in range(0, S, 50)), f"final {free[-1]:.2e}")
print("teacher-forced one-step error: max %.2e at t=%d; median %.2e; the ten largest: %s" % (
    float(forced_t.max()), S - 1 - int(forced_t.argmax()), float(forced_t.median()),
    ", ".join(f"t={S - 1 - int(i)}: {forced[int(i)]:.1e}" for i in torch.topk(forced_t, 10).indices)))
d = (x8 - x3).abs()[0]
fr, ft = int(d.amax(1).argmax()), int(d.amax(0).argmax())
print(f"final distance: worst frame {fr}, worst feature {ft} ({'joint position' if ft < 66 else 'rot6d'} of joint {ft // 3 if ft < 66 else (ft - 66) // 6}); "
      f"per-frame max: {' '.join(f'{float(v):.0e}' for v in d.amax(1)[::8])}")
# the state of the split-bf16 chain at the worst teacher-forced step, and both engines' stage taps there
tw = S - 1 - int(forced_t.argmax())
x = xT.clone()
if tw < S - 1:
    e3.sample_loop_(x, xc, S - 1, S - 1 - tw, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
tt = torch.full((1,), tw, device="cuda", dtype=torch.long)
print(f"stage taps at t={tw} from the split-bf16 state (|x|max {float(x.abs().max()):.2f}):")
for li in range(4):
    for st in (("embed",) if li == 0 else ()) + ("attn_out", "attn_ln", "ffn_hidden", "out"):
        t3, t8 = e3.debug_stage(x, xc, tt, li, st), e8.debug_stage(x, xc, tt, li, st)
        dd = (t3 - t8).abs()
        print(f"  layer {li} {st:10s} |tap|max {float(t3.abs().max()):7.2f}  row max of |diff| {float(dd.amax(-1).max()):.2e} (row {int(dd.amax(-1).argmax())})  "
              f"mean |diff| {float(dd.mean()):.1e}")
y3, y8 = e3.denoise(x, xc, tt), e8.denoise(x, xc, tt)
print(f"  x0 prediction: |y|max {float(y3.abs().max()):.2f}, |diff|max {float((y3 - y8).abs().max()):.2e}, clamped |diff|max {float((y3.clamp(-1, 1) - y8.clamp(-1, 1)).abs().max()):.2e}")

# ---- is the growth over the last steps the CHAIN's (an expanding direction of the map x_t -> x_{t-1} near t = 0 that the accumulated difference
# happens to point along) or the int8 engine's?  From the states at t = TS: continue the int8 state in SPLIT-BF16; continue the split-bf16 state
# moved by a fraction of the accumulated difference, and by a random perturbation of the same size, in split-bf16.
TS = 50
s3, s8 = xT.clone(), xT.clone()
e3.sample_loop_(s3, xc, S - 1, S - TS, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
e8.sample_loop_(s8, xc, S - 1, S - TS, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
diff = s8 - s3


def tail3(state):
    y = state.clone()
    e3.sample_loop_(y, xc, TS - 1, TS, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
    return y


def tail8(state):
    y = state.clone()
    e8.sample_loop_(y, xc, TS - 1, TS, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=W)
    return y


base = tail3(s3)
d0 = float(diff.abs().max())
print(f"at t={TS}: |x8 - x3|max = {d0:.2e}")
print(f"  int8 state continued in split-bf16      : final distance {float((tail3(s8) - base).abs().max()):.2e}   (int8 continued in int8: {float((tail8(s8) - base).abs().max()):.2e})")
print(f"  split-bf16 state continued in int8      : final distance {float((tail8(s3) - base).abs().max()):.2e}")
for frac in (1.0, 0.1, 0.01):
    print(f"  split-bf16 state + {frac:4.2f} x the difference : final distance {float((tail3(s3 + frac * diff) - base).abs().max()):.2e}  (x{float((tail3(s3 + frac * diff) - base).abs().max()) / (frac * d0):.1f})")
rnd = torch.randn(diff.shape, generator=torch.Generator().manual_seed(3)).cuda()
rnd = rnd * (diff.norm() / rnd.norm())
print(f"  split-bf16 state + a random move of the same norm (|.|max {float(rnd.abs().max()):.1e}): final distance {float((tail3(s3 + rnd) - base).abs().max()):.2e}")
fr_d = diff.abs()[0].amax(1)
print("  per-frame |difference| at t=%d: %s" % (TS, " ".join(f"{float(v):.0e}" for v in fr_d[::8])))
