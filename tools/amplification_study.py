#!/usr/bin/env python3
"""How the precision gate's two figures move along TRAINING: the chain's amplification of operand rounding (plan.AMPLIFICATION_LIMIT) and the
worst window of a B = 256 batch, for checkpoints of the same recipe stopped after 0 / 100 / 300 / 1000 / 3000 / 10000 Adam steps, for
other learning rates, and for a second data seed (the limits were derived from ONE recipe at 3000 steps).

For every checkpoint: what `auto` picks (no cache), the plan's stage-1 / whole-chain / amplification figures of every candidate it tried,
and — whatever it picked — the worst window of 256 whole 1000-step Philox chains of "9 as is" and of "8 prepared + ffn16" against split-bf16
(the int8 forms at the two ends of the ladder).

    python tools/amplification_study.py [--steps 0,100,300,1000,3000,10000] [--lrs 2e-4] [--seeds 0] [--window 120] [--out profiles/r06_amplification_vs_training.txt]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from egoego_release_amd import ModelConfig, make_weights, plan  # noqa: E402
from chain_tail_b256 import chain_tail  # noqa: E402
from make_trained_like_checkpoint import train_like  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", default="0,100,300,1000,3000,10000")
    ap.add_argument("--lrs", default="2e-4")
    ap.add_argument("--seeds", default="0")
    ap.add_argument("--window", type=int, default=120)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--forms", default="auto,9,8pn")
    ap.add_argument("--data-seeds", default="31337", help="seeds of the caller's batch (synthetic.make_motion_windows + x_T): every (checkpoint, batch) pair is measured afresh, "
                                                          "so 'auto' is what that batch as a FIRST call would get and '9' what a verdict reached elsewhere would do to it")
    ap.add_argument("--probe-windows", type=int, default=0, help="windows of the plan's whole-chain stage (default: plan.CHAIN_WINDOWS)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.probe_windows:
        plan.CHAIN_WINDOWS = args.probe_windows
    T = args.window
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
    log(f"# tools/amplification_study.py --steps {args.steps} --lrs {args.lrs} --seeds {args.seeds} --window {T} --batch {args.batch} --forms {args.forms}")
    log(f"# limits in force: stage 1 {plan.PROBE_LIMIT:.1e}, whole chain {plan.CHAIN_LIMIT:.2e} on {plan.CHAIN_WINDOWS} windows, amplification {plan.AMPLIFICATION_LIMIT:.1f}x; {torch.cuda.get_device_name(0)}")
    rows = []
    for seed in [int(v) for v in args.seeds.split(",")]:
        for lr in [float(v) for v in args.lrs.split(",")]:
            for steps in [int(v) for v in args.steps.split(",")]:
                if steps == 0:
                    sd, info = make_weights(ModelConfig(max_timesteps=T + 1), seed), {"steps": 0, "loss_first": None, "loss_last": None}
                else:
                    sd, info = train_like(steps, seed, "cuda", T, lr=lr)
                    sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
                log(f"== seed {seed}, lr {lr:g}, {steps} Adam steps: l1 {info.get('loss_first')} -> {info.get('loss_last')}")
                for ds in [int(v) for v in args.data_seeds.split(",")]:
                    if "," in args.data_seeds:
                        log(f"-- batch (data seed) {ds}")
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        res = chain_tail(sd, T, args.batch, args.forms.split(","), data_seed=ds, cache=False, log=log)
                    a = res["auto"]
                    amp = {k.replace(", amplification", ""): v for k, v in (a["probe"] or {}).items() if k.endswith("amplification") or k.startswith("chain gain")}
                    rows.append((seed if "," not in args.data_seeds else f"{seed}/batch {ds}", lr, steps, info.get("loss_last"), a["precision"], a["form"], amp,
                                 {f: r["vs3"]["max"] for f, r in res.items()}))
    log("== summary: what auto runs, the amplification its probe measured (per candidate that reached stage 2), the worst of 256 windows against split-bf16 per form")
    for seed, lr, steps, loss, prec, form, amp, worst in rows:
        log(f"   seed {seed} lr {lr:g} steps {steps:6d}  l1 {loss if loss is None else format(loss, '.3f')}  auto -> {prec} {form or ''}  amplification "
            + ", ".join(f"{k}: {v:.2f}x" for k, v in amp.items()) + "  | worst of 256: " + ", ".join(f"{f} {v:.2e}" for f, v in worst.items()))
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
