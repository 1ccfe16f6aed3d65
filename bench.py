#!/usr/bin/env python3
"""Headline benchmark: stage-2 diffusion sampling steps/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 10
    python bench.py --gpus N ...            (starts its own ranks: a child `python -m torch.distributed.run`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one p_sample (denoiser forward + DDPM posterior update, in-kernel Philox noise) applied to
the GLOBAL batch of B=256 windows of T=120 frames x 198 features with inputs resident in HBM.  With N GPUs
the batch is split (egoego_release_amd.dist: contiguous window shards, no data-path collective; STRONG
scaling) and one RCCL all_gather of the poses to every rank closes the timed region.
Rank 0 prints ONE JSON line.

Environment: EGOEGO_DIST_BACKEND=gloo lets several ranks share one GPU (tests on a 1-GPU box; default "nccl" = RCCL);
EGOEGO_FORCE_COLLECTIVE=1 initialises the process group and runs the all_gather even with ONE rank, so that RCCL
init + the collective execute on a 1-GPU box.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA: twice the bf16 rate (same guide; its micro-benchmark ceiling is >= 3944)


def flops_per_window_step(T, d_feats=198, d_model=512, n_head=4, d_k=256, n_layers=4):
    """Algorithmic FLOPs (2/MAC; softmax/LN/elementwise excluded) — BASELINE.md §3."""
    L, HD = T + 1, n_head * d_k
    layer = 2 * L * d_model * 3 * HD + 4 * L * L * HD + 2 * L * HD * d_model + 4 * L * d_model * d_model
    return 2 * T * 2 * d_feats * d_model + n_layers * layer + 2 * T * d_model * d_feats + 2 * (64 * 256 + 256 * d_model)


def qkv_attn_flops_per_launch(B, T, d_model=512, n_head=4, d_k=256):
    """Algorithmic FLOPs of one launch of the attention-layer kernel: Q/K/V projections + QK^T + PV (one layer)."""
    L, HD = T + 1, n_head * d_k
    return B * (2 * L * d_model * 3 * HD + 4 * L * L * HD)


def tail_flops_per_launch(B, T, d_model=512, n_head=4, d_k=256):
    """Algorithmic FLOPs of one launch of the layer-tail kernel: fc + FFN-1 + FFN-2 (one layer)."""
    L, HD = T + 1, n_head * d_k
    return B * L * (2 * HD * d_model + 4 * d_model * d_model)


def cpu_baseline(cfg, sd, B, T, budget_s=25.0):
    """Time the CPU oracle (fp32 PyTorch restatement of the reference, bit-identical to it) on the
    host cores: whole-batch p_sample steps until ~budget_s of work."""
    import torch
    from oracle import egoego_oracle as O
    from egoego_release_amd import make_head_windows
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sched = O.make_schedule(cfg.timesteps)
    g = torch.Generator().manual_seed(5)
    xs, cm = make_head_windows(B, T, seed=3)
    x = torch.randn(xs.shape, generator=g)
    xc = xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)
    with torch.no_grad():
        # pick the thread count that is fastest on a small slice (all-cores oversubscription is often slower)
        wb = min(B, 16)
        best, best_t = None, None
        for nt in sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail}):
            torch.set_num_threads(nt)
            O.p_sample(sd, sched, x[:wb], torch.full((wb,), 999), xc[:wb], x[:wb])
            t1 = time.perf_counter()
            O.p_sample(sd, sched, x[:wb], torch.full((wb,), 999), xc[:wb], x[:wb])
            dt = time.perf_counter() - t1
            if best_t is None or dt < best_t:
                best, best_t = nt, dt
        torch.set_num_threads(best)
        n, t0 = 0, time.perf_counter()
        while True:
            x = O.p_sample(sd, sched, x, torch.full((B,), 999 - n), xc, torch.randn(x.shape, generator=g))
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 50 or el + el / n > 1.5 * budget_s:
                break
    return {"value": n / el, "unit": "diffusion-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} whole-batch p_sample steps of the CPU oracle at B={B},T={T} (fp32, torch {torch.__version__}; "
                      f"thread count picked from a B={wb} probe, {avail} cpus visible)",
            "ms_per_step": 1e3 * el / n}


def kernel_launch_times(eng, x_l, xc_l, t_start, n_steps, lo):
    """Average launch duration (us) of the attention-layer and layer-tail launch sites over `n_steps` extra steps, by HIP event pairs around
    every launch of that site on the launch stream (egoego_profile_begin / _end): {"qkv": (us, launches), "fc_ln": (us, launches)}."""
    import torch
    from egoego_release_amd import _lib
    kern = {"qkv": (0.0, 0), "fc_ln": (0.0, 0)}
    for name in (("qkv", "fc_ln") if x_l.shape[0] else ()):  # (a rank can be empty when there are fewer windows than ranks)
        eng.profile_begin(name)
        eng.sample_loop_(x_l, xc_l, t_start, n_steps, noise_mode=_lib.NOISE_PHILOX, seed=7, window_offset=lo)
        torch.cuda.synchronize()
        kern[name] = eng.profile_end()
    return kern


def load_traffic(Bl, T, prec, weights):
    """HBM bytes per launch per kernel from the separate rocprofv3 --pmc passes of this command summarised under profiles/ (newest round
    first; `rNN_traffic.json` = precision 9, `rNN_traffic_p3.json` = split-bf16) -> ({kernel: {...}}, source) or ({}, None)."""
    import glob
    try:
        for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
            with open(tp) as f:
                tj = json.load(f)
            if (Bl, T, prec) == (tj.get("batch"), tj.get("window"), tj.get("precision")) and tj.get("weights", "synthetic") in (weights, "any"):
                return tj["kernels"], f"profiles/{os.path.basename(tp)} (rocprofv3 --pmc passes of this command; not measured in this run)"
    except Exception:
        pass
    return {}, None


def roofline_objects(eng, prec, Bl, T, kern, ms_step, traffic, traffic_src):
    """(dominant, other): the roofline objects of the attention-layer and the layer-tail kernel — algorithmic operations per launch over
    the HIP-event launch time, against the peak of the MFMAs the kernel issues."""
    L = T + 1
    (k_us, k_n), (t_us, t_n) = kern["qkv"], kern["fc_ln"]
    if True:
        # which kernels ran comes from the library (egoego_last_kernel_name: it records what its dispatch picked for this shape and
        # precision during the profiled steps above) — no copy of the dispatch rules here
        attn_full, tail_full = eng.last_kernel("qkv"), eng.last_kernel("fc_ln")
        attn_name, tail_name = attn_full.split("<")[0], tail_full.split("<")[0]
        i8_layer = attn_name in ("attn_layer_i8w_kernel", "attn_layer_i8h_kernel", "attn_proj_i8_kernel", "attn_proj6_i8_kernel")
        i8_long = attn_name == "qkv_i8q_kernel"
        o8 = prec == 9 and L > 64                      # the attention kernels hand O over as int8 rows (2 B per value)
        ATTN_TXT = {
            "attn_layer_i8w_kernel": "Q/K/V projections, softmax and PV of one window x head per 8-wave workgroup, int8 slices; K, V, Q and the probabilities stay in LDS/registers",
            "attn_layer_i8h_kernel": "the one-kernel int8 attention layer as two half-query workgroups per window x head (small grids)",
            "attn_proj_i8_kernel": "Q/K/V projections of a window x head as three workgroups writing int8 images (+ attn_core_s_kernel, not in this figure); projection operations only",
            "attn_proj6_i8_kernel": "Q/K/V projections of a window x head as six workgroups writing int8 images (+ attn_core_s_kernel, not in this figure); projection operations only",
            "qkv_i8q_kernel": "Q/K/V projections on int8 slices, quantised into the int8 operand images of attn_core_i8w_kernel; projection operations only",
            "qkv_i8_kernel": "Q/K/V projections on int8 slices for the split-bf16 attention core; projection operations only",
            "qkv_attn_kernel": "fused Q/K/V projection + attention, split-bf16",
            "qkv_kernel": "Q/K/V projections, split-bf16; projection operations only"}
        proj_only = attn_name not in ("attn_layer_i8w_kernel", "attn_layer_i8h_kernel", "qkv_attn_kernel")
        attn_i8 = i8_layer or i8_long or attn_name == "qkv_i8_kernel"
        attn_peak = PEAK_I8_TOPS if attn_i8 else PEAK_BF16_TFLOPS
        attn_flops = Bl * 2 * L * 512 * 3 * 1024 if proj_only else qkv_attn_flops_per_launch(Bl, T)
        attn_ach = attn_flops / (k_us * 1e-6) / 1e12 if k_n else None
        tail_ach = tail_flops_per_launch(Bl, T) / (t_us * 1e-6) / 1e12 if t_n else None
        # algorithmic HBM bytes of one attention-layer launch: the layer input rows in (int8 slices: 2 B per value; split-bf16: 4),
        # O out (int8 rows with precision 9: 2 B per value; split-bf16: 4) + the three projections' weights once
        attn_bytes = ((2 if attn_i8 else 4) * Bl * L * 512 + (2 if o8 else 4) * Bl * L * 1024 + (2 if attn_i8 else 4) * 3 * 512 * 1024) if not proj_only else None
        core_flops = Bl * 4 * L * L * 1024  # QK^T + PV: what north_star words as the "attention-GEMM roofline"
        attn_roof = {
            "bound": "mfma", "kernel": f"{attn_full} ({ATTN_TXT.get(attn_name, '')})",
            "achieved": attn_ach, "peak": attn_peak, "unit": "TOP/s (int8 MFMA, 2 per MAC)" if attn_i8 else "TFLOP/s",
            "frac": (attn_ach / attn_peak) if attn_ach else None,
            "traffic": (traffic.get(attn_name) or {}).get("hbm_bytes_per_launch"), "traffic_source": traffic_src,
            "algorithmic_bytes": attn_bytes,
            "launch_us": k_us, "launches": k_n, "share_of_step": 4 * k_us / (1e3 * ms_step) if k_n else None,
            "attention_core_share_of_operations": None if proj_only else core_flops / attn_flops,
            "note": "algorithmic operations (one per MAC x 2) over the HIP-event launch time, measured on rank 0's shard right after "
                    "the timed region; three MFMAs are issued per product (two int8 slices / two bf16 planes per operand), so matrix-pipe "
                    "utilisation is 3x this fraction.  The kernel is one launch for projections + QK^T + softmax + PV: the attention "
                    "core alone (QK^T + PV) is `attention_core_share_of_operations` of its operations and has no launch time of its own"}
        fc8 = prec == 9 and L > 64
        TAIL_TXT = {
            "tail_kernel": "fc+residual+LayerNorm -> FFN-1 -> FFN-2+residual+LayerNorm per 32/64 tokens: weights streamed into registers, activations by "
                           "LDS-DMA chunks (precision 9: all three contractions on int8 slices, one integer chain per head in fc, LayerNorm-1 rows and hidden rows resident in LDS)",
            "layer_tail_i8_kernel": "the same three GEMMs per 64 tokens, two workgroups per CU, LDS-ring operands; fc split-bf16, FFN on int8 slices in two passes into one int32 accumulator",
            "layer_tail_kernel": "the same three GEMMs per 64 tokens, two workgroups per CU, LDS-ring operands, split-bf16",
            "layer_tail_kernel:128": "the same three GEMMs per 128 tokens, one eight-wave workgroup per CU, LDS-ring operands, split-bf16",
            "gemm_kernel:EpiResLN": "fc + residual + LayerNorm alone (unfused small-batch form; FFN-1 / FFN-2 are separate launches not in this figure)"}
        # the peak of the MFMAs the kernel issues: all int8 (precision 9), fc bf16 + FFN int8 (precision 8: the two halves of its
        # FLOPs at 2.5 and 5 P, i.e. 3333 T together), all bf16 (precisions 3, 1)
        tail_peak = PEAK_I8_TOPS if fc8 else (2.0 / (1.0 / PEAK_BF16_TFLOPS + 1.0 / PEAK_I8_TOPS) if prec == 8 else PEAK_BF16_TFLOPS)
        tail_unit = "TOP/s (int8 MFMA, 2 per MAC)" if fc8 else ("TFLOP/s (fc on bf16 MFMAs, FFN on int8 MFMAs)" if prec == 8 else "TFLOP/s")
        tail_roof = {
            "bound": "mfma", "kernel": f"{tail_full} ({TAIL_TXT.get(tail_name, '')})",
            "achieved": tail_ach, "peak": tail_peak, "unit": tail_unit, "frac": (tail_ach / tail_peak) if tail_ach else None,
            "traffic": (traffic.get(tail_full) or traffic.get(tail_name) or {}).get("hbm_bytes_per_launch"), "traffic_source": traffic_src,
            # precision 9: the attention output (1024 x 2 B), the residual rows in and the layer's rows out (512 x 2 B each) per token
            # + the int8 weights once; the other precisions: split-bf16 rows (4 B per value) incl. the hidden activations
            "algorithmic_bytes": (2 * Bl * L * (1024 + 512 + 512) + 2.1e6) if fc8 else (4 * Bl * L * (1024 + 512 + 512) + 4.2e6),
            "launch_us": t_us, "launches": t_n, "share_of_step": 4 * t_us / (1e3 * ms_step) if t_n else None,
            "note": "measured like the attention-layer kernel; 3 MFMAs are issued per product (split-bf16: K=16 per MFMA; int8 slices: K=32 per "
                    "MFMA at the same issue time); normalised by the peak of the MFMAs the kernel actually issues"}
    return (attn_roof, tail_roof) if (k_us or 0) >= (t_us or 0) else (tail_roof, attn_roof)


def trained_like_line(args, cfg, dev, B, T, K, W):
    """The same timed region (K graph-replayed steps after W warm-up steps, inputs resident) on a TRAINED-LIKE checkpoint — the
    module's own training loss optimised for --train-steps Adam steps on synthetic motion, on this GPU, right here
    (tools/make_trained_like_checkpoint.py) — in what hip_precision = "auto" picks for it by measurement: what a checkpoint that is
    no longer the initialisation runs at (the headline `value` is the seeded initialisation)."""
    import torch
    from egoego_release_amd import _lib, make_head_windows
    from egoego_release_amd import dist as D
    from egoego_release_amd.model import CondGaussianDiffusion
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_trained_like_checkpoint import train_like
    t_train = time.perf_counter()
    sd, info = train_like(args.train_steps, 0, dev, T)
    t_train = time.perf_counter() - t_train
    sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    model = CondGaussianDiffusion(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=False)
    model.hip_graph = not args.no_graph
    model = model.to(dev)
    import warnings
    xs, cm = make_head_windows(B, T, seed=100)
    gen = torch.Generator().manual_seed(1234)
    noise = {"x_T": torch.randn(xs.shape, generator=gen).to(dev), "cond": torch.randn(xs.shape, generator=gen).to(dev)}
    xs, cm = xs.to(dev), cm.to(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        t_pack = time.perf_counter()
        model.hip_engine(verify=True, job=(B, T, cfg.timesteps), conditions=(xs * (1 - cm) + cm * noise["cond"]).contiguous())
        t_pack = time.perf_counter() - t_pack
    S = cfg.timesteps
    timed_fn = D.hip_steps_fn(model, S - 1 - W, K, seed=7, verify=False, guard=False)
    D.sample_local(D.hip_steps_fn(model, S - 1, max(W, 1), seed=7), xs, cm, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D.sample_local(timed_fn, xs, cm, noise)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    guarded_fn = D.hip_steps_fn(model, S - 1 - W, K, seed=7)  # (the same steps with the checksum and the runtime guard's read-back, like a caller's chain)
    tg = time.perf_counter()
    D.sample_local(guarded_fn, xs, cm, noise)
    torch.cuda.synchronize()
    ms_guarded = 1e3 * (time.perf_counter() - tg) / K
    eng = model.hip_engine()
    prec = int(model.hip_precision_used)
    x_l = noise["x_T"].contiguous().clone()
    xc_l = (xs * (1 - cm) + cm * noise["cond"]).contiguous()
    kern = kernel_launch_times(eng, x_l, xc_l, S - 1 - W, min(K, 20), 0)
    traffic, traffic_src = load_traffic(B, T, prec, "any")
    dominant, other = roofline_objects(eng, prec, B, T, kern, 1e3 * el / K, traffic, traffic_src)
    probe = model.hip_precision_probe or {}
    return {"precision": prec, "form": probe.get("form"), "plan_source": probe.get("source"),
            "ms_per_step": 1e3 * el / K, "steps_per_s": K / el, "steps": K, "warmup": W, "ms_per_step_with_checksum_and_guard": ms_guarded,
            "step_frac_of_bf16_peak": flops_per_window_step(T) * B * K / el / 1e12 / PEAK_BF16_TFLOPS,
            "roofline": dominant, "roofline_second_kernel": other,
            "probe_errors": probe.get("errors"), "probe_limits": {"stage1": probe.get("limit"), "whole_chain": probe.get("chain_limit"),
                                                                  "chain_windows": probe.get("chain_windows")},
            "attention_kernel": eng.last_kernel("qkv"), "tail_kernel": eng.last_kernel("fc_ln"),
            "checkpoint": {k: info[k] for k in ("steps", "seed", "lr", "loss_first", "loss_last", "gain_spread")},
            "train_s": t_train, "pack_s": t_pack, "warnings": [str(w.message)[:200] for w in rec]}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """Run this very command under torch.distributed.run with n ranks in a child process and relay its output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL / cross-process tensors)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="GLOBAL number of windows (split over the GPUs)")
    ap.add_argument("--window", type=int, default=120)
    ap.add_argument("--precision", default="auto", choices=("auto", "1", "3", "8", "9"),
                    help="auto (default) = what the module's default picks for these weights by measuring them (plan.py: the ladder "
                         "9 ... 8 ... 3); 9 = int8-slice attention layer, fc, FFN and linear_out + split-bf16 embed (parity-grade, the "
                         "fastest mode inside the 1e-3 bar), 8 = the same with fc / linear_out on split-bf16 (parity-grade, half the error), "
                         "3 = split-bf16 everywhere (parity-grade), 1 = plain bf16 (NOT parity-grade)")
    ap.add_argument("--weights", default="synthetic", choices=("synthetic", "trained-like"),
                    help="synthetic: the seeded initialisation-distribution weights; trained-like: the same after --train-steps Adam steps of the "
                         "module's own training loss on synthetic motion (tools/make_trained_like_checkpoint.py; trained on this GPU before the bench)")
    ap.add_argument("--train-steps", type=int, default=3000)
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the pack-time precision probe (profiling runs: its small-batch launches would mix into per-kernel averages); "
                         "needs an explicit --precision")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of every step (no hipGraph replay)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trained-like", action="store_true",
                    help="skip the extra `trained_like` object (N=1 only: the same timed region on the trained-like checkpoint, in what 'auto' picks for it)")
    ap.add_argument("--dump", default=None, help="rank 0 saves the gathered poses of the timed call here (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (this parent never touches the
        # GPU and never execs), relay what they print and exit with the launcher's code
        return self_launch(args.gpus)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    backend = os.environ.get("EGOEGO_DIST_BACKEND", "nccl")  # "gloo" lets two ranks share one GPU (1-GPU test of this path)
    force_coll = os.environ.get("EGOEGO_FORCE_COLLECTIVE", "0") == "1"
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, {ndev} visible (EGOEGO_DIST_BACKEND=gloo shares GPUs between ranks)")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or force_coll:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from egoego_release_amd import ModelConfig, make_weights, make_head_windows
    from egoego_release_amd.model import CondGaussianDiffusion
    from egoego_release_amd import _lib
    from egoego_release_amd import dist as D

    B, T = args.batch, args.window
    cfg = ModelConfig(max_timesteps=T + 1)
    if args.weights == "trained-like":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from make_trained_like_checkpoint import train_like
        sd, train_info = train_like(args.train_steps, 0, dev, T)
        sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    else:
        sd, train_info = make_weights(cfg, 0), None
    model = CondGaussianDiffusion(**cfg.ctor_kwargs())
    model.load_state_dict(sd, strict=False)
    model.hip_precision = "auto" if args.precision == "auto" else int(args.precision)
    if args.no_probe:
        if args.precision == "auto":
            raise SystemExit("--no-probe needs an explicit --precision")
        model.hip_probe_at_pack = False
        model.hip_outlier_guard = False
    model.hip_graph = not args.no_graph
    model = model.to(dev)
    # the GLOBAL batch (every rank holds it: 24 MB per tensor at B=256); rank r samples the contiguous slice
    # dist.shard_bounds gives it (strong scaling: BASELINE configs[2] is B=256 split over the GPUs of the node)
    xs, cm = make_head_windows(B, T, seed=100)
    gen = torch.Generator().manual_seed(1234)
    noise = {"x_T": torch.randn(xs.shape, generator=gen).to(dev), "cond": torch.randn(xs.shape, generator=gen).to(dev)}
    xs, cm = xs.to(dev), cm.to(dev)
    # pack: under "auto" the precision is MEASURED here, stage 2 on this job's own conditions (N > 1: group rank 0 measures on its shard's, every rank packs its plan)
    lo0, hi0 = D.shard_bounds(B, rank, world)
    xc0 = (xs[lo0:hi0] * (1 - cm[lo0:hi0]) + cm[lo0:hi0] * noise["cond"][lo0:hi0]).contiguous() if hi0 > lo0 else None
    eng = model.hip_engine(verify=True, group=D._group_of(None), job=(B, T, cfg.timesteps), conditions=xc0)
    prec = int(model.hip_precision_used)  # what runs: the probe's pick under "auto"
    lo, hi = D.shard_bounds(B, rank, world)
    S = cfg.timesteps
    K, W = args.steps, args.warmup
    if K + W > S:
        raise SystemExit(f"steps + warmup must not exceed the {S}-step chain")

    # warm-up: W untimed steps through the very path that is timed (packs the workspace, captures the step graph,
    # and runs the collective once so that RCCL's lazy communicator set-up is not in the timed region)
    # the timed call neither checksums the weights nor reads the runtime guard's monitors back (host round trips): the warm-up call
    # just before it does both (and agrees on ONE plan over the ranks, plan.py); the guard of the timed chain runs after t1
    timed_fn = D.hip_steps_fn(model, S - 1 - W, K, seed=7, verify=False, guard=False)
    D.sample_sharded(D.hip_steps_fn(model, S - 1, max(W, 1), seed=7), xs, cm, noise, force_collective=force_coll)
    prec = int(model.hip_precision_used)  # (under torch.distributed: what group rank 0 resolved)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # K steps of this rank's shard (no collective in the loop) ...
    shard = D.sample_local(timed_fn, xs, cm, noise)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # ... then the ONE all_gather of the path (timed separately as well)
    out = D.gather_windows(shard, B, force=force_coll) if dist else shard
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    model._outlier_guard(*timed_fn.last, group=D._group_of(None))  # the timed chain's LayerNorm monitors (collective verdict), outside the timed region
    gather_ms, sample_ms = 1e3 * (t2 - t1), 1e3 * (t1 - t0)
    if dist:
        tmax = torch.tensor([el, gather_ms, sample_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el, gather_ms, sample_ms = tmax.tolist()
    finite = bool(torch.isfinite(out).all().item()) and tuple(out.shape) == (B, T, cfg.d_feats)
    if args.dump and rank == 0:
        torch.save(out.cpu(), args.dump)

    # the product path as a caller runs it — weights checksummed, plan agreed on, the runtime guard's read-back at the end of the chain —
    # timed next to the bare region (the headline's timed call does none of the three)
    guarded_fn = D.hip_steps_fn(model, S - 1 - W, K, seed=7)
    torch.cuda.synchronize()
    tg = time.perf_counter()
    D.sample_local(guarded_fn, xs, cm, noise)
    torch.cuda.synchronize()
    ms_step_guarded = 1e3 * (time.perf_counter() - tg) / K
    # per-kernel launch durations (HIP events on the launch stream), measured AFTER the timed region on this rank's
    # shard: a few extra steps per kernel with event pairs around every launch of that kernel
    Bl = hi - lo
    eng = model.hip_engine()  # (the runtime outlier guard may have re-packed in another precision after the timed call)
    guard_demoted = int(model.hip_precision_used) != prec
    x_l = noise["x_T"][lo:hi].contiguous().clone()
    xc_l = (xs[lo:hi] * (1 - cm[lo:hi]) + cm[lo:hi] * noise["cond"][lo:hi]).contiguous()
    kern = kernel_launch_times(eng, x_l, xc_l, S - 1 - W, min(K, 20), lo)
    traffic, traffic_src = load_traffic(Bl, T, int(model.hip_precision_used), args.weights)

    if rank == 0:
        steps_per_s = K / el
        fl_step = flops_per_window_step(T) * B
        ms_step = 1e3 * el / K
        dominant, other = roofline_objects(eng, int(model.hip_precision_used), Bl, T, kern, ms_step, traffic, traffic_src)
        probe = model.hip_precision_probe
        out_json = {
            "metric": f"diffusion-steps/sec (B={B}, T={T}, 22-joint)",
            "value": steps_per_s,
            "unit": "diffusion-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": ms_step,
            # the same K steps as a caller's chain runs them (weights checksummed, plan agreed on, the runtime guard's LayerNorm monitors read
            # back at the end): `value` times the bare region (those three run in the warm-up call before it / after t1)
            "ms_per_step_with_checksum_and_guard": ms_step_guarded,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {8: "i8x3 + bf16x3 (attention layer and FFN: 2 x int8 slices per operand, int32 accumulate; embed, fc, linear_out: split-bf16, fp32 accumulate)",
                      9: "i8x3 + bf16x3 (attention layer, fc, FFN and linear_out: 2 x int8 slices per operand, int32 accumulate; embed: split-bf16, fp32 accumulate)",
                      3: "bf16x3 (split-bf16 MFMA, fp32 accumulate)", 1: "bf16"}[prec],
            "precision": prec,
            "precision_requested": args.precision,
            # what hip_precision = "auto" (the module default) picks for THESE weights, and the measurement it picked it on
            "precision_auto": prec if args.precision == "auto" else None,
            "precision_probe": probe,
            "weights": args.weights if train_info is None else {"kind": args.weights, **train_info},
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: B={B} windows x T={T} frames x 198 feats split over {world} GPU(s) "
                                   f"({Bl} windows on rank 0), 1000-step DDPM chain (steps {S - 1 - W}..{S - W - K} timed), in-kernel "
                                   f"Philox noise keyed by the global window index, {'synthetic seeded' if train_info is None else 'trained-like'} weights",
                       "global_windows": B, "windows_per_gpu": Bl, "window_len": T,
                       "parallelism": f"window-sharded x{world} (dist.sample_sharded), one all_gather of the poses inside the timed region",
                       "hip_graph": not args.no_graph},
            "window_steps_per_s": steps_per_s * B,
            "step_tflops_algorithmic": fl_step * steps_per_s / 1e12,
            "step_frac_of_bf16_peak": fl_step * steps_per_s / world / 1e12 / PEAK_BF16_TFLOPS,
            "output_finite": finite,
            "rccl_ranks": (dist.get_world_size() if dist and backend == "nccl" else 0),
            "collective_backend": (backend if dist else None),
            "gather_ms": gather_ms if dist else None,
            "sample_ms": sample_ms,
            "outlier_row_max": model.hip_outlier_seen,
            "outlier_guard_demoted_after_timed_call": guard_demoted,
            "roofline": dominant,
            "roofline_second_kernel": other,
        }
        if world == 1 and not args.no_trained_like and args.weights == "synthetic" and args.precision == "auto":
            try:
                out_json["trained_like"] = trained_like_line(args, cfg, dev, B, T, K, W)
            except Exception as e:  # the headline stands on its own
                out_json["trained_like"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out_json["cpu_baseline"] = cpu_baseline(cfg, sd, B, T)
                out_json["speedup_vs_cpu_baseline"] = steps_per_s / out_json["cpu_baseline"]["value"]
            except Exception as e:  # the GPU number stands on its own
                out_json["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out_json), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
