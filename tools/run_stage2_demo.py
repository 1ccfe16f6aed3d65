#!/usr/bin/env python3
"""Stage-2 entry point: head-pose trajectory -> full-body SMPL-H motion through the MI355X diffusion sampler.

The counterpart, for the stage this repo accelerates, of the reference's drivers
    eval_stage2.py:58-222 (test_diffusion / full_body_gen_cond_head_pose_sliding_window on ground-truth head poses)
    run_egoego.py:55-192  (the same call on stage-1 head poses, then fk_smpl and the MPJPE of
                           kinpoly/scripts/eval_metrics_imu_rec.py:297-301)
with the reference's `--diffusion_*` flags (eval_stage2.py:395-404).  Stage 1 (HeadNet / GravityNet / DROID-SLAM), the
AMASS / ARES datasets, SMPL-H and the pretrained weights are not part of this repo: every asset is an
input file, and without `--weight` the run uses the seeded synthetic weights.

    python tools/run_stage2_demo.py --head_pose head_qpos.npy --stats cano_min_max_mean_std_data_window_120.p \\
        --rest_offsets rest_offsets.npy [--weight model-10.pt] [--gt_jpos gt.npy] --diffusion_window 120 --out out.npz

  --head_pose     .npy [T,7] or [B,T,7]: xyz + quaternion (w,x,y,z), what trainer.full_body_gen_cond_head_pose_sliding_window
                  takes (trainer_amass_cond_motion_diffusion.py:261-276); or the reference's demo pickle
                  (test_data/ares/demo_ares_data.p: its 'head_qpos')
  --stats         the dataset's min/max statistics pickle (global_jpos_min / global_jpos_max, amass_diffusion_dataset.py:232-239)
  --rest_offsets  .npy [22,3] rest-pose joint offsets (AMASSDataset.rest_human_offsets; needs the licensed SMPL-H model to
                  produce); --parents optionally overrides the SMPL-H kintree
  --gt_jpos       optional .npy [T,22,3] ground-truth global joints -> MPJPE (mm)
"""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egoego_release_amd import harness, make_weights, ModelConfig  # noqa: E402


def mpjpe_mm(pred_global_jpos, gt_global_jpos):
    """kinpoly/scripts/eval_metrics_imu_rec.py:297-301: root-relative mean per-joint position error in millimetres.
    pred / gt: [T, J, 3] (arrays or tensors)."""
    p = torch.as_tensor(pred_global_jpos, dtype=torch.float64)
    g = torch.as_tensor(gt_global_jpos, dtype=torch.float64)
    p = p - p[:, 0:1]
    g = g - g[:, 0:1]
    return float(torch.linalg.norm(p - g, dim=2).mean() * 1000.0)


def _load_any(path):
    if path.endswith(".npy"):
        return np.load(path)
    try:
        import joblib
        return joblib.load(path)
    except Exception:
        with open(path, "rb") as f:
            return pickle.load(f)


def load_head_pose(path):
    d = _load_any(path)
    if isinstance(d, dict):  # the reference's demo pickle: {0: {'head_qpos': (T,7), ...}}
        d = d[sorted(d.keys())[0]] if "head_qpos" not in d else d
        d = d["head_qpos"]
    hp = torch.as_tensor(np.asarray(d), dtype=torch.float32)
    return hp[None] if hp.dim() == 2 else hp


def parse_opt(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--device", default="0", help="cuda device")
    p.add_argument("--weight", default="", help="reference checkpoint ({'step','model','ema','scaler'}); empty = synthetic weights")
    # the reference's diffusion flags (eval_stage2.py:395-404)
    p.add_argument("--diffusion_window", type=int, default=120, help="horizon")
    p.add_argument("--diffusion_batch_size", type=int, default=1, help="samples drawn per trajectory (sample_bs, run_egoego.py:146)")
    p.add_argument("--diffusion_n_dec_layers", type=int, default=4)
    p.add_argument("--diffusion_n_head", type=int, default=4)
    p.add_argument("--diffusion_d_k", type=int, default=256)
    p.add_argument("--diffusion_d_v", type=int, default=256)
    p.add_argument("--diffusion_d_model", type=int, default=512)
    p.add_argument("--use_min_max", action="store_true", help="accepted for flag compatibility (always on in the shipped configs)")
    p.add_argument("--canonicalize_init_head", action="store_true", help="accepted for flag compatibility (always on)")
    # assets
    p.add_argument("--head_pose", required=True)
    p.add_argument("--stats", required=True)
    p.add_argument("--rest_offsets", required=True)
    p.add_argument("--parents", default="", help="comma-separated 22 parent indices (default: SMPL-H kintree)")
    p.add_argument("--gt_jpos", default="")
    p.add_argument("--timesteps", type=int, default=1000, help="diffusion steps (lower = truncated chain, for smoke runs)")
    p.add_argument("--sampling_rng", default="torch", choices=("torch", "philox"))
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--noise", default="", help="torch-saved dict {'x_all','cond','steps'} of injected draws (harness.py "
                                               "p_sample_loop_sliding_window_w_canonical): reproducibility / parity runs")
    p.add_argument("--gpus", type=int, default=0,
                   help="N >= 1: shard the (trajectory, sample) pairs over N GPUs by SEQUENCE (egoego_release_amd.dist.harness_sharded: "
                        "Philox draws keyed by the global pair index, one all_gather of the result; the output does not depend on N). "
                        "N > 1 starts its own ranks (a child `python -m torch.distributed.run`); EGOEGO_DIST_BACKEND=gloo lets ranks share a GPU. "
                        "0 (default): the single-process path with the reference's RNG order")
    p.add_argument("--out", default="stage2_out.npz")
    return p.parse_args(argv)


def self_launch(n, argv):
    """`--gpus N` outside a launcher: run this very command under torch.distributed.run in a CHILD process (never an exec)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        raise SystemExit(proc.returncode or 1)
    return json.loads(lines[-1])


def main(argv=None):
    opt = parse_opt(argv)
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(opt.gpus, sys.argv[1:] if argv is None else argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("EGOEGO_DIST_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and world > ndev:
            raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, {ndev} visible (EGOEGO_DIST_BACKEND=gloo shares GPUs between ranks)")
        opt.device = str(int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev))
        torch.cuda.set_device(int(opt.device))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(opt.device)))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", int(opt.device))
    stats = _load_any(opt.stats)
    rest = np.load(opt.rest_offsets)
    parents = tuple(int(v) for v in opt.parents.split(",")) if opt.parents else harness.SMPLH_PARENTS_22
    ds = harness.SkeletonStats(stats["global_jpos_min"], stats["global_jpos_max"], rest, parents)

    model_kw = dict(window=opt.diffusion_window, d_model=opt.diffusion_d_model, n_head=opt.diffusion_n_head,
                    n_dec_layers=opt.diffusion_n_dec_layers, d_k=opt.diffusion_d_k, d_v=opt.diffusion_d_v)
    if opt.weight:
        model, info = harness.load_stage2_checkpoint(opt.weight, device=dev, **model_kw)
    else:
        model = harness.build_stage2_model(device=None, **model_kw)
        cfg = ModelConfig(max_timesteps=opt.diffusion_window + 1, d_model=opt.diffusion_d_model, n_head=opt.diffusion_n_head,
                          n_dec_layers=opt.diffusion_n_dec_layers, d_k=opt.diffusion_d_k, d_v=opt.diffusion_d_v)
        model.load_state_dict(make_weights(cfg, 0), strict=False)
        model = model.to(dev)
        info = {"step": None, "missing": [], "unexpected": [], "note": "synthetic seeded weights"}
    model.num_timesteps = opt.timesteps
    model.sampling_rng = opt.sampling_rng
    model.philox_seed = opt.seed

    torch.manual_seed(opt.seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if opt.gpus >= 1:
        # sequence-sharded: every (trajectory, sample) pair stays on one rank (its windows depend on each other), one all_gather at the end
        from egoego_release_amd import dist as D
        head_pose = load_head_pose(opt.head_pose)
        aa, root = D.harness_sharded(model, ds, head_pose, sample_bs=opt.diffusion_batch_size, seed=opt.seed, parents=parents)
        head_pose = head_pose.repeat_interleave(opt.diffusion_batch_size, 0)
    else:
        head_pose = load_head_pose(opt.head_pose).repeat_interleave(opt.diffusion_batch_size, 0).to(dev)
        noise = torch.load(opt.noise, map_location="cpu") if opt.noise else None
        aa, root = harness.full_body_gen_cond_head_pose_sliding_window(model, ds, head_pose, noise=noise)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return None
    # global joints through FK, as run_egoego.py:152-158 does with ds.fk_smpl
    b, t = aa.shape[:2]
    gq, gj = ds.fk_smpl(root.reshape(-1, 3), aa.reshape(-1, 22, 3))
    gj = gj.reshape(b, t, 22, 3)
    out = {"local_aa": aa.cpu().numpy(), "root_trans": root.cpu().numpy(), "global_jpos": gj.cpu().numpy()}
    rep = {"frames": int(t), "samples": int(b), "windows": len(range(0, head_pose.shape[1], opt.diffusion_window - harness.OVERLAP)),
           "diffusion_steps": opt.timesteps, "seconds": round(el, 3), "checkpoint": info, "ranks": world,
           "sharding": "by sequence (dist.harness_sharded)" if opt.gpus >= 1 else None}
    if opt.gt_jpos:
        gt = np.load(opt.gt_jpos)
        rep["mpjpe_mm"] = [mpjpe_mm(gj[i, : gt.shape[0]].cpu(), gt[:t]) for i in range(b)]
    np.savez_compressed(opt.out, **out)
    print(json.dumps(rep), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return rep


if __name__ == "__main__":
    main()
