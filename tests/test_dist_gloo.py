"""N > 1 path on CPU: two gloo ranks shard the window axis, 'sample' their slice with a stand-in
sampler, and all_gather; the result must equal the single-process result (shard invariance)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egoego_release_amd import dist as D


def test_shard_bounds_cover_everything():
    for n in (1, 2, 7, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _fake_sampler(xs, cm, noise, window_offset):
    # depends on the GLOBAL window index only, like the Philox-keyed HIP sampler
    idx = torch.arange(window_offset, window_offset + xs.shape[0], dtype=torch.float32)[:, None, None]
    return noise["x_T"] * 0.5 + xs * (1 - cm) + idx


def _worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(B, 6, 10, generator=g)
    cm = (torch.rand(B, 6, 10, generator=g) > 0.5).float()
    noise = {"x_T": torch.randn(B, 6, 10, generator=g), "cond": torch.randn(B, 6, 10, generator=g)}
    res = D.sample_sharded(_fake_sampler, xs, cm, noise)
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for B in (8, 7, 1):  # even and ragged shards, and fewer windows than ranks (one rank samples nothing)
        out = str(tmp_path / f"r{B}.pt")
        mp.spawn(_worker, args=(2, port, B, out), nprocs=2, join=True)
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(B, 6, 10, generator=g)
        cm = (torch.rand(B, 6, 10, generator=g) > 0.5).float()
        noise = {"x_T": torch.randn(B, 6, 10, generator=g), "cond": torch.randn(B, 6, 10, generator=g)}
        want = _fake_sampler(xs, cm, noise, 0)
        assert torch.equal(torch.load(out), want)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_one_rank_bit_for_bit(tmp_path):
    """The N > 1 path of bench.py on real kernels: two gloo ranks share the one GPU of the test box, each samples its
    half of a 64-window batch through dist.sample_sharded / hip_steps_fn (in-kernel Philox keyed by the GLOBAL window
    index) and the gathered poses must equal the single-rank run bit for bit."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EGOEGO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "6", "--warmup", "2", "--batch", "64", "--no-cpu-baseline"]
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dump", one] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dump", two]
                        + common, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["windows_per_gpu"] == 32
    a, b = torch.load(one), torch.load(two)
    assert a.shape == (64, 120, 198) and torch.equal(a, b)
