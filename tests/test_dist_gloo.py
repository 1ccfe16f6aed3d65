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


def _bench(args, env, timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    import json
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` (the shape of the driver's 1-GPU command) must start two ranks by itself — as a child
    torch.distributed.run, never an exec — and hand back their exit code.  Without a GPU each rank stops with the
    'needs an MI355X' message; that message arriving proves the launch path ran."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher")
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], dict(os.environ, EGOEGO_DIST_BACKEND="gloo"), 300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_one_rank_bit_for_bit(tmp_path):
    """The N > 1 path of bench.py on real kernels, started the way the driver starts the 1-GPU run (`python bench.py --gpus 2`,
    no torchrun on the command line): two gloo ranks share the one GPU of the test box, each samples its half of a 64-window
    batch through dist.sample_local / hip_steps_fn (in-kernel Philox keyed by the GLOBAL window index) and the gathered
    poses must equal the single-rank run bit for bit.  Also: fewer windows than ranks (one rank samples nothing)."""
    env = dict(os.environ, EGOEGO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    common = ["--steps", "6", "--warmup", "2", "--batch", "64", "--no-cpu-baseline"]
    one, two, three = str(tmp_path / "one.pt"), str(tmp_path / "two.pt"), str(tmp_path / "three.pt")
    r1 = _bench(["--gpus", "1", "--dump", one] + common, env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _bench(["--gpus", "2", "--dump", two] + common, env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line = _json_line(r2.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["windows_per_gpu"] == 32
    assert line["collective_backend"] == "gloo" and line["gather_ms"] is not None and line["rccl_ranks"] == 0
    a, b = torch.load(one), torch.load(two)
    assert a.shape == (64, 120, 198) and torch.equal(a, b)
    r3 = _bench(["--gpus", "2", "--dump", three, "--steps", "3", "--warmup", "1", "--batch", "1", "--no-cpu-baseline"], env)
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert torch.load(three).shape == (1, 120, 198) and _json_line(r3.stdout)["output_finite"]


@pytest.mark.gpu
def test_rccl_init_and_all_gather_execute_with_one_rank():
    """RCCL on hardware: bench.py under `torchrun --nproc-per-node 1` with backend nccl and EGOEGO_FORCE_COLLECTIVE=1
    initialises the communicator and runs the path's all_gather (world 1) inside the timed region."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EGOEGO_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EGOEGO_DIST_BACKEND", None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--batch", "32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["rccl_ranks"] == 1 and line["collective_backend"] == "nccl" and line["gather_ms"] is not None
    assert line["output_finite"] and line["n_gpus"] == 1
