"""N > 1 path on CPU: two gloo ranks shard the window axis, 'sample' their slice with a stand-in
sampler, and all_gather; the result must equal the single-process result (shard invariance)."""
import os
import socket

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
    for B in (8, 7):  # even and ragged shards
        out = str(tmp_path / f"r{B}.pt")
        mp.spawn(_worker, args=(2, port, B, out), nprocs=2, join=True)
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(B, 6, 10, generator=g)
        cm = (torch.rand(B, 6, 10, generator=g) > 0.5).float()
        noise = {"x_T": torch.randn(B, 6, 10, generator=g), "cond": torch.randn(B, 6, 10, generator=g)}
        want = _fake_sampler(xs, cm, noise, 0)
        assert torch.equal(torch.load(out), want)
