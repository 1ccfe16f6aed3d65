"""Window-sharded sampling over the GPUs of one node.

Windows are independent for the whole chain, so rank r samples a contiguous slice of
the batch with no collective in the loop; ONE all_gather of the final poses (RCCL over xGMI when the
backend is "nccl") ends the call.  Per-step noise comes from the in-kernel Philox stream keyed by the
GLOBAL window index, so the result does not depend on the number of ranks.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_windows, rank, world):
    """Contiguous, balanced [lo, hi) slice of the window axis for `rank`."""
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_windows(local, n_windows, group=None, force=False):
    """all_gather the per-rank [b_r, T, D] slices back into [n_windows, T, D] on every rank.
    force=True runs the collective even with one rank (exercises RCCL init + all_gather on a 1-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    sizes = [shard_bounds(n_windows, r, world) for r in range(world)]
    pad = max(1, max(hi - lo for lo, hi in sizes))
    buf = local
    if local.shape[0] < pad:  # uneven shards: pad to the largest so all_gather sees equal shapes
        buf = torch.cat((local, local.new_zeros(pad - local.shape[0], *local.shape[1:])), 0)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf.contiguous(), group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], 0)


def sample_local(sample_fn, x_start, cond_mask, init_noise, group=None):
    """This rank's part of `sample_sharded`: run `sample_fn` on its slice of the global batch (no collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(x_start.shape[0], rank, world)
    sl = slice(lo, hi)
    # fewer windows than ranks: a rank with an empty slice still calls sample_fn (which returns an empty tensor on
    # ITS device and dtype) so that what it hands to the collective matches the other ranks'
    return sample_fn(x_start[sl], cond_mask[sl], {k: v[sl] for k, v in init_noise.items()}, lo)


def sample_sharded(sample_fn, x_start, cond_mask, init_noise, group=None, force_collective=False):
    """Run `sample_fn(x_start_slice, cond_mask_slice, init_noise_slice, window_offset)` on this rank's
    slice of the global batch and gather the result.

    x_start, cond_mask: the GLOBAL [B, T, D] tensors (every rank holds them; they are small).
    init_noise: dict with global 'x_T' and 'cond' draws [B, T, D] so that x_T and the condition noise do
    not depend on the sharding either.  sample_fn must accept an empty slice (a rank with no windows) and return an
    empty [0, T, D] tensor on the device the other ranks return theirs on.
    """
    local = sample_local(sample_fn, x_start, cond_mask, init_noise, group)
    if not dist.is_initialized():
        return local
    return gather_windows(local, x_start.shape[0], group, force=force_collective)


def hip_sample_fn(model, seed=0):
    """sample_fn for `sample_sharded` that runs the HIP path with in-kernel Philox noise."""
    from . import _lib

    def fn(xs, cm, noise, window_offset):
        eng = model.hip_engine(verify=True)  # chain-level entry point: in-place weight updates are picked up
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        if x.shape[0] == 0:
            return x
        xs, cm = xs.to(dev), cm.to(dev)
        x_cond = (xs * (1.0 - cm) + cm * noise["cond"].to(dev)).float().contiguous()
        S = model.num_timesteps
        eng.sample_loop_(x, x_cond, S - 1, S, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
        return x

    return fn


def hip_steps_fn(model, t_start, n_steps, seed=0):
    """Like hip_sample_fn, but `n_steps` ancestral steps from timestep `t_start` downwards (a slice of the chain):
    what bench.py times."""
    from . import _lib

    def fn(xs, cm, noise, window_offset):
        eng = model.hip_engine(verify=True)
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        if x.shape[0] == 0:
            return x
        x_cond = (xs.to(dev) * (1.0 - cm.to(dev)) + cm.to(dev) * noise["cond"].to(dev)).float().contiguous()
        eng.sample_loop_(x, x_cond, t_start, n_steps, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
        return x

    return fn
