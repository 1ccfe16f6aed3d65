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


def gather_windows(local, n_windows, group=None):
    """all_gather the per-rank [b_r, T, D] slices back into [n_windows, T, D] on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(n_windows, r, world) for r in range(world)]
    pad = max(1, max(hi - lo for lo, hi in sizes))
    buf = local
    if local.shape[0] < pad:  # uneven shards: pad to the largest so all_gather sees equal shapes
        buf = torch.cat((local, local.new_zeros(pad - local.shape[0], *local.shape[1:])), 0)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf.contiguous(), group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], 0)


def sample_sharded(sample_fn, x_start, cond_mask, init_noise, group=None):
    """Run `sample_fn(x_start_slice, cond_mask_slice, init_noise_slice, window_offset)` on this rank's
    slice of the global batch and gather the result.

    x_start, cond_mask: the GLOBAL [B, T, D] tensors (every rank holds them; they are small).
    init_noise: dict with global 'x_T' and 'cond' draws [B, T, D] so that x_T and the condition noise do
    not depend on the sharding either.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x_start.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    sl = slice(lo, hi)
    if hi > lo:
        local = sample_fn(x_start[sl], cond_mask[sl], {k: v[sl] for k, v in init_noise.items()}, lo)
    else:
        # fewer windows than ranks: this rank has nothing to sample but must still enter the collective
        dev = next(iter(init_noise.values())).device if init_noise else x_start.device
        local = torch.empty((0,) + tuple(x_start.shape[1:]), dtype=torch.float32, device=dev)
    return gather_windows(local, B, group) if world > 1 else local


def hip_sample_fn(model, seed=0):
    """sample_fn for `sample_sharded` that runs the HIP path with in-kernel Philox noise."""
    from . import _lib

    def fn(xs, cm, noise, window_offset):
        eng = model.hip_engine()
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
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
        eng = model.hip_engine()
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        x_cond = (xs.to(dev) * (1.0 - cm.to(dev)) + cm.to(dev) * noise["cond"].to(dev)).float().contiguous()
        eng.sample_loop_(x, x_cond, t_start, n_steps, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
        return x

    return fn
