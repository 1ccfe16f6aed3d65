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
    """sample_fn for `sample_sharded` that runs the HIP path with in-kernel Philox noise.

    ALWAYS Philox: the per-step draws are keyed by (seed; global window index, timestep, frame, feature), which is what makes the
    result independent of the number of ranks.  The reference's torch-RNG draw order (`model.sampling_rng = "torch"`, one
    `randn_like` of the WHOLE batch per step, M:253) cannot be reproduced by a shard that only holds part of the batch; a caller
    who needs the reference's exact draws samples unsharded through `model.sample()`.

    The weights are checksummed against the packed copy once, here (in-place updates made before this call are picked up); the
    returned function then reuses the context without re-checksumming — make a new function after updating weights."""
    from . import _lib

    model.hip_engine(verify=True)

    def fn(xs, cm, noise, window_offset):
        eng = model.hip_engine()
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        if x.shape[0] == 0:
            return x
        xs, cm = xs.to(dev), cm.to(dev)
        x_cond = (xs * (1.0 - cm) + cm * noise["cond"].to(dev)).float().contiguous()
        S = model.num_timesteps
        eng.sample_loop_(x, x_cond, S - 1, S, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
        model._outlier_guard(eng, x, x_cond)
        return x

    return fn


def hip_steps_fn(model, t_start, n_steps, seed=0):
    """Like hip_sample_fn (Philox draws, weights checksummed once at creation), but `n_steps` ancestral steps from timestep
    `t_start` downwards (a slice of the chain): what bench.py times."""
    from . import _lib

    model.hip_engine(verify=True)

    def fn(xs, cm, noise, window_offset):
        eng = model.hip_engine()
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        if x.shape[0] == 0:
            return x
        x_cond = (xs.to(dev) * (1.0 - cm.to(dev)) + cm.to(dev) * noise["cond"].to(dev)).float().contiguous()
        eng.sample_loop_(x, x_cond, t_start, n_steps, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
        model._outlier_guard(eng, x, x_cond)
        return x

    return fn


# ------------------------------------------------------------------------------------------ sequence-level sharding of the harness
def harness_noise(n_pairs, n_frames, seq_len, seed=0, d_feats=198):
    """The initial draws of the sliding-window harness for ALL (sequence, sample) pairs, from a private seeded CPU generator:
    {'x_all': [n, T, D], 'cond': [per window [n, Tw, D]]} — every rank draws the same and slices its pairs, so x_T and the
    condition noise of a pair do not depend on the sharding (the per-step draws are Philox keyed by the global pair index)."""
    from . import harness
    g = torch.Generator().manual_seed(int(seed) * 104729 + 7)
    x_all = torch.randn((n_pairs, n_frames, d_feats), generator=g)
    cond = [torch.randn((n_pairs, n, d_feats), generator=g) for _, n in harness.window_spans(n_frames, seq_len)]
    return {"x_all": x_all, "cond": cond}


def harness_sharded(model, ds, head_pose, sample_bs=1, seed=0, parents=None, group=None, force_collective=False, harness_fn=None):
    """`Trainer.full_body_gen_cond_head_pose_sliding_window` (trainer_amass_cond_motion_diffusion.py:261-276) over the GPUs of a
    node, sharded by SEQUENCE: head_pose [Bseq, T, 7] (every rank holds all of it: it is small) holds Bseq
    head trajectories, each sampled `sample_bs` times (run_egoego.py:146-148 repeats a sequence's head pose sample_bs times).
    The Bseq * sample_bs (sequence, sample) pairs — sequence-major, like that repeat — are split contiguously over the ranks;
    the windows of one pair never leave its rank (window k + 1 in-paints the tail of window k, M:395-467).  No collective in
    the loop; ONE all_gather of the stitched (axis-angle, root) result ends the call.  Draws: `harness_noise` + Philox keyed by
    the global pair index, so the result is the same for any number of ranks.
    Returns (local axis-angle [Bseq * sample_bs, T', 22, 3], root [Bseq * sample_bs, T', 3]) on every rank.
    harness_fn(head_pose_slice, noise_slice, pair_offset) -> (aa, root): stand-in for tests; default = the HIP harness."""
    from . import harness
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    pairs = head_pose.repeat_interleave(sample_bs, dim=0)
    n, n_frames = pairs.shape[0], pairs.shape[1]
    seq_len = model.seq_len
    noise = harness_noise(n, n_frames, seq_len, seed)
    lo, hi = shard_bounds(n, rank, world)
    if harness_fn is None:
        dev = model.betas.device

        def harness_fn(hp, nz, off):
            return harness.full_body_gen_cond_head_pose_sliding_window(model, ds, hp.to(dev), noise=nz, parents=parents, window_offset=off)
    t_out = harness.output_frames(n_frames, seq_len)
    if hi > lo:
        nz = {"x_all": noise["x_all"][lo:hi], "cond": [c[lo:hi] for c in noise["cond"]]}
        aa, root = harness_fn(pairs[lo:hi], nz, lo)
        local = torch.cat((aa.reshape(hi - lo, t_out, 66), root.reshape(hi - lo, t_out, 3)), dim=-1).float().contiguous()
    else:  # more ranks than pairs: this rank still joins the collective, with an empty slice on its device
        local = torch.zeros((0, t_out, 69), device=model.betas.device, dtype=torch.float32)
    if dist.is_initialized():
        local = gather_windows(local, n, group, force=force_collective)
    return local[..., :66].reshape(n, t_out, 22, 3), local[..., 66:]
