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
    """all_gather the per-rank [b_r, T, D] slices back into [n_windows, T, D] on every rank: ONE collective
    (`all_gather_into_tensor`) into one preallocated [world * pad, T, D] buffer, pad = the largest shard; with even shards the
    buffer IS the result (no copy), ragged ones are compacted once.
    force=True runs the collective even with one rank (exercises RCCL init + all_gather on a 1-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    sizes = [shard_bounds(n_windows, r, world) for r in range(world)]
    pad = max(1, max(hi - lo for lo, hi in sizes))
    send = local.contiguous()
    if send.shape[0] < pad:  # ragged shards: the collective needs equal shapes
        send = torch.cat((send, send.new_zeros(pad - send.shape[0], *send.shape[1:])), 0)
    out = torch.empty((world * pad,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(hi - lo == pad for lo, hi in sizes):
        return out
    return torch.cat([out[r * pad: r * pad + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def sample_local(sample_fn, x_start, cond_mask, init_noise, group=None):
    """This rank's part of `sample_sharded`: run `sample_fn` on its slice of the global batch (no collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(x_start.shape[0], rank, world)
    sl = slice(lo, hi)
    # fewer windows than ranks: a rank with an empty slice still calls sample_fn (which returns an empty tensor on
    # ITS device and dtype) so that what it hands to the collective matches the other ranks'
    args = (x_start[sl], cond_mask[sl], {k: v[sl] for k, v in init_noise.items()}, lo)
    if getattr(sample_fn, "wants_global", False):  # (the HIP sample functions size their plan by the GLOBAL job)
        return sample_fn(*args, global_windows=x_start.shape[0])
    return sample_fn(*args)


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


def _group_of(group):
    """The group whose ranks must agree on one plan: `group`, or the default group once torch.distributed is initialised."""
    if not dist.is_initialized():
        return None
    return group if group is not None else dist.group.WORLD


def hip_steps_fn(model, t_start, n_steps, seed=0, group=None, verify=True, guard=True):
    """sample_fn for `sample_sharded`: `n_steps` ancestral steps from timestep `t_start` downwards on the HIP path with in-kernel
    Philox noise (hip_sample_fn = the whole chain; bench.py times a slice of it).

    ALWAYS Philox: the per-step draws are keyed by (seed; global window index, timestep, frame, feature), which is what makes the
    result independent of the number of ranks.  The reference's torch-RNG draw order (`model.sampling_rng = "torch"`, one
    `randn_like` of the WHOLE batch per step, M:253) cannot be reproduced by a shard that only holds part of the batch; a caller
    who needs the reference's exact draws samples unsharded through `model.sample()`.

    ONE plan for all ranks (plan.py): every call checksums the weights (in-place updates between calls are picked up; `verify=False`
    skips that — bench.py's timed call, whose weights were checked just before), agrees over the group on whether to re-pack, and
    packs what group rank 0 resolved; the runtime guard's verdict at the end of the chain is collective too (`guard=False`: the
    caller runs `model._outlier_guard(eng, x, x_cond, group)` itself, outside its timed region).  Every rank of the group must make
    the same calls — with an empty slice too (fewer windows than ranks)."""
    from . import _lib

    grp = _group_of(group)

    def fn(xs, cm, noise, window_offset, global_windows=None):
        dev = model.betas.device
        x = noise["x_T"].to(dev, torch.float32).contiguous().clone()
        job = (int(global_windows if global_windows is not None else x.shape[0]), int(x.shape[1]), int(n_steps))
        x_cond = x
        if x.shape[0]:
            xs, cm = xs.to(dev), cm.to(dev)
            x_cond = (xs * (1.0 - cm) + cm * noise["cond"].to(dev)).float().contiguous()
        # (this shard's conditions shape stage 2 of the plan's measurement: group rank 0's, which holds the batch's first windows)
        eng = model.hip_engine(verify=verify, job=job, group=grp if verify else None, conditions=x_cond if x.shape[0] else None)
        if x.shape[0]:
            eng.sample_loop_(x, x_cond, t_start, n_steps, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=window_offset)
            model._note_job((x.shape[0], x.shape[1], n_steps))
        fn.last = (eng, x, x_cond)
        if guard:
            model._outlier_guard(eng, x, x_cond, group=grp)
        return x

    fn.last = None
    fn.wants_global = True
    return fn


def hip_sample_fn(model, seed=0, group=None):
    """The whole `num_timesteps` chain (see hip_steps_fn)."""
    S = model.num_timesteps
    return hip_steps_fn(model, S - 1, S, seed=seed, group=group)


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
    `seed` keys EVERY draw of the call: the initial ones (`harness_noise`) and the per-step Philox stream (the model's
    `philox_seed` is set from it for the call), so two calls with different seeds share no noise.
    All ranks sample from one plan and the runtime guard's verdict is collective (plan.py; a rank without pairs joins in).
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
    own = harness_fn is None
    if own:
        dev = model.betas.device
        grp = _group_of(group)

        def harness_fn(hp, nz, off):
            keep = model.philox_seed
            model.philox_seed = int(seed) * 1000003 + 11  # (+ the window index inside the harness)
            try:
                return harness.full_body_gen_cond_head_pose_sliding_window(model, ds, hp.to(dev), noise=nz, parents=parents,
                                                                           window_offset=off, group=grp, global_pairs=n)
            finally:
                model.philox_seed = keep
    t_out = harness.output_frames(n_frames, seq_len)
    if hi > lo or own:  # (the HIP harness of a rank without pairs still walks the group's collective calls)
        nz = {"x_all": noise["x_all"][lo:hi], "cond": [c[lo:hi] for c in noise["cond"]]}
        aa, root = harness_fn(pairs[lo:hi], nz, lo)
        local = torch.cat((aa.reshape(hi - lo, t_out, 66), root.reshape(hi - lo, t_out, 3)), dim=-1).float().contiguous()
    else:  # more ranks than pairs: this rank still joins the collective, with an empty slice on its device
        local = torch.zeros((0, t_out, 69), device=model.betas.device, dtype=torch.float32)
    if dist.is_initialized():
        local = gather_windows(local, n, group, force=force_collective)
    return local[..., :66].reshape(n, t_out, 22, 3), local[..., 66:]
