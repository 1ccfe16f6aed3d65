"""Callers either side of the sampling loop: head-condition mask, the overlapping
sliding-window harness with heading canonicalisation, and the post-loop conversion of a sampled window
back to SMPL-H parameters.

Mirrors, with the same names and argument meaning,
  Trainer.prep_head_condition_mask                      trainer_amass_cond_motion_diffusion.py:210-221
  Trainer.full_body_gen_cond_head_pose_sliding_window   trainer_amass_cond_motion_diffusion.py:261-276
  CondGaussianDiffusion.p_sample_loop_sliding_window_w_canonical / sample_sliding_window_w_canonical /
  convert_model_res_to_data                             egoego/model/transformer_cond_diffusion_model.py:329-555
  rotate_at_frame_smplh                                 egoego/lafan1/utils.py:111-137
  AMASSDataset.{normalize,de_normalize}_jpos_min_max, fk_smpl, quat_ik_torch
                                                        egoego/data/amass_diffusion_dataset.py:109-125, 265-293, 379-392

Differences from the reference, on purpose: everything stays on the GPU (the reference round-trips through
numpy twice per window, M:362-368, 435-440); the 1000 diffusion steps of a window run inside ONE call of the
HIP sample loop, with the per-step overwrite of the first 10 frames (M:395-397) done by the step kernel's
`prefix` argument; 6D -> matrix uses the HIP kernel.  Parity: the loop and the conversion chain are checked against a run of
the reference's own code on its demo trajectory (tests/golden/make_window_loop_golden.py -> tests/test_window_loop_golden.py);
only the bodies of the pytorch3d.transforms functions (absent here) are unpinned in the sense of SURVEY.md §8c.
"""
import numpy as np
import ctypes as C

import torch

from . import rotations as R
from . import _lib

# first 22 entries of the SMPL-H kintree (amass_diffusion_dataset.py:83-90 reads them from the licensed npz)
SMPLH_PARENTS_22 = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19)
HEAD_IDX = 15
OVERLAP = 10  # frames shared by consecutive windows (M:350)


def prep_head_condition_mask(data, joint_idx=HEAD_IDX):
    """1 = to be generated, 0 = given head position (3) and 6D rotation (6) dims."""
    mask = torch.ones_like(data)
    mask[:, :, joint_idx * 3:joint_idx * 3 + 3] = 0
    mask[:, :, 22 * 3 + joint_idx * 6:22 * 3 + joint_idx * 6 + 6] = 0
    return mask


def prep_padding_mask(val_data, seq_len, window):
    """Trainer.prep_padding_mask (trainer_amass_cond_motion_diffusion.py:223-231): [B, 1, window + 1] bool, True on the time token and the
    first seq_len[b] frames of window b (what `p_sample_loop(..., padding_mask=)` / `denoise_fn(..., padding_mask=)` take; `sample()` itself
    drops its padding_mask like the reference's, M:528-532)."""
    actual = seq_len.to(torch.long) + 1  # + 1: the time token
    keep = torch.arange(window + 1)[None, :].expand(val_data.shape[0], window + 1) < actual.cpu()[:, None]
    return keep[:, None, :].to(val_data.device)


class SkeletonStats:
    """The three `ds` methods the harness needs, for users without the reference's AMASSDataset
    (which requires human_body_prior + licensed SMPL-H): min/max joint normalisation and quaternion FK
    over given rest-pose offsets.  Any object with the same three methods can be passed instead."""

    def __init__(self, global_jpos_min, global_jpos_max, rest_human_offsets, parents=SMPLH_PARENTS_22):
        self.global_jpos_min = torch.as_tensor(global_jpos_min, dtype=torch.float32).reshape(1, 22, 3)
        self.global_jpos_max = torch.as_tensor(global_jpos_max, dtype=torch.float32).reshape(1, 22, 3)
        self.rest_human_offsets = torch.as_tensor(rest_human_offsets, dtype=torch.float32).reshape(1, 22, 3)
        self.parents = tuple(int(p) for p in parents)

    def normalize_jpos_min_max(self, ori_jpos):
        lo, hi = self.global_jpos_min.to(ori_jpos.device), self.global_jpos_max.to(ori_jpos.device)
        return (ori_jpos - lo) / (hi - lo) * 2 - 1

    def de_normalize_jpos_min_max(self, normalized_jpos):
        lo, hi = self.global_jpos_min.to(normalized_jpos.device), self.global_jpos_max.to(normalized_jpos.device)
        return (normalized_jpos + 1) * 0.5 * (hi - lo) + lo

    def fk_smpl(self, root_trans, lrot_aa):
        """root_trans [N,3], local axis-angle [N,22,3] -> (global quaternions [N,22,4], global joints [N,22,3])."""
        lrot = R.matrix_to_quaternion(R.axis_angle_to_matrix(lrot_aa))
        lpos = self.rest_human_offsets.to(lrot.device).repeat(lrot.shape[0], 1, 1)
        gp, gr = [lpos[..., :1, :]], [lrot[..., :1, :]]
        for i in range(1, len(self.parents)):
            p = self.parents[i]
            gp.append(R.quaternion_apply(gr[p], lpos[..., i:i + 1, :]) + gp[p])
            gr.append(R.quaternion_multiply(gr[p], lrot[..., i:i + 1, :]))
        return torch.cat(gr, dim=-2), torch.cat(gp, dim=-2) + root_trans[:, None, :]


def _parents_of(ds, parents=None):
    """The kinematic tree to use: an explicit argument, else `ds.parents` (SkeletonStats carries it; set it on any
    other `ds` to override), else the first 22 entries of the SMPL-H kintree (an input, not a constant)."""
    if parents is None:
        parents = getattr(ds, "parents", None)
    return SMPLH_PARENTS_22 if parents is None else tuple(int(p) for p in parents)


def _hip_stats(ds, *names):
    """The statistics tensors of `ds` as flat fp32 lists when the per-window HIP kernels may stand in for ds's METHODS
    (normalize / de_normalize / fk_smpl), else None (the torch chain then calls the methods).  The kernels implement
    exactly SkeletonStats' and the reference AMASSDataset's methods, so they are used for those two types, or for any
    `ds` that opts in with `ds.use_hip_harness = True`; a `ds` with methods of its own keeps them.  Shapes must match
    the 22-joint layout (66 values each)."""
    if not (isinstance(ds, SkeletonStats) or type(ds).__name__ == "AMASSDataset" or getattr(ds, "use_hip_harness", False)):
        return None
    if getattr(ds, "use_hip_harness", True) is False:
        return None
    out = []
    for n in names:
        v = getattr(ds, n, None)
        if v is None:
            return None
        v = torch.as_tensor(v)
        if v.numel() != 66:
            return None
        out.append(v)
    return out


def rotate_at_frame(trans, quat, cano_t_idx=0):
    """Heading canonicalisation on the device: rotate about z so that the facing direction of frame
    `cano_t_idx` projects onto +x.  trans [B,T,3], quat [B,T,4] -> (trans', quat', yrot [B,1,1,4])."""
    key = quat[:, cano_t_idx:cano_t_idx + 1, :]
    ex = torch.zeros_like(key[..., :3])
    ex[..., 0] = 1
    t = 2.0 * torch.cross(key[..., 1:], ex, dim=-1)
    fwd = ex + key[..., :1] * t + torch.cross(key[..., 1:], t, dim=-1)
    fwd = fwd * fwd.new_tensor([1, 1, 0])
    fwd = fwd / (fwd.norm(dim=-1, keepdim=True) + 1e-8)
    # quaternion taking +x onto fwd: (|x||f| + x.f, x cross f), normalised
    w = torch.sqrt((ex * ex).sum(-1) * (fwd * fwd).sum(-1))[..., None] + (ex * fwd).sum(-1, keepdim=True)
    yrot = torch.cat((w, torch.cross(ex, fwd, dim=-1)), -1)
    yrot = yrot / (yrot.norm(dim=-1, keepdim=True) + 1e-8)
    inv = yrot * yrot.new_tensor([1, -1, -1, -1])
    new_q = R.quaternion_raw_multiply(inv.expand_as(quat), quat)
    ti = 2.0 * torch.cross(inv[..., 1:].expand_as(trans), trans, dim=-1)
    new_x = trans + inv[..., :1] * ti + torch.cross(inv[..., 1:].expand_as(trans), ti, dim=-1)
    return new_x, new_q, yrot[:, None]


def quat_ik(grot_mat, parents=SMPLH_PARENTS_22):
    """Global -> local joint rotations (amass_diffusion_dataset.py:109-125)."""
    grot = R.matrix_to_quaternion(grot_mat)
    par = list(parents[1:])
    res = torch.cat((grot[..., :1, :],
                     R.quaternion_multiply(R.quaternion_invert(grot[..., par, :]), grot[..., 1:, :])), dim=-2)
    return R.quaternion_to_matrix(res)


def convert_model_res_to_data(ds, all_res_list, recover_rot_quat, curr_global_head_jpos=None, parents=None):
    """M:469-525: normalised window [B,T,198] -> (local axis-angle [B,T,22,3], root position [B,T,3],
    head position [B,T,3]) in the ORIGINAL (un-canonicalised) heading.  recover_rot_quat: [B,1,1,4]
    (tensor or numpy)."""
    bs = all_res_list.shape[0]
    dev = all_res_list.device
    parents = _parents_of(ds, parents)
    st = _hip_stats(ds, "global_jpos_min", "global_jpos_max") if all_res_list.is_cuda and all_res_list.shape[-1] == 198 else None
    if st is not None and len(parents) == 22:
        jmin, jmax = st
        # one HIP kernel for the whole chain (egoego_convert_model_res); the torch expressions below are for CPU tensors
        from . import _lib
        lib = _lib.load()
        x = all_res_list.to(torch.float32).contiguous()
        n = x.shape[1]
        rec = torch.as_tensor(recover_rot_quat).to(dev, torch.float32).reshape(bs, 4).contiguous()
        lo = torch.as_tensor(jmin).to(dev, torch.float32).reshape(66).contiguous()
        hi = torch.as_tensor(jmax).to(dev, torch.float32).reshape(66).contiguous()
        aa = torch.empty(bs, n, 22, 3, device=dev, dtype=torch.float32)
        root = torch.empty(bs, n, 3, device=dev, dtype=torch.float32)
        head = torch.empty(bs, n, 3, device=dev, dtype=torch.float32)
        par = (C.c_int32 * 22)(*[int(p) for p in parents])
        _lib.check(lib.egoego_convert_model_res(x.data_ptr(), rec.data_ptr(), lo.data_ptr(), hi.data_ptr(), par, HEAD_IDX, bs, n,
                                                aa.data_ptr(), root.data_ptr(), head.data_ptr(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return aa, root, head
    jpos = ds.de_normalize_jpos_min_max(all_res_list[:, :, :66].reshape(-1, 22, 3)).reshape(bs, -1, 22, 3)
    n = jpos.shape[1]
    rot6d = all_res_list[:, :, 66:].reshape(bs, n, 22, 6)
    rec = torch.as_tensor(recover_rot_quat).to(dev, jpos.dtype).reshape(bs, 1, 1, 4)
    gquat = R.matrix_to_quaternion(R.rotation_6d_to_matrix(rot6d))
    ori_gquat = R.quaternion_multiply(rec.expand_as(gquat), gquat)
    rec_t = rec.reshape(bs, 1, 4).expand(bs, n, 4)
    root = R.quaternion_apply(rec_t, jpos[:, :, 0, :])
    head = R.quaternion_apply(rec_t, jpos[:, :, HEAD_IDX, :])
    local = quat_ik(R.quaternion_to_matrix(ori_gquat).reshape(-1, 22, 3, 3), parents).reshape(bs, n, 22, 3, 3)
    return R.matrix_to_axis_angle(local), root, head


def _window_condition_hip(ds, head_jpos, head_jquat):
    """egoego_window_condition (rotate_at_frame + x_start assembly + normalisation in one HIP kernel, M:355-378) for ROCm tensors
    when the kernels may stand in for `ds` (_hip_stats); None otherwise.  Returns (x_start [B,Tw,198], recover [B,1,1,4])."""
    st = _hip_stats(ds, "global_jpos_min", "global_jpos_max") if head_jpos.is_cuda else None
    if st is None:
        return None
    jmin, jmax = st
    from . import _lib
    lib = _lib.load()
    dev = head_jpos.device
    b, tw = head_jpos.shape[0], head_jpos.shape[1]
    jp, jq = head_jpos.to(torch.float32).contiguous(), head_jquat.to(torch.float32).contiguous()
    lo = torch.as_tensor(jmin).to(dev, torch.float32).reshape(66).contiguous()
    hi = torch.as_tensor(jmax).to(dev, torch.float32).reshape(66).contiguous()
    x_start = torch.empty(b, tw, 198, device=dev, dtype=torch.float32)
    rec = torch.empty(b, 4, device=dev, dtype=torch.float32)
    _lib.check(lib.egoego_window_condition(jp.data_ptr(), jq.data_ptr(), lo.data_ptr(), hi.data_ptr(), HEAD_IDX, b, tw, x_start.data_ptr(),
                                           rec.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return x_start, rec.reshape(b, 1, 1, 4)


def _window_prefix_hip(ds, aa, root, n_last, parents=None):
    """egoego_window_prefix (one HIP kernel for fk_smpl + rotate_at_frame + normalisation + 6D, M:399-467) when the tensors
    live on a ROCm device and the kernels may stand in for `ds` (_hip_stats); None otherwise (the torch chain then runs)."""
    parents = _parents_of(ds, parents)
    st = _hip_stats(ds, "rest_human_offsets", "global_jpos_min", "global_jpos_max") if aa.is_cuda else None
    if st is None or len(parents) != 22 or n_last < 1 or n_last > aa.shape[1]:
        return None
    rest, jmin, jmax = st
    from . import _lib
    lib = _lib.load()
    dev = aa.device
    b, tw = aa.shape[0], aa.shape[1]
    f32 = lambda t, n: torch.as_tensor(t).to(dev, torch.float32).reshape(n).contiguous()
    aa_c, root_c = aa.to(torch.float32).contiguous(), root.to(torch.float32).contiguous()
    rest_c, lo, hi = f32(rest, 66), f32(jmin, 66), f32(jmax, 66)
    out = torch.empty(b, n_last, 198, device=dev, dtype=torch.float32)
    par = (C.c_int32 * 22)(*[int(p) for p in parents])
    _lib.check(lib.egoego_window_prefix(aa_c.data_ptr(), root_c.data_ptr(), rest_c.data_ptr(), lo.data_ptr(), hi.data_ptr(), par,
                                        HEAD_IDX, b, tw, n_last, out.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def window_spans(num_frames, seq_len):
    """(first frame, length) of every window the sliding-window loop runs over a `num_frames` trajectory (M:350-356): stride
    seq_len - 10, a trailing window of at most 10 frames is dropped."""
    stride = seq_len - OVERLAP
    spans = []
    for t_idx in range(0, num_frames, stride):
        n = min(seq_len, num_frames - t_idx)
        if n <= seq_len - stride:
            break
        spans.append((t_idx, n))
    return spans


def output_frames(num_frames, seq_len):
    """Frames of the stitched result of the sliding-window loop: the first window whole, the later ones without their 10 overlap frames."""
    spans = window_spans(num_frames, seq_len)
    return sum(n for _, n in spans) - OVERLAP * max(0, len(spans) - 1)


@torch.no_grad()
def p_sample_loop_sliding_window_w_canonical(model, ds, shape, global_head_jpos, global_head_jquat, cond_mask,
                                             noise=None, parents=None, window_offset=0, group=None, global_pairs=None):
    """M:329-467.  Windows of `model.seq_len` frames, stride seq_len-10; window k+1 is conditioned on the
    last 10 frames of window k, re-canonicalised, by overwriting its first 10 frames after every step.

    noise: {'x_all': [B,T,D], 'cond': [per-window [B,Tw,D]], 'steps': [per-window [S,B,Tw,D]]} injects every draw (tests);
    without 'steps' the per-step draws are in-kernel Philox keyed by (philox_seed + window; window_offset + b, ...), i.e. by the
    GLOBAL sequence index — what dist.harness_sharded uses so that the result does not depend on how sequences are sharded.
    group (dist.harness_sharded): the process group whose ranks sample from one plan (model.hip_engine / _outlier_guard are
    collective then); a rank without sequences (b = 0) walks the same collective calls and returns empty tensors.
    global_pairs (dist.harness_sharded): the (sequence, sample) pairs of the WHOLE call over all ranks — the plan's small-job rule is
    sized by the global job, so that the precision picked does not depend on how many ranks share it.
    """
    b = shape[0]
    S = model.num_timesteps
    seq_len = model.seq_len
    stride = seq_len - OVERLAP
    spans = window_spans(global_head_jpos.shape[1], seq_len)
    job = ((b if global_pairs is None else int(global_pairs)) * len(spans), seq_len, S)
    eng = model.hip_engine(verify=True, job=job, group=group)
    device = model.betas.device
    if b == 0:
        empty = torch.zeros((0, seq_len, 198), device=device)
        eng = model.hip_engine(verify=True, job=job, group=group)  # (mirrors the other ranks' second call, the one that hands the first window's conditions over)
        for _ in spans:
            model._outlier_guard(eng, empty, empty, group=group)
            eng = model.hip_engine()  # (a collective step-down re-packs here too: this rank must leave the int8 precision with the others)
        t_out = output_frames(global_head_jpos.shape[1], seq_len)
        return torch.zeros((0, t_out, 22, 3), device=device), torch.zeros((0, t_out, 3), device=device)
    parents = _parents_of(ds, parents)
    x_all = noise["x_all"].to(device).float() if noise is not None else torch.randn(shape, device=device)
    jpos_all = global_head_jpos.to(device)
    jquat_all = global_head_jquat.to(device)
    num_steps = jpos_all.shape[1]
    whole_aa = whole_root = whole_head = None
    prefix = None
    w_idx = 0
    for t_idx in range(0, num_steps, stride):
        curr_x = x_all[:, t_idx:t_idx + seq_len].contiguous().clone()
        if curr_x.shape[1] <= seq_len - stride:
            break
        cur_quat = jquat_all[:, t_idx:t_idx + seq_len]
        cur_jpos = jpos_all[:, t_idx:t_idx + seq_len]
        cond = _window_condition_hip(ds, cur_jpos, cur_quat)
        if cond is not None:
            x_start, recover = cond
        else:
            al_trans, al_quat, recover = rotate_at_frame(cur_jpos, cur_quat, 0)
            move0 = al_trans[:, 0:1, :].clone()
            move0[:, :, 2] = 0
            al_trans = al_trans - move0
            al_6d = R.matrix_to_rotation_6d(R.quaternion_to_matrix(al_quat))
            x_start = torch.zeros(b, al_6d.shape[1], 198, device=device)
            x_start[:, :, HEAD_IDX * 3:HEAD_IDX * 3 + 3] = al_trans.float()
            x_start[:, :, 66 + HEAD_IDX * 6:66 + HEAD_IDX * 6 + 6] = al_6d.float()
            x_start[:, :, :66] = ds.normalize_jpos_min_max(x_start[:, :, :66].reshape(-1, 22, 3)).reshape(b, -1, 66)
        cm = cond_mask[:, t_idx:t_idx + seq_len].to(device)
        cn = noise["cond"][w_idx].to(device) if noise is not None else torch.randn_like(x_start)
        x_cond = (x_start * (1.0 - cm) + cm * cn).float().contiguous()
        pfx = prefix if t_idx > 0 else None
        if t_idx == 0:  # (the first window's conditions shape stage 2 of the plan's measurement; collective like the call above)
            eng = model.hip_engine(verify=True, job=job, group=group, conditions=x_cond)
        if noise is not None and "steps" in noise:
            eng.sample_loop_(curr_x, x_cond, S - 1, S, noise=noise["steps"][w_idx].to(device).float().contiguous(), prefix=pfx)
        elif model.sampling_rng == "philox" or noise is not None:
            eng.sample_loop_(curr_x, x_cond, S - 1, S, noise_mode=_lib.NOISE_PHILOX, seed=model.philox_seed + w_idx,
                             window_offset=window_offset, prefix=pfx)
        else:
            model._torch_rng_chain(eng, curr_x, x_cond, S, pfx)
        model._note_job((b, curr_x.shape[1], S))
        model._outlier_guard(eng, curr_x, x_cond, group=group)  # (may re-pack in another precision — on every rank of the group alike —: take the engine afresh for the next window)
        eng = model.hip_engine()
        aa, root, head = convert_model_res_to_data(ds, curr_x, recover, cur_jpos, parents)
        if t_idx == 0:
            whole_aa, whole_root, whole_head = aa, root, head
        else:
            move = whole_head[:, -1:, :] - head[:, seq_len - stride - 1:seq_len - stride, :]
            root = root + move
            head = head + move
            whole_aa = torch.cat((whole_aa, aa[:, seq_len - stride:]), dim=1)
            whole_root = torch.cat((whole_root, root[:, seq_len - stride:]), dim=1)
            whole_head = torch.cat((whole_head, head[:, seq_len - stride:]), dim=1)
        # condition for the next window: the last `OVERLAP` frames, re-canonicalised and re-normalised
        hip_prefix = _window_prefix_hip(ds, aa, root, seq_len - stride, parents)
        if hip_prefix is not None:
            prefix = hip_prefix
            w_idx += 1
            continue
        gq, gj = ds.fk_smpl(root.reshape(-1, 3), aa.reshape(-1, 22, 3))
        gq = gq.reshape(b, -1, 22, 4)[:, -seq_len + stride:]
        gj = gj.reshape(b, -1, 22, 3)[:, -seq_len + stride:]
        t_trans, _, t_rec = rotate_at_frame(gj[:, :, HEAD_IDX, :], gq[:, :, HEAD_IDX, :], 0)
        t_move = t_trans[:, 0:1, :].clone()
        t_move[:, :, 2] = 0
        inv = R.quaternion_invert(t_rec.float()).expand(b, gj.shape[1], 22, 4)
        pj = R.quaternion_apply(inv, gj) - t_move[:, :, None, :]
        pj = ds.normalize_jpos_min_max(pj.reshape(-1, 22, 3)).reshape(b, -1, 66)
        p6 = R.matrix_to_rotation_6d(R.quaternion_to_matrix(R.quaternion_multiply(inv, gq))).reshape(b, -1, 132)
        prefix = torch.cat((pj, p6), dim=-1).float().contiguous()
        w_idx += 1
    return whole_aa, whole_root


@torch.no_grad()
def sample_sliding_window_w_canonical(model, ds, global_head_jpos, global_head_jquat, x_start, cond_mask, noise=None,
                                      parents=None, window_offset=0, group=None, global_pairs=None):
    model.denoise_fn.eval()
    res = p_sample_loop_sliding_window_w_canonical(model, ds, x_start.shape, global_head_jpos, global_head_jquat,
                                                   cond_mask, noise=noise, parents=parents, window_offset=window_offset, group=group,
                                                   global_pairs=global_pairs)
    model.denoise_fn.train()
    return res


@torch.no_grad()
def full_body_gen_cond_head_pose_sliding_window(model, ds, head_pose, noise=None, parents=None, window_offset=0, group=None, global_pairs=None):
    """head_pose [B,T,7] = xyz + quaternion (w,x,y,z) -> (local axis-angle [B,T',22,3], root [B,T',3])."""
    jpos, jquat = head_pose[:, :, :3], head_pose[:, :, 3:]
    data = torch.zeros(head_pose.shape[0], head_pose.shape[1], 198, device=head_pose.device)
    return sample_sliding_window_w_canonical(model, ds, jpos, jquat, data, prep_head_condition_mask(data), noise=noise,
                                             parents=parents, window_offset=window_offset, group=group, global_pairs=global_pairs)


# ------------------------------------------------------------------------------------------ checkpoints
def build_stage2_model(window=120, d_model=512, n_head=4, n_dec_layers=4, d_k=256, d_v=256, repr_dim=22 * 3 + 22 * 6,
                       device=None):
    """The model `get_trainer()` builds (trainer_amass_cond_motion_diffusion.py:458-475): pred_x0, l1, 1000 cosine
    steps, max_timesteps = window + 1."""
    from .model import CondGaussianDiffusion
    m = CondGaussianDiffusion(d_feats=repr_dim, d_model=d_model, n_head=n_head, n_dec_layers=n_dec_layers, d_k=d_k, d_v=d_v,
                              max_timesteps=window + 1, out_dim=repr_dim, timesteps=1000, objective="pred_x0",
                              loss_type="l1")
    return m.to(device) if device is not None else m


def load_stage2_checkpoint(path_or_dict, model=None, use_ema=True, device=None, **model_kw):
    """Load a reference checkpoint — {'step', 'model', 'ema', 'scaler'} as written by Trainer.save
    (trainer:99-106) — into the HIP-backed module.  Inference in the reference always goes through
    `trainer.ema.ema_model` (trainer:243,263,273), so by default the EMA weights are taken: ema-pytorch stores them
    under the 'ema_model.' prefix (next to 'online_model.*', 'initted', 'step').  Like the reference
    (`strict=False`, trainer:120-121) unknown keys are ignored; missing ones are reported."""
    data = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, "read") else path_or_dict
    if model is None:
        model = build_stage2_model(device=device, **model_kw)
    sd = None
    if use_ema and "ema" in data:
        sd = {k[len("ema_model."):]: v for k, v in data["ema"].items() if k.startswith("ema_model.")}
    if not sd:
        sd = data["model"] if "model" in data else data
    missing, unexpected = model.load_state_dict(sd, strict=False)
    return model, {"step": data.get("step"), "missing": list(missing), "unexpected": list(unexpected)}
