"""Deterministic synthetic weights and head-pose windows.

The reference ships no pretrained weights, datasets or tests, so
parity fixtures, the GPU parity tests and bench.py all use seeded random
weights in the reference's checkpoint layout
(/root/reference/egoego/model/transformer_cond_diffusion_model.py:143-214 and
egoego/model/transformer_module.py:36-186 define the tensors and their init
scales).  Values come from numpy's PCG64 keyed by (seed, tensor name), so the
same state dict can be rebuilt anywhere without shipping 44 MB of weights.
"""
import math
import zlib

import numpy as np
import torch


class ModelConfig:
    """Shapes of the stage-2 denoiser as run_egoego.py / eval_stage2.py build it
    (/root/reference/trainer_amass_cond_motion_diffusion.py:458-475)."""

    def __init__(self, d_feats=198, d_model=512, n_head=4, n_dec_layers=4, d_k=256, d_v=256,
                 max_timesteps=121, timesteps=1000, objective="pred_x0"):
        self.d_feats, self.d_model, self.n_head = d_feats, d_model, n_head
        self.n_dec_layers, self.d_k, self.d_v = n_dec_layers, d_k, d_v
        self.max_timesteps, self.timesteps, self.objective = max_timesteps, timesteps, objective

    def ctor_kwargs(self):
        return dict(d_feats=self.d_feats, d_model=self.d_model, n_head=self.n_head,
                    n_dec_layers=self.n_dec_layers, d_k=self.d_k, d_v=self.d_v,
                    max_timesteps=self.max_timesteps, out_dim=self.d_feats,
                    timesteps=self.timesteps, objective=self.objective)


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    return torch.from_numpy((_rng(seed, name).standard_normal(shape) * std).astype(np.float32))


def _uniform(seed, name, shape, bound):
    return torch.from_numpy(_rng(seed, name).uniform(-bound, bound, shape).astype(np.float32))


def sinusoid_position_table(n_position, d_hid):
    """Frozen table of transformer_module.py:6-24 (float64 math, row 0 zeroed)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab = np.where(j % 2 == 0, np.sin(ang), np.cos(ang))
    tab[0] = 0.0
    return torch.from_numpy(tab.astype(np.float32))


def make_weights(cfg, seed=0):
    """Learnable tensors + the frozen position table, keyed like the reference state dict
    (without the 13 schedule buffers, which the module computes itself)."""
    D, dm, H = cfg.d_feats, cfg.d_model, cfg.n_head
    sd = {}
    tr = "denoise_fn.motion_transformer."

    def lin(name, out_f, in_f, conv=False):
        b = 1.0 / np.sqrt(in_f)  # PyTorch default kaiming_uniform(a=sqrt(5)) bound
        shape = (out_f, in_f, 1) if conv else (out_f, in_f)
        sd[name + ".weight"] = _uniform(seed, name + ".weight", shape, b)
        sd[name + ".bias"] = _uniform(seed, name + ".bias", (out_f,), b)

    lin(tr + "start_conv", dm, 2 * D, conv=True)
    sd[tr + "position_vec.weight"] = sinusoid_position_table(cfg.max_timesteps + 1, dm)
    for i in range(cfg.n_dec_layers):
        a = tr + f"layer_stack.{i}.self_attn."
        for nm, dd in (("w_q", cfg.d_k), ("w_k", cfg.d_k), ("w_v", cfg.d_v)):
            sd[a + nm + ".weight"] = _normal(seed, a + nm + ".weight", (H * dd, dm), np.sqrt(2.0 / (dm + dd)))
            sd[a + nm + ".bias"] = _uniform(seed, a + nm + ".bias", (H * dd,), 1.0 / np.sqrt(dm))
        sd[a + "fc.weight"] = _normal(seed, a + "fc.weight", (dm, H * cfg.d_v), np.sqrt(2.0 / (dm + H * cfg.d_v)))
        sd[a + "fc.bias"] = _uniform(seed, a + "fc.bias", (dm,), 1.0 / np.sqrt(H * cfg.d_v))
        f = tr + f"layer_stack.{i}.pos_ffn."
        lin(f + "w_1", dm, dm, conv=True)
        lin(f + "w_2", dm, dm, conv=True)
        for ln in (a + "layer_norm", f + "layer_norm"):
            # perturbed affine so a gamma/beta mix-up cannot hide behind the 1/0 default init
            sd[ln + ".weight"] = 1.0 + _normal(seed, ln + ".weight", (dm,), 0.1)
            sd[ln + ".bias"] = _normal(seed, ln + ".bias", (dm,), 0.1)
    lin("denoise_fn.linear_out", D, dm)
    lin("denoise_fn.time_mlp.1", 256, 64)
    lin("denoise_fn.time_mlp.3", dm, 256)
    return sd


def head_condition_mask(shape, device="cpu"):
    """1 on dims the model must synthesise, 0 on the head joint's position (45:48) and 6D
    rotation (156:162) — trainer_amass_cond_motion_diffusion.py:210-221."""
    m = torch.ones(shape, device=device)
    m[..., 45:48] = 0
    m[..., 156:162] = 0
    return m


def make_head_windows(B, T, seed=0, d_feats=198):
    """Synthetic normalised head-pose windows: zeros except a clipped random
    walk on the head position dims and the first two rows of random rotations on the head
    rot6d dims.  Returns (x_start [B,T,D], cond_mask [B,T,D])."""
    g = np.random.Generator(np.random.PCG64([seed, 7]))
    x = np.zeros((B, T, d_feats), dtype=np.float32)
    walk = np.cumsum(g.standard_normal((B, T, 3)) * 0.02, axis=1) + g.uniform(-0.5, 0.5, (B, 1, 3))
    x[..., 45:48] = np.clip(walk, -1, 1)
    q = g.standard_normal((B, T, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, a, b, c = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r0 = np.stack([1 - 2 * (b * b + c * c), 2 * (a * b - c * w), 2 * (a * c + b * w)], -1)
    r1 = np.stack([2 * (a * b + c * w), 1 - 2 * (a * a + c * c), 2 * (b * c - a * w)], -1)
    x[..., 156:159], x[..., 159:162] = r0, r1
    xs = torch.from_numpy(x)
    return xs, head_condition_mask(xs.shape)


# ------------------------------------------------------------------------------------------ synthetic motion (training-like data)
# first 22 entries of the SMPL-H kintree (harness.SMPLH_PARENTS_22)
_PARENTS_22 = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19)


def make_motion_windows(B, T, seed=0, device="cpu", d_feats=198):
    """Seeded synthetic full-body windows in the model's data layout: [B, T, 198] = 22 normalised
    global joint positions + 22 global rotations as 6D.  NOT real motion — the reference's AMASS data cannot ship — but it has
    the structure a denoiser can learn from: a fixed seeded skeleton driven through forward kinematics by smooth per-joint
    rotations (a few low-frequency sinusoids per axis) on a smooth root walk, the head at the xy origin in the first frame
    like the reference's canonicalised windows (lafan1/utils.py:111-137), positions min/max-normalised to about [-1, 1] and 6D =
    the first two rows of each global rotation (amass_diffusion_dataset.py:446-447).  Used to optimise the module's own training
    loss for a few thousand steps (tools/make_trained_like_checkpoint.py): weights that are no longer the initialisation."""
    assert d_feats == 198
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(int(seed) * 7919 + 13)
    sk = np.random.Generator(np.random.PCG64([1234, 22]))  # the skeleton is the same for every seed
    offs = sk.standard_normal((22, 3))
    offs = offs / np.linalg.norm(offs, axis=1, keepdims=True) * sk.uniform(0.08, 0.35, (22, 1))
    offs[0] = 0.0
    offs = torch.from_numpy(offs.astype(np.float32)).to(dev)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev)

    def uni(*shape):
        return torch.rand(*shape, generator=g).to(dev)

    tt = torch.arange(T, device=dev, dtype=torch.float32) / max(T, 1)
    freqs = torch.tensor([0.5, 1.0, 2.0], device=dev)
    amp = 0.35 * rnd(B, 22, 3, 3) / freqs                                         # [B, joint, axis, harmonic]
    ph = 2 * math.pi * uni(B, 22, 3, 3)
    aa = (amp[..., None] * torch.sin(2 * math.pi * freqs[:, None] * tt + ph[..., None])).sum(-2)  # [B, 22, 3, T]
    aa = aa.permute(0, 3, 1, 2).contiguous()                                      # [B, T, 22, 3]
    aa[:, :, 0, 2] += 2 * math.pi * uni(B, 1)                                      # random heading of the root
    ang = aa.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    ax = aa / ang
    K = torch.zeros(B, T, 22, 3, 3, device=dev)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -ax[..., 2], ax[..., 1], ax[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -ax[..., 0], -ax[..., 1], ax[..., 0]
    eye = torch.eye(3, device=dev).expand(B, T, 22, 3, 3)
    # batched matmul in pieces: one call over B*T*22 > 2^24 matrices faults inside the BLAS on this stack (4096 x 196 x 22, measured)
    KK = torch.cat([k @ k for k in K.split(512)])
    Rl = eye + torch.sin(ang)[..., None] * K + (1 - torch.cos(ang))[..., None] * KK  # Rodrigues
    step = 0.01 * rnd(B, T, 3)
    step = torch.nn.functional.avg_pool1d(step.transpose(1, 2), 9, 1, 4, count_include_pad=False).transpose(1, 2)
    root = torch.cumsum(step, dim=1) * 3.0
    root[..., 2] = 0.9 + 0.05 * torch.sin(2 * math.pi * (tt[None] + uni(B, 1)))
    Rg, pos = [Rl[:, :, 0]], [root]
    for j in range(1, 22):
        p = _PARENTS_22[j]
        Rg.append(Rg[p] @ Rl[:, :, j])
        pos.append(pos[p] + (Rg[p] @ offs[j][None, None, :, None])[..., 0])
    Rg, pos = torch.stack(Rg, 2), torch.stack(pos, 2)                               # [B, T, 22, 3, 3], [B, T, 22, 3]
    shift = pos[:, :1, 15:16, :].clone()
    shift[..., 2] = 0
    pos = pos - shift
    lo = torch.tensor([-1.5, -1.5, 0.0], device=dev)
    hi = torch.tensor([1.5, 1.5, 2.0], device=dev)
    pos = ((pos - lo) / (hi - lo) * 2 - 1).clamp(-1, 1)
    return torch.cat((pos.reshape(B, T, 66), Rg[..., :2, :].reshape(B, T, 132)), -1).float()
