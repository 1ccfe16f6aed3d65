"""CPU oracle for EgoEgo's stage-2 conditional motion-diffusion sampling path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``egoego_release_amd``) never routes through here.

It is a functional, fp32, CPU-PyTorch restatement of the reference algorithm,
operating directly on a state dict in the reference's checkpoint layout.  It
issues the same aten ops in the same order as the reference modules, so on
the same torch build its outputs are bit-identical to the reference
(checked in ``tests/golden/make_golden.py``, which imports the reference in
the authoring container and writes the committed fixtures; parity status:
PINNED for the sampling loop / denoiser, see ``tests/test_oracle_golden.py``).

``rotation_6d_to_matrix`` restates pytorch3d (absent from /root/reference,
version unpinned by the reference's requirements.txt); no reference test pins
it, so that one function is "parity unpinned" and is anchored on mathematical
known answers instead (tests/test_rot6d.py).

Reference files (all under /root/reference):
  M  = egoego/model/transformer_cond_diffusion_model.py
  TM = egoego/model/transformer_module.py
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PFX = "denoise_fn."
TR = PFX + "motion_transformer."


# --------------------------------------------------------------------------- schedule
def cosine_betas(timesteps, s=0.008):
    """M:47-57 — cos^2 schedule in float64, clipped to [0, 0.999]."""
    u = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    abar = torch.cos(((u / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)


def linear_betas(timesteps):
    """M:41-45."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def make_schedule(timesteps=1000, kind="cosine", p2_gamma=0.0, p2_k=1):
    """M:173-214 — the 13 registered buffers (float64 math, stored fp32)."""
    if kind == "cosine":
        betas = cosine_betas(timesteps)
    elif kind == "linear":
        betas = linear_betas(timesteps)
    else:
        raise ValueError(f"unknown beta schedule {kind}")
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    out = {
        "betas": betas,
        "alphas_cumprod": abar,
        "alphas_cumprod_prev": abar_prev,
        "sqrt_alphas_cumprod": torch.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - abar),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - abar),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / abar),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / abar - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(abar_prev) / (1.0 - abar),
        "posterior_mean_coef2": (1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar),
        "p2_loss_weight": (p2_k + abar / (1 - abar)) ** -p2_gamma,
    }
    return {k: v.to(torch.float32) for k, v in out.items()}


# --------------------------------------------------------------------------- tables
def sinusoid_table(n_position, d_hid, padding_idx=0):
    """TM:6-24 — frozen position table, float64 numpy then cast; row padding_idx zeroed."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab = np.array(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    if padding_idx is not None:
        tab[padding_idx] = 0.0
    return torch.FloatTensor(tab)


def time_embed(sd, t):
    """M:61-73 + M:111-116 — sinusoidal(64) -> Linear -> exact-erf GELU -> Linear."""
    half = 32
    c = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -c)
    e = t[:, None] * f[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    h = F.linear(e, sd[PFX + "time_mlp.1.weight"], sd[PFX + "time_mlp.1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd[PFX + "time_mlp.3.weight"], sd[PFX + "time_mlp.3.bias"])


# --------------------------------------------------------------------------- denoiser
def _mha(sd, pre, h, n_head, d_k, d_v, taps=None):
    """TM:61-95 — projections, head split, scaled softmax attention, fc, +res, LN."""
    bs, n, _ = h.shape
    res = h

    def split(name, dd):
        y = F.linear(h, sd[pre + name + ".weight"], sd[pre + name + ".bias"])
        return y.view(bs, n, n_head, dd).permute(2, 0, 1, 3).contiguous().view(-1, n, dd)

    q, k, v = split("w_q", d_k), split("w_k", d_k), split("w_v", d_v)
    attn = torch.bmm(q, k.transpose(1, 2))
    attn = attn / np.power(d_k, 0.5)
    attn = F.softmax(attn, dim=2)
    o = torch.bmm(attn, v)
    o = o.view(n_head, bs, n, d_v).permute(1, 2, 0, 3).contiguous().view(bs, n, -1)
    if taps is not None:
        taps["q"], taps["k"], taps["v"], taps["attn_out"] = q, k, v, o
    y = F.linear(o, sd[pre + "fc.weight"], sd[pre + "fc.bias"])
    d = y.shape[-1]
    return F.layer_norm(y + res, (d,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5)


def _ffn(sd, pre, h, taps=None):
    """TM:107-116 — Conv1d(k=1) -> ReLU -> Conv1d(k=1), +res, LN."""
    res = h
    y = h.transpose(1, 2)
    y = F.relu(F.conv1d(y, sd[pre + "w_1.weight"], sd[pre + "w_1.bias"]))
    if taps is not None:
        taps["ffn_hidden"] = y.transpose(1, 2)
    y = F.conv1d(y, sd[pre + "w_2.weight"], sd[pre + "w_2.bias"])
    y = y.transpose(1, 2)
    d = y.shape[-1]
    return F.layer_norm(y + res, (d,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5)


def denoise(sd, x_all, t, padding_mask=None, n_head=4, d_k=256, d_v=256, taps=None):
    """M:118-141 + TM:188-225 — x_all [B,T,2D], t int64 [B] -> [B,T,D].

    padding_mask: optional bool/float [B,1,T+1]; rows are multiplied by it after
    attention and after the FFN (TM:135,139).  Attention itself is unmasked
    (use_full_attention=True, TM:210-211).
    taps: optional dict that receives per-layer intermediates (test use).
    """
    tau = time_embed(sd, t)[:, None, :]
    bs, n_frames = x_all.shape[0], x_all.shape[1]
    L = n_frames + 1
    pm = None
    if padding_mask is not None:
        pm = padding_mask.squeeze(1).unsqueeze(-1).float()
    pos = (torch.arange(L) + 1)[None, :].repeat(bs, 1)
    e = F.conv1d(x_all.transpose(1, 2), sd[TR + "start_conv.weight"], sd[TR + "start_conv.bias"])
    e = e.transpose(1, 2)
    h = torch.cat((tau, e), dim=1) + F.embedding(pos, sd[TR + "position_vec.weight"])
    if taps is not None:
        taps["embed"] = h
    li = 0
    while (TR + f"layer_stack.{li}.self_attn.w_q.weight") in sd:
        lt = {} if taps is not None else None
        base = TR + f"layer_stack.{li}."
        h = _mha(sd, base + "self_attn.", h, n_head, d_k, d_v, lt)
        if pm is not None:
            h = h * pm
        if lt is not None:
            lt["attn_ln"] = h
        h = _ffn(sd, base + "pos_ffn.", h, lt)
        if pm is not None:
            h = h * pm
        if lt is not None:
            lt["out"] = h
            taps[f"layer{li}"] = lt
        li += 1
    return F.linear(h[:, 1:], sd[PFX + "linear_out.weight"], sd[PFX + "linear_out.bias"])


# --------------------------------------------------------------------------- diffusion step
def _gather(a, t, ndim):
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def p_sample(sd, sched, x, t, x_cond, noise, objective="pred_x0", clip_denoised=True,
             padding_mask=None, **kw):
    """M:231-256 — one ancestral step with the noise tensor supplied by the caller."""
    out = denoise(sd, torch.cat((x, x_cond), dim=-1), t, padding_mask=padding_mask, **kw)
    if objective == "pred_noise":
        x0 = (_gather(sched["sqrt_recip_alphas_cumprod"], t, x.dim()) * x
              - _gather(sched["sqrt_recipm1_alphas_cumprod"], t, x.dim()) * out)
    elif objective == "pred_x0":
        x0 = out
    else:
        raise ValueError(f"unknown objective {objective}")
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    mean = (_gather(sched["posterior_mean_coef1"], t, x.dim()) * x0
            + _gather(sched["posterior_mean_coef2"], t, x.dim()) * x)
    logvar = _gather(sched["posterior_log_variance_clipped"], t, x.dim())
    nonzero = (1 - (t == 0).float()).reshape(x.shape[0], *((1,) * (x.dim() - 1)))
    return mean + nonzero * (0.5 * logvar).exp() * noise


def p_sample_loop(sd, sched, x_start, cond_mask, gen, objective="pred_x0", num_timesteps=None,
                  trajectory=None, **kw):
    """M:258-270 — RNG draw order: x_T, condition noise, then one draw per step (t=0 included).

    gen: a CPU torch.Generator; all draws come from it in the reference's order.
    trajectory: optional list receiving x after every step (test diagnostics).
    """
    S = num_timesteps if num_timesteps is not None else sched["betas"].shape[0]
    b = x_start.shape[0]
    x = torch.randn(x_start.shape, generator=gen)
    x_cond = x_start * (1.0 - cond_mask) + cond_mask * torch.randn(x_start.shape, generator=gen)
    for i in reversed(range(S)):
        noise = torch.randn(x.shape, generator=gen)
        x = p_sample(sd, sched, x, torch.full((b,), i, dtype=torch.long), x_cond, noise, objective, **kw)
        if trajectory is not None:
            trajectory.append(x.clone())
    return x


def head_condition_mask(shape):
    """trainer_amass_cond_motion_diffusion.py:210-221 — 1 = missing, 0 = head-pose dims."""
    m = torch.ones(shape)
    m[..., 15 * 3:15 * 3 + 3] = 0
    m[..., 22 * 3 + 15 * 6:22 * 3 + 15 * 6 + 6] = 0
    return m


# --------------------------------------------------------------------------- rot6d (pytorch3d restatement)
def rotation_6d_to_matrix(d6):
    """Zhou et al. 2019 Gram-Schmidt as pytorch3d.transforms.rotation_6d_to_matrix defines it
    (call site M:493): b1 = norm(a1); b2 = norm(a2 - <b1,a2> b1); b3 = b1 x b2; rows stacked.
    Third-party, unpinned: parity unpinned for this function."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


# --------------------------------------------------------------------------- DDIM (NOT in the reference)
def ddim_loop(sd, sched, x, x_cond, timesteps, objective="pred_x0", eta=0.0, noise=None):
    """DDIM (Song et al. 2021 eq. 12 / 16) on a descending list of timesteps, with the reference's x0 clamp; eta = 0 is
    deterministic, eta > 0 adds sig_t * noise[i] with sig_t = eta sqrt((1 - abar_prev) / (1 - abar_t)) sqrt(1 - abar_t / abar_prev).
    The reference has no DDIM sampler, so this restates the published update rule, not reference
    code; it is the checker for egoego_ddim_loop only.  With eta = 1 on the full list 999..0 the update IS the
    reference's ancestral step (sig_t^2 = posterior variance, same mean), which test_oracle_golden.py checks against
    p_sample — the one tie between this sampler and the reference's chain."""
    abar = sched["alphas_cumprod"].double()
    b = x.shape[0]
    for i, t in enumerate(timesteps):
        tt = torch.full((b,), int(t), dtype=torch.long)
        out = denoise(sd, torch.cat((x, x_cond), dim=-1), tt)
        if objective == "pred_x0":
            x0 = out
        else:
            x0 = sched["sqrt_recip_alphas_cumprod"][t] * x - sched["sqrt_recipm1_alphas_cumprod"][t] * out
        x0 = x0.clamp(-1.0, 1.0)
        a_t = abar[t]
        a_prev = abar[timesteps[i + 1]] if i + 1 < len(timesteps) else torch.tensor(1.0, dtype=torch.float64)
        sig = torch.tensor(0.0, dtype=torch.float64)
        if eta > 0 and a_prev < 1 and a_t < 1:
            sig = eta * ((1 - a_prev) / (1 - a_t)).sqrt() * (1 - a_t / a_prev).clamp(min=0).sqrt()
        eps = (x - a_t.sqrt().float() * x0) / (1 - a_t).float().clamp(min=1e-20).sqrt()
        x = a_prev.sqrt().float() * x0 + (1 - a_prev - sig * sig).clamp(min=0).sqrt().float() * eps
        if eta > 0:
            x = x + sig.float() * noise[i]
    return x
