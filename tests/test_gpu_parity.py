"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle and the golden vectors the
reference produced.  Tolerance: the north star's 1e-3 max-abs on pose tensors (written below); the
split-bf16 path actually lands around 1e-5..1e-4, the int8-slice path (i8x3) around 1.5e-4.
Every oracle / golden comparison runs for both parity-grade operand precisions."""
import numpy as np
import pytest
import torch

from egoego_release_amd import ModelConfig, make_weights, make_head_windows, _lib
from egoego_release_amd.model import CondGaussianDiffusion
from oracle import egoego_oracle as O

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-3      # BASELINE.json north_star: <= 1e-3 max-abs on the final pose tensor
STAGE_TOL = 3e-4     # per-stage intermediates (values up to ~6)


@pytest.fixture(params=[_lib.PREC_BF16X3, _lib.PREC_I8X3, _lib.PREC_I8X3_FC], ids=["bf16x3", "i8x3", "i8x3fc"])
def prec(request):
    return request.param


def _model(T=120, objective="pred_x0", precision=3):
    cfg = ModelConfig(max_timesteps=T + 1, objective=objective)
    sd = make_weights(cfg, 0)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision = precision
    return cfg, sd, m.cuda()


def _ref_noise(shape, S, seed=123):
    """The draws reference.sample() makes after torch.manual_seed(seed) on CPU, in its order."""
    g = torch.Generator().manual_seed(seed)
    return {"x_T": torch.randn(shape, generator=g), "cond": torch.randn(shape, generator=g),
            "steps": torch.stack([torch.randn(shape, generator=g) for _ in range(S)])}


def test_stagewise_against_oracle(prec):
    B, T, H = 2, 120, 4
    # precision 9: the stops run the PRODUCT kernels and tap their int8 rows: 16-bit FIXED point per row, so the error of a tap scales
    # with its row's maximum (values up to ~5 -> steps of 1.5e-4).  Measured (gain_cases.py (round-4/5 experiment, removed; results: HISTORY.md), round 5): <= 2.2e-4 of the
    # row maximum at every stop, 6.4e-4 absolute at the last layer's output; split-bf16: <= 5.4e-5 absolute.
    cfg, sd, m = _model(T, precision=prec)
    eng = m.hip_engine()
    x_all = torch.randn(B, T, 396, generator=torch.Generator().manual_seed(1120))
    t = torch.tensor([3, 977])
    taps = {}
    with torch.no_grad():
        O.denoise(sd, x_all, t, taps=taps)
    xd, xcd, td = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda(), t.cuda()
    L = T + 1

    seen = {}

    def ok(got, want, name=None):
        d = (got.cpu() - want).abs()
        if prec == _lib.PREC_I8X3_FC:  # (the Q/K/V stops run split-bf16 projections, but on the product path's int8 layer input)
            rel = float((d.amax(-1) / want.abs().amax(-1).clamp_min(1.0)).max())
            seen[name] = (float(d.max()), rel)
            return rel <= 3e-4 and float(d.max()) < 8e-4
        return float(d.max()) < STAGE_TOL

    assert ok(eng.debug_stage(xd, xcd, td, 0, "embed"), taps["embed"], "embed")
    for li in (0, 3):
        lt = taps[f"layer{li}"]
        hm = lambda a: a.view(H, B, L, 256).permute(1, 0, 2, 3)
        # Q/K/V taps: the i8x3 attention kernel keeps them on-chip, so these three stops run the split-bf16 projections
        assert ok(eng.debug_stage(xd, xcd, td, li, "q"), hm(lt["q"]) / 16.0, f"{li}.q"), seen
        assert ok(eng.debug_stage(xd, xcd, td, li, "k"), hm(lt["k"]), f"{li}.k"), seen
        assert ok(eng.debug_stage(xd, xcd, td, li, "v"), hm(lt["v"]), f"{li}.v"), seen
        for st in ("attn_out", "attn_ln", "ffn_hidden", "out"):
            assert ok(eng.debug_stage(xd, xcd, td, li, st), lt[st], f"{li}.{st}"), (li, st, seen)
    print("stage taps (max abs, max relative to the row maximum):", {k: (f"{a:.1e}", f"{r:.1e}") for k, (a, r) in seen.items()})


@pytest.mark.parametrize("T,tags", [(120, ("t0", "t500", "t999", "tmix")), (30, ("t0", "tmix")), (196, ("t0", "tmix"))])
def test_denoise_against_reference_golden(golden, T, tags, prec):
    cfg, sd, m = _model(T, precision=prec)
    x_all = torch.randn(2, T, 396, generator=torch.Generator().manual_seed(int(golden[f"denoise_T{T}_seed"])))
    x, xc = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda()
    tt = {"t0": [0, 0], "t500": [500, 500], "t999": [999, 999], "tmix": [3, 977]}
    for tag in tags:
        y = m.denoise(x, torch.tensor(tt[tag]).cuda(), xc).cpu().numpy()
        assert np.abs(y - golden[f"denoise_T{T}_{tag}"]).max() < POSE_TOL, tag


def test_denoise_padding_mask_golden(golden, prec):
    cfg, sd, m = _model(precision=prec)
    x_all = torch.randn(2, 120, 396, generator=torch.Generator().manual_seed(77))
    pm = torch.ones(2, 1, 121).bool()
    pm[0, 0, 100:] = False
    pm[1, 0, 61:] = False
    y = m.denoise(x_all[..., :198].contiguous().cuda(), torch.tensor([10, 700]).cuda(),
                  x_all[..., 198:].contiguous().cuda(), padding_mask=pm.cuda()).cpu().numpy()
    assert np.abs(y - golden["denoise_padmask"]).max() < POSE_TOL


@pytest.mark.parametrize("objective", ["pred_x0", "pred_noise"])
def test_p_sample_golden(golden, objective, prec):
    cfg, sd, m = _model(objective=objective, precision=prec)
    g = torch.Generator().manual_seed(2024)
    x = torch.randn(2, 120, 198, generator=g)
    xc = torch.randn(2, 120, 198, generator=g)
    for tval in (500, 0):
        noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(555))
        x_in = x.cuda()
        y = m.p_sample(x_in, torch.full((2,), tval).cuda(), xc.cuda(), noise=noise.cuda())
        assert torch.equal(x_in.cpu(), x)  # like the reference, p_sample returns a new tensor
        assert np.abs(y.cpu().numpy() - golden[f"p_sample_{objective}_t{tval}"]).max() < POSE_TOL, (objective, tval)


@pytest.mark.parametrize("objective", ["pred_x0", "pred_noise"])
def test_surface_methods_against_oracle(objective):
    """The reference's own decomposition of a step, method by method (M:216-246): `predict_start_from_noise`, `q_posterior` and
    `p_mean_variance` (= denoiser on the HIP path + the two, with the in-place clamp) against the oracle's restatement of the same
    lines, per-row timesteps included (the API takes a [B] tensor)."""
    cfg, sd, m = _model(objective=objective)
    sched = O.make_schedule(1000)
    g = torch.Generator().manual_seed(41)
    x, xc, eps = (torch.randn(3, 120, 198, generator=g) for _ in range(3))
    t = torch.tensor([0, 417, 999])

    def gat(name):
        return sched[name].gather(-1, t).reshape(3, 1, 1)
    # predict_start_from_noise (M:216-220) and q_posterior (M:222-229): schedule gathers + elementwise arithmetic
    want_x0 = gat("sqrt_recip_alphas_cumprod") * x - gat("sqrt_recipm1_alphas_cumprod") * eps
    got_x0 = m.predict_start_from_noise(x.cuda(), t.cuda(), eps.cuda()).cpu()
    assert (got_x0 - want_x0).abs().max() <= 1e-6 * want_x0.abs().max()
    x0 = torch.rand(3, 120, 198, generator=g) * 2 - 1
    mean, var, logvar = m.q_posterior(x0.cuda(), x.cuda(), t.cuda())
    want_mean = gat("posterior_mean_coef1") * x0 + gat("posterior_mean_coef2") * x
    assert (mean.cpu() - want_mean).abs().max() <= 1e-6 * want_mean.abs().max()
    assert torch.equal(var.cpu(), gat("posterior_variance")) and torch.equal(logvar.cpu(), gat("posterior_log_variance_clipped"))
    assert var.shape == (3, 1, 1) and float(logvar[0]) == pytest.approx(-46.0517, abs=1e-3)  # (clamp at 1e-20)
    # p_mean_variance (M:231-246): the denoiser pass, the objective's x0, clamp_(-1, 1), q_posterior
    with torch.no_grad():
        out = O.denoise(sd, torch.cat((x, xc), -1), t)
    w0 = out if objective == "pred_x0" else gat("sqrt_recip_alphas_cumprod") * x - gat("sqrt_recipm1_alphas_cumprod") * out
    for clip in (True, False):
        w = w0.clamp(-1.0, 1.0) if clip else w0
        want_mean = gat("posterior_mean_coef1") * w + gat("posterior_mean_coef2") * x
        mean, var, logvar = m.p_mean_variance(x.cuda(), t.cuda(), xc.cuda(), clip_denoised=clip)
        # (pred_noise at t = 999 multiplies the denoiser's error by sqrt(1 / abar - 1) ~ 2e4 before the clamp bounds it: the clamped form is what sampling uses)
        tol = POSE_TOL if (clip or objective == "pred_x0") else POSE_TOL * float(w0.abs().max())
        assert (mean.cpu() - want_mean).abs().max() < tol, (objective, clip, float((mean.cpu() - want_mean).abs().max()))
        assert torch.equal(var.cpu(), gat("posterior_variance")) and torch.equal(logvar.cpu(), gat("posterior_log_variance_clipped"))
    with pytest.raises(ValueError, match="unknown objective"):
        m.objective = "pred_v"
        m.p_mean_variance(x.cuda(), t.cuda(), xc.cuda(), clip_denoised=True)


def test_p_sample_default_noise_uses_torch_generator():
    cfg, sd, m = _model()
    g = torch.Generator().manual_seed(8)
    x, xc = torch.randn(2, 120, 198, generator=g).cuda(), torch.randn(2, 120, 198, generator=g).cuda()
    t = torch.full((2,), 300).cuda()
    torch.manual_seed(42)
    a = m.p_sample(x, t, xc)
    torch.manual_seed(42)
    noise = torch.randn_like(x)  # the draw the reference would make at M:253 on this device
    b = m.p_sample(x, t, xc, noise=noise)
    assert torch.equal(a, b)
    want = O.p_sample(sd, O.make_schedule(1000), x.cpu(), t.cpu(), xc.cpu(), noise.cpu())
    assert (a.cpu() - want).abs().max() < POSE_TOL


@pytest.mark.parametrize("tag,B,S", [("b1_s10", 1, 10), ("b2_s50", 2, 50), ("b1_s1000", 1, 1000)])
def test_sample_chain_against_reference_golden(golden, tag, B, S, prec):
    """Full sample() with the reference's own noise draws: BASELINE config 1 (B=1, 10 steps), a
    50-step B=2 chain, and the full 1000-step chain."""
    cfg, sd, m = _model(precision=prec)
    m.num_timesteps = S  # the same truncation the reference fixture used
    xs, cm = make_head_windows(B, 120, seed=11)
    y = m.sample(xs.cuda(), cm.cuda(), noise=_ref_noise(xs.shape, S)).cpu().numpy()
    assert m.denoise_fn.training  # sample() leaves the denoiser in train mode like the reference
    err = np.abs(y - golden[f"sample_{tag}"])
    assert err.max() < POSE_TOL, (tag, err.max())
    assert np.abs(y).max() <= 1.0


def test_trajectory_stays_close_over_many_steps(prec):
    """Early divergence check: per-step max error against the oracle over a 25-step window of the chain."""
    cfg, sd, m = _model(precision=prec)
    eng = m.hip_engine()
    sched = O.make_schedule(1000)
    B, T = 2, 120
    xs, cm = make_head_windows(B, T, seed=5)
    nz = _ref_noise(xs.shape, 25, seed=9)
    x = nz["x_T"].clone()
    xc = xs * (1 - cm) + cm * nz["cond"]
    xd, xcd = x.cuda(), xc.cuda()
    worst = 0.0
    for i, t in enumerate(range(999, 974, -1)):
        x = O.p_sample(sd, sched, x, torch.full((B,), t), xc, nz["steps"][i])
        eng.sample_loop_(xd, xcd, t, 1, noise=nz["steps"][i:i + 1].cuda())
        worst = max(worst, (xd.cpu() - x).abs().max().item())
    assert worst < POSE_TOL, worst


def test_full_size_properties_b256(prec):
    """BASELINE config 3 size (B=256, T=120): determinism, shard invariance of the Philox noise,
    prefix in-painting, finiteness and the final clamp — properties that need no CPU oracle run."""
    cfg, sd, m = _model(precision=prec)
    eng = m.hip_engine()
    B, T = 256, 120
    xs, cm = make_head_windows(B, T, seed=21)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(xs.shape, generator=g).cuda()
    xc = (xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)).cuda()
    a, b = x0.clone(), x0.clone()
    eng.sample_loop_(a, xc, 999, 6, noise_mode=_lib.NOISE_PHILOX, seed=3)
    eng.sample_loop_(b, xc, 999, 6, noise_mode=_lib.NOISE_PHILOX, seed=3)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    c = x0.clone()
    eng.sample_loop_(c, xc, 999, 6, noise_mode=_lib.NOISE_PHILOX, seed=4)
    assert not torch.equal(a, c)
    # a shard of 32 windows at global offset 96 reproduces rows 96:128 of the full batch bit-for-bit
    sh = x0[96:128].clone()
    eng.sample_loop_(sh, xc[96:128].contiguous(), 999, 6, noise_mode=_lib.NOISE_PHILOX, seed=3, window_offset=96)
    assert torch.equal(sh, a[96:128])
    # the in-kernel noise is standard normal
    z = torch.zeros_like(x0)
    eng.sample_loop_(z, xc, 999, 1, noise_mode=_lib.NOISE_PHILOX, seed=11)
    z0 = torch.zeros_like(x0)
    eng.sample_loop_(z0, xc, 999, 1, noise_mode=_lib.NOISE_NONE)
    sigma = float(np.exp(0.5 * O.make_schedule(1000)["posterior_log_variance_clipped"][999].item()))
    n = ((z - z0) / sigma).flatten()
    assert abs(n.mean().item()) < 3e-3 and abs(n.std().item() - 1) < 3e-3
    assert abs((n ** 4).mean().item() - 3) < 0.05
    # prefix in-painting (sliding-window harness, M:395-397) overwrites the first frames after every step
    pre = torch.rand(B, 10, 198, generator=torch.Generator().manual_seed(2)).cuda() * 2 - 1
    d = x0.clone()
    eng.sample_loop_(d, xc, 999, 3, noise_mode=_lib.NOISE_PHILOX, seed=3, prefix=pre)
    assert torch.equal(d[:, :10], pre) and not torch.equal(d[:, 10:], x0[:, 10:])
    # last step of the chain returns clamp(x0): |x| <= 1
    e = x0.clone()
    eng.sample_loop_(e, xc, 0, 1, noise_mode=_lib.NOISE_PHILOX, seed=3)
    assert e.abs().max().item() <= 1.0


def test_plain_bf16_mode_runs_and_is_less_accurate():
    """precision=1 (one MFMA per product) is a reported speed mode; it must run, and it is expected NOT
    to meet the 1e-3 bar — which is why split-bf16 is the default."""
    x_all = torch.randn(2, 120, 396, generator=torch.Generator().manual_seed(1120))
    t = torch.tensor([3, 977])
    errs = {}
    for prec in (3, 1):
        cfg, sd, m = _model(precision=prec)
        with torch.no_grad():
            ref = O.denoise(sd, x_all, t)
        y = m.denoise(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda())
        errs[prec] = (y.cpu() - ref).abs().max().item()
    assert errs[3] < 2e-4 and errs[1] < 0.2 and errs[1] > 5 * errs[3], errs


def test_error_paths():
    cfg, sd, m = _model()
    eng = m.hip_engine()
    x = torch.zeros(1, 130, 198, device="cuda")
    with pytest.raises(_lib.EgoEgoHipError, match="exceeds max_timesteps"):
        eng.denoise(x, x, torch.zeros(1, dtype=torch.long, device="cuda"))
    x = torch.zeros(1, 120, 198, device="cuda")
    with pytest.raises(_lib.EgoEgoHipError, match="bad step range"):
        eng.sample_loop_(x, x, 5, 7)
    with pytest.raises(_lib.EgoEgoHipError):
        eng.denoise(x.cpu(), x, torch.zeros(1, dtype=torch.long, device="cuda"))
    m.objective = "pred_v"
    with pytest.raises(ValueError, match="unknown objective"):
        m.hip_engine()


def test_ddim_against_oracle_restatement(prec):
    """BASELINE config 4's sampler.  No reference oracle exists for DDIM (the reference only has the ancestral
    chain): the check is against oracle.ddim_loop, a restatement of the published eta=0 update."""
    cfg, sd, m = _model(precision=prec)
    sched = O.make_schedule(1000)
    B, T = 2, 120
    xs, cm = make_head_windows(B, T, seed=8)
    nz = _ref_noise(xs.shape, 1, seed=4)
    y = m.ddim_sample(xs.cuda(), cm.cuda(), n_steps=12, noise=nz).cpu()
    ts = sorted({int(round(v)) for v in np.linspace(0, 999, 12)}, reverse=True)
    xc = xs * (1 - cm) + cm * nz["cond"]
    with torch.no_grad():
        want = O.ddim_loop(sd, sched, nz["x_T"].clone(), xc, ts)
    assert (y - want).abs().max().item() < POSE_TOL
    assert y.abs().max().item() <= 1.0 + 1e-6  # last step lands on clamp(x0)
    with pytest.raises(_lib.EgoEgoHipError, match="strictly descending"):
        m.hip_engine().ddim_loop_(xs.cuda(), xs.cuda(), [5, 7])


@pytest.mark.parametrize("B,T,n_head,n_layers", [(1, 1, 4, 4), (3, 2, 4, 4), (5, 31, 4, 4), (3, 63, 4, 4), (2, 95, 4, 4), (3, 96, 4, 4), (2, 127, 4, 4),
                                                  (2, 128, 4, 4), (3, 191, 4, 4), (2, 207, 4, 4), (2, 208, 4, 4), (1, 223, 4, 4), (2, 50, 2, 1), (2, 120, 8, 2)])
def test_shape_edge_cases_against_oracle(B, T, n_head, n_layers, prec):
    """Minimum window (T=1), every key-tile boundary (L = 32/64/65/96/97/128/129; 65..128 is the one-kernel i8x3 attention layer's range), the maximum supported window
    (T=223), the 16-row packing of long windows in the int8 precisions (L = 129 / 192 / 208 pack at 208 rows per window, so odd
    windows start in the middle of a 32-row tile; L = 209 falls back to 224), odd batches, and other head / layer counts than the
    shipped checkpoint's."""
    cfg = ModelConfig(max_timesteps=T + 1, n_head=n_head, n_dec_layers=n_layers)
    sd = make_weights(cfg, 3)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision = prec  # i8x3: 64 < L <= 128 runs the one-kernel attention layer (one workgroup per window and head), other lengths the int8 projection kernels
    m = m.cuda()
    g = torch.Generator().manual_seed(100 * T + B)
    x_all = torch.randn(B, T, 396, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        want = O.denoise(sd, x_all, t, n_head=n_head)
    got = m.denoise(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda()).cpu()
    assert (got - want).abs().max().item() < POSE_TOL
    # one posterior step on top, ragged per-row timesteps included
    noise = torch.randn(B, T, 198, generator=g)
    want = O.p_sample(sd, O.make_schedule(1000), x_all[..., :198], t, x_all[..., 198:], noise, n_head=n_head)
    got = m.p_sample(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda(), noise=noise.cuda()).cpu()
    assert (got - want).abs().max().item() < POSE_TOL


@pytest.mark.parametrize("d_feats", [30, 64, 100, 248])
def test_other_feature_widths_against_oracle(d_feats, prec):
    """The embed operand has 2 * ceil8(d_feats) columns padded to 32: 26 k-blocks of 16 for the shipped d_feats = 198, which is
    what the small-batch embed kernel's chunking (8 + 8 + 8 + 2) is built for.  Other widths (4 / 8 / 14 / 32 k-blocks here)
    must take the general kernels and still match the oracle, at a small batch (where the direct-operand kernels run)."""
    B, T = 3, 40
    cfg = ModelConfig(d_feats=d_feats, max_timesteps=T + 1)
    sd = make_weights(cfg, 5)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision = prec
    m = m.cuda()
    g = torch.Generator().manual_seed(d_feats)
    x_all = torch.randn(B, T, 2 * d_feats, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        want = O.denoise(sd, x_all, t)
    got = m.denoise(x_all[..., :d_feats].contiguous().cuda(), t.cuda(), x_all[..., d_feats:].contiguous().cuda()).cpu()
    assert (got - want).abs().max().item() < POSE_TOL
    noise = torch.randn(B, T, d_feats, generator=g)
    want = O.p_sample(sd, O.make_schedule(1000), x_all[..., :d_feats], t, x_all[..., d_feats:], noise)
    got = m.p_sample(x_all[..., :d_feats].contiguous().cuda(), t.cuda(), x_all[..., d_feats:].contiguous().cuda(), noise=noise.cuda()).cpu()
    assert (got - want).abs().max().item() < POSE_TOL


def test_unsupported_shapes_fail_loudly():
    with pytest.raises(_lib.EgoEgoHipError, match="d_model"):
        m = CondGaussianDiffusion(198, 256, 4, 4, 256, 256, 121, 198, objective="pred_x0").cuda()
        m.hip_engine()
    cfg = ModelConfig(max_timesteps=300)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs()).cuda()
    x = torch.zeros(1, 224, 198, device="cuda")
    with pytest.raises(_lib.EgoEgoHipError, match="not supported"):
        m.denoise(x, torch.zeros(1, dtype=torch.long, device="cuda"), x)


def test_batch_size_invariance_covers_large_batch_kernels(prec):
    """Windows are independent, and every kernel accumulates a given output element in the same order whatever
    its tiling: the first windows of a B=256 run (fused QKV+attention, fused layer tail, 128-token LayerNorm
    tiles) must equal a B=3 run (unfused kernels, 64-token tiles) — and the B=3 run is checked against the oracle."""
    cfg, sd, m = _model(precision=prec)
    eng = m.hip_engine()
    g = torch.Generator().manual_seed(77)
    B = 256
    x = torch.randn(B, 120, 198, generator=g)
    xc = torch.randn(B, 120, 198, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    big = m.denoise(x.cuda(), t.cuda(), xc.cuda()).cpu()
    small = m.denoise(x[:3].contiguous().cuda(), t[:3].cuda(), xc[:3].contiguous().cuda()).cpu()
    assert (big[:3] - small).abs().max().item() <= 1e-6
    with torch.no_grad():
        want = O.denoise(sd, torch.cat((x[:3], xc[:3]), -1), t[:3])
    assert (small - want).abs().max().item() < POSE_TOL
    # a few steps of the chain as well (posterior epilogue + embed operand refresh at full size)
    a, b = x.cuda().clone(), x[:3].contiguous().cuda().clone()
    nz = torch.randn(4, B, 120, 198, generator=g)
    eng.sample_loop_(a, xc.cuda(), 500, 4, noise=nz.cuda())
    eng.sample_loop_(b, xc[:3].contiguous().cuda(), 500, 4, noise=nz[:, :3].contiguous().cuda())
    assert (a[:3] - b).abs().max().item() <= 1e-6


def test_small_grid_kernels_give_the_bits_of_the_large_grid_ones():
    """Default precision: which kernels run depends on the batch — up to 10 windows the attention projections as six workgroups per
    (window, head) and a core launch, up to 21 windows as three, up to 32 as two (1.5 projections each; round 3's two half-query attention workgroups covered 22..24), and the eight-wave tail, up to 64 windows the full attention workgroup and the eight-wave tail, beyond that
    the two-workgroups-per-CU tail, embed / linear_out on the direct-operand kernels up to 128 windows — and all of them must
    produce the SAME bits for a window (integer contractions; one summation order for the LayerNorm and softmax row sums)."""
    cfg, sd, m = _model(precision=_lib.PREC_I8X3_FC)
    eng = m.hip_engine()
    g = torch.Generator().manual_seed(5)
    B = 256
    x = torch.randn(B, 120, 198, generator=g).cuda()
    xc = torch.randn(B, 120, 198, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    big = m.denoise(x, t, xc)
    for n in (1, 3, 10, 11, 16, 21, 22, 24, 25, 32, 33, 64, 100):  # (22..32: the two-workgroup projection split + core, round 4)
        small = m.denoise(x[:n].contiguous(), t[:n].contiguous(), xc[:n].contiguous())
        assert torch.equal(small, big[:n]), n
    a = x.clone()
    eng.sample_loop_(a, xc, 999, 5, noise_mode=_lib.NOISE_PHILOX, seed=9)
    for n in (2, 10, 11, 21, 24, 32, 48):
        b = x[:n].contiguous().clone()
        eng.sample_loop_(b, xc[:n].contiguous(), 999, 5, noise_mode=_lib.NOISE_PHILOX, seed=9)
        assert torch.equal(b, a[:n]), n


@pytest.mark.parametrize("T", [120, 196])
def test_fc_weights_as_three_slices(T):
    """EGOEGO_FLAG_FC24 (precision 9): fc's weights as three int8 slices — a second contraction per feature pass with the third slice.
    On the reference's initialisation the denoiser stays inside precision 9's error (the third slice only removes weight rounding),
    the result differs from the two-slice engine (the pass really runs), a window has the same bits in a small call (eight-wave tail)
    and in a large one (two 256-register workgroups per CU), and the flag is refused for any other precision."""
    from egoego_release_amd.engine import HipEngine
    from egoego_release_amd.precision import _engine_cfg
    cfg, sd, m = _model(T=T, precision=_lib.PREC_I8X3_FC)
    ecfg = _engine_cfg(m)
    dev = torch.device("cuda")
    e24 = HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3_FC, _lib.FLAG_NO_GRAPH | _lib.FLAG_FC24)
    e16 = HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3_FC, _lib.FLAG_NO_GRAPH)
    try:
        g = torch.Generator().manual_seed(11)
        B = 200
        x = torch.randn(B, T, 198, generator=g)
        xc = torch.randn(B, T, 198, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        with torch.no_grad():
            want = O.denoise(sd, torch.cat((x[:2], xc[:2]), -1), t[:2])
        big = e24.denoise(x.cuda(), xc.cuda(), t.cuda())
        small = e24.denoise(x[:3].contiguous().cuda(), xc[:3].contiguous().cuda(), t[:3].contiguous().cuda())
        plain = e16.denoise(x[:3].contiguous().cuda(), xc[:3].contiguous().cuda(), t[:3].contiguous().cuda())
        assert torch.equal(small, big[:3])
        assert not torch.equal(small, plain)
        d24, d16 = small[:2].cpu() - want, plain[:2].cpu() - want
        e24_err, e16_err = float(d24.abs().max()), float(d16.abs().max())
        r24, r16 = float(d24.pow(2).mean().sqrt()), float(d16.pow(2).mean().sqrt())
        print(f"T={T}: max error fc24 {e24_err:.2e} / two slices {e16_err:.2e}; rms {r24:.2e} / {r16:.2e}")
        # the maxima are single elements and move either way by a third; the rms may only fall (the third slice removes rounding)
        assert e24_err < 6e-4 and r24 < 1.03 * r16
    finally:
        e24.close()
        e16.close()
    with pytest.raises(_lib.EgoEgoHipError):
        HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3, _lib.FLAG_FC24)


@pytest.mark.parametrize("T", [120, 196])
def test_ffn_on_split_bf16_in_precision_8(T):
    """EGOEGO_FLAG_FFN16 (precision 8): the FFN contractions on split-bf16, int8 slices in the attention layer only.  The denoiser's error
    against the oracle may only fall against plain precision 8 (two int8 sites fewer), the result differs from it (the flag is honoured), a
    window has the same bits in a small call (the direct-operand tail) and in a large one (the fused 64-token tail), the stage taps of
    the FFN read the split-bf16 tensors, and the flag is refused for any other precision."""
    from egoego_release_amd.engine import HipEngine
    from egoego_release_amd.precision import _engine_cfg
    cfg, sd, m = _model(T=T, precision=_lib.PREC_I8X3)
    ecfg = _engine_cfg(m)
    dev = torch.device("cuda")
    en = HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3, _lib.FLAG_NO_GRAPH | _lib.FLAG_FFN16)
    e8 = HipEngine(ecfg, m.state_dict(), dev, _lib.PREC_I8X3, _lib.FLAG_NO_GRAPH)
    try:
        g = torch.Generator().manual_seed(12)
        B = 200
        x = torch.randn(B, T, 198, generator=g)
        xc = torch.randn(B, T, 198, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        taps = {}
        with torch.no_grad():
            want = O.denoise(sd, torch.cat((x[:2], xc[:2]), -1), t[:2], taps=taps)
        big = en.denoise(x.cuda(), xc.cuda(), t.cuda())
        sm = [v[:3].contiguous().cuda() for v in (x, xc, t)]
        small = en.denoise(*sm)
        plain = e8.denoise(*sm)
        assert torch.equal(small, big[:3]) and not torch.equal(small, plain)
        dn, d8 = small[:2].cpu() - want, plain[:2].cpu() - want
        print(f"T={T}: max error ffn16 {float(dn.abs().max()):.2e} / precision 8 {float(d8.abs().max()):.2e}; rms {float(dn.pow(2).mean().sqrt()):.2e} / {float(d8.pow(2).mean().sqrt()):.2e}")
        assert float(dn.abs().max()) < 3e-4 and float(dn.pow(2).mean().sqrt()) < 1.03 * float(d8.pow(2).mean().sqrt())
        two = [v[:2].contiguous().cuda() for v in (x, xc, t)]
        for st in ("attn_ln", "ffn_hidden", "out"):
            got = en.debug_stage(two[0], two[1], two[2], 1, st).cpu()
            assert float((got - taps["layer1"][st]).abs().max()) < 3e-4, st
        y = x[:70].contiguous().cuda()
        en.sample_loop_(y, xc[:70].contiguous().cuda(), 999, 4, noise_mode=_lib.NOISE_PHILOX, seed=5)
        z = x[:3].contiguous().cuda()
        en.sample_loop_(z, xc[:3].contiguous().cuda(), 999, 4, noise_mode=_lib.NOISE_PHILOX, seed=5)
        assert torch.equal(z, y[:3])
    finally:
        en.close()
        e8.close()
    for bad in (_lib.PREC_I8X3_FC, _lib.PREC_BF16X3):
        with pytest.raises(_lib.EgoEgoHipError):
            HipEngine(ecfg, m.state_dict(), dev, bad, _lib.FLAG_FFN16)


@pytest.mark.filterwarnings("ignore:LayerNorm gains span")  # (explicit int8 precisions on such a checkpoint: the module says so)
@pytest.mark.parametrize("T", [120, 196, 48])
def test_outlier_heavy_weights_stay_within_the_bar(prec, T):
    """The synthetic weights follow the reference's initialisation; a trained checkpoint may not.  One scale per row makes
    the int8-slice precision sensitive to outliers in principle, so this pins its behaviour on a hostile variant: 8x
    LayerNorm gains on six features of every LayerNorm, shifted LayerNorm biases and 4x heavy tails on 0.2 % of the Q/K/V
    projection weights (tools/hostile_weights_check.py sweeps further: 25x gains and 12x tails together give 8e-4).
    T = 120: the one-kernel attention layer; T = 196: the long-window pair (V scaled per key); T = 48: int8 projections
    in front of the split-bf16 attention core."""
    cfg = ModelConfig(max_timesteps=T + 1)
    sd = make_weights(cfg, 0)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):
        if "layer_norm.weight" in k:
            sd[k] = sd[k].clone()
            sd[k][torch.randperm(512, generator=g)[:6]] *= 8.0
        if "layer_norm.bias" in k:
            sd[k] = sd[k].clone() + 0.5 * torch.randn(512, generator=g)
        if any(s in k for s in ("w_q.weight", "w_k.weight", "w_v.weight")):
            w = sd[k].clone()
            w[torch.rand(w.shape, generator=g) < 0.002] *= 4.0
            sd[k] = w
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m.hip_precision = prec
    m = m.cuda()
    x_all = torch.randn(2, T, 396, generator=g)
    t = torch.tensor([3, 977])
    with torch.no_grad():
        want = O.denoise(sd, x_all, t)
    got = m.denoise(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda()).cpu()
    err = (got - want).abs().max().item()
    assert err < (1e-4 if prec == _lib.PREC_BF16X3 else 6e-4), (T, err)


def test_degenerate_inputs_give_finite_results(prec):
    """All-zero pose and condition tensors, and all-zero Q/K/V projection weights and biases (every row maximum the int8
    quantisers divide by is then zero): results stay finite and match the oracle."""
    for T in (120, 196):
        cfg = ModelConfig(max_timesteps=T + 1)
        sd = make_weights(cfg, 0)
        for k in list(sd):
            if any(s in k for s in ("w_q.", "w_k.", "w_v.")) and "layer_stack.1." in k:
                sd[k] = torch.zeros_like(sd[k])
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(sd, strict=False)
        m.hip_precision = prec
        m = m.cuda()
        x_all = torch.zeros(2, T, 396)
        t = torch.tensor([0, 999])
        with torch.no_grad():
            want = O.denoise(sd, x_all, t)
        got = m.denoise(x_all[..., :198].contiguous().cuda(), t.cuda(), x_all[..., 198:].contiguous().cuda()).cpu()
        assert torch.isfinite(got).all()
        err = (got - want).abs().max().item()
        assert err < (5e-4 if prec == _lib.PREC_I8X3_FC else 3e-4), (T, err)


def test_outlier_heavy_layernorm_gains_step_the_default_precision_down():
    """One scale per row is 16-bit fixed point: LayerNorm gains far above the rest cost the other features their bits
    (DESIGN.md 3c).  `hip_precision = "auto"` (the default) MEASURES each checkpoint when it is packed (plan.resolve / plan.run_ladder:
    int8 slices against split-bf16 on a probe batch): the reference's initialisation runs precision 9 with no warning; the SAME
    six features amplified 25x in every LayerNorm (the worst case found, tools/gain_sweep.py) — and already 3x — step down to
    split-bf16, with a warning, and the stepped-down result is inside the bar; 2x runs precision 8 as is, silently, inside the bar
    (every outcome below is what gain_cases.py (round-4/5 experiment, removed; results: HISTORY.md) measured in round 5, pinned); an explicit int8 precision is kept and
    warned about."""
    import warnings
    cfg = ModelConfig(max_timesteps=121)
    sd = make_weights(cfg, 0)
    g = torch.Generator().manual_seed(3)
    x_all = torch.randn(2, 120, 396, generator=g)
    t = torch.tensor([7, 900])
    xa, xb = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda()

    def build(state, prec=None):
        m = CondGaussianDiffusion(**cfg.ctor_kwargs())
        m.load_state_dict(state, strict=False)
        if prec is not None:
            m.hip_precision = prec
        return m.cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the reference's initialisation: no warning, the int8 default
        m = build(sd)
        m.denoise(xa, t.cuda(), xb)
        assert m.hip_precision == "auto" and m.hip_precision_used == _lib.PREC_I8X3_FC
        pr = m.hip_precision_probe
        assert pr["errors"]["9 as is"] <= pr["limit"] and not pr["prepared"] and len(pr["row_max"]) == 8 and min(pr["row_max"]) > 1.0
    hot = {k: v.clone() for k, v in sd.items()}
    for k in hot:
        if k.endswith("layer_norm.weight"):
            hot[k][:6] *= 25.0
    with torch.no_grad():
        want = O.denoise(hot, x_all, t)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m = build(hot)
        got = m.denoise(xa, t.cuda(), xb).cpu()
    pr = m.hip_precision_probe
    # every int8 form measures outside the limit on this checkpoint (round 5: 2.0 each — the rows' one scale is spent on six features):
    # split-bf16 runs, with a warning
    assert m.hip_precision_used == _lib.PREC_BF16X3 and any("falling back to split-bf16" in str(w.message) for w in rec)
    assert {"9 as is", "9 prepared", "9 prepared + fc24", "8 as is", "8 prepared", "8 prepared + ffn16"} == set(pr["errors"]) and min(pr["errors"].values()) > pr["limit"]
    # (these gains blow the outputs up to |y| ~ 50: the bar relative to that)
    assert (got - want).abs().max().item() < POSE_TOL * max(1.0, want.abs().max().item())
    # gains of 3x: the same outcome (measured: the int8 forms 1.3e-3 ... 2.9e-3 on the probe), and the stepped-down result is inside the bar
    warm = {k: v.clone() for k, v in sd.items()}
    for k in warm:
        if k.endswith("layer_norm.weight"):
            warm[k][:6] *= 3.0
    with torch.no_grad():
        want = O.denoise(warm, x_all, t)
    with pytest.warns(RuntimeWarning, match="falling back to split-bf16"):
        m = build(warm)
        got = m.denoise(xa, t.cuda(), xb).cpu()
    assert m.hip_precision_used == _lib.PREC_BF16X3 and min(m.hip_precision_probe["errors"].values()) > m.hip_precision_probe["limit"]
    assert (got - want).abs().max().item() < POSE_TOL
    # gains of 2x: precision 9 fails both packings (1.2e-3 / 7.7e-4 / 8.2e-4), precision 8 as is measures 4.5e-4 and 4.6e-4 on the whole
    # chains: it runs, without a warning, ~40 % slower than 9, inside the bar
    mild = {k: v.clone() for k, v in sd.items()}
    for k in mild:
        if k.endswith("layer_norm.weight"):
            mild[k][:6] *= 2.0
    with torch.no_grad():
        want = O.denoise(mild, x_all, t)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m = build(mild)
        got = m.denoise(xa, t.cuda(), xb).cpu()
    pr = m.hip_precision_probe
    assert m.hip_precision_used == _lib.PREC_I8X3 and pr["form"] == "as is", pr
    assert pr["errors"]["9 as is"] > pr["limit"] and pr["errors"]["8 as is"] <= pr["limit"] and pr["errors"]["8 as is, full chain"] <= pr["chain_limit"]
    assert (got - want).abs().max().item() < POSE_TOL * max(1.0, want.abs().max().item())
    with pytest.warns(RuntimeWarning, match="differs from split-bf16"):  # an explicit int8 precision, packed as is, is kept and warned about
        m = build(hot, _lib.PREC_I8X3_FC)
        m.hip_int8_prep = "never"
        m.denoise(xa, t.cuda(), xb)
    assert m.hip_precision_used == _lib.PREC_I8X3_FC


def test_graph_replay_equals_individual_launches(prec):
    """egoego_sample_loop captures one step into a hipGraph and replays it (timestep and step index live in device
    memory); the result must be bit-identical to launching every kernel of every step, for Philox and for injected
    noise, with prefix in-painting, across repeated calls that reuse the cached graph, and for the DDIM loop."""
    B, T, S = 5, 120, 9
    xs, cm = make_head_windows(B, T, seed=2)
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(xs.shape, generator=g).cuda()
    xc = (xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)).cuda()
    nz = torch.randn(S, B, T, 198, generator=g).cuda()
    pre = (torch.rand(B, 10, 198, generator=g) * 2 - 1).cuda()
    res = {}
    for graph in (True, False):
        cfg, sd, m = _model(precision=prec)
        m.hip_graph = graph
        eng = m.hip_engine()
        out = []
        for rep in range(2):  # the second round hits the graph cache
            a = x0.clone()
            eng.sample_loop_(a, xc, 999, S, noise_mode=_lib.NOISE_PHILOX, seed=3)
            b = x0.clone()
            eng.sample_loop_(b, xc, 400, S, noise=nz, prefix=pre)
            c = x0.clone()
            eng.ddim_loop_(c, xc, [900, 700, 500, 300, 100, 0])
            out += [a, b, c]
        assert torch.equal(out[0], out[3]) and torch.equal(out[1], out[4]) and torch.equal(out[2], out[5])
        res[graph] = out[:3]
    for u, v in zip(res[True], res[False]):
        assert torch.equal(u, v)


def test_captured_step_serves_other_buffers_and_seeds(prec):
    """The caller's x / noise / prefix pointers and the Philox key reach the kernels through the device-resident step
    state, so a step captured for one call replays for calls on OTHER tensors, seeds and window offsets: results must
    equal those of a fresh engine that never saw the first call, and of the no-graph engine."""
    B, T, S = 3, 120, 6
    xs, cm = make_head_windows(B, T, seed=12)
    g = torch.Generator().manual_seed(16)
    xa, xb = torch.randn(xs.shape, generator=g).cuda(), torch.randn(xs.shape, generator=g).cuda()
    xc = (xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)).cuda()
    nz1, nz2 = torch.randn(S, B, T, 198, generator=g).cuda(), torch.randn(S, B, T, 198, generator=g).cuda()
    cfg, sd, m = _model(precision=prec)
    eng = m.hip_engine()
    r = []
    for x0, seed, off, nz in ((xa, 3, 0, nz1), (xb, 11, 64, nz2)):
        a = x0.clone()
        eng.sample_loop_(a, xc, 999, S, noise_mode=_lib.NOISE_PHILOX, seed=seed, window_offset=off)
        b = x0.clone()
        eng.sample_loop_(b, xc, 500, S, noise=nz.clone())
        r.append((a, b))
    cfg, sd, m2 = _model(precision=prec)
    m2.hip_graph = False
    e2 = m2.hip_engine()
    a = xb.clone()
    e2.sample_loop_(a, xc, 999, S, noise_mode=_lib.NOISE_PHILOX, seed=11, window_offset=64)
    b = xb.clone()
    e2.sample_loop_(b, xc, 500, S, noise=nz2)
    assert torch.equal(r[1][0], a) and torch.equal(r[1][1], b)
    assert not torch.equal(r[0][0], r[1][0])


def test_p_sample_loop_with_padding_mask(golden, prec):
    """M:259, 268: p_sample_loop(padding_mask=...) masks every step's denoiser pass.  Fixture: the reference's own 10-step
    chain with its own draws; then the default torch-RNG path must consume the generator like the reference does and run as
    one HIP loop (same result as feeding the same draws through p_sample step by step)."""
    cfg, sd, m = _model(precision=prec)
    m.num_timesteps = 10
    xs, cm = make_head_windows(2, 120, seed=13)
    pm = torch.ones(2, 1, 121).bool()
    pm[0, 0, 100:] = False
    pm[1, 0, 61:] = False
    y = m.p_sample_loop(xs.shape, xs.cuda(), cm.cuda(), padding_mask=pm.cuda(), noise=_ref_noise(xs.shape, 10, seed=321)).cpu().numpy()
    assert np.abs(y - golden["sample_padmask_b2_s10"]).max() < POSE_TOL
    xs, cm, pm = xs.cuda(), cm.cuda(), pm.cuda()
    torch.manual_seed(5)
    got = m.p_sample_loop(xs.shape, xs, cm, padding_mask=pm)
    torch.manual_seed(5)
    x = torch.randn(xs.shape, device="cuda")
    xc = xs * (1.0 - cm) + cm * torch.randn_like(xs)
    for i in reversed(range(10)):
        x = m.p_sample(x, torch.full((2,), i, device="cuda", dtype=torch.long), xc, padding_mask=pm)
    assert torch.equal(got, x)
    unmasked = m.p_sample_loop(xs.shape, xs, cm, noise=_ref_noise(xs.shape, 10, seed=321))
    assert np.abs(unmasked.cpu().numpy() - golden["sample_padmask_b2_s10"]).max() > 1e-2  # the mask matters


def test_ddim_eta1_full_chain_is_the_ddpm_chain(prec):
    """SURVEY.md §8f #3's self-consistency check.  DDIM with eta = 1 on ALL timesteps 999..0 is algebraically the
    reference's ancestral chain (sig_t^2 = posterior variance, DDIM mean = posterior mean), so with the same Philox
    key it must land on egoego_sample_loop's result up to rounding — the tie between the strided sampler and the
    chain the reference pins.  Also eta = 0.5 on a 12-step stride against the oracle restatement with injected noise."""
    cfg, sd, m = _model(precision=prec)
    eng = m.hip_engine()
    B, T = 2, 120
    xs, cm = make_head_windows(B, T, seed=21)
    g = torch.Generator().manual_seed(22)
    x0 = torch.randn(xs.shape, generator=g).cuda()
    xc_cpu = xs * (1 - cm) + cm * torch.randn(xs.shape, generator=g)
    xc = xc_cpu.cuda()
    a = x0.clone()
    eng.sample_loop_(a, xc, 999, 1000, noise_mode=_lib.NOISE_PHILOX, seed=9, window_offset=5)
    b = x0.clone()
    eng.ddim_loop_(b, xc, list(range(999, -1, -1)), eta=1.0, seed=9, window_offset=5)
    assert (a - b).abs().max().item() < 3e-4, (a - b).abs().max().item()
    c = x0.clone()
    eng.ddim_loop_(c, xc, list(range(999, -1, -1)), eta=0.0)
    assert (a - c).abs().max().item() > 1e-3  # eta matters
    ts = sorted({int(round(v)) for v in np.linspace(0, 999, 12)}, reverse=True)
    nz = torch.randn(len(ts), B, T, 198, generator=g)
    d = x0.clone()
    eng.ddim_loop_(d, xc, ts, eta=0.5, noise=nz.cuda())
    with torch.no_grad():
        want = O.ddim_loop(sd, O.make_schedule(1000), x0.cpu(), xc_cpu, ts, eta=0.5, noise=nz)
    assert (d.cpu() - want).abs().max().item() < POSE_TOL
    with pytest.raises(_lib.EgoEgoHipError, match="eta"):
        eng.ddim_loop_(d, xc, ts, eta=1.5)


def test_default_torch_rng_path_draws_like_the_reference():
    """sampling_rng='torch' (the drop-in default): after torch.manual_seed(s), sample() consumes the device generator
    exactly as the reference's loop does — x_T, condition noise, then one randn_like(x) per step, t = S-1 .. 0
    (M:263-268) — although the draws are made a chunk of steps ahead.  Checked against the same draws made by hand
    and fed through p_sample step by step."""
    cfg, sd, m = _model()
    S, B, T = 7, 3, 120
    m.num_timesteps = S
    xs, cm = make_head_windows(B, T, seed=4)
    xs, cm = xs.cuda(), cm.cuda()
    torch.manual_seed(77)
    got = m.sample(xs, cm)
    torch.manual_seed(77)
    x = torch.randn(xs.shape, device="cuda")
    xc = xs * (1.0 - cm) + cm * torch.randn_like(xs)
    for i in reversed(range(S)):
        x = m.p_sample(x, torch.full((B,), i, device="cuda", dtype=torch.long), xc)  # draws randn_like(x) itself
    assert torch.equal(got, x)


def test_in_place_weight_update_is_picked_up():
    """`p.data.copy_` (ema_pytorch) changes neither data_ptr nor _version; sample() must still use the new
    weights (device-side fingerprint), and invalidate_engine() serves the per-step API."""
    cfg, sd, m = _model()
    xs, cm = make_head_windows(2, 120, seed=9)
    m.num_timesteps = 3
    nz = _ref_noise(xs.shape, 3)
    a = m.sample(xs.cuda(), cm.cuda(), noise=nz)
    w = m.denoise_fn.linear_out.weight
    w.data.copy_(w.data * 0.5)
    b = m.sample(xs.cuda(), cm.cuda(), noise=nz)
    assert not torch.equal(a, b)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["denoise_fn.linear_out.weight"] = sd["denoise_fn.linear_out.weight"] * 0.5
    with torch.no_grad():
        x = nz["x_T"].clone()
        xc = xs * (1 - cm) + cm * nz["cond"]
        for i, t in enumerate((2, 1, 0)):
            x = O.p_sample(sd2, O.make_schedule(1000), x, torch.full((2,), t), xc, nz["steps"][i])
    assert (b.cpu() - x).abs().max().item() < POSE_TOL
    t = torch.zeros(2, dtype=torch.long, device="cuda")
    y0 = m.denoise(xs.cuda(), t, xs.cuda())
    w.data.mul_(2.0)
    m.invalidate_engine()
    y1 = m.denoise(xs.cuda(), t, xs.cuda())
    assert not torch.equal(y0, y1)
    with pytest.raises(IndexError, match="out of range"):
        m.denoise(xs.cuda(), t + 1000, xs.cuda())
