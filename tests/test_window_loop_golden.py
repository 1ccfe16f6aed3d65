"""The sliding-window LOOP and the post-loop CONVERSION CHAIN pinned to a run of the reference's own code.

tests/golden/window_loop_golden.npz was written by tests/golden/make_window_loop_golden.py, which EXECUTES
  CondGaussianDiffusion.sample_sliding_window_w_canonical / p_sample_loop_sliding_window_w_canonical   M:548-555, 329-467
  CondGaussianDiffusion.convert_model_res_to_data                                                      M:469-525
  quat_ik_torch, AMASSDataset.fk_smpl / normalize / de_normalize (bound to a stand-in ds), rotate_at_frame_smplh
of /root/reference on the reference's demo head trajectory (140 frames: windows of 120 + 30) and the real window-120 statistics —
only the nine pytorch3d.transforms function BODIES (absent from the image) were supplied from scipy there, so those bodies stay
unpinned, the loop's control flow, draw order, in-painting, stitching and re-canonicalisation do not.

CPU part: oracle/harness_oracle.py and the product's torch chains against the fixture.
GPU part: the product harness (HIP loop + the three per-window HIP kernels) with the same draws against the fixture:
1e-3 in position (metres; normalised units for a sampled window), 1e-3 rad in rotation angle.
"""
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation as Rot

from egoego_release_amd import ModelConfig, make_weights, harness
from oracle import egoego_oracle as O
from oracle import harness_oracle as HO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wl():
    return np.load(os.path.join(ROOT, "tests", "golden", "window_loop_golden.npz"))


@pytest.fixture(scope="module")
def hg():
    return np.load(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"))


def _angle(aa_a, aa_b):
    d = Rot.from_rotvec(np.asarray(aa_a, np.float64).reshape(-1, 3)) * Rot.from_rotvec(np.asarray(aa_b, np.float64).reshape(-1, 3)).inv()
    return np.abs(d.magnitude())


def _setup(wl, hg):
    seq_len, S = int(wl["seq_len"]), int(wl["num_timesteps"])
    cfg = ModelConfig(max_timesteps=seq_len + 1)
    sd = make_weights(cfg, int(wl["weight_seed"]))
    sd["denoise_fn.linear_out.bias"] = torch.from_numpy(wl["linear_out_bias"]).float()
    sd["denoise_fn.linear_out.weight"] = sd["denoise_fn.linear_out.weight"] * float(wl["linear_out_scale"])
    lo, hi = hg["stats_global_jpos_min"], hg["stats_global_jpos_max"]
    ds = harness.SkeletonStats(lo, hi, wl["rest_offsets"], parents=tuple(int(p) for p in wl["parents"]))
    dso = HO.SkeletonOracle(lo, hi, wl["rest_offsets"])
    return cfg, sd, ds, dso, seq_len, S


def _draws(wl, b, T, spans, n_steps):
    """The reference's draws, replayed from the fixture's seed in the reference's order (M:341, 390, 253): the CPU generator is
    deterministic; the fixture's checksums say whether this torch build still produces the stream the fixture was made with."""
    torch.manual_seed(int(wl["seed"]))
    noise = {"x_all": torch.randn(b, T, 198), "cond": [], "steps": []}
    for _, n in spans:
        noise["cond"].append(torch.randn(b, n, 198))
        noise["steps"].append(torch.stack([torch.randn(b, n, 198) for _ in range(n_steps)]))
    vals = [noise["x_all"]] + noise["cond"] + noise["steps"]
    got = np.array([[float(v.double().sum()), float(v.double().abs().sum()), float(v.reshape(-1)[0]), float(v.reshape(-1)[-1])] for v in vals])
    if not np.allclose(got, wl["draw_checksums"], rtol=0, atol=1e-6):
        pytest.skip("this torch build's CPU generator does not reproduce the draws the fixture was generated with")
    return noise


# ------------------------------------------------------------------------------------------ CPU: oracle vs the reference's run
def test_oracle_sliding_window_equals_the_reference_loop(wl, hg):
    """oracle/harness_oracle.py::sliding_window against what the reference's own p_sample_loop_sliding_window_w_canonical returned,
    and the two sampled windows — bit for bit (they come out of the pinned p_sample only) — and the in-paint prefix in between."""
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    hp = wl["head_pose"]
    b, T = hp.shape[:2]
    spans = harness.window_spans(T, seq_len)
    assert spans == [(0, 120), (110, 30)] and harness.output_frames(T, seq_len) == 140
    noise = _draws(wl, b, T, spans, S)
    cm = O.head_condition_mask((b, T, 198))
    aa, root = HO.sliding_window(sd, O.make_schedule(1000), dso, seq_len, S, hp[..., :3], hp[..., 3:], cm, noise)
    assert aa.shape == wl["loop_aa"].shape and root.shape == wl["loop_root"].shape
    assert np.abs(root - wl["loop_root"]).max() < 5e-6, np.abs(root - wl["loop_root"]).max()
    assert _angle(aa, wl["loop_aa"]).max() < 5e-6
    # window 0 of the loop is a plain conditioned chain: the oracle's p_sample on the replayed draws gives the reference's window
    sched = O.make_schedule(1000)
    a_t, a_q, rec = HO.rotate_at_frame_smplh(hp[:, :120, :3], hp[:, :120, 3:], 0)
    assert np.array_equal(rec, wl["w0_recover"])
    mv = a_t[:, 0:1].copy()
    mv[:, :, 2] = 0
    xs = np.zeros((b, 120, 198))
    xs[:, :, 45:48] = a_t - mv
    xs[:, :, 156:162] = HO.quat_to_mat(a_q)[..., :2, :].reshape(b, -1, 6)
    xs[:, :, :66] = dso.norm(xs[:, :, :66].reshape(-1, 22, 3)).reshape(b, -1, 66)
    xs = torch.from_numpy(xs).float()
    xc = xs * (1.0 - cm[:, :120]) + cm[:, :120] * noise["cond"][0]
    x = noise["x_all"][:, :120].clone()
    for i, t in enumerate(reversed(range(S))):
        x = O.p_sample(sd, sched, x, torch.full((b,), t, dtype=torch.long), xc, noise["steps"][0][i])
    assert np.abs(x.numpy() - wl["w0_x"]).max() < 2e-6, np.abs(x.numpy() - wl["w0_x"]).max()


def test_oracle_conversion_chain_equals_the_reference(wl, hg):
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    for tag, tol in (("conv", 1e-6), ("conv_rand", 1e-6), ("w0", 2e-6), ("w1", 2e-6)):
        x, rec = (wl[f"{tag}_x"], wl[f"{tag}_recover"])
        aa, root, head = HO.convert_model_res_to_data(dso, x.astype(np.float64), rec)
        assert np.abs(root - wl[f"{tag}_root"]).max() < tol and np.abs(head - wl[f"{tag}_head"]).max() < tol, tag
        assert _angle(aa, wl[f"{tag}_aa"]).max() < 1e-6, tag
    gq, gj = dso.fk(hg["demo_trans"], np.concatenate([hg["demo_root_orient"][:, None], hg["demo_body_pose"].reshape(-1, 21, 3)], 1))
    assert np.abs(gj - wl["fk_demo_jpos"]).max() < 1e-6 and np.abs(gq - wl["fk_demo_quat"]).max() < 1e-6
    for w in (0, 1):
        gq, gj = dso.fk(wl[f"w{w}_fk_root"], wl[f"w{w}_fk_aa"])
        assert np.abs(gj - wl[f"w{w}_fk_jpos"]).max() < 2e-6 and np.abs(gq - wl[f"w{w}_fk_quat"]).max() < 1e-6


# ------------------------------------------------------------------------------------------ CPU: product's torch chains
def test_product_torch_chains_equal_the_reference(wl, hg):
    """harness.convert_model_res_to_data (CPU tensors: the torch expressions over rotations.py) and SkeletonStats.fk_smpl."""
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    for tag in ("conv", "conv_rand", "w0", "w1"):
        aa, root, head = harness.convert_model_res_to_data(ds, torch.from_numpy(wl[f"{tag}_x"]), wl[f"{tag}_recover"])
        assert np.abs(root.numpy() - wl[f"{tag}_root"]).max() < 2e-6 and np.abs(head.numpy() - wl[f"{tag}_head"]).max() < 2e-6, tag
        ang = _angle(aa.numpy(), wl[f"{tag}_aa"])
        # single rows of the uniform random windows have nearly parallel 6D halves: fp32 Gram-Schmidt (the product, like pytorch3d)
        # against the fixture's float64 stand-in moves those by up to ~1e-3 rad; every well-conditioned row sits at 1e-6
        assert ang.max() < (2e-3 if tag == "conv_rand" else 2e-5) and np.median(ang) < 1e-6, (tag, ang.max())
    for w in (0, 1):
        gq, gj = ds.fk_smpl(torch.from_numpy(wl[f"w{w}_fk_root"]).float(), torch.from_numpy(wl[f"w{w}_fk_aa"]).float())
        assert np.abs(gj.numpy() - wl[f"w{w}_fk_jpos"]).max() < 2e-6
        d = np.minimum(np.abs(gq.numpy() - wl[f"w{w}_fk_quat"]), np.abs(gq.numpy() + wl[f"w{w}_fk_quat"])).max()
        assert d < 2e-6


# ------------------------------------------------------------------------------------------ GPU: the product harness
def _model(cfg, sd, S):
    from egoego_release_amd.model import CondGaussianDiffusion
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    m.num_timesteps = S
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["auto", 3])
def test_gpu_harness_equals_the_reference_loop(wl, hg, precision):
    """full_body_gen_cond_head_pose_sliding_window on the GPU (one HIP loop per window with the prefix in-paint inside, the
    three per-window HIP kernels) with the reference's draws against the reference's own result."""
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    m = _model(cfg, sd, S)
    m.hip_precision = precision
    hp = torch.from_numpy(wl["head_pose"])
    b, T = hp.shape[:2]
    noise = _draws(wl, b, T, harness.window_spans(T, seq_len), S)
    aa, root = harness.full_body_gen_cond_head_pose_sliding_window(m, ds, hp.float().cuda(), noise=noise)
    assert aa.shape == wl["loop_aa"].shape and root.shape == wl["loop_root"].shape
    d_root = np.abs(root.cpu().numpy() - wl["loop_root"]).max()
    ang = _angle(aa.cpu().numpy(), wl["loop_aa"])
    assert d_root < 1e-3, d_root
    assert ang.max() < 1e-3, (ang.max(), np.median(ang))
    # the same through the module's method, like trainer:273 calls it
    data = torch.zeros(b, T, 198, device="cuda")
    aa2, root2 = m.sample_sliding_window_w_canonical(ds, hp[..., :3].float().cuda(), hp[..., 3:].float().cuda(), data,
                                                    harness.prep_head_condition_mask(data), noise=noise)
    assert torch.equal(aa2, aa) and torch.equal(root2, root)
    assert m.denoise_fn.training


@pytest.mark.gpu
def test_gpu_sampled_windows_equal_the_reference(wl, hg):
    """The two windows the reference's loop sampled (what it handed to convert_model_res_to_data), from p_sample_loop with the
    reference's draws: window 0 a plain conditioned chain, window 1 with the 10-frame prefix overwritten after every step."""
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    m = _model(cfg, sd, S)
    hp = torch.from_numpy(wl["head_pose"]).float().cuda()
    b, T = hp.shape[:2]
    spans = harness.window_spans(T, seq_len)
    noise = _draws(wl, b, T, spans, S)
    for w, (t0, n) in enumerate(spans):
        x_start, rec = harness._window_condition_hip(ds, hp[:, t0:t0 + n, :3], hp[:, t0:t0 + n, 3:])
        assert np.abs(rec.cpu().numpy() - wl[f"w{w}_recover"]).max() < 1e-6
        cm = harness.prep_head_condition_mask(x_start)
        prefix = None
        if w > 0:
            prefix = torch.from_numpy(np.concatenate([wl["w1_prefix_jpos"], wl["w1_prefix_6d"]], -1)).float().cuda()
        got = m.p_sample_loop(x_start.shape, x_start, cm, noise={"x_T": noise["x_all"][:, t0:t0 + n], "cond": noise["cond"][w], "steps": noise["steps"][w]},
                              prefix=prefix)
        d = np.abs(got.cpu().numpy() - wl[f"w{w}_x"]).max()
        assert d < 1e-3, (w, d)


@pytest.mark.gpu
def test_gpu_per_window_kernels_equal_the_reference(wl, hg):
    """egoego_convert_model_res and egoego_window_prefix on what the reference's instrumented calls saw."""
    cfg, sd, ds, dso, seq_len, S = _setup(wl, hg)
    for tag in ("conv", "w0", "w1"):
        aa, root, head = harness.convert_model_res_to_data(ds, torch.from_numpy(wl[f"{tag}_x"]).cuda(), torch.from_numpy(wl[f"{tag}_recover"]).cuda())
        assert aa.is_cuda
        assert np.abs(root.cpu().numpy() - wl[f"{tag}_root"]).max() < 1e-5 and np.abs(head.cpu().numpy() - wl[f"{tag}_head"]).max() < 1e-5, tag
        assert _angle(aa.cpu().numpy(), wl[f"{tag}_aa"]).max() < 1e-4, tag
    # window 0's conversion -> the prefix window 1 is in-painted with (fk_smpl + rotate_at_frame + normalisation + 6D, M:423-464)
    b = wl["w0_aa"].shape[0]
    got = harness._window_prefix_hip(ds, torch.from_numpy(wl["w0_aa"]).float().cuda(), torch.from_numpy(wl["w0_root"]).float().cuda(), 10)
    assert got is not None and got.shape == (b, 10, 198)
    assert np.abs(got[:, :, :66].cpu().numpy() - wl["w1_prefix_jpos"]).max() < 2e-5
    assert np.abs(got[:, :, 66:].cpu().numpy() - wl["w1_prefix_6d"]).max() < 2e-5
    gq, gj = ds.fk_smpl(torch.from_numpy(wl["w0_fk_root"]).float().cuda(), torch.from_numpy(wl["w0_fk_aa"]).float().cuda())
    assert np.abs(gj.cpu().numpy() - wl["w0_fk_jpos"]).max() < 5e-6
