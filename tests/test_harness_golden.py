"""Harness pieces pinned to the reference (tests/golden/harness_golden.npz, written by make_harness_golden.py from
/root/reference/egoego/lafan1/utils.py and AMASSDataset's min/max methods on the reference's own test_data/ares
fixtures) and exercised on real data: the 140-frame head trajectory and body motion of demo_ares_data.p with the real
window-120 statistics.

CPU part: oracle/harness_oracle.py and the product's torch chains against the goldens.
GPU part: the three per-window HIP kernels against the numpy/scipy oracle (NOT against the product's own torch
chain), and the demo trajectory through the whole GPU sliding-window harness (windows of 120 + 30 frames).
"""
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation as Rot

from egoego_release_amd import ModelConfig, make_weights, harness, rotations as R
from oracle import egoego_oracle as O
from oracle import harness_oracle as HO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# A plausible SMPL-H-like rest skeleton (metres, child minus parent, body frame).  The licensed SMPL-H model is not
# available: the offsets are an input of the harness, any consistent set exercises the same code.
REST_OFFSETS = np.array([
    [0.0, 0.0, 0.0], [0.07, -0.09, -0.01], [-0.07, -0.09, -0.01], [0.0, 0.11, -0.03], [0.03, -0.38, 0.0], [-0.03, -0.38, 0.0],
    [0.0, 0.14, 0.02], [-0.01, -0.40, -0.04], [0.01, -0.40, -0.04], [0.0, 0.05, 0.03], [0.03, -0.06, 0.12], [-0.03, -0.06, 0.12],
    [0.0, 0.21, -0.04], [0.08, 0.12, -0.03], [-0.08, 0.12, -0.03], [0.0, 0.09, 0.05], [0.10, 0.03, -0.01], [-0.10, 0.03, -0.01],
    [0.26, -0.01, -0.02], [-0.26, -0.01, -0.02], [0.25, 0.01, 0.0], [-0.25, 0.01, 0.0]])


@pytest.fixture(scope="module")
def hg():
    return np.load(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"))


def _stats(hg):
    lo, hi = hg["stats_global_jpos_min"], hg["stats_global_jpos_max"]
    return harness.SkeletonStats(lo, hi, REST_OFFSETS), HO.SkeletonOracle(lo, hi, REST_OFFSETS)


def _angle(aa_a, aa_b):
    """Rotation angle between two axis-angle arrays (insensitive to the sign ambiguity near pi)."""
    d = Rot.from_rotvec(np.asarray(aa_a, np.float64).reshape(-1, 3)) * Rot.from_rotvec(np.asarray(aa_b, np.float64).reshape(-1, 3)).inv()
    return np.abs(d.magnitude())


# ------------------------------------------------------------------------------------------ CPU: oracle vs reference
def test_oracle_rotate_at_frame_equals_reference(hg):
    qp = hg["demo_head_qpos"]
    for tag, (a, b) in (("w0", (0, 120)), ("w1", (110, 140))):
        x, q, y = HO.rotate_at_frame_smplh(qp[None, a:b, :3], qp[None, a:b, 3:], 0)
        assert np.array_equal(x, hg[f"raf_demo_{tag}_trans"]) and np.array_equal(q, hg[f"raf_demo_{tag}_quat"])
        assert np.array_equal(y, hg[f"raf_demo_{tag}_yrot"])
    p, q = hg["raf_rand_in_trans"], hg["raf_rand_in_quat"]
    for idx in (0, 9):
        x, nq, y = HO.rotate_at_frame_smplh(p, q, idx)
        assert np.array_equal(x, hg[f"raf_rand_t{idx}_trans"]) and np.array_equal(nq, hg[f"raf_rand_t{idx}_quat"])
        assert np.array_equal(y, hg[f"raf_rand_t{idx}_yrot"])
    assert np.array_equal(HO.quat_mul(q, hg["quat_in_b"]), hg["quat_mul"])
    assert np.array_equal(HO.quat_mul_vec(q, p), hg["quat_mul_vec"])
    assert np.array_equal(HO.quat_inv(q), hg["quat_inv"])
    assert np.array_equal(HO.quat_between(np.array([1.0, 0, 0]), p), hg["quat_between"])


def test_oracle_minmax_equals_reference(hg):
    _, dso = _stats(hg)
    assert np.abs(dso.norm(hg["norm_in"].astype(np.float64)) - hg["norm_out"]).max() < 1e-6
    assert np.abs(dso.denorm(hg["norm_out"].astype(np.float64)) - hg["denorm_out"]).max() < 1e-6
    assert np.abs(dso.denorm(hg["denorm_unit_in"].astype(np.float64)) - hg["denorm_unit_out"]).max() < 1e-6


# ------------------------------------------------------------------------------------------ CPU: product vs reference
def test_product_rotate_at_frame_equals_reference(hg):
    qp = torch.from_numpy(hg["demo_head_qpos"])
    for tag, (a, b) in (("w0", (0, 120)), ("w1", (110, 140))):
        x, q, y = harness.rotate_at_frame(qp[None, a:b, :3], qp[None, a:b, 3:], 0)
        assert np.abs(x.numpy() - hg[f"raf_demo_{tag}_trans"]).max() < 1e-12
        assert np.abs(q.numpy() - hg[f"raf_demo_{tag}_quat"]).max() < 1e-12
        assert np.abs(y.numpy() - hg[f"raf_demo_{tag}_yrot"]).max() < 1e-12
    p, q = torch.from_numpy(hg["raf_rand_in_trans"]), torch.from_numpy(hg["raf_rand_in_quat"])
    for idx in (0, 9):
        x, nq, y = harness.rotate_at_frame(p, q, idx)
        assert np.abs(x.numpy() - hg[f"raf_rand_t{idx}_trans"]).max() < 1e-12
        assert np.abs(nq.numpy() - hg[f"raf_rand_t{idx}_quat"]).max() < 1e-12
        assert np.abs(y.numpy() - hg[f"raf_rand_t{idx}_yrot"]).max() < 1e-12
    x, nq, y = harness.rotate_at_frame(p.float(), q.float(), 0)
    assert np.abs(x.numpy() - hg["raf_rand_f32_trans"]).max() < 2e-6 and np.abs(nq.numpy() - hg["raf_rand_f32_quat"]).max() < 2e-6


def test_product_minmax_equals_reference(hg):
    ds, _ = _stats(hg)
    assert np.abs(ds.normalize_jpos_min_max(torch.from_numpy(hg["norm_in"])).numpy() - hg["norm_out"]).max() < 1e-6
    assert np.abs(ds.de_normalize_jpos_min_max(torch.from_numpy(hg["norm_out"])).numpy() - hg["denorm_out"]).max() < 1e-6
    assert np.abs(ds.de_normalize_jpos_min_max(torch.from_numpy(hg["denorm_unit_in"])).numpy() - hg["denorm_unit_out"]).max() < 1e-6


# ------------------------------------------------------------------------------------------ real poses -> windows
def _demo_windows(hg, dso, frames):
    """The demo's real body motion as model-space windows: FK with the oracle, canonicalised about the head heading of
    the first frame, min/max-normalised with the real statistics, rotations as 6D — what a trained denoiser would emit.
    Returns (x [1,n,198] float64, recover [1,1,1,4], local axis-angle [n,22,3], root [n,3], head [n,3])."""
    a, b = frames
    aa = np.concatenate([hg["demo_root_orient"][a:b, None], hg["demo_body_pose"][a:b].reshape(-1, 21, 3)], 1)
    root = hg["demo_trans"][a:b]
    gq, gj = dso.fk(root, aa)
    n = b - a
    ct, cq, rec = HO.rotate_at_frame_smplh(gj[None, :, HO.HEAD], gq[None, :, HO.HEAD], 0)
    inv = np.broadcast_to(HO.quat_inv(rec)[0], gq.shape)
    cj = HO.quat_mul_vec(inv, gj)
    cr = HO.std_mul(inv, gq)
    x = np.concatenate([dso.norm(cj).reshape(n, 66), HO.quat_to_mat(cr)[..., :2, :].reshape(n, 132)], -1)[None]
    return x, rec, aa, gj[:, 0], gj[:, HO.HEAD]


def test_convert_model_res_roundtrip_on_real_poses_cpu(hg):
    """Real SMPL-H poses -> FK -> canonical window -> convert_model_res_to_data must return the poses: a known-answer
    test of the whole M:469-525 chain (6D -> matrix -> quaternion -> un-canonicalise -> IK -> axis-angle) on
    well-conditioned rotations, for the oracle and for the product's torch chain."""
    ds, dso = _stats(hg)
    x, rec, aa, root, head = _demo_windows(hg, dso, (0, 120))
    aa_o, root_o, head_o = HO.convert_model_res_to_data(dso, x, rec)
    # 1e-7, not round-off: rotate_at_frame_smplh normalises with x / (|x| + 1e-8) (lafan1/utils.py:17-27)
    assert _angle(aa_o[0], aa).max() < 1e-7 and np.abs(root_o[0] - root).max() < 1e-7 and np.abs(head_o[0] - head).max() < 1e-7
    aa_p, root_p, head_p = harness.convert_model_res_to_data(ds, torch.from_numpy(x).float(), rec)
    assert _angle(aa_p[0].numpy(), aa).max() < 2e-5
    assert np.abs(root_p[0].numpy() - root).max() < 1e-5 and np.abs(head_p[0].numpy() - head).max() < 1e-5
    assert np.abs(x[..., :66]).max() <= 1.0  # the demo motion lies inside the real statistics' range


# ------------------------------------------------------------------------------------------ GPU: kernels vs oracle
@pytest.mark.gpu
def test_convert_model_res_hip_kernel_vs_oracle(hg):
    """egoego_convert_model_res on the real-pose windows (known answer + oracle) and on random but well-conditioned
    windows (oracle), batch of 3 with different recover rotations."""
    ds, dso = _stats(hg)
    xs, recs = [], []
    for fr in ((0, 40), (50, 90), (100, 140)):
        x, rec, aa, root, head = _demo_windows(hg, dso, fr)
        xs.append(x)
        recs.append(rec)
    x, rec = np.concatenate(xs, 0), np.concatenate(recs, 0)
    g = np.random.default_rng(3)
    x = x + g.standard_normal(x.shape) * 0.01  # off the manifold, still well-conditioned
    aa_o, root_o, head_o = HO.convert_model_res_to_data(dso, x, rec)
    aa_h, root_h, head_h = harness.convert_model_res_to_data(ds, torch.from_numpy(x).float().cuda(), torch.from_numpy(rec).cuda())
    assert aa_h.is_cuda and aa_h.shape == (3, 40, 22, 3)
    assert _angle(aa_h.cpu().numpy(), aa_o).max() < 1e-4
    assert np.abs(root_h.cpu().numpy() - root_o).max() < 1e-5 and np.abs(head_h.cpu().numpy() - head_o).max() < 1e-5
    with pytest.raises(Exception, match="earlier joint"):
        harness.convert_model_res_to_data(ds, torch.from_numpy(x).float().cuda(), torch.from_numpy(rec).cuda(),
                                          parents=(-1,) + (5,) * 21)


@pytest.mark.gpu
def test_window_condition_hip_kernel_vs_oracle(hg):
    """egoego_window_condition against the reference-pinned numpy chain on the demo head trajectory's two windows."""
    ds, dso = _stats(hg)
    qp = hg["demo_head_qpos"]
    for a, b in ((0, 120), (110, 140)):
        p = np.stack([qp[a:b, :3], qp[a:b, :3][::-1]])
        q = np.stack([qp[a:b, 3:], qp[a:b, 3:][::-1]])
        x_start, rec = harness._window_condition_hip(ds, torch.from_numpy(p).cuda(), torch.from_numpy(q).cuda())
        a_t, a_q, yrot = HO.rotate_at_frame_smplh(p, q, 0)
        mv = a_t[:, 0:1].copy()
        mv[:, :, 2] = 0
        want = np.zeros((2, b - a, 198))
        want[:, :, 45:48] = a_t - mv
        want[:, :, 156:162] = HO.quat_to_mat(a_q)[..., :2, :].reshape(2, -1, 6)
        want[:, :, :66] = dso.norm(want[:, :, :66].reshape(-1, 22, 3)).reshape(2, -1, 66)
        assert np.abs(x_start.cpu().numpy() - want).max() < 1e-5
        assert np.abs(rec.cpu().numpy() - yrot).max() < 1e-6 and rec.shape == yrot.shape


@pytest.mark.gpu
def test_window_prefix_hip_kernel_vs_oracle(hg):
    """egoego_window_prefix (fk_smpl + rotate_at_frame + normalisation + 6D, M:399-467) against the numpy/scipy oracle
    on the demo's real poses."""
    ds, dso = _stats(hg)
    n_last = 10
    aas, roots = [], []
    for a, b in ((0, 30), (60, 90), (110, 140)):
        aas.append(np.concatenate([hg["demo_root_orient"][a:b, None], hg["demo_body_pose"][a:b].reshape(-1, 21, 3)], 1))
        roots.append(hg["demo_trans"][a:b])
    aa, root = np.stack(aas), np.stack(roots)
    aa[0, :, 5] = 0.0  # zero rotation: the small-angle branch
    got = harness._window_prefix_hip(ds, torch.from_numpy(aa).float().cuda(), torch.from_numpy(root).float().cuda(), n_last)
    assert got is not None and got.shape == (3, n_last, 198)
    B = 3
    gq, gj = dso.fk(root.reshape(-1, 3), aa.reshape(-1, 22, 3))
    gq, gj = gq.reshape(B, -1, 22, 4)[:, -n_last:], gj.reshape(B, -1, 22, 3)[:, -n_last:]
    t_t, _, t_rec = HO.rotate_at_frame_smplh(gj[:, :, HO.HEAD], gq[:, :, HO.HEAD], 0)
    t_mv = t_t[:, 0:1].copy()
    t_mv[:, :, 2] = 0
    inv = np.broadcast_to(HO.quat_inv(t_rec), gq.shape)
    pj = dso.norm((HO.quat_mul_vec(inv, gj) - t_mv[:, :, None, :]).reshape(-1, 22, 3)).reshape(B, -1, 66)
    p6 = HO.quat_to_mat(HO.std_mul(inv, gq))[..., :2, :].reshape(B, -1, 132)
    want = np.concatenate([pj, p6], -1)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-5
    assert harness._window_prefix_hip(ds, torch.from_numpy(aa).float(), torch.from_numpy(root).float(), n_last) is None


def _trained_like_weights(hg, dso, cfg, seed=0):
    """Synthetic weights whose output head behaves like a trained one: linear_out's bias is a REAL canonical pose of
    the demo motion (valid, orthonormal 6D; joints inside the statistics' range) and its weight is scaled down, so the
    denoiser emits that pose plus a small input-dependent perturbation.  The post-loop rotation chain is then
    well-conditioned (random 6D outputs are not: M:493's Gram-Schmidt amplifies them arbitrarily)."""
    sd = make_weights(cfg, seed)
    x, _, _, _, _ = _demo_windows(hg, dso, (20, 21))
    sd["denoise_fn.linear_out.bias"] = torch.from_numpy(x[0, 0]).float()
    sd["denoise_fn.linear_out.weight"] = sd["denoise_fn.linear_out.weight"] * 0.05
    return sd


@pytest.mark.gpu
def test_demo_trajectory_through_gpu_harness(hg):
    """The reference demo's 140-frame head trajectory (test_data/ares/demo_ares_data.p) through
    full_body_gen_cond_head_pose_sliding_window on the GPU: seq_len 120 -> windows of 120 and 30 frames with the 10-frame
    overlap in-painted after every step, real statistics, trained-like output head, 8 diffusion steps, injected noise.
    Row 1 of the batch is the same trajectory rotated by 1.1 rad about z and shifted: canonicalisation must make its
    result the rotated/shifted result of row 0.  Checked against oracle/harness_oracle.py (numpy + scipy)."""
    from egoego_release_amd.model import CondGaussianDiffusion
    ds, dso = _stats(hg)
    seq_len, S, T, B = 120, 8, 140, 2
    cfg = ModelConfig(max_timesteps=seq_len + 1)
    sd = _trained_like_weights(hg, dso, cfg)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    m.num_timesteps = S
    qp = hg["demo_head_qpos"]
    zrot = np.array([np.cos(0.55), 0.0, 0.0, np.sin(0.55)])
    p1 = HO.quat_mul_vec(np.broadcast_to(zrot, (T, 4)), qp[:, :3]) + np.array([0.7, -0.4, 0.0])
    q1 = HO.quat_mul(np.broadcast_to(zrot, (T, 4)), qp[:, 3:])
    head_pose = torch.from_numpy(np.stack([qp, np.concatenate([p1, q1], -1)])).float()
    g = torch.Generator().manual_seed(5)
    wins = [(0, 120), (110, 140)]
    dup = lambda t: t.repeat_interleave(2, dim=-3) if t.dim() == 3 else t.repeat_interleave(2, dim=1)
    noise = {"x_all": torch.randn(1, T, 198, generator=g).repeat(2, 1, 1),
             "cond": [torch.randn(1, b - a, 198, generator=g).repeat(2, 1, 1) for a, b in wins],
             "steps": [torch.randn(S, 1, b - a, 198, generator=g).repeat(1, 2, 1, 1) for a, b in wins]}
    aa, root = harness.full_body_gen_cond_head_pose_sliding_window(m, ds, head_pose.cuda(), noise=noise)
    assert aa.shape == (B, T, 22, 3) and root.shape == (B, T, 3)
    cm = O.head_condition_mask((B, T, 198))
    aa2, root2 = HO.sliding_window(sd, O.make_schedule(1000), dso, seq_len, S, head_pose[..., :3].double().numpy(),
                                   head_pose[..., 3:].double().numpy(), cm, noise)
    aa, root = aa.cpu().numpy(), root.cpu().numpy()
    assert np.abs(root - root2).max() < 1e-4, np.abs(root - root2).max()
    ang = _angle(aa, aa2)
    assert ang.max() < 1e-3, (ang.max(), np.median(ang))
    # equivariance: row 1 is row 0 rotated about z (root orientation composes with the rotation, the other joints'
    # local rotations are unchanged).  The xy shift does NOT come back: like the reference, convert_model_res_to_data
    # un-rotates the canonical window (M:498-501) but never restores the translation that moved the head to the
    # origin (M:369-373), so both rows live in a frame centred on their own first head position.
    r0 = HO.quat_mul_vec(np.broadcast_to(zrot, (T, 4)), root[0].astype(np.float64))
    assert np.abs(r0 - root[1]).max() < 2e-4
    assert _angle(aa[0, :, 1:], aa[1, :, 1:]).max() < 1e-3
    rot0 = Rot.from_quat(np.array([[0.0, 0.0, np.sin(0.55), np.cos(0.55)]])) * Rot.from_rotvec(aa[0, :, 0].astype(np.float64))
    assert np.abs((rot0 * Rot.from_rotvec(aa[1, :, 0].astype(np.float64)).inv()).magnitude()).max() < 1e-3


# ------------------------------------------------------------------------------------------ entry point
def test_mpjpe_matches_the_reference_definition():
    """kinpoly/scripts/eval_metrics_imu_rec.py:297-301: root-relative, mean over frames and joints, millimetres."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("run_stage2_demo", os.path.join(ROOT, "tools", "run_stage2_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.random.default_rng(0)
    gt = g.standard_normal((7, 22, 3))
    pred = gt + np.array([0.3, -0.2, 0.1])            # a pure root offset costs nothing
    assert mod.mpjpe_mm(pred, gt) < 1e-9
    pred = gt.copy()
    pred[:, 5] += np.array([0.003, 0.0, 0.004])       # one joint off by 5 mm -> 5/22 mm on average
    assert abs(mod.mpjpe_mm(pred, gt) - 5.0 / 22) < 1e-9
    want = np.linalg.norm((pred - pred[:, 0:1]) - (gt - gt[:, 0:1]), axis=2).mean() * 1000
    assert abs(mod.mpjpe_mm(pred, gt) - want) < 1e-12


@pytest.mark.gpu
def test_stage2_entry_point_on_the_demo_trajectory(hg, tmp_path):
    """tools/run_stage2_demo.py with the reference's --diffusion_* flags on the 140-frame demo head trajectory and the real
    statistics: two windows (120 + 30 frames), outputs of the reference's shapes, an MPJPE against the demo's own FK
    joints — and THE SAME NUMBERS as (a) calling the harness directly (bit for bit, Philox run on synthetic weights) and
    (b) the numpy/scipy oracle on injected draws (a checkpoint in the reference's file layout with a trained-like output
    head; 1e-4 m / 1e-3 rad like the direct harness test)."""
    import importlib.util
    import pickle
    spec = importlib.util.spec_from_file_location("run_stage2_demo", os.path.join(ROOT, "tools", "run_stage2_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ds, dso = _stats(hg)
    np.save(tmp_path / "head.npy", hg["demo_head_qpos"])
    np.save(tmp_path / "rest.npy", REST_OFFSETS)
    with open(tmp_path / "stats.p", "wb") as f:
        pickle.dump({"global_jpos_min": hg["stats_global_jpos_min"], "global_jpos_max": hg["stats_global_jpos_max"]}, f)
    aa_gt = np.concatenate([hg["demo_root_orient"][:, None], hg["demo_body_pose"].reshape(-1, 21, 3)], 1)
    _, gj = dso.fk(hg["demo_trans"], aa_gt)
    np.save(tmp_path / "gt.npy", gj)
    common = ["--head_pose", str(tmp_path / "head.npy"), "--stats", str(tmp_path / "stats.p"), "--rest_offsets", str(tmp_path / "rest.npy"),
              "--gt_jpos", str(tmp_path / "gt.npy"), "--diffusion_window", "120", "--diffusion_batch_size", "2", "--use_min_max",
              "--canonicalize_init_head"]
    # ---- (a) Philox run on the synthetic weights == the harness called directly
    rep = mod.main(common + ["--timesteps", "5", "--sampling_rng", "philox", "--seed", "3", "--out", str(tmp_path / "out.npz")])
    assert rep["frames"] == 140 and rep["samples"] == 2 and rep["windows"] == 2 and len(rep["mpjpe_mm"]) == 2
    out = np.load(tmp_path / "out.npz")
    assert out["local_aa"].shape == (2, 140, 22, 3) and out["root_trans"].shape == (2, 140, 3) and out["global_jpos"].shape == (2, 140, 22, 3)
    assert all(np.isfinite(out[k]).all() for k in out.files) and all(np.isfinite(v) and v > 0 for v in rep["mpjpe_mm"])
    # the two samples share the head trajectory and (Philox keyed by window index) differ in their noise
    assert not np.array_equal(out["local_aa"][0], out["local_aa"][1])
    cfg = ModelConfig(max_timesteps=121)
    m = harness.build_stage2_model(window=120)
    m.load_state_dict(make_weights(cfg, 0), strict=False)
    m = m.cuda()
    m.num_timesteps, m.sampling_rng, m.philox_seed = 5, "philox", 3
    head_pose = torch.from_numpy(hg["demo_head_qpos"]).float()[None].repeat_interleave(2, 0).cuda()
    torch.manual_seed(3)
    aa, root = harness.full_body_gen_cond_head_pose_sliding_window(m, ds, head_pose)
    assert np.array_equal(out["local_aa"], aa.cpu().numpy()) and np.array_equal(out["root_trans"], root.cpu().numpy())
    gq_d, gj_d = ds.fk_smpl(root.reshape(-1, 3), aa.reshape(-1, 22, 3))
    assert np.array_equal(out["global_jpos"], gj_d.reshape(2, 140, 22, 3).cpu().numpy())
    assert abs(rep["mpjpe_mm"][0] - mod.mpjpe_mm(out["global_jpos"][0], gj)) < 1e-9
    # ---- (b) checkpoint file in the reference's layout + injected draws == the oracle's sliding window
    S, T = 6, 140
    sd = _trained_like_weights(hg, dso, cfg)
    ref_model = harness.build_stage2_model(window=120)
    ref_model.load_state_dict(sd, strict=False)
    full = ref_model.state_dict()
    torch.save({"step": 7, "model": {k: torch.zeros_like(v) for k, v in full.items()},  # the EMA weights are the ones used
                "ema": {**{"ema_model." + k: v for k, v in full.items()}, "initted": torch.tensor(True), "step": torch.tensor(7)},
                "scaler": {}}, tmp_path / "model-7.pt")
    g = torch.Generator().manual_seed(15)
    wins = [(0, 120), (110, 140)]
    noise = {"x_all": torch.randn(2, T, 198, generator=g), "cond": [torch.randn(2, b - a, 198, generator=g) for a, b in wins],
             "steps": [torch.randn(S, 2, b - a, 198, generator=g) for a, b in wins]}
    torch.save(noise, tmp_path / "noise.pt")
    rep = mod.main(common + ["--timesteps", str(S), "--weight", str(tmp_path / "model-7.pt"), "--noise", str(tmp_path / "noise.pt"),
                             "--out", str(tmp_path / "out2.npz")])
    assert rep["checkpoint"]["step"] == 7 and not rep["checkpoint"]["unexpected"]
    out2 = np.load(tmp_path / "out2.npz")
    hp = head_pose.cpu()
    aa2, root2 = HO.sliding_window(sd, O.make_schedule(1000), dso, 120, S, hp[..., :3].double().numpy(), hp[..., 3:].double().numpy(),
                                   O.head_condition_mask((2, T, 198)), noise)
    assert np.abs(out2["root_trans"] - root2).max() < 1e-4, np.abs(out2["root_trans"] - root2).max()
    ang = _angle(out2["local_aa"], aa2)
    assert ang.max() < 1e-3, (ang.max(), np.median(ang))
