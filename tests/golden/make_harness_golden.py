#!/usr/bin/env python3
"""Golden vectors for the callers either side of the sampling loop, produced by THE REFERENCE ITSELF where it can run.

Runs only in the authoring container (/root/reference present).  What of the harness is runnable here:

  * egoego/lafan1/utils.py (pure numpy once its unused pytorch3d import is stubbed): rotate_at_frame_smplh and the
    quaternion helpers it is built from (:5-137) — called on the real head trajectory of the reference's demo
    fixture test_data/ares/demo_ares_data.p (head_qpos, 140 x 7), cut into the two windows the sliding-window
    harness makes of it (0:120 and 110:140, M:350-356), and on seeded random batches;
  * AMASSDataset.normalize_jpos_min_max / de_normalize_jpos_min_max (egoego/data/amass_diffusion_dataset.py:379-392),
    called unbound on a stand-in `self` that carries the reference's real statistics
    test_data/ares/cano_min_max_mean_std_data_window_120.p exactly as the dataset loads them (:236-239).

fk_smpl, quat_ik_torch, convert_model_res_to_data and the sliding-window loop as a whole need `pytorch3d.transforms`: they are
run by make_window_loop_golden.py (next to this file) with those nine function bodies supplied from scipy.

Outputs are data only (inputs + expected outputs + the two statistics vectors): tests/golden/harness_golden.npz.

    python tests/golden/make_harness_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_pieces():
    _stub("pytorch3d")
    _stub("pytorch3d.transforms")
    _stub("human_body_prior")
    _stub("human_body_prior.body_model")
    _stub("human_body_prior.body_model.body_model", BodyModel=object)
    sys.path.insert(0, REF)
    import egoego.lafan1.utils as U
    import egoego.data.amass_diffusion_dataset as DS
    return U, DS


def main():
    import joblib
    U, DS = import_reference_pieces()
    from oracle import harness_oracle as HO
    out = {}

    # ---- the reference's real fixtures
    stats = joblib.load(os.path.join(REF, "test_data/ares/cano_min_max_mean_std_data_window_120.p"))
    demo = joblib.load(os.path.join(REF, "test_data/ares/demo_ares_data.p"))[0]
    jmin, jmax = stats["global_jpos_min"].astype(np.float32), stats["global_jpos_max"].astype(np.float32)
    out["stats_global_jpos_min"], out["stats_global_jpos_max"] = jmin, jmax
    head_qpos = np.asarray(demo["head_qpos"], np.float64)  # (140, 7): xyz + quaternion wxyz (trainer:261-266)
    out["demo_head_qpos"] = head_qpos
    # the demo's body motion (SMPL-H parameters of the same 140 frames): well-conditioned real rotations for the
    # conversion-chain tests
    out["demo_root_orient"] = np.asarray(demo["root_orient"], np.float64)
    out["demo_body_pose"] = np.asarray(demo["body_pose"], np.float64)
    out["demo_trans"] = np.asarray(demo["trans"], np.float64)

    # ---- rotate_at_frame_smplh on the demo trajectory's two sliding windows (seq_len 120, stride 110)
    for tag, (a, b) in (("w0", (0, 120)), ("w1", (110, 140))):
        p, q = head_qpos[None, a:b, :3], head_qpos[None, a:b, 3:]
        x, nq, yrot = U.rotate_at_frame_smplh(p, q, 0)
        out[f"raf_demo_{tag}_trans"], out[f"raf_demo_{tag}_quat"], out[f"raf_demo_{tag}_yrot"] = x, nq, yrot
        x2, nq2, yrot2 = HO.rotate_at_frame_smplh(p, q, 0)
        for r, o in ((x, x2), (nq, nq2), (yrot, yrot2)):
            assert np.array_equal(r, o), (tag, np.abs(r - o).max())
    # ---- seeded random batches (B=5, T=37), both dtypes the callers use, and a non-zero canonical frame
    g = np.random.default_rng(20241002)
    q = g.standard_normal((5, 37, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    p = g.standard_normal((5, 37, 3))
    out["raf_rand_in_trans"], out["raf_rand_in_quat"] = p, q
    for idx in (0, 9):
        x, nq, yrot = U.rotate_at_frame_smplh(p, q, idx)
        out[f"raf_rand_t{idx}_trans"], out[f"raf_rand_t{idx}_quat"], out[f"raf_rand_t{idx}_yrot"] = x, nq, yrot
        x2, nq2, yrot2 = HO.rotate_at_frame_smplh(p, q, idx)
        assert np.array_equal(x, x2) and np.array_equal(nq, nq2) and np.array_equal(yrot, yrot2)
    x32 = U.rotate_at_frame_smplh(p.astype(np.float32), q.astype(np.float32), 0)
    out["raf_rand_f32_trans"], out["raf_rand_f32_quat"], out["raf_rand_f32_yrot"] = x32
    # ---- the quaternion helpers themselves
    q2 = g.standard_normal((5, 37, 4))
    q2 /= np.linalg.norm(q2, axis=-1, keepdims=True)
    out["quat_in_b"] = q2
    out["quat_mul"] = U.quat_mul(q, q2)
    out["quat_mul_vec"] = U.quat_mul_vec(q, p)
    out["quat_inv"] = U.quat_inv(q)
    out["quat_between"] = U.quat_between(np.array([1.0, 0, 0]), p)
    out["quat_normalize"] = U.quat_normalize(U.quat_between(np.array([1.0, 0, 0]), p))
    assert np.array_equal(out["quat_mul"], HO.quat_mul(q, q2))
    assert np.array_equal(out["quat_mul_vec"], HO.quat_mul_vec(q, p))
    assert np.array_equal(out["quat_between"], HO.quat_between(np.array([1.0, 0, 0]), p))

    # ---- min/max normalisation with the real statistics, through the reference's own methods
    fake = types.SimpleNamespace(
        global_jpos_min=torch.from_numpy(jmin).float().reshape(22, 3)[None],   # dataset:236-239
        global_jpos_max=torch.from_numpy(jmax).float().reshape(22, 3)[None])
    tg = torch.Generator().manual_seed(7)
    jp = torch.randn(9, 22, 3, generator=tg) * 0.8 + torch.tensor([0.0, 0.0, 0.9])
    nrm = DS.AMASSDataset.normalize_jpos_min_max(fake, jp)
    den = DS.AMASSDataset.de_normalize_jpos_min_max(fake, nrm)
    out["norm_in"], out["norm_out"], out["denorm_out"] = jp.numpy(), nrm.numpy(), den.numpy()
    unit = torch.rand(9, 22, 3, generator=tg) * 2 - 1
    out["denorm_unit_in"] = unit.numpy()
    out["denorm_unit_out"] = DS.AMASSDataset.de_normalize_jpos_min_max(fake, unit).numpy()
    dso = HO.SkeletonOracle(jmin, jmax, np.zeros((22, 3)))
    assert np.abs(dso.norm(jp.double().numpy()) - nrm.numpy()).max() < 1e-6
    assert np.abs(dso.denorm(unit.double().numpy()) - out["denorm_unit_out"]).max() < 1e-6

    path = os.path.join(HERE, "harness_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
