"""CPU checks of the pack-time preparation for the int8-slice precisions (egoego_release_amd/precision.py) against the oracle:
the mean shift of the LayerNorm rows is function-preserving, and the compensated rounding stays on the library's integer grid."""
import torch

from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.precision import QMAX, compensated_rounding, prepare_int8_state
from oracle import egoego_oracle as O


def _calibration(sd, B=2, T=24):
    g = torch.Generator().manual_seed(3)
    rows = {}
    for tv in (0, 500, 999):
        x_all = torch.randn(B, T, 396, generator=g)
        taps = {}
        with torch.no_grad():
            O.denoise(sd, x_all, torch.full((B,), tv, dtype=torch.long), taps=taps)
        prev = taps["embed"]
        for li in range(4):
            lt = taps[f"layer{li}"]
            for key, v in (((li, "qkv"), prev), ((li, "fc"), lt["attn_out"]), ((li, "w_1"), lt["attn_ln"]), ((li, "w_2"), lt["ffn_hidden"])):
                rows.setdefault(key, []).append(v.reshape(-1, v.shape[-1]))
            for st in ("k", "v"):  # the oracle's head-major [H * B, L, 256] -> rows of H * 256 features
                v = lt[st].view(4, B, T + 1, 256).permute(1, 2, 0, 3).reshape(-1, 1024)
                rows.setdefault((li, st), []).append(v)
            prev = lt["out"]
        rows.setdefault(("out", "linear_out"), []).append(prev[:, 1:].reshape(-1, 512))
    return {"rows": {k: torch.cat(v, 0) for k, v in rows.items()}, "n_layers": 4}


def test_mean_shift_is_function_preserving_and_rows_are_shifted():
    cfg = ModelConfig(max_timesteps=25)
    sd = make_weights(cfg, 0)
    calib = _calibration(sd)
    sd2, shift = prepare_int8_state(sd, calib, _lib.PREC_I8X3_FC, shift=True, rounding=False, shift_kv=True)
    assert set(shift) == {"embed"} | {(li, s) for li in range(4) for s in ("attn_ln", "out", "k", "v", "attn_out")}  # (shift_kv=True below)
    assert float(shift[(0, "attn_ln")].abs().max()) > 0.05  # the rows do have a common component to remove
    x_all = torch.randn(2, 24, 396, generator=torch.Generator().manual_seed(9))
    t = torch.tensor([5, 700])
    ta, tb = {}, {}
    with torch.no_grad():
        want = O.denoise(sd, x_all, t, taps=ta)
        got = O.denoise(sd2, x_all, t, taps=tb)
    assert float((got - want).abs().max()) < 2e-5
    # the stored rows are the true rows minus the constant, at every LayerNorm site and at the embed output
    assert float((tb["embed"] + shift["embed"] - ta["embed"]).abs().max()) < 2e-5
    for li in range(4):
        for s in ("attn_ln", "out"):
            assert float((tb[f"layer{li}"][s] + shift[(li, s)] - ta[f"layer{li}"][s]).abs().max()) < 5e-5, (li, s)
        for s in ("q", "ffn_hidden"):  # untouched
            assert float((tb[f"layer{li}"][s] - ta[f"layer{li}"][s]).abs().max()) < 5e-5, (li, s)
        # K and V (and with V the attention output) lack their mean rows: softmax and fc's bias take care of it
        hm = lambda c: c.view(4, 1, 1, 256).expand(4, 2, 25, 256).reshape(8, 25, 256)
        for s in ("k", "v"):
            assert float((tb[f"layer{li}"][s] + hm(shift[(li, s)]) - ta[f"layer{li}"][s]).abs().max()) < 5e-5, (li, s)
        assert float((tb[f"layer{li}"]["attn_out"] + shift[(li, "attn_out")] - ta[f"layer{li}"]["attn_out"]).abs().max()) < 5e-5, li
        assert float(shift[(li, "k")].abs().max()) > 0.05


def test_compensated_rounding_stays_on_the_grid_and_lowers_the_output_error():
    g = torch.Generator().manual_seed(1)
    W = torch.randn(96, 64, generator=g) * 0.05
    X = torch.randn(500, 64, generator=g) + 3.0 * torch.randn(1, 64, generator=g)  # rows with a strong common component
    Q = compensated_rounding(W, X, block=16)
    scale = W.abs().amax(1, keepdim=True) / QMAX
    q = Q / scale
    assert float((q - q.round()).abs().max()) < 5e-3 and float(q.abs().max()) <= QMAX + 1e-2  # integers of the library's grid
    assert torch.equal(Q.abs().amax(1), W.abs().amax(1))                                       # the scale-setting entry is untouched
    assert float(((Q - W) / scale).abs().max()) < 8                                            # a few steps at most from the weights
    nearest = torch.round(W / scale) * scale
    Xt = torch.randn(400, 64, generator=g) + X.mean(0, keepdim=True)                           # fresh rows of the same distribution
    e_near = float(((nearest - W) @ Xt.T).pow(2).mean().sqrt())
    e_comp = float(((Q - W) @ Xt.T).pow(2).mean().sqrt())
    assert e_comp < 0.6 * e_near, (e_comp, e_near)
    # what the library's packing kernel computes from Q (k_pack_rows_i8: rint(w * 32639 / max|w|)) is Q's own integer
    inv = QMAX / Q.abs().amax(1, keepdim=True)
    assert torch.equal(torch.round(Q * inv), q.round())


def test_prepared_state_only_touches_what_the_precision_contracts_on_int8():
    cfg = ModelConfig(max_timesteps=25)
    sd = make_weights(cfg, 0)
    calib = _calibration(sd)
    s8, _ = prepare_int8_state(sd, calib, _lib.PREC_I8X3, shift=False, rounding=True, shift_kv=False)
    s9, _ = prepare_int8_state(sd, calib, _lib.PREC_I8X3_FC, shift=False, rounding=True, shift_kv=False)
    changed8 = {k for k in sd if not torch.equal(sd[k], s8[k])}
    changed9 = {k for k in sd if not torch.equal(sd[k], s9[k])}
    assert all(k.endswith(".weight") for k in changed9)
    assert not any("fc.weight" in k or "linear_out" in k for k in changed8) and any("fc.weight" in k for k in changed9) and "denoise_fn.linear_out.weight" in changed9
    assert any("w_q.weight" in k for k in changed8) and any("w_2.weight" in k for k in changed8)
    assert not any("start_conv" in k or "time_mlp" in k or "position_vec" in k for k in changed9)
