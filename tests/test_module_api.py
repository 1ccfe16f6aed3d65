"""Host logic of the Python mirror: checkpoint layout, EMA-style deepcopy, loud failure off-GPU,
and the plain-PyTorch training forward agreeing with the oracle."""
import copy

import pytest
import torch

from egoego_release_amd import ModelConfig, make_weights, _lib
from egoego_release_amd.model import CondGaussianDiffusion
from oracle import egoego_oracle as O


def _model(**kw):
    cfg = ModelConfig(**kw)
    m = CondGaussianDiffusion(**cfg.ctor_kwargs())
    sd = make_weights(cfg, 0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    return cfg, m, sd


def test_state_dict_layout():
    cfg, m, sd = _model()
    keys = m.state_dict()
    assert len(keys) == 86  # 73 learnable/frozen tensors + 13 schedule buffers
    assert keys["denoise_fn.motion_transformer.start_conv.weight"].shape == (512, 396, 1)
    assert keys["denoise_fn.motion_transformer.position_vec.weight"].shape == (122, 512)
    assert keys["denoise_fn.motion_transformer.layer_stack.3.self_attn.w_q.weight"].shape == (1024, 512)
    assert keys["denoise_fn.motion_transformer.layer_stack.0.pos_ffn.w_2.weight"].shape == (512, 512, 1)
    assert keys["denoise_fn.linear_out.weight"].shape == (198, 512)
    assert keys["denoise_fn.time_mlp.1.weight"].shape == (256, 64)
    assert keys["denoise_fn.time_mlp.3.weight"].shape == (512, 256)
    sched = O.make_schedule(1000)
    for k, v in sched.items():
        assert torch.equal(keys[k], v), k
    assert m.seq_len == 120 and m.num_timesteps == 1000 and m.out_dim == 198
    assert sum(p.numel() for p in m.parameters()) == 11027910


def test_ctor_errors_match_reference():
    with pytest.raises(ValueError, match="unknown beta schedule"):
        CondGaussianDiffusion(198, 512, 4, 4, 256, 256, 121, 198, beta_schedule="sigmoid")
    m = CondGaussianDiffusion(198, 512, 4, 4, 256, 256, 121, 198, beta_schedule="linear", timesteps=50)
    assert m.num_timesteps == 50 and m.betas.shape == (50,)


def test_deepcopy_and_cpu_sampling_fails_loudly():
    cfg, m, sd = _model()
    m2 = copy.deepcopy(m)  # ema_pytorch.EMA deep-copies the module
    assert m2._slot is not m._slot
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    x = torch.zeros(1, 120, 198)
    with pytest.raises(_lib.EgoEgoHipError, match="no CPU fallback"):
        m.sample(x, torch.ones_like(x))
    with pytest.raises(_lib.EgoEgoHipError):
        m.p_sample(x, torch.zeros(1, dtype=torch.long), x)


def test_training_forward_matches_oracle():
    cfg, m, sd = _model()
    m.eval()
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(2, 40, 396, generator=g)
    t = torch.tensor([7, 900])
    pm = torch.ones(2, 1, 41).bool()
    pm[1, 0, 30:] = False
    with torch.no_grad():
        for mask in (None, pm):
            a = m.denoise_fn(x_all, t, mask)
            b = O.denoise(sd, x_all, t, padding_mask=mask)
            assert (a - b).abs().max() < 2e-5
    m.train()
    loss = m(x_all[..., :198], torch.ones(2, 40, 198))
    assert torch.isfinite(loss)
    loss.backward()
    assert m.denoise_fn.linear_out.weight.grad is not None


def test_reference_checkpoint_format_loads(tmp_path):
    """A file in the reference's checkpoint layout (trainer:99-106; EMA weights under 'ema_model.') loads into the
    HIP-backed module; the EMA copy is the one used for inference."""
    from egoego_release_amd import harness
    cfg, m, sd = _model()
    full = m.state_dict()
    ema = {"ema_model." + k: (v + 1.0 if k.endswith("linear_out.bias") else v.clone()) for k, v in full.items()}
    ema.update({"online_model." + k: v.clone() for k, v in full.items()})
    ema.update({"initted": torch.tensor(True), "step": torch.tensor(10)})
    path = tmp_path / "model-4.pt"
    torch.save({"step": 123, "model": full, "ema": ema, "scaler": {}}, path)
    m2, info = harness.load_stage2_checkpoint(str(path))
    assert info["step"] == 123 and not info["missing"] and not info["unexpected"]
    assert torch.equal(m2.state_dict()["denoise_fn.linear_out.bias"], full["denoise_fn.linear_out.bias"] + 1.0)
    assert torch.equal(m2.state_dict()["denoise_fn.time_mlp.3.weight"], full["denoise_fn.time_mlp.3.weight"])
    m3, _ = harness.load_stage2_checkpoint(str(path), use_ema=False)
    assert torch.equal(m3.state_dict()["denoise_fn.linear_out.bias"], full["denoise_fn.linear_out.bias"])
    assert m2.seq_len == 120 and m2.objective == "pred_x0" and m2.num_timesteps == 1000


def test_weight_updates_through_data_are_detected():
    """ema_pytorch writes the EMA weights with `ma_params.data.copy_` / `.data.lerp_`, which bump neither
    `data_ptr` nor `_version`: the cheap key cannot see them, the device-side fingerprint every chain-level entry
    point takes must, and load_state_dict / .to() drop the packed copy explicitly."""
    cfg, m, sd = _model()
    key, fp = m._engine_key(), m._weights_fingerprint()
    p = m.denoise_fn.linear_out.weight
    p.data.lerp_(torch.zeros_like(p), 0.01)
    assert m._engine_key() == key            # the blind spot ...
    assert m._weights_fingerprint() != fp    # ... that the fingerprint covers
    fp = m._weights_fingerprint()
    m.posterior_mean_coef1.data.mul_(1.001)  # schedule buffers are packed too
    assert m._weights_fingerprint() != fp

    class Eng:
        closed = False

        def close(self):
            self.closed = True

    e = Eng()
    m._slot.engine, m._slot.key = e, key
    m.load_state_dict(sd, strict=False)
    assert e.closed and m._slot.engine is None
    e2 = Eng()
    m._slot.engine = e2
    m.double()
    assert e2.closed and m._slot.engine is None
    m.float()
    m2 = copy.deepcopy(m)
    assert m2._slot.engine is None and m2._slot.fingerprint is None


def test_timestep_range_is_checked_like_the_reference():
    cfg, m, sd = _model()
    with pytest.raises(IndexError, match="out of range"):
        m._check_t(torch.tensor([3, 1000]))
    with pytest.raises(IndexError, match="out of range"):
        m._check_t(torch.tensor([-1]))
    m._check_t(torch.tensor([0, 999]))
