import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from egoego_release_amd import ModelConfig, make_weights, make_head_windows, _lib
from egoego_release_amd.model import CondGaussianDiffusion
from oracle import egoego_oracle as O
golden = np.load("tests/golden/stage2_golden.npz")
def ref_noise(shape, S, seed=123):
    g = torch.Generator().manual_seed(seed)
    return {"x_T": torch.randn(shape, generator=g), "cond": torch.randn(shape, generator=g), "steps": torch.stack([torch.randn(shape, generator=g) for _ in range(S)])}
for prec in (3, 8, 9):
    out = {}
    for T in (120, 196):
        cfg = ModelConfig(max_timesteps=T + 1); sd = make_weights(cfg, 0)
        m = CondGaussianDiffusion(**cfg.ctor_kwargs()); m.load_state_dict(sd, strict=False); m.hip_precision = prec; m = m.cuda()
        x_all = torch.randn(2, T, 396, generator=torch.Generator().manual_seed(int(golden[f"denoise_T{T}_seed"])))
        x, xc = x_all[..., :198].contiguous().cuda(), x_all[..., 198:].contiguous().cuda()
        for tag, tt in (("t0", [0, 0]), ("tmix", [3, 977])):
            y = m.denoise(x, torch.tensor(tt).cuda(), xc).cpu().numpy()
            out[f"fwd T{T} {tag}"] = np.abs(y - golden[f"denoise_T{T}_{tag}"]).max()
        if T == 120:
            for tag, B, S in (("b2_s50", 2, 50), ("b1_s1000", 1, 1000)):
                m.num_timesteps = S
                xs, cm = make_head_windows(B, 120, seed=11)
                y = m.sample(xs.cuda(), cm.cuda(), noise=ref_noise(xs.shape, S)).cpu().numpy()
                out[f"chain {tag}"] = np.abs(y - golden[f"sample_{tag}"]).max()
    print(prec, {k: f"{v:.2e}" for k, v in out.items()})
