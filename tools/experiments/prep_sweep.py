"""Whole-chain probe value of the prepared precision 9 on the trained-like checkpoint for other calibration sizes / GPTQ dampings (round 4:
4.6e-4 ... 7.6e-4 with no trend: the whole-chain figure of one packing is a noisy statistic, tuning the preparation does not move it).
    python tools/experiments/prep_sweep.py   (GPU)"""
import os, sys, time, warnings, functools, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))  # run from the repo root
warnings.simplefilter("ignore")
from egoego_release_amd import ModelConfig
from egoego_release_amd.model import CondGaussianDiffusion
from egoego_release_amd import precision as P
from make_trained_like_checkpoint import train_like
T = 120
cfg = ModelConfig(max_timesteps=T + 1)
sd, info = train_like(3000, 0, "cuda", T)
m = CondGaussianDiffusion(**cfg.ctor_kwargs()); m.load_state_dict(sd, strict=False); m = m.cuda()
m.hip_precision = 3; m.hip_engine()
orig_cr = P.compensated_rounding
fr_sets = {"5 timesteps (shipped)": (0.0, 0.02, 0.1, 0.5, 1.0),
           "12 timesteps": (0.0, 0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.7, 0.85, 1.0),
           "24 timesteps": tuple(i / 23 for i in range(24))}
for fname, fr in fr_sets.items():
    P.CAL_TIMESTEP_FRACTIONS = fr
    probe = P.PrecisionProbe(m, tail=50)
    t0 = time.time(); calib = probe.calibration(); tc = time.time() - t0
    for damp in (0.01, 0.05, 0.2):
        P.compensated_rounding = functools.partial(orig_cr, damp=damp)
        for prec in (9,):
            t0 = time.time()
            sd_s, rs = P.prepare_int8_state(probe.sd, calib, prec, shift=True)
            tp = time.time() - t0
            e1, _ = probe.error(sd_s, prec, rs)
            e2 = probe.chain_error(sd_s, prec, rs)
            print(f"{fname:22s} damp {damp:4.2f} precision {prec}: stage 1 {e1:.2e}  whole chain {e2:.2e}   (calibration {tc:.1f} s, preparation {tp:.1f} s)", flush=True)
    probe.close()
