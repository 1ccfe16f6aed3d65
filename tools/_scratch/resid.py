import sys, os, warnings, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from egoego_release_amd import ModelConfig, make_weights
from chain_tail_b256 import chain_tail
from make_trained_like_checkpoint import train_like
T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
print(f"# residual risk of ACCEPTED checkpoints: what auto accepts on one batch (data seed 31337), run on four OTHER batches of 256 windows; T={T}")
for steps in (0, 10, 30, 70):
    if steps == 0: sd = make_weights(ModelConfig(max_timesteps=T + 1), 0)
    else:
        sd, _ = train_like(steps, 0, "cuda", T); sd = {k: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    for ds in (31337, 1, 2, 3, 4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = chain_tail(sd, T, 256, ("auto", "9"), data_seed=ds, cache=False, log=lambda s: None)
        a, n = r["auto"], r["9"]
        print(f"  {steps:3d} Adam steps, batch seed {ds:5d}: auto -> {a['precision']} {a['form'] or ''} (gain {a['probe'].get('chain gain, max', 0):.2f}/{a['probe'].get('chain gain, median', 0):.2f}); '9 as is' worst of 256 {n['vs3']['max']:.2e} p99 {n['vs3']['p99']:.2e}", flush=True)
