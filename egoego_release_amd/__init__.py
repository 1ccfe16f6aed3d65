"""egoego_release_amd — MI355X-native stage-2 motion-diffusion sampling for EgoEgo.

Only the sampling hot path lives here: HIP kernels + C ABI in csrc/, the Python
mirror of the reference's CondGaussianDiffusion interface in model.py.
"""
from .synthetic import ModelConfig, make_weights, make_head_windows, head_condition_mask  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch.cuda or the .so
    if name in ("CondGaussianDiffusion", "TransformerDiffusionModel"):
        from . import model
        return getattr(model, name)
    if name == "HipEngine":
        from .engine import HipEngine
        return HipEngine
    raise AttributeError(name)
