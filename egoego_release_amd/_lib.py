"""ctypes binding of libegoego_hip.so (C ABI: include/egoego_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libegoego_hip.so")
PERFDEBUG_LIB_PATH = os.path.join(os.path.dirname(_PKG), "tools", "_build", "libegoego_hip_perfdebug.so")  # tools/ only: `build --perfdebug`

ABI_VERSION = 6
FLAG_NO_GRAPH = 1
FLAG_FC24 = 2  # precision 9 only: fc's weights as three int8 slices (include/egoego_hip.h)
FLAG_FFN16 = 4  # precision 8 only: the FFN contractions on split-bf16 (int8 slices in the attention layer only)
PRED_NOISE, PRED_X0 = 0, 1
NOISE_INJECTED, NOISE_PHILOX, NOISE_NONE = 0, 1, 2
PREC_BF16X3, PREC_BF16X1, PREC_I8X3, PREC_I8X3_FC = 3, 1, 8, 9
K_QKV, K_ATTN, K_FC_LN, K_FFN1, K_FFN2_LN, K_EMBED, K_OUT = range(7)
KERNEL_NAMES = {"qkv": K_QKV, "attn": K_ATTN, "fc_ln": K_FC_LN, "ffn1": K_FFN1, "ffn2_ln": K_FFN2_LN,
                "embed": K_EMBED, "out": K_OUT}
DBG = {"embed": 0, "q": 1, "k": 2, "v": 3, "attn_out": 4, "attn_ln": 5, "ffn_hidden": 6, "out": 7}

c_float_p = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_feats", "d_model", "n_head", "n_dec_layers", "d_k", "d_v",
                                          "max_timesteps", "num_timesteps", "objective", "precision", "flags")]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_q", "b_q", "w_k", "b_k", "w_v", "b_v", "w_fc", "b_fc", "ln1_g", "ln1_b",
                                          "w_1", "b_1", "w_2", "b_2", "ln2_g", "ln2_b")]


class Weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("start_conv_w", "start_conv_b", "position_vec", "linear_out_w",
                                          "linear_out_b", "time_mlp1_w", "time_mlp1_b", "time_mlp3_w",
                                          "time_mlp3_b")] + [("layers", C.POINTER(LayerWeights))]


class Schedule(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("posterior_mean_coef1", "posterior_mean_coef2",
                                         "posterior_log_variance_clipped", "sqrt_recip_alphas_cumprod",
                                         "sqrt_recipm1_alphas_cumprod", "alphas_cumprod")]


EXPORTS = ["egoego_abi_version", "egoego_last_error", "egoego_ctx_create", "egoego_ctx_destroy",
           "egoego_load_weights", "egoego_load_schedule", "egoego_workspace_bytes", "egoego_denoise",
           "egoego_p_sample", "egoego_sample_loop", "egoego_ddim_loop", "egoego_rot6d_to_matrix", "egoego_convert_model_res", "egoego_window_prefix", "egoego_window_condition",
           "egoego_profile_begin", "egoego_profile_end", "egoego_debug_stage", "egoego_last_kernel_name", "egoego_outlier_stats"]
OUTLIER_SITES = 16

_lib = None


class EgoEgoHipError(RuntimeError):
    pass


def use_perfdebug_build(tag=None):
    """tools/*_trace.py only: bind the perf-debug build (per-block timestamps, stage ablation; `tag` selects a variant
    built with `build --perfdebug --tag=X -D...`) instead of the product library.  Must be called before the first load()."""
    global LIB_PATH
    if _lib is not None:
        raise EgoEgoHipError("use_perfdebug_build() must be called before the library is loaded")
    tag = tag if tag is not None else os.environ.get("EGOEGO_PERFDEBUG_TAG", "")
    LIB_PATH = PERFDEBUG_LIB_PATH.replace(".so", f"_{tag}.so") if tag else PERFDEBUG_LIB_PATH


def load():
    """dlopen the library and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgoEgoHipError(
            f"{LIB_PATH} is missing: build it with `python -m egoego_release_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for the sampling path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
    lib.egoego_abi_version.restype = i32
    lib.egoego_last_error.restype = C.c_char_p
    lib.egoego_ctx_create.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    lib.egoego_ctx_destroy.argtypes = [vp]
    lib.egoego_ctx_destroy.restype = None
    lib.egoego_load_weights.argtypes = [vp, C.POINTER(Weights), vp]
    lib.egoego_load_schedule.argtypes = [vp, C.POINTER(Schedule), vp]
    lib.egoego_workspace_bytes.argtypes = [vp, i32, i32]
    lib.egoego_workspace_bytes.restype = sz
    lib.egoego_denoise.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]
    lib.egoego_p_sample.argtypes = [vp, vp, vp, vp, vp, vp, i32, u64, i64, i32, i32, i32, vp, sz, vp]
    lib.egoego_sample_loop.argtypes = [vp, vp, vp, i32, i32, vp, i32, u64, i64, vp, i32, vp, i32, i32, vp, sz, vp]
    lib.egoego_ddim_loop.argtypes = [vp, vp, vp, C.POINTER(C.c_int32), i32, C.c_float, vp, i32, u64, i64, i32, i32, vp, sz, vp]
    lib.egoego_rot6d_to_matrix.argtypes = [vp, vp, i64, vp]
    lib.egoego_convert_model_res.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int32), i32, i32, i32, vp, vp, vp, vp]
    lib.egoego_window_condition.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.egoego_window_prefix.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int32), i32, i32, i32, i32, vp, vp]
    lib.egoego_profile_begin.argtypes = [vp, i32]
    lib.egoego_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i32)]
    lib.egoego_debug_stage.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, i32, i32, vp, sz, vp]
    lib.egoego_last_kernel_name.argtypes = [vp, i32]
    lib.egoego_last_kernel_name.restype = C.c_char_p
    lib.egoego_outlier_stats.argtypes = [vp, i32, i32, vp, sz, c_float_p, i32, i32, vp]
    if lib.egoego_abi_version() != ABI_VERSION:
        raise EgoEgoHipError(f"ABI mismatch: library {lib.egoego_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EgoEgoHipError(f"libegoego_hip error {rc}: {load().egoego_last_error().decode()}")
