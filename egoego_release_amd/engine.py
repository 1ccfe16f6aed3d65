"""HipEngine: owns one libegoego_hip context on one GPU and feeds it torch tensors.

PyTorch is plumbing here (device memory, the current HIP stream); all arithmetic of the sampling
step happens in the HIP library.
"""
import ctypes as C

import torch

from . import _lib

TR = "denoise_fn.motion_transformer."


def _f32c(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class HipEngine:
    def __init__(self, cfg_dict, state_dict, device, precision=_lib.PREC_I8X3_FC, flags=0, row_shift=None):
        """cfg_dict: d_feats, d_model, n_head, n_dec_layers, d_k, d_v, max_timesteps, num_timesteps,
        objective ('pred_x0' | 'pred_noise').  state_dict: reference-layout tensors (any device).
        row_shift: the per-feature constants a mean-shifted state dict (precision.prepare_int8_state) leaves out of the rows it
        stores — {'embed' | (layer, 'attn_ln') | (layer, 'out'): [512]}; only the debug taps need them (added back there)."""
        self.row_shift = row_shift or {}
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EgoEgoHipError("HipEngine needs a cuda (ROCm) device; there is no CPU path")
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.cfg = dict(cfg_dict)
        self.precision = precision
        obj = {"pred_x0": _lib.PRED_X0, "pred_noise": _lib.PRED_NOISE}.get(self.cfg["objective"])
        if obj is None:
            raise ValueError(f"unknown objective {self.cfg['objective']}")
        c = _lib.Config(self.cfg["d_feats"], self.cfg["d_model"], self.cfg["n_head"], self.cfg["n_dec_layers"],
                        self.cfg["d_k"], self.cfg["d_v"], self.cfg["max_timesteps"], self.cfg["num_timesteps"],
                        obj, precision, flags)
        self._ctx = C.c_void_p()
        _lib.check(self.lib.egoego_ctx_create(C.byref(c), self.dev_index, C.byref(self._ctx)))
        self._ws = {}
        self.load(state_dict)

    # ------------------------------------------------------------------ weights
    def load(self, sd):
        dev = self.device
        keep = []  # fp32 device copies must outlive the packing kernels

        def p(name):
            t = _f32c(sd[name], dev)
            keep.append(t)
            return t.data_ptr()

        L = self.cfg["n_dec_layers"]
        layers = (_lib.LayerWeights * L)()
        for i in range(L):
            a, f = TR + f"layer_stack.{i}.self_attn.", TR + f"layer_stack.{i}.pos_ffn."
            lw = layers[i]
            lw.w_q, lw.b_q = p(a + "w_q.weight"), p(a + "w_q.bias")
            lw.w_k, lw.b_k = p(a + "w_k.weight"), p(a + "w_k.bias")
            lw.w_v, lw.b_v = p(a + "w_v.weight"), p(a + "w_v.bias")
            lw.w_fc, lw.b_fc = p(a + "fc.weight"), p(a + "fc.bias")
            lw.ln1_g, lw.ln1_b = p(a + "layer_norm.weight"), p(a + "layer_norm.bias")
            lw.w_1, lw.b_1 = p(f + "w_1.weight"), p(f + "w_1.bias")
            lw.w_2, lw.b_2 = p(f + "w_2.weight"), p(f + "w_2.bias")
            lw.ln2_g, lw.ln2_b = p(f + "layer_norm.weight"), p(f + "layer_norm.bias")
        w = _lib.Weights()
        w.start_conv_w, w.start_conv_b = p(TR + "start_conv.weight"), p(TR + "start_conv.bias")
        w.position_vec = p(TR + "position_vec.weight")
        w.linear_out_w, w.linear_out_b = p("denoise_fn.linear_out.weight"), p("denoise_fn.linear_out.bias")
        w.time_mlp1_w, w.time_mlp1_b = p("denoise_fn.time_mlp.1.weight"), p("denoise_fn.time_mlp.1.bias")
        w.time_mlp3_w, w.time_mlp3_b = p("denoise_fn.time_mlp.3.weight"), p("denoise_fn.time_mlp.3.bias")
        w.layers = layers
        stream = self._stream()
        with torch.cuda.device(self.dev_index):
            _lib.check(self.lib.egoego_load_weights(self._ctx, C.byref(w), stream))
            torch.cuda.current_stream().synchronize()
        sch = _lib.Schedule()
        hold = []
        for name in ("posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped",
                     "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "alphas_cumprod"):
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if t.numel() != self.cfg["num_timesteps"]:
                raise ValueError(f"schedule buffer {name} has {t.numel()} entries, expected {self.cfg['num_timesteps']}")
            hold.append(t)
            setattr(sch, name, C.cast(t.data_ptr(), _lib.c_float_p))
        _lib.check(self.lib.egoego_load_schedule(self._ctx, C.byref(sch), stream))
        del keep, hold

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev_index).cuda_stream)

    def workspace(self, B, T):
        key = (B, T)
        ws = self._ws.get(key)
        if ws is None:
            n = self.lib.egoego_workspace_bytes(self._ctx, B, T)
            if n == 0:
                raise _lib.EgoEgoHipError(f"unsupported shape B={B} T={T}: {self.lib.egoego_last_error().decode()}")
            if len(self._ws) >= 4:
                self._ws.clear()
            ws = torch.empty(n + 256, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
            off = (-ws.data_ptr()) % 256
            # a fresh workspace holds undefined bytes: clear its outlier monitor (no sync)
            _lib.check(self.lib.egoego_outlier_stats(self._ctx, B, T, ws.data_ptr() + off, ws.numel() - off, None, 0, 1, self._stream()))
        off = (-ws.data_ptr()) % 256
        return ws.data_ptr() + off, ws.numel() - off

    def outlier_stats(self, B, T, reset=True):
        """Largest |value| every row-quantising LayerNorm epilogue has seen on the (B, T) workspace since the last reset:
        a list of 2 * n_dec_layers floats (layer-major: self_attn.layer_norm, pos_ffn.layer_norm); synchronises the stream."""
        n = min(_lib.OUTLIER_SITES, 2 * self.cfg["n_dec_layers"])
        out = (C.c_float * n)()
        ws, nb = self.workspace(B, T)
        _lib.check(self.lib.egoego_outlier_stats(self._ctx, B, T, ws, nb, out, n, 1 if reset else 0, self._stream()))
        return [float(v) for v in out]

    def last_kernel(self, kernel):
        """Name of the kernel variant the launch site `kernel` ('qkv', 'attn', 'fc_ln', 'embed', 'out', ...) last dispatched to."""
        return self.lib.egoego_last_kernel_name(self._ctx, _lib.KERNEL_NAMES[kernel]).decode()

    def _chk(self, t, shape=None, dtype=torch.float32):
        if not t.is_cuda or t.device.index != self.dev_index:
            raise _lib.EgoEgoHipError(f"tensor on {t.device}, engine on cuda:{self.dev_index}")
        if t.dtype != dtype or not t.is_contiguous():
            raise _lib.EgoEgoHipError(f"tensor must be contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise _lib.EgoEgoHipError(f"tensor shape {tuple(t.shape)} != expected {tuple(shape)}")
        return t.data_ptr()

    def _mask(self, row_mask, B, T):
        if row_mask is None:
            return None, None
        m = row_mask.reshape(B, T + 1).to(device=self.device, dtype=torch.float32).contiguous()
        return m, m.data_ptr()

    # ------------------------------------------------------------------ entry points
    def denoise(self, x, x_cond, t, row_mask=None):
        B, T, D = x.shape
        out = torch.empty_like(x)
        ws, n = self.workspace(B, T)
        m, mp = self._mask(row_mask, B, T)
        _lib.check(self.lib.egoego_denoise(self._ctx, self._chk(x), self._chk(x_cond, x.shape),
                                           self._chk(t, (B,), torch.int64), mp, self._chk(out), B, T, ws, n,
                                           self._stream()))
        return out

    def p_sample_(self, x, x_cond, t, noise=None, row_mask=None, clip_denoised=True, noise_mode=None, seed=0,
                  window_offset=0):
        """In-place x <- p_sample(x, t, x_cond)."""
        B, T, D = x.shape
        if noise_mode is None:
            noise_mode = _lib.NOISE_INJECTED if noise is not None else _lib.NOISE_NONE
        ws, n = self.workspace(B, T)
        m, mp = self._mask(row_mask, B, T)
        npnt = self._chk(noise, x.shape) if noise is not None else None
        _lib.check(self.lib.egoego_p_sample(self._ctx, self._chk(x), self._chk(x_cond, x.shape),
                                            self._chk(t, (B,), torch.int64), mp, npnt, noise_mode, seed, window_offset,
                                            1 if clip_denoised else 0, B, T, ws, n, self._stream()))
        return x

    def sample_loop_(self, x, x_cond, t_start, n_steps, noise=None, noise_mode=None, seed=0, window_offset=0,
                     prefix=None, row_mask=None):
        """In-place: n_steps ancestral steps from timestep t_start downwards (row_mask: the padding mask every
        step's denoiser pass applies, [B, 1, T+1] / [B, T+1])."""
        B, T, D = x.shape
        if noise_mode is None:
            noise_mode = _lib.NOISE_INJECTED if noise is not None else _lib.NOISE_PHILOX
        ws, n = self.workspace(B, T)
        npnt = self._chk(noise, (n_steps, B, T, D)) if noise is not None else None
        pp, plen = None, 0
        if prefix is not None:
            plen = prefix.shape[1]
            pp = self._chk(prefix, (B, plen, D))
        m, mp = self._mask(row_mask, B, T)
        _lib.check(self.lib.egoego_sample_loop(self._ctx, self._chk(x), self._chk(x_cond, x.shape), t_start, n_steps,
                                               npnt, noise_mode, seed, window_offset, pp, plen, mp, B, T, ws, n,
                                               self._stream()))
        return x

    def ddim_loop_(self, x, x_cond, timesteps, eta=0.0, noise=None, noise_mode=None, seed=0, window_offset=0):
        """In-place DDIM over the descending `timesteps`; eta > 0 adds eta-weighted noise (injected [n, B, T, D] or Philox)."""
        B, T, D = x.shape
        ws, n = self.workspace(B, T)
        arr = (C.c_int32 * len(timesteps))(*[int(v) for v in timesteps])
        if noise_mode is None:
            noise_mode = _lib.NOISE_INJECTED if noise is not None else (_lib.NOISE_PHILOX if eta > 0 else _lib.NOISE_NONE)
        npnt = self._chk(noise, (len(timesteps), B, T, D)) if noise is not None else None
        _lib.check(self.lib.egoego_ddim_loop(self._ctx, self._chk(x), self._chk(x_cond, x.shape), arr, len(timesteps),
                                             float(eta), npnt, noise_mode, seed, window_offset, B, T, ws, n, self._stream()))
        return x

    def debug_stage(self, x, x_cond, t, layer, stage, row_mask=None):
        B, T, D = x.shape
        H, L = self.cfg["n_head"], T + 1
        sid = _lib.DBG[stage]
        if stage in ("q", "k", "v"):
            out = torch.empty(B, H, L, 256, device=self.device)
        elif stage == "attn_out":
            out = torch.empty(B, L, H * 256, device=self.device)
        else:
            out = torch.empty(B, L, 512, device=self.device)
        ws, n = self.workspace(B, T)
        m, mp = self._mask(row_mask, B, T)
        _lib.check(self.lib.egoego_debug_stage(self._ctx, self._chk(x), self._chk(x_cond, x.shape),
                                               self._chk(t, (B,), torch.int64), mp, layer, sid, out.data_ptr(), B, T,
                                               ws, n, self._stream()))
        sh = self.row_shift.get("embed" if stage == "embed" else (layer, stage))
        if sh is not None:
            sh = sh.to(out.device, out.dtype)
            out += sh.view(H, 1, 256) if stage in ("q", "k", "v") else sh
        return out

    def rot6d_to_matrix(self, d6):
        d6 = d6.contiguous()
        n = d6.numel() // 6
        out = torch.empty(*d6.shape[:-1], 3, 3, device=d6.device, dtype=torch.float32)
        _lib.check(self.lib.egoego_rot6d_to_matrix(self._chk(d6), out.data_ptr(), n, self._stream()))
        return out

    def profile_begin(self, kernel):
        _lib.check(self.lib.egoego_profile_begin(self._ctx, _lib.KERNEL_NAMES[kernel]))

    def profile_end(self):
        us, n = C.c_double(), C.c_int()
        _lib.check(self.lib.egoego_profile_end(self._ctx, C.byref(us), C.byref(n)))
        return us.value, n.value

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.egoego_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self._ws = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
