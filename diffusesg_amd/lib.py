"""ctypes binding of libdsg.so (include/dsg.h).  There is NO fallback: if the HIP library is
missing or no GPU is visible, construction fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdsg.so")
MAX_LAYERS = 8

# every symbol include/dsg.h declares (tests/test_abi.py checks the built library exports them)
EXPORTS = [
    "dsg_create", "dsg_destroy", "dsg_last_error", "dsg_version", "dsg_abi_version", "dsg_set_weight", "dsg_finalize_weights",
    "dsg_num_weight_keys", "dsg_weight_key", "dsg_workspace_bytes", "dsg_denoise", "dsg_precond", "dsg_sample",
    "dsg_sigma_schedule", "dsg_debug_tap", "dsg_debug_clear_taps", "dsg_decode_bits", "dsg_decode", "dsg_profile_forward", "dsg_set_option",
    "dsg_get_option", "dsg_gen_noise", "dsg_train_inputs", "dsg_rainbow_loss", "dsg_rainbow_loss_backward", "dsg_noise_embed", "dsg_affine_width", "dsg_block_train", "dsg_train_grads", "dsg_train_step_grads", "dsg_train_self_cond", "dsg_train_bind_params", "dsg_adam_step", "dsg_ema_update", "dsg_debug_gemm", "dsg_debug_gemm_bx", "dsg_debug_attn_bx", "dsg_debug_qkv_attn_bx", "dsg_debug_projmlp_bx", "dsg_debug_mlp_bx", "dsg_profile_clock_ghz",
]


class DsgError(RuntimeError):
    pass


class DsgConfig(C.Structure):
    _fields_ = [("max_node_num", C.c_int32), ("c_adj", C.c_int32), ("c_node", C.c_int32), ("embed_dim", C.c_int32),
                ("num_layers", C.c_int32), ("depths", C.c_int32 * MAX_LAYERS), ("num_heads", C.c_int32 * MAX_LAYERS),
                ("window_size", C.c_int32), ("mlp_ratio", C.c_int32), ("self_condition", C.c_int32)]


class DsgSamplerCfg(C.Structure):
    _fields_ = [("num_steps", C.c_int32), ("heun", C.c_int32),
                ("S_churn", C.c_float), ("S_min", C.c_float), ("S_max", C.c_float), ("S_noise", C.c_float),
                ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("rho", C.c_double),
                ("use_graph", C.c_int32), ("reserved", C.c_int32)]


class DsgSampleStats(C.Structure):
    _fields_ = [("precond_calls", C.c_int64), ("net_forwards", C.c_int64), ("graph_replays", C.c_int64)]


_lib = None


# ABI generation of include/dsg.h this binding was written against (DSG_ABI_VERSION there): argument lists changed between rounds
# (e.g. iou_loss_type in the loss entries), and a stale libdsg.so would take shifted arguments without any error
ABI_VERSION = 4


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen libdsg.so and declare the prototypes of include/dsg.h.  `path`: another build of the library -- the kernel-variant
    A/B builds of the dev tools pass it explicitly (tools/bx_bench.py); the product and the tests always load the in-tree one, and
    no environment variable redirects the load."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != getattr(_lib, "_dsg_path", None):
            raise DsgError(f"libdsg.so is already loaded from {_lib._dsg_path}; cannot switch to {path}")
        return _lib
    lib_path = os.path.abspath(path) if path is not None else LIB_PATH
    if not os.path.exists(lib_path):
        raise DsgError(f"{lib_path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch first makes the
    # dynamic loader resolve libdsg.so's NEEDED libamdhip64.so.7 to that already-loaded copy, so streams and device
    # pointers are shared with torch.  (Loaded the other way round, two runtimes end up in the process and the second
    # one to initialise cannot see the GPU.)
    import torch  # noqa: F401
    L = C.CDLL(lib_path)
    L._dsg_path = lib_path
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    try:
        L.dsg_abi_version.restype = i32
        abi = int(L.dsg_abi_version())
    except AttributeError:
        abi = -1
    if abi != ABI_VERSION:
        raise DsgError(f"{lib_path} has ABI generation {abi}, this binding needs {ABI_VERSION}: rebuild it "
                       f"(python -c 'import __graft_entry__ as g; g.build()')")
    L.dsg_create.argtypes = [C.POINTER(DsgConfig), C.POINTER(vp)]
    L.dsg_destroy.argtypes = [vp]
    L.dsg_destroy.restype = None
    L.dsg_last_error.argtypes = [vp]
    L.dsg_last_error.restype = C.c_char_p
    L.dsg_version.restype = C.c_char_p
    L.dsg_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32]
    L.dsg_finalize_weights.argtypes = [vp]
    L.dsg_num_weight_keys.argtypes = [vp]
    L.dsg_weight_key.argtypes = [vp, i32]
    L.dsg_weight_key.restype = C.c_char_p
    L.dsg_workspace_bytes.argtypes = [vp, i32]
    L.dsg_workspace_bytes.restype = C.c_size_t
    L.dsg_denoise.argtypes = [vp, i32] + [vp] * 9
    L.dsg_precond.argtypes = [vp, i32] + [vp] * 6 + [i32] + [vp] * 3
    L.dsg_sample.argtypes = [vp, C.POINTER(DsgSamplerCfg), i32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp,
                             vp, i32, vp, vp, vp, vp, C.POINTER(DsgSampleStats), vp]
    L.dsg_sigma_schedule.argtypes = [C.POINTER(DsgSamplerCfg), vp, vp, vp, vp]
    L.dsg_debug_tap.argtypes = [vp, C.c_char_p, vp, i64]
    L.dsg_debug_clear_taps.argtypes = [vp]
    L.dsg_debug_clear_taps.restype = None
    L.dsg_set_option.argtypes = [vp, C.c_char_p, i32]
    L.dsg_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i32)]
    L.dsg_gen_noise.argtypes = [vp, i32, vp, C.c_uint64, C.c_uint32, vp, vp, vp]
    L.dsg_profile_clock_ghz.argtypes = [vp]
    L.dsg_profile_clock_ghz.restype = C.c_double
    L.dsg_debug_gemm.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    L.dsg_debug_gemm_bx.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, C.POINTER(C.c_float), vp]
    L.dsg_debug_attn_bx.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_float), vp]
    L.dsg_debug_projmlp_bx.argtypes = [i32, i32] + [vp] * 9 + [i32, vp, i32, C.POINTER(C.c_float), vp]
    L.dsg_debug_qkv_attn_bx.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, C.POINTER(C.c_float), vp]
    L.dsg_debug_mlp_bx.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, C.POINTER(C.c_float), vp]
    L.dsg_train_inputs.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp]
    L.dsg_rainbow_loss.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, i32, vp, vp, vp]
    L.dsg_block_train.argtypes = [vp, C.c_char_p, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), vp]
    L.dsg_train_grads.argtypes = [vp, i32] + [vp] * 10 + [i32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), vp]
    L.dsg_train_step_grads.argtypes = [vp, i32] + [vp] * 9 + [C.c_float] * 3 + [i32] + [vp] * 4 + [i32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), vp]
    L.dsg_train_self_cond.argtypes = [vp, i32] + [vp] * 7
    L.dsg_train_bind_params.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
    L.dsg_adam_step.argtypes = [i32] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_int64), i32] + [C.c_float] * 6 + [C.POINTER(C.c_float), vp]
    L.dsg_ema_update.argtypes = [i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_float, vp]
    L.dsg_noise_embed.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.dsg_affine_width.argtypes = [vp]
    L.dsg_affine_width.restype = i32
    L.dsg_rainbow_loss_backward.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, i32, vp, vp, vp, vp, vp, vp]
    L.dsg_profile_forward.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.dsg_decode_bits.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.dsg_decode.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    _lib = L
    return L


# `--edge_encoding` / `--node_encoding` of the reference (R/utils/arg_parser.py) -> DSG_ENC_* (include/dsg.h)
ENCODINGS = {"bits": 0, "one_hot": 1, "ddpm": 2}


# iou_loss_type of the trainer's bounding-box term (trainer_node_adj.py:138-153) -> DSG_IOU_* (include/dsg.h)
IOU_LOSS_TYPES = {"iou": 0, "giou": 1, "giou_squared": 2, "diou": 3, "ciou": 4}


def iou_loss_type_code(name: str) -> int:
    if name not in IOU_LOSS_TYPES:
        raise NotImplementedError(name)   # the reference's `else: raise NotImplementedError` (trainer_node_adj.py:153-154)
    return IOU_LOSS_TYPES[name]


def make_config(cfg) -> DsgConfig:
    c = DsgConfig()
    c.max_node_num, c.c_adj, c.c_node, c.embed_dim = cfg.max_node_num, cfg.c_adj, cfg.c_node, cfg.embed_dim
    c.num_layers = cfg.num_layers
    for i in range(cfg.num_layers):
        c.depths[i] = cfg.depths[i]
        c.num_heads[i] = cfg.num_heads[i]
    c.window_size, c.mlp_ratio, c.self_condition = cfg.window_size, cfg.mlp_ratio, int(cfg.self_condition)
    return c


def make_sampler_cfg(num_steps: int, solver: str = "heun", S_churn: float = 40.0, S_min: float = 0.05,
                     S_max: float = 50.0, S_noise: float = 1.003, sigma_min: float = 0.002, sigma_max: float = 80.0,
                     rho: float = 7.0, use_graph: bool = True) -> DsgSamplerCfg:
    if solver not in ("heun", "euler"):
        raise ValueError(solver)
    return DsgSamplerCfg(int(num_steps), 1 if solver == "heun" else 0, S_churn, S_min, S_max, S_noise,
                         sigma_min, sigma_max, rho, int(bool(use_graph)), 0)


class Handle:
    """Owns one dsg_handle.  Raises DsgError with the library's message on any non-zero status."""

    def __init__(self, cfg):
        self.L = load()
        self.cfg = cfg
        self._c = make_config(cfg)
        self._h = C.c_void_p()
        rc = self.L.dsg_create(C.byref(self._c), C.byref(self._h))
        if rc != 0:
            why = self.L.dsg_last_error(None)
            raise DsgError(f"dsg_create failed with status {rc}: {why.decode() if why else ''} (there is no CPU fallback)")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.L.dsg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != 0:
            raise DsgError(f"{what}: status {rc}: {self.L.dsg_last_error(self._h).decode()}")

    @property
    def raw(self):
        return self._h

    def weight_keys(self):
        return [self.L.dsg_weight_key(self._h, i).decode() for i in range(self.L.dsg_num_weight_keys(self._h))]

    def set_weight(self, key: str, ptr: int, shape, is_device: bool):
        shp = (C.c_int64 * len(shape))(*shape)
        self.check(self.L.dsg_set_weight(self._h, key.encode(), C.c_void_p(ptr), shp, len(shape), int(is_device)),
                   f"dsg_set_weight({key})")

    def set_option(self, name: str, value: int):
        self.check(self.L.dsg_set_option(self._h, name.encode(), int(value)), f"dsg_set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_int32(0)
        self.check(self.L.dsg_get_option(self._h, name.encode(), C.byref(v)), f"dsg_get_option({name})")
        return int(v.value)

    def precision_mode(self) -> str:
        """'f32' | 'f32-split' | 'bf16': the GEMM arithmetic the handle will actually run (options or DSG_* env defaults)."""
        return "f32-split" if self.get_option("gemm_split") else ("bf16" if self.get_option("gemm_bf16") else "f32")

    def finalize(self):
        self.check(self.L.dsg_finalize_weights(self._h), "dsg_finalize_weights")


def sigma_schedule(scfg: DsgSamplerCfg):
    """(sigma_steps f64, t_hat f32, noise_coef f32, h f32) of the loop -- host-only, no GPU needed."""
    import numpy as np
    T = scfg.num_steps
    sg, th, nz, hs = np.empty(T, np.float64), np.empty(T, np.float32), np.empty(T, np.float32), np.empty(T, np.float32)
    rc = load().dsg_sigma_schedule(C.byref(scfg), sg.ctypes.data, th.ctypes.data, nz.ctypes.data, hs.ctypes.data)
    if rc != 0:
        raise DsgError(f"dsg_sigma_schedule: status {rc}")
    return sg, th, nz, hs
