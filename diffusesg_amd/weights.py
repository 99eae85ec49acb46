"""Portable counter-based generator for synthetic weights, noise and flags.

There is no network (no trained checkpoints) and torch's CPU RNG stream is not
reproducible on the device, so every synthetic tensor used by tests, goldens and the
bench comes from this generator: value(seed, name, i) is a pure function computed with
64-bit integer hashing in numpy, identical on every machine (SURVEY §7 step 0, §8d
"Synthetic inputs").
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .spec import ModelConfig, TensorSpec, state_dict_spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def _stream_key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return _mix(np.array([(int(seed) << 32) ^ h], dtype=np.uint64))[0]


def uniform01(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n float64 values in (0,1), a pure function of (seed, name, offset+i)."""
    key = _stream_key(seed, name)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _mix(idx * np.uint64(0xD1342543DE82EF95) + key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """float32 N(mean, std^2) tensor of `shape` (Box-Muller on two hashed uniforms)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name + "/u1", n)
    u2 = uniform01(seed, name + "/u2", n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def coins(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 draws in (0,1): the stand-in for the reference's `np.random.rand()` coin (precond.py:90)."""
    return uniform01(seed, name + "/coin", n)


def synth_tensor(seed: int, t: TensorSpec) -> np.ndarray:
    """One synthetic parameter (fan-in scaled matrices so that activations stay O(1)).

    LayerNorm affine parameters are deliberately non-trivial (1 + 0.1 n, 0.1 n): with the
    reference's (1, 0) init a wrong gamma/beta wiring would go unnoticed.
    """
    if t.kind == "buffer":
        return t.buffer.copy()
    if t.kind in ("matrix", "conv", "convT"):
        return normal(seed, t.key, t.shape, std=1.0 / np.sqrt(max(t.fan_in, 1)))
    if t.kind == "bias":
        return normal(seed, t.key, t.shape, std=0.1)
    if t.kind == "ln_w":
        return normal(seed, t.key, t.shape, std=0.1, mean=1.0)
    if t.kind == "ln_b":
        return normal(seed, t.key, t.shape, std=0.1)
    if t.kind == "relbias":
        return normal(seed, t.key, t.shape, std=0.5)
    raise ValueError(t.kind)


def synth_state_dict(cfg: ModelConfig, seed: int = 0, prefix: str = "") -> Dict[str, np.ndarray]:
    """Synthetic `state_dict` (numpy) for `DiffuseSG` (prefix='') or the precond wrapper (prefix='model.')."""
    return {prefix + t.key: synth_tensor(seed, t) for t in state_dict_spec(cfg)}


def synth_flags(batch: int, n: int, valid) -> np.ndarray:
    """node_flags [B,N] bool with valid[b] leading True entries (ragged when `valid` is a sequence)."""
    if np.isscalar(valid):
        valid = [int(valid)] * batch
    f = np.zeros((batch, n), dtype=bool)
    for b in range(batch):
        f[b, : int(valid[b % len(valid)])] = True
    return f


def mask_adj(x: np.ndarray, flags: np.ndarray) -> np.ndarray:
    """Zero rows and columns of padded nodes; x [B,C,N,N] (restates graph_utils.py:5-38)."""
    f = flags.astype(x.dtype)
    return x * f[:, None, :, None] * f[:, None, None, :]


def mask_node(x: np.ndarray, flags: np.ndarray) -> np.ndarray:
    """Zero rows of padded nodes; x [B,N,C] (restates graph_utils.py:41-86)."""
    return x * flags.astype(x.dtype)[:, :, None]
