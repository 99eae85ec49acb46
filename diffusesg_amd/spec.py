"""Structural description of the DiffuseSG denoiser: config, channel table and state-dict layout.

Everything here is *data about* the reference network (names, shapes, constant index
buffers), restated from the reference source so that the HIP path, the oracle and the
golden generator agree on one description.  No arithmetic of the hot path lives here.

Reference citations (R/ = /root/reference/DiffuseSG/):
  * module tree / parameter names: R/model/diffusesg/diffusesg.py:9-26,60-106,158-230,298-372,516-560,587-720
  * channel table:                 R/utils/sg_utils.py:348-428
  * ctor kwargs used by the repo:  R/utils/learning_utils.py:47-64
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

NOISE_EMB = 512  # R/model/diffusesg/diffusesg.py:643
HEAD_DIM = 32    # embed_dim / num_heads[0] = 96 / 3 at every level (learning_utils.py:56)


@dataclass(frozen=True)
class ModelConfig:
    """Mirror of the kwargs `get_network` passes to `DiffuseSG` (learning_utils.py:47-64)."""
    max_node_num: int            # img_size == N (patch_size is always 1 in the reference configs)
    c_adj: int                   # out_chans_adj == adjacency channels of the state
    c_node: int                  # out_chans_node == node channels of the state
    embed_dim: int = 96
    depths: Tuple[int, ...] = (1, 1, 3, 1)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    window_size: int = 8
    mlp_ratio: int = 4
    self_condition: bool = True
    patch_size: int = 1

    def __post_init__(self):
        if self.patch_size != 1:
            raise NotImplementedError("only patch_size=1 (both reference YAMLs) is supported")
        if len(self.depths) > len(self.num_heads):
            raise ValueError("depths longer than num_heads")
        for i in range(len(self.depths)):
            c = self.embed_dim << i
            if c % self.num_heads[i] != 0:
                raise ValueError("dim not divisible by heads")
        if self.max_node_num % (1 << (len(self.depths) - 1)) != 0:
            raise ValueError("max_node_num must be divisible by 2**(num_layers-1)")

    @property
    def num_layers(self) -> int:
        return len(self.depths)

    @property
    def in_chans(self) -> int:
        """Channels of the assembled [B,C,N,N] input (diffusesg.py:790-802)."""
        base = self.c_adj + 2 * self.c_node
        return base * 2 if self.self_condition else base

    def level_dim(self, lvl: int) -> int:
        return self.embed_dim << lvl

    def level_res(self, lvl: int) -> int:
        return self.max_node_num >> lvl

    def level_window(self, lvl: int) -> int:
        """Effective window (diffusesg.py:189-192): clipped to the resolution."""
        r = self.level_res(lvl)
        return r if r <= self.window_size else self.window_size

    def block_shift(self, lvl: int, j: int) -> int:
        """Shift of block j at level lvl (diffusesg.py:459, :189-192)."""
        r = self.level_res(lvl)
        if r <= self.window_size:
            return 0
        return 0 if j % 2 == 0 else self.window_size // 2


def sg_channels(dataset_name: str, encoding: str, node_bbox: bool = True) -> Dict[str, int]:
    """Channel table for scene graphs (restates R/utils/sg_utils.py:348-409, non node-only branch)."""
    if "visual_genome" in dataset_name:
        raw_node, raw_adj = 150, 51
    elif "coco_stuff" in dataset_name:
        raw_node, raw_adj = 171, 7
    else:
        raise NotImplementedError(dataset_name)
    if encoding == "one_hot":
        n_node, n_adj = raw_node, raw_adj
    elif encoding == "bits":
        n_node, n_adj = int(math.ceil(math.log2(raw_node))), int(math.ceil(math.log2(raw_adj)))
    elif encoding == "ddpm":
        n_node, n_adj = 1, 1
    else:
        raise NotImplementedError(encoding)
    c_node = n_node + (4 if node_bbox else 0)
    return {
        "raw_num_node_type": raw_node, "raw_num_adj_type": raw_adj,
        "c_adj": n_adj, "c_node": c_node,
        "in_chans": n_adj + 2 * c_node,   # before the x2 of self-conditioning (diffusesg.py:631-632)
        "out_chans_adj": n_adj, "out_chans_node": c_node,
    }


# ----------------------------------------------------------------------------------------------
# named configurations used by tests / bench

def tiny_config() -> ModelConfig:
    return ModelConfig(max_node_num=8, c_adj=6, c_node=12, depths=(1, 1), num_heads=(3, 6),
                       window_size=4, self_condition=True)


def vg_config() -> ModelConfig:
    ch = sg_channels("visual_genome", "bits")
    return ModelConfig(max_node_num=64, c_adj=ch["c_adj"], c_node=ch["c_node"],
                       depths=(1, 1, 3, 1), window_size=8, self_condition=True)


def coco_config() -> ModelConfig:
    ch = sg_channels("coco_stuff", "bits")
    return ModelConfig(max_node_num=40, c_adj=ch["c_adj"], c_node=ch["c_node"],
                       depths=(1, 2, 6), num_heads=(3, 6, 12), window_size=10, self_condition=True)


NAMED_CONFIGS = {"tiny": tiny_config, "vg": vg_config, "coco": coco_config}


# ----------------------------------------------------------------------------------------------
# constant buffers of the reference (recomputed here, compared with the reference by tools/gen_golden.py)

def relative_position_index(ws: int) -> np.ndarray:
    """[W,W] int64 index into the (2ws-1)^2 bias table (diffusesg.py:88-98)."""
    p = np.arange(ws * ws)
    pi, pj = p // ws, p % ws
    di = pi[:, None] - pi[None, :] + ws - 1
    dj = pj[:, None] - pj[None, :] + ws - 1
    return (di * (2 * ws - 1) + dj).astype(np.int64)


def shift_attn_mask(res: int, ws: int, shift: int) -> Optional[np.ndarray]:
    """[nW,W,W] float32 mask in {0,-100} for a shifted block, None if shift == 0 (diffusesg.py:207-228)."""
    if shift == 0:
        return None
    region = np.zeros((res, res), dtype=np.int64)
    bounds = [(0, res - ws), (res - ws, res - shift), (res - shift, res)]
    cnt = 0
    for (h0, h1) in bounds:
        for (w0, w1) in bounds:
            region[h0:h1, w0:w1] = cnt
            cnt += 1
    nw = res // ws
    # window_partition of the region map (diffusesg.py:28-40)
    win = region.reshape(nw, ws, nw, ws).transpose(0, 2, 1, 3).reshape(nw * nw, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return np.where(diff != 0, np.float32(-100.0), np.float32(0.0)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# state-dict layout

@dataclass
class TensorSpec:
    key: str                      # name inside DiffuseSG.state_dict()
    shape: Tuple[int, ...]
    kind: str                     # 'matrix' | 'bias' | 'ln_w' | 'ln_b' | 'relbias' | 'conv' | 'convT' | 'buffer'
    fan_in: int = 0
    buffer: Optional[np.ndarray] = field(default=None, repr=False)


def _block_specs(prefix: str, c: int, heads: int, ws: int, res: int, shift: int, mlp_ratio: int) -> List[TensorSpec]:
    w = ws * ws
    out = [
        TensorSpec(f"{prefix}.affine.weight", (2 * c, NOISE_EMB), "matrix", NOISE_EMB),
        TensorSpec(f"{prefix}.affine.bias", (2 * c,), "bias"),
        TensorSpec(f"{prefix}.norm1.weight", (c,), "ln_w"),
        TensorSpec(f"{prefix}.norm1.bias", (c,), "ln_b"),
        TensorSpec(f"{prefix}.attn.relative_position_bias_table", ((2 * ws - 1) ** 2, heads), "relbias"),
        TensorSpec(f"{prefix}.attn.relative_position_index", (w, w), "buffer", buffer=relative_position_index(ws)),
        TensorSpec(f"{prefix}.attn.qkv.weight", (3 * c, c), "matrix", c),
        TensorSpec(f"{prefix}.attn.qkv.bias", (3 * c,), "bias"),
        TensorSpec(f"{prefix}.attn.proj.weight", (c, c), "matrix", c),
        TensorSpec(f"{prefix}.attn.proj.bias", (c,), "bias"),
        TensorSpec(f"{prefix}.norm2.weight", (c,), "ln_w"),
        TensorSpec(f"{prefix}.norm2.bias", (c,), "ln_b"),
        TensorSpec(f"{prefix}.mlp.fc1.weight", (mlp_ratio * c, c), "matrix", c),
        TensorSpec(f"{prefix}.mlp.fc1.bias", (mlp_ratio * c,), "bias"),
        TensorSpec(f"{prefix}.mlp.fc2.weight", (c, mlp_ratio * c), "matrix", mlp_ratio * c),
        TensorSpec(f"{prefix}.mlp.fc2.bias", (c,), "bias"),
    ]
    mask = shift_attn_mask(res, ws, shift)
    if mask is not None:
        out.append(TensorSpec(f"{prefix}.attn_mask", mask.shape, "buffer", buffer=mask))
    return out


def state_dict_spec(cfg: ModelConfig) -> List[TensorSpec]:
    """All tensors of `DiffuseSG(...).state_dict()` for this config, in a deterministic order."""
    e = cfg.embed_dim
    L = cfg.num_layers
    s: List[TensorSpec] = [
        TensorSpec("patch_embed.affine.weight", (2 * e, NOISE_EMB), "matrix", NOISE_EMB),
        TensorSpec("patch_embed.affine.bias", (2 * e,), "bias"),
        TensorSpec("patch_embed.proj.weight", (e, cfg.in_chans, 1, 1), "conv", cfg.in_chans),
        TensorSpec("patch_embed.proj.bias", (e,), "bias"),
        TensorSpec("patch_embed.norm.weight", (e,), "ln_w"),
        TensorSpec("patch_embed.norm.bias", (e,), "ln_b"),
    ]
    for lvl in range(L):
        c, res, ws = cfg.level_dim(lvl), cfg.level_res(lvl), cfg.level_window(lvl)
        for j in range(cfg.depths[lvl]):
            s += _block_specs(f"down_layers.{lvl}.blocks.{j}", c, cfg.num_heads[lvl], ws, res,
                              cfg.block_shift(lvl, j), cfg.mlp_ratio)
        if lvl < L - 1:
            p = f"down_layers.{lvl}.downsample"
            s += [TensorSpec(f"{p}.reduction.weight", (2 * c, 4 * c), "matrix", 4 * c),
                  TensorSpec(f"{p}.norm.weight", (4 * c,), "ln_w"),
                  TensorSpec(f"{p}.norm.bias", (4 * c,), "ln_b")]
    for i in range(L):
        lvl = L - 1 - i
        c, res, ws = cfg.level_dim(lvl), cfg.level_res(lvl), cfg.level_window(lvl)
        if i > 0:
            d = 4 * c  # PatchBreakup(dim=dim*4) at the coarser resolution (diffusesg.py:450)
            p = f"up_layers.{i}.upsample"
            s += [TensorSpec(f"{p}.pre_linear.weight", (d, d), "matrix", d),
                  TensorSpec(f"{p}.norm.weight", (d,), "ln_w"),
                  TensorSpec(f"{p}.norm.bias", (d,), "ln_b"),
                  TensorSpec(f"{p}.post_linear.weight", (d // 4, d // 4), "matrix", d // 4),
                  TensorSpec(f"{p}.post_norm.weight", (d // 4,), "ln_w"),
                  TensorSpec(f"{p}.post_norm.bias", (d // 4,), "ln_b")]
        for j in range(cfg.depths[lvl]):
            s += _block_specs(f"up_layers.{i}.blocks.{j}", c, cfg.num_heads[lvl], ws, res,
                              cfg.block_shift(lvl, j), cfg.mlp_ratio)
    s += [
        TensorSpec("read_out.0.weight", (e, e, 1, 1), "convT", e),   # ConvTranspose2d: [in,out,1,1]
        TensorSpec("read_out.0.bias", (e,), "bias"),
        TensorSpec("read_out.1.weight", (e, e, 1, 1), "conv", e),
        TensorSpec("read_out.1.bias", (e,), "bias"),
        TensorSpec("read_out.2.weight", (e, e, 1, 1), "conv", e),
        TensorSpec("read_out.2.bias", (e,), "bias"),
        TensorSpec("map_layer0.weight", (NOISE_EMB, e), "matrix", e),
        TensorSpec("map_layer0.bias", (NOISE_EMB,), "bias"),
        TensorSpec("map_layer1.weight", (NOISE_EMB, NOISE_EMB), "matrix", NOISE_EMB),
        TensorSpec("map_layer1.bias", (NOISE_EMB,), "bias"),
        TensorSpec("norm.weight", (e,), "ln_w"),
        TensorSpec("norm.bias", (e,), "ln_b"),
        TensorSpec("readout_adj_mlp.fc1.weight", (e, e), "matrix", e),
        TensorSpec("readout_adj_mlp.fc1.bias", (e,), "bias"),
        TensorSpec("readout_adj_mlp.fc2.weight", (cfg.c_adj, e), "matrix", e),
        TensorSpec("readout_adj_mlp.fc2.bias", (cfg.c_adj,), "bias"),
        TensorSpec("readout_node_mlp.fc1.weight", (e, e), "matrix", e),
        TensorSpec("readout_node_mlp.fc1.bias", (e,), "bias"),
        TensorSpec("readout_node_mlp.fc2.weight", (cfg.c_node, e), "matrix", e),
        TensorSpec("readout_node_mlp.fc2.bias", (cfg.c_node,), "bias"),
    ]
    return s


def num_parameters(cfg: ModelConfig) -> int:
    return sum(int(np.prod(t.shape)) for t in state_dict_spec(cfg) if t.kind != "buffer")


def flops_per_forward(cfg: ModelConfig) -> int:
    """2*MAC over linears, 1x1 convs, QK^T and PV for one sample, one network forward."""
    e, n = cfg.embed_dim, cfg.max_node_num
    t0 = n * n
    f = 0
    f += 2 * (e * NOISE_EMB + NOISE_EMB * NOISE_EMB)                 # noise-embedding MLP
    f += 2 * t0 * cfg.in_chans * e                                   # patch-embed conv
    f += 2 * NOISE_EMB * 2 * e                                       # patch-embed affine

    def block(lvl):
        c, t, w = cfg.level_dim(lvl), cfg.level_res(lvl) ** 2, cfg.level_window(lvl) ** 2
        lin = 2 * t * c * (3 * c + c + 2 * cfg.mlp_ratio * c)
        att = 2 * 2 * t * w * c
        return lin + att + 2 * NOISE_EMB * 2 * c
    L = cfg.num_layers
    for lvl in range(L):
        f += 2 * cfg.depths[lvl] * block(lvl)                        # down + up use the same depths
        if lvl < L - 1:
            c, t = cfg.level_dim(lvl), cfg.level_res(lvl) ** 2
            f += 2 * (t // 4) * 4 * c * 2 * c                        # PatchMerging
            d, tc = 4 * c, cfg.level_res(lvl + 1) ** 2
            f += 2 * tc * d * d + 2 * (4 * tc) * (d // 4) ** 2       # PatchBreakup
    f += 3 * 2 * t0 * e * e                                          # read_out
    f += 2 * t0 * (e * e + e * cfg.c_adj)                            # adj head
    f += 2 * n * (e * e + e * cfg.c_node)                            # node head
    return f
