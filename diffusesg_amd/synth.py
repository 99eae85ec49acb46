"""Synthetic test / bench cases built from the portable generator (weights.py).

The golden generator (tools/gen_golden.py, dev container) and the tests (GPU box) must feed
byte-identical inputs to the reference, the oracle and the HIP path; they all call these helpers.
"""
from __future__ import annotations

import numpy as np

from . import spec as S
from . import weights as W


def small_config() -> S.ModelConfig:
    """16 nodes, window 4, depths (2,1): the smallest net with a shifted, masked block."""
    return S.ModelConfig(max_node_num=16, c_adj=3, c_node=5, depths=(2, 1), num_heads=(3, 6),
                         window_size=4, self_condition=True)


def nosc_config() -> S.ModelConfig:
    """No self-conditioning, one adjacency and one node channel (exercises the squeezed layouts)."""
    return S.ModelConfig(max_node_num=8, c_adj=1, c_node=1, depths=(1, 1), num_heads=(3, 6),
                         window_size=4, self_condition=False)


def onehot_config() -> S.ModelConfig:
    """VG with the 'one_hot' encoding's channel widths (sg_utils.py:348-409: C_adj = 51, C_node = 150 + 4, 718 input
    channels with self-conditioning) on a small grid: the widest channel counts the channel table can produce."""
    ch = S.sg_channels("visual_genome", "one_hot")
    return S.ModelConfig(max_node_num=8, c_adj=ch["c_adj"], c_node=ch["c_node"], depths=(1, 1), num_heads=(3, 6),
                         window_size=4, self_condition=True)


CONFIGS = {"tiny": S.tiny_config, "small": small_config, "nosc": nosc_config,
           "vg": S.vg_config, "coco": S.coco_config, "onehot": onehot_config}

# ragged numbers of valid nodes per sample used by the forward / precond goldens
VALID = {"tiny": [8, 5], "small": [16, 9], "nosc": [8, 3], "vg": [30, 11], "coco": [20, 40], "onehot": [8, 5]}

FWD_C_NOISE = np.array([0.35, -1.2], dtype=np.float32)
PRECOND_SIGMAS = (80.0, 1.5, 0.002)


def case_inputs(cfg: S.ModelConfig, batch: int, valid, seed: int, tag: str, sigma_scale: float = 1.0):
    """flags, adj, node, sc_adj, sc_node (all masked like the sampler would hand them over)."""
    n = cfg.max_node_num
    flags = W.synth_flags(batch, n, valid)
    adj = W.mask_adj(W.normal(seed, f"{tag}/adj", (batch, cfg.c_adj, n, n)) * np.float32(sigma_scale), flags)
    node = W.mask_node(W.normal(seed, f"{tag}/node", (batch, n, cfg.c_node)) * np.float32(sigma_scale), flags)
    sc_adj = W.mask_adj(W.normal(seed, f"{tag}/sc_adj", (batch, cfg.c_adj, n, n)), flags)
    sc_node = W.mask_node(W.normal(seed, f"{tag}/sc_node", (batch, n, cfg.c_node)), flags)
    return flags, adj, node, sc_adj, sc_node


def fwd_case(name: str):
    cfg = CONFIGS[name]()
    return (cfg,) + case_inputs(cfg, 2, VALID[name], 1, f"fwd/{name}")


def precond_case(name: str, si: int):
    cfg = CONFIGS[name]()
    sigma = PRECOND_SIGMAS[si]
    return (cfg, sigma) + case_inputs(cfg, 2, VALID[name], 2, f"pre/{name}/{si}",
                                      sigma_scale=float(np.sqrt(sigma ** 2 + 0.25)))


def sampler_case(cfg: S.ModelConfig, T: int, B: int, valid, seed: int, tag: str, solver: str = "heun"):
    """flags, init_adj, init_node, noise_adj [T,B,..], noise_node [T,B,..], coin draws in (0,1)."""
    n = cfg.max_node_num
    flags = W.synth_flags(B, n, valid)
    init_adj = W.mask_adj(W.normal(seed, f"{tag}/init_adj", (B, cfg.c_adj, n, n)), flags)
    init_node = W.mask_node(W.normal(seed, f"{tag}/init_node", (B, n, cfg.c_node)), flags)
    noise_adj = np.stack([W.normal(seed, f"{tag}/churn_adj/{i}", (B, cfg.c_adj, n, n)) for i in range(T)])
    noise_node = np.stack([W.normal(seed, f"{tag}/churn_node/{i}", (B, n, cfg.c_node)) for i in range(T)])
    ncalls = T if solver == "euler" else 2 * T - 1
    return flags, init_adj, init_node, noise_adj, noise_node, W.coins(seed, tag, ncalls)


def gt_case(cfg: S.ModelConfig, B: int, valid, seed: int = 3):
    """+-1 'bits' ground truth for the sanity-check (known-answer) sampler run."""
    n = cfg.max_node_num
    flags = W.synth_flags(B, n, valid)
    gt_adj = W.mask_adj(np.sign(W.normal(seed, "smp/gt/adj", (B, cfg.c_adj, n, n))).astype(np.float32), flags)
    gt_node = W.mask_node(np.sign(W.normal(seed, "smp/gt/node", (B, n, cfg.c_node))).astype(np.float32), flags)
    return gt_adj, gt_node


INTER_ROWS = 12


def inter_rows(n_rows: int) -> np.ndarray:
    """The token rows of a [B*T, C] tap kept by the per-module fixtures of the full-size nets (fwd_vg / fwd_coco 'rows/<tap>'):
    a fixed pseudo-random, sorted, duplicate-free selection covering both samples."""
    idx = (np.arange(1, 4 * INTER_ROWS + 1, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(n_rows)
    return np.sort(np.unique(idx.astype(np.int64))[:INTER_ROWS]) if n_rows > INTER_ROWS else np.arange(n_rows)


def tap_shapes(cfg: S.ModelConfig) -> dict:
    """name -> (tokens per sample, channels) of every named intermediate activation (dsg_debug_tap / the oracle's taps)."""
    n, E, L = cfg.max_node_num, cfg.embed_dim, len(cfg.depths)
    out = {"patch_embed": (n * n, E), "read_out": (n * n, E)}
    for l in range(L):
        T, C = (n >> l) ** 2, E << l
        for j in range(cfg.depths[l]):
            out[f"down{l}.block{j}"] = (T, C)
        out[f"down{l}"] = (T // 4, 2 * C) if l < L - 1 else (T, C)
    for i in range(L):
        l = L - 1 - i
        T, C = (n >> l) ** 2, E << l
        if i > 0:
            out[f"up{i}.upsample"] = (T, C)
        for j in range(cfg.depths[l]):
            out[f"up{i}.block{j}"] = (T, C)
    return out


# short trajectories of the full-size nets through the reference's sampler (tests/golden/traj_big.npz):
# tag -> (config, T, solver, S_churn, valid nodes per sample, seed, stream tag, explicit coins or None = drawn from the stream)
BIG_TRAJ = {
    "vg_heun6": ("vg", 6, "heun", 40.0, [30, 30], 17, "vg/smp3", [1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0]),
    "vg_euler6": ("vg", 6, "euler", 0.0, [30, 11], 41, "vg/euler6", None),
    "coco_heun6": ("coco", 6, "heun", 40.0, [20, 40], 53, "coco/smp6", None),
}


# Trajectories of REALISTIC length on the full-size networks ("looser stated bound + decoded-bit agreement rate at
# T >= 50"): 50 Heun + churn steps = 99 preconditioned calls + the coins' extra forwards through the reference's own sampler;
# the fixture holds the final raw (adj, node) and the reference-decoded integer graphs ('bits'; dataset, #edge types, #node types)
LONG_TRAJ = {
    "vg_heun50": ("vg", 50, "heun", 40.0, [30, 11], 71, "vg/long50", None, ("visual_genome", 51, 150)),
    "coco_heun50": ("coco", 50, "heun", 40.0, [20, 40], 73, "coco/long50", None, ("coco_stuff", 7, 171)),
}


def long_traj_case(tag: str):
    """cfg, T, solver, S_churn, flags, init_adj, init_node, noise_adj, noise_node, coins (uint8), (dataset, n_adj_type, n_node_type)"""
    name, T, solver, churn, valid, seed, stream, coins, types = LONG_TRAJ[tag]
    cfg = CONFIGS[name]()
    flags, ia, inn, na, nn, cv = sampler_case(cfg, T, len(valid), valid, seed, stream, solver)
    c = np.asarray(coins, np.uint8) if coins is not None else (cv < 0.5).astype(np.uint8)
    return cfg, T, solver, churn, flags, ia, inn, na, nn, c, types


def big_traj_case(tag: str):
    """cfg, T, solver, S_churn, flags, init_adj, init_node, noise_adj, noise_node, coins (uint8) of a BIG_TRAJ case"""
    name, T, solver, churn, valid, seed, stream, coins = BIG_TRAJ[tag]
    cfg = CONFIGS[name]()
    flags, ia, inn, na, nn, cv = sampler_case(cfg, T, len(valid), valid, seed, stream, solver)
    c = np.asarray(coins, np.uint8) if coins is not None else (cv < 0.5).astype(np.uint8)
    return cfg, T, solver, churn, flags, ia, inn, na, nn, c


# (name in sampler.npz, T, solver, S_churn)
SAMPLER_RUNS = (("t8_heun", 8, "heun", 40.0), ("t50_heun", 50, "heun", 40.0), ("t8_euler", 8, "euler", 0.0))
SAMPLER_VALID = [8, 5, 3, 8]


# ---- post-decode fixture (sampler_node_adj.py:222-285): 'bits' samples -> integer graphs + bbox ----
# (dataset name, raw #edge types, raw #node types) per config; B = 3 with full / ragged / nearly-empty flags
DECODE_CASES = {"vg": ("visual_genome", 51, 150, [30, 64, 2]), "coco": ("coco_stuff", 7, 171, [20, 40, 1])}


def decode_case(name: str):
    """flags, raw adj [B,C_adj,N,N], raw node [B,N,C_node] as the sampler would return them: values beyond [-1,1] (the
    decode clamps first), exact zeros on valid entries (0 is NOT > 0: bit 0) and every bit pattern, so codes above
    n_type-1 occur (6 adjacency bits reach 63 > 50, 8 node bits reach 255 > 149) and must be clamped."""
    cfg = CONFIGS[name]()
    n, B = cfg.max_node_num, 3
    flags = W.synth_flags(B, n, DECODE_CASES[name][3])
    adj = (W.normal(9, f"dec/{name}/adj", (B, cfg.c_adj, n, n)) * np.float32(1.5)).astype(np.float32)
    node = (W.normal(9, f"dec/{name}/node", (B, n, cfg.c_node)) * np.float32(1.5)).astype(np.float32)
    adj.reshape(-1)[::7] = 0.0
    node.reshape(-1)[::5] = 0.0
    return cfg, flags, adj, node


# ---- post-decode fixture for the other two encodings (`--edge_encoding` / `--node_encoding` = 'one_hot' | 'ddpm';
# sampler_node_adj.py:222-285 -> attribute_converter, attribute_code.py:13).  (dataset, edge enc, node enc, N, valid counts) ----
DECODE_ENC_CASES = {
    "vg_onehot": ("visual_genome", "one_hot", "one_hot", 64, [30, 64, 2]),
    "vg_ddpm": ("visual_genome", "ddpm", "ddpm", 64, [30, 64, 2]),
    "coco_onehot": ("coco_stuff", "one_hot", "one_hot", 40, [20, 40, 1]),
    "coco_ddpm": ("coco_stuff", "ddpm", "ddpm", 40, [20, 40, 1]),
    "vg_bits_ddpm": ("visual_genome", "bits", "ddpm", 64, [30, 64, 2]),      # mixed: the two flags are independent in the reference
    "coco_onehot_bits": ("coco_stuff", "one_hot", "bits", 40, [20, 40, 1]),
}


def _ddpm_edges(k: int) -> np.ndarray:
    """fp32 values on and right next to every interval edge of attribute_ddpm_to_int's k classes (the reference builds the edges
    with Python floats and compares in fp32), plus the class centres"""
    L = 2.0 / (k - 1)
    vals = []
    for i in range(k):
        c = -1.0 + i * L
        for e in (c - 0.5 * L, c + L * 0.5, c):
            f = np.float32(e)
            vals += [f, np.nextafter(f, np.float32(-2)), np.nextafter(f, np.float32(2))]
    return np.array(vals, np.float32)


def decode_enc_case(name: str):
    """cfg (channel counts of the encoding: one channel per type, or a single one, + 4 bbox channels), flags, raw adj, raw node as the
    sampler would return them.  one_hot: several channels positive (the FIRST wins), none positive (class 0), exact zeros (0 is not
    > 0), values beyond [-1,1].  ddpm: values on and one ulp either side of every interval edge, exact -1 / +1, values beyond the range."""
    dataset, e_adj, e_node, n, valid = DECODE_ENC_CASES[name]
    ch_a, ch_n = S.sg_channels(dataset, e_adj), S.sg_channels(dataset, e_node)
    raw = S.sg_channels(dataset, "one_hot")
    n_adj_type, n_node_type = raw["c_adj"], raw["c_node"] - 4
    cfg = S.ModelConfig(max_node_num=n, c_adj=ch_a["c_adj"], c_node=ch_n["c_node"], depths=(1, 1), num_heads=(3, 6),
                        window_size=8 if n == 64 else 10, self_condition=True)
    B = 3
    flags = W.synth_flags(B, n, valid)
    adj = (W.normal(11, f"decenc/{name}/adj", (B, cfg.c_adj, n, n)) * np.float32(1.5)).astype(np.float32)
    node = (W.normal(11, f"decenc/{name}/node", (B, n, cfg.c_node)) * np.float32(1.5)).astype(np.float32)
    if e_adj == "one_hot":
        adj -= np.float32(2.2)                     # mostly negative: a handful of positive channels per entry, sometimes none
        adj.reshape(-1)[::7] = 0.0
    elif e_adj == "ddpm":
        adj *= np.float32(0.6)
        ed = _ddpm_edges(n_adj_type)
        st = adj.size // len(ed)
        adj.reshape(-1)[: st * len(ed): st] = ed
        adj.reshape(-1)[3::97] = 1.0
        adj.reshape(-1)[5::89] = -1.0
    else:
        adj.reshape(-1)[::7] = 0.0
    attr = node[..., :-4]
    if e_node == "one_hot":
        attr -= np.float32(2.6)
        attr.reshape(B, -1)[:, ::5] = 0.0
    elif e_node == "ddpm":
        attr *= np.float32(0.6)
        ed = _ddpm_edges(n_node_type)
        flat = attr.reshape(-1)
        take = min(len(ed), len(flat))
        flat[:take] = ed[np.linspace(0, len(ed) - 1, take).astype(int)]
        flat[1::13] = 1.0
        flat[2::17] = -1.0
        attr = flat.reshape(attr.shape)
    else:
        attr.reshape(-1)[::5] = 0.0
    node[..., :-4] = attr
    return cfg, flags, adj, node, e_adj, e_node, n_adj_type, n_node_type


# ---- training-time forward fixture (trainer_node_adj.py:96-163 in 'test' mode: objective -> model -> loss) ----
TRAIN_VALID = [8, 5, 3, 6]


def train_case(name: str = "tiny", B: int = 4, seed: int = 7):
    """clean +-1 'bits' graphs with bbox channels in (-1,1), the N(0,1) draws of the objective generator in its draw order
    (sigma [B], adjacency noise, node noise) and the self-conditioning coin of the one preconditioned call."""
    cfg = CONFIGS[name]()
    n = cfg.max_node_num
    flags = W.synth_flags(B, n, TRAIN_VALID)
    clean_adj = W.mask_adj(np.sign(W.normal(seed, f"trn/{name}/adj", (B, cfg.c_adj, n, n))).astype(np.float32), flags)
    node = np.sign(W.normal(seed, f"trn/{name}/node", (B, n, cfg.c_node))).astype(np.float32)
    node[..., -4:] = (2.0 * W.uniform01(seed, f"trn/{name}/bbox", B * n * 4) - 1.0).astype(np.float32).reshape(B, n, 4) * 0.8
    clean_node = W.mask_node(node, flags)
    rnd = W.normal(seed, f"trn/{name}/rnd", (B,))
    eps_adj = W.normal(seed, f"trn/{name}/eps_adj", (B, cfg.c_adj, n, n))
    eps_node = W.normal(seed, f"trn/{name}/eps_node", (B, n, cfg.c_node))
    coin = float(W.coins(seed, f"trn/{name}", 1)[0])
    return cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin


IOU_TYPES = ("iou", "giou", "giou_squared", "diou", "ciou")   # trainer_node_adj.py:138-153


def iou_case(seed: int = 23):
    """Inputs of the bounding-box loss fixtures (tests/golden/iou_losses.npz): the tiny config's shapes (N = 8, C_adj = 6, C_node = 12),
    B = 4 with ragged flags; targets = clean +-1 graphs with bbox channels in (-0.8, 0.8); predictions = targets + noise, with the
    bbox channels of the prediction spread wide enough that some corners leave [0, 1] (the clamp's zero-gradient branch), some boxes do
    not overlap their target (the masked intersection) and some enclose / are enclosed by it; per-sample loss weights and sigmas."""
    cfg = CONFIGS["tiny"]()
    n, B = cfg.max_node_num, 4
    flags = W.synth_flags(B, n, TRAIN_VALID)
    tgt_adj = W.mask_adj(np.sign(W.normal(seed, "iou/adj", (B, cfg.c_adj, n, n))).astype(np.float32), flags)
    node = np.sign(W.normal(seed, "iou/node", (B, n, cfg.c_node))).astype(np.float32)
    node[..., -4:] = (2.0 * W.uniform01(seed, "iou/bbox", B * n * 4) - 1.0).astype(np.float32).reshape(B, n, 4) * 0.8
    tgt_node = W.mask_node(node, flags)
    pred_adj = W.mask_adj(tgt_adj + 0.4 * W.normal(seed, "iou/eps_adj", (B, cfg.c_adj, n, n)), flags)
    eps = W.normal(seed, "iou/eps_node", (B, n, cfg.c_node))
    pred = tgt_node + 0.4 * eps
    pred[..., -4:] = tgt_node[..., -4:] + np.float32(0.7) * eps[..., -4:]
    pred[..., -2:] = np.maximum(pred[..., -2:], np.float32(-0.9))        # width / height stay positive: (w + 1)/2 >= 0.05
    pred[..., -4:-2] = np.clip(pred[..., -4:-2], np.float32(-0.85), np.float32(0.85))   # centres inside the image: no box collapses to zero
    # width or height under the clamp (complete_box_iou_loss divides w by h: a collapsed box is NaN in torchvision itself)
    pred_node = W.mask_node(pred.astype(np.float32), flags)
    weights = (0.5 + 2.0 * W.uniform01(seed, "iou/w", B)).astype(np.float32)
    sigmas = np.exp(1.2 * W.normal(seed, "iou/sig", (B,)) - 1.2).astype(np.float32)
    return cfg, flags, pred_adj, pred_node, tgt_adj, tgt_node, weights, sigmas


def block_case(cfg, prefix: str, B: int, seed: int = 11):
    """inputs of one SwinTransformerBlock on its own: x [B, T, C] of the block's level, a mapped noise embedding [B, 512] of
    realistic scale, and the upstream gradient dY [B, T, C] (tools/gen_golden.py::gen_block_backward, tests)."""
    parts = prefix.split(".")
    lvl = int(parts[1]) if parts[0] == "down_layers" else cfg.num_layers - 1 - int(parts[1])
    res, C = cfg.max_node_num >> lvl, cfg.embed_dim << lvl
    x = W.normal(seed, f"blk/{prefix}/x", (B, res * res, C))
    emb = (0.3 * W.normal(seed, f"blk/{prefix}/emb", (B, 512))).astype(np.float32)
    dy = W.normal(seed, f"blk/{prefix}/dy", (B, res * res, C))
    return x, emb, dy
