"""Hand-off formats either side of the hot path: checkpoints in, decoded graphs + .npz out.

  * checkpoint layout  R/runner/trainer/trainer_utils.py:168-185 : {'model': state_dict, 'config': dict, 'epoch', 'train_loss',
    'test_loss', 'model_ema_beta_0.9000': state_dict, ...}; keys are 'model.<DiffuseSG key>' (Precond holds the net as .model),
    optionally with DDP's 'module.' prefix.  Loaded with weights_only=True (nothing from the file is executed).
  * weight selection   R/eval.py:15-40 (get_ema_weight_keywords), R/utils/sampling_utils.py:34-60 (load_model).
  * post-decode        R/runner/sampler/sampler_node_adj.py:194-311 -> dsg_decode_bits on the device.
  * sample archive     R/runner/sampler/sampler_node_adj.py:395-407 (final_samples_array_before_eval.npz keys), which
    R/helper/eval_sg_samples.py and the reference's CPU metrics consume as-is.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch


def load_checkpoint(path: str) -> Dict:
    """torch.load with the safe loader only (weights_only=True); raises if the file needs unpickling of code."""
    return torch.load(path, map_location="cpu", weights_only=True)


def ema_weight_keywords(ckp_data: Dict, use_ema=None) -> List[str]:
    """Which state dicts of a checkpoint to evaluate (eval.py:15-40): None -> ['model']; 'all' -> every 'model*' entry;
    a list of betas -> 'model_ema_beta_{beta:.4f}' (1.0 selects the online weights)."""
    all_kw = [k for k in ckp_data.keys() if k.startswith("model")]
    if use_ema is None:
        return ["model"]
    if use_ema == "all":
        return all_kw
    betas = list(use_ema)
    out = ["model"] if 1.0 in betas else []
    for b in betas:
        if b == 1.0:
            continue
        kw = "model_ema_beta_{:.4f}".format(b)
        assert kw in all_kw, "{} not found in the model data!".format(kw)
        out.append(kw)
    return out


def load_model(ckp_data: Dict, model: torch.nn.Module, weight_keyword: str = "model") -> torch.nn.Module:
    """Strict load with the reference's 'module.' prefix fix-up (sampling_utils.py:34-60)."""
    assert weight_keyword in ckp_data
    sd = ckp_data[weight_keyword]
    cur = set(model.state_dict().keys())
    fixed = {}
    for k, v in sd.items():
        if k in cur:
            fixed[k] = v
        elif k.startswith("module.") and k[len("module."):] in cur:
            fixed[k[len("module."):]] = v
        elif "module." + k in cur:
            fixed["module." + k] = v
        else:
            raise NotImplementedError(f"unexpected key {k}")
    model.load_state_dict(fixed, strict=True)
    return model


def decode_bits(net, adj: torch.Tensor, node: torch.Tensor, node_flags: torch.Tensor, n_adj_type: int, n_node_type: int,
                bbox: bool = True):
    """On-device post-decode of 'bits' samples -> (q_adj [B,N,N] int32, q_node [B,N] int32, bbox [B,N,4] | None).
    `net` is a DiffuseSGHip (or the precond wrapper)."""
    m = getattr(net, "model", net)
    h = m._ensure_handle()
    cfg = m.config
    B, n = node_flags.shape[0], cfg.max_node_num
    dev = m._dev
    a = adj.to(device=dev, dtype=torch.float32).reshape(B, cfg.c_adj, n, n).contiguous()
    x = node.to(device=dev, dtype=torch.float32).reshape(B, n, cfg.c_node).contiguous()
    fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
    node_bits = cfg.c_node - 4 if bbox else cfg.c_node
    qa = torch.empty((B, n, n), dtype=torch.int32, device=dev)
    qn = torch.empty((B, n), dtype=torch.int32, device=dev)
    bb = torch.empty((B, n, 4), dtype=torch.float32, device=dev) if bbox else None
    st = torch.cuda.current_stream(dev).cuda_stream
    h.check(h.L.dsg_decode_bits(h.raw, B, a.data_ptr(), x.data_ptr(), fl.data_ptr(), int(n_adj_type), int(n_node_type), node_bits,
                                qa.data_ptr(), qn.data_ptr(), None if bb is None else bb.data_ptr(), C.c_void_p(st)), "dsg_decode_bits")
    return qa, qn, bb


def pack_decoded(q_adj: torch.Tensor, q_node: torch.Tensor, bbox: Optional[torch.Tensor], node_flags: torch.Tensor) -> torch.Tensor:
    """Decoded graph as one int16/float-free byte row per sample for the single all-gather: [B, N*N + N + N] int16 + bbox.
    ~5 KB per VG graph instead of 101 KB of raw fp32."""
    B = q_adj.shape[0]
    parts = [q_adj.reshape(B, -1).to(torch.float32), q_node.reshape(B, -1).to(torch.float32),
             node_flags.reshape(B, -1).to(torch.float32)]
    if bbox is not None:
        parts.append(bbox.reshape(B, -1))
    return torch.cat(parts, dim=1).contiguous()


def unpack_decoded(packed: torch.Tensor, n: int, with_bbox: bool):
    B = packed.shape[0]
    o = 0
    q_adj = packed[:, o:o + n * n].reshape(B, n, n).to(torch.int32); o += n * n
    q_node = packed[:, o:o + n].to(torch.int32); o += n
    flags = packed[:, o:o + n] > 0.5; o += n
    bbox = packed[:, o:o + 4 * n].reshape(B, n, 4) if with_bbox else None
    return q_adj, q_node, flags, bbox


def save_samples_npz(path: str, *, samples_node_flags, samples_a, samples_x, raw_a, raw_x, samples_x_bbox=None,
                     gt_node_flags=None, gt_a=None, gt_x=None, gt_x_bbox=None, gt_image_ids=None):
    """Write `final_samples_array_before_eval.npz` with the reference's keys (sampler_node_adj.py:395-407)."""
    def npy(t):
        if t is None:
            return None
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    np.savez_compressed(path,
                        samples_node_flags=npy(samples_node_flags).astype(bool),
                        samples_a=npy(samples_a).astype(np.float32), samples_x=npy(samples_x).astype(np.float32),
                        raw_a=npy(raw_a), raw_x=npy(raw_x),
                        gt_node_flags=None if gt_node_flags is None else npy(gt_node_flags).astype(bool),
                        gt_a=npy(gt_a), gt_x=npy(gt_x), samples_x_bbox=npy(samples_x_bbox), gt_x_bbox=npy(gt_x_bbox),
                        gt_image_ids=npy(gt_image_ids))
