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


def _checkpoint_allow_list():
    """The only non-tensor objects a reference checkpoint holds besides plain containers: the NumPy scalar losses
    `train_loss` / `test_loss` (np.concatenate(...).mean(), trainer_utils.py:160-176).  A NumPy scalar pickles as
    `numpy.core.multiarray.scalar(dtype, bytes)` (NumPy 1.x name) or `numpy._core.multiarray.scalar` (NumPy 2.x); both
    names are mapped to this interpreter's reconstruction function.  These are data constructors, not code from the file."""
    import numpy as _np
    scalar = _np.core.multiarray.scalar if not hasattr(_np, "_core") else _np._core.multiarray.scalar
    allow = [scalar, _np.dtype, (scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar")]
    for name in ("float64", "float32", "float16", "int64", "int32", "bool_"):
        allow.append(type(_np.dtype(getattr(_np, name))))      # numpy.dtypes.Float64DType, ... (NumPy >= 1.25 pickles these)
    return allow


def load_checkpoint(path: str) -> Dict:
    """Read a checkpoint written by the reference (`get_ckpt_data`, trainer_utils.py:168-185: state dicts, the nested
    `config.to_dict()`, `epoch`, NumPy-scalar `train_loss` / `test_loss`, one state dict per EMA beta).
    Safe loader only: torch.load(weights_only=True) with an explicit allow-list for the NumPy scalars; nothing from
    the file is executed, and a file that needs anything else is refused."""
    with torch.serialization.safe_globals(_checkpoint_allow_list()):
        return torch.load(path, map_location="cpu", weights_only=True)


def get_ckpt_data(model: torch.nn.Module, ema_helper, epoch: int, train_loss, test_loss, config: Dict) -> Dict:
    """The dictionary the reference's trainer saves (`get_ckpt_data`, trainer_utils.py:168-185): 'model' = model.state_dict() of the
    preconditioned wrapper ('model.' prefixed keys), the nested config dict, the epoch, NumPy-scalar losses, and one
    'model_ema_beta_{beta:.4f}' state dict per EMA helper (`diffusesg_amd.train.EMAHip`: its shadow weights, buffers copied from the
    online model) -- what `load_checkpoint` / `load_model` / the reference's own eval.py read back."""
    import numpy as np
    online = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    to_save = {"model": online, "config": dict(config), "epoch": int(epoch), "train_loss": np.float64(train_loss),
               "test_loss": np.float64(test_loss)}
    prefix = "model." if any(k.startswith("model.") for k in online) else ""
    for ema in (ema_helper or []):
        sd = {k: v.clone() for k, v in online.items()}                      # buffers (relative_position_index, attn_mask) as in the online model
        for k, v in ema.shadow.items():
            sd[prefix + k] = v.detach().cpu().clone()
        to_save["model_ema_beta_{:.4f}".format(ema.beta)] = sd
    return to_save


def save_checkpoint(path: str, model: torch.nn.Module, ema_helper, epoch: int, train_loss, test_loss, config: Dict) -> str:
    """`torch.save(get_ckpt_data(...), path)` (trainer_utils.py:160-165, file name `{dataset}_{epoch:05d}.pth` is the caller's)."""
    torch.save(get_ckpt_data(model, ema_helper, epoch, train_loss, test_loss, config), path)
    return path


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


def decode(net, adj: torch.Tensor, node: torch.Tensor, node_flags: torch.Tensor, n_adj_type: int, n_node_type: int,
           edge_encoding: str = "bits", node_encoding: str = "bits", bbox: bool = True):
    """On-device post-decode for any of the reference's attribute encodings ('bits' | 'one_hot' | 'ddpm';
    R/runner/sampler/sampler_node_adj.py:222-285 -> attribute_converter, R/utils/attribute_code.py:13) -> (q_adj [B,N,N] int32,
    q_node [B,N] int32, bbox [B,N,4] | None).  'ddpm' adjacencies may come squeezed to [B,N,N] as the sampler returns them."""
    from .lib import ENCODINGS
    for e in (edge_encoding, node_encoding):
        if e not in ENCODINGS:
            raise ValueError("encoding should be 'int', 'ddpm', 'bits' or 'one_hot'")   # attribute_code.py:43
    from .lib import Handle
    if isinstance(net, Handle):   # the decode needs the handle only for N / C_adj / C_node: a bare handle (no weights) will do
        h, cfg, dev = net, net.cfg, torch.device("cuda", torch.cuda.current_device())
    else:
        m = getattr(net, "model", net)
        h, cfg, dev = m._ensure_handle(), m.config, m._dev
    B, n = node_flags.shape[0], cfg.max_node_num
    a = adj.to(device=dev, dtype=torch.float32).reshape(B, cfg.c_adj, n, n).contiguous()
    x = node.to(device=dev, dtype=torch.float32).reshape(B, n, cfg.c_node).contiguous()
    fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
    node_chans = cfg.c_node - 4 if bbox else cfg.c_node
    qa = torch.empty((B, n, n), dtype=torch.int32, device=dev)
    qn = torch.empty((B, n), dtype=torch.int32, device=dev)
    bb = torch.empty((B, n, 4), dtype=torch.float32, device=dev) if bbox else None
    st = torch.cuda.current_stream(dev).cuda_stream
    h.check(h.L.dsg_decode(h.raw, B, a.data_ptr(), x.data_ptr(), fl.data_ptr(), ENCODINGS[edge_encoding], ENCODINGS[node_encoding],
                           int(n_adj_type), int(n_node_type), node_chans, qa.data_ptr(), qn.data_ptr(),
                           None if bb is None else bb.data_ptr(), C.c_void_p(st)), "dsg_decode")
    return qa, qn, bb


def pack_decoded(q_adj: torch.Tensor, q_node: torch.Tensor, bbox: Optional[torch.Tensor], node_flags: torch.Tensor) -> torch.Tensor:
    """Decoded graphs as one int16 row per sample, the unit of the single all-gather of decoded results:
    [ q_adj N*N | q_node N | flags N | bbox 4N fp32 viewed as 8N int16 ]  -- VG: 2*(4096+64+64)+1024 = 9.3 KB, COCO 4.6 KB
    per graph instead of 101 KB / 21 KB of raw fp32.  Type ids fit int16 (at most 171 node / 51 edge types)."""
    B = q_adj.shape[0]
    parts = [q_adj.reshape(B, -1).to(torch.int16), q_node.reshape(B, -1).to(torch.int16),
             node_flags.reshape(B, -1).to(torch.int16)]
    if bbox is not None:
        parts.append(bbox.reshape(B, -1).to(torch.float32).contiguous().view(torch.int16))
    return torch.cat(parts, dim=1).contiguous()


def unpack_decoded(packed: torch.Tensor, n: int, with_bbox: bool):
    B = packed.shape[0]
    o = 0
    q_adj = packed[:, o:o + n * n].reshape(B, n, n).to(torch.int32); o += n * n
    q_node = packed[:, o:o + n].to(torch.int32); o += n
    flags = packed[:, o:o + n] != 0; o += n
    bbox = packed[:, o:o + 8 * n].contiguous().view(torch.float32).reshape(B, n, 4) if with_bbox else None
    return q_adj, q_node, flags, bbox


def save_samples_npz(path: str, *, samples_node_flags, samples_a, samples_x, raw_a, raw_x, gt_node_flags, gt_a, gt_x,
                     samples_x_bbox=None, gt_x_bbox=None, gt_image_ids=None):
    """Write `final_samples_array_before_eval.npz` with the reference's keys and dtypes (sampler_node_adj.py:395-407): flags bool,
    quantised graphs float32 (`bin2dec` of float tensors), raw samples / bbox float32, image ids int64.  The consumer
    (R/helper/eval_sg_samples.py:248-253) reads samples_x, samples_a, gt_x, gt_a, samples_x_bbox, gt_x_bbox and gt_node_flags with
    a plain `np.load` (no pickle), so every entry is a typed array: the ground-truth entries are mandatory (the reference always
    has them: they are the data loader's batch, decoded like the samples), and where the reference would store a Python `None`
    (bbox arrays of a run without bbox channels; `np.load` cannot read that object array back without pickle) a typed
    zero-width array [B, N, 0] is written instead.  Sample and ground-truth bbox arrays come together or not at all."""
    def npy(t, dtype):
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        return np.ascontiguousarray(a.astype(dtype))
    for name, v in (("gt_node_flags", gt_node_flags), ("gt_a", gt_a), ("gt_x", gt_x)):
        if v is None:
            raise ValueError(f"save_samples_npz: {name} is required (the evaluator reads it without pickle)")
    if (samples_x_bbox is None) != (gt_x_bbox is None):
        raise ValueError("save_samples_npz: samples_x_bbox and gt_x_bbox come together")
    flags = npy(samples_node_flags, bool)
    B, n = flags.shape[:2]
    empty_bbox = np.zeros((B, n, 0), np.float32)
    arrays = dict(samples_node_flags=flags, samples_a=npy(samples_a, np.float32), samples_x=npy(samples_x, np.float32),
                  raw_a=npy(raw_a, np.float32), raw_x=npy(raw_x, np.float32),
                  gt_node_flags=npy(gt_node_flags, bool), gt_a=npy(gt_a, np.float32), gt_x=npy(gt_x, np.float32),
                  samples_x_bbox=empty_bbox if samples_x_bbox is None else npy(samples_x_bbox, np.float32),
                  gt_x_bbox=empty_bbox if gt_x_bbox is None else npy(gt_x_bbox, np.float32),
                  gt_image_ids=np.zeros((0,), np.int64) if gt_image_ids is None else npy(gt_image_ids, np.int64))
    for k in ("samples_a", "gt_a"):
        if arrays[k].shape != (B, n, n):
            raise ValueError(f"save_samples_npz: {k} must be [B, N, N], got {arrays[k].shape}")
    for k in ("samples_x", "gt_x", "gt_node_flags"):
        if arrays[k].shape != (B, n):
            raise ValueError(f"save_samples_npz: {k} must be [B, N], got {arrays[k].shape}")
    np.savez_compressed(path, **arrays)
    return path
