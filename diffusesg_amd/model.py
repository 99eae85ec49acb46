"""Drop-in counterparts of the reference's `model/` callables, backed by libdsg.so.

  DiffuseSGHip      <->  model.diffusesg.diffusesg.DiffuseSG        (R/model/diffusesg/diffusesg.py:587-830)
  NodeAdjPrecondHip <->  model.precond.precond.NodeAdjPrecond        (R/model/precond/precond.py:60-114)

Same constructor meaning, same `forward` signatures and argument conventions (squeezed singleton
channel dims, `None` self-conditioning), same `state_dict()` key names (so the reference's
`load_model(strict=True)` works unchanged), same error behaviour for wrong shapes (AssertionError).
The arithmetic runs in hand-written gfx950 kernels; PyTorch only owns the device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as _lib
from .spec import ModelConfig, state_dict_spec


class _Holder(nn.Module):
    """Parameter container: gives the flat spec the reference's dotted module-tree names."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_buffer: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Holder())
        mod = getattr(mod, p)
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class DiffuseSGHip(nn.Module):
    """`DiffuseSG` on MI355X.  Constructor kwargs mirror R/utils/learning_utils.py:47-64."""

    def __init__(self, img_size=64, patch_size=1, in_chans=30, embed_dim=96, depths=(1, 1, 3, 1),
                 num_heads=(3, 6, 12, 24), window_size=8, mlp_ratio=4., out_chans_adj=6, out_chans_node=12,
                 self_condition=True, symmetric_noise=False, device="cuda", **unused):
        super().__init__()
        if symmetric_noise:
            raise NotImplementedError("symmetric_noise=True is the adjacency-only ablation; scene graphs use False "
                                      "(learning_utils.py:61)")
        assert in_chans == out_chans_adj + 2 * out_chans_node, "in_chans must be c_adj + 2*c_node (sg_utils.py:412-428)"
        self.config = ModelConfig(max_node_num=int(img_size), c_adj=int(out_chans_adj), c_node=int(out_chans_node),
                                  embed_dim=int(embed_dim), depths=tuple(int(d) for d in depths),
                                  num_heads=tuple(int(x) for x in num_heads)[:len(depths)], window_size=int(window_size),
                                  mlp_ratio=int(mlp_ratio), self_condition=bool(self_condition), patch_size=int(patch_size))
        self.self_condition = bool(self_condition)
        self.symmetric_noise = False
        self.out_chans_adj, self.out_chans_node = int(out_chans_adj), int(out_chans_node)
        self._dev = torch.device(device)
        # parameters under the reference's names; zeros until a state dict is loaded
        for t in state_dict_spec(self.config):
            if t.kind == "buffer":
                _attach(self, t.key, torch.from_numpy(t.buffer.copy()), True)
            else:
                _attach(self, t.key, torch.zeros(t.shape, dtype=torch.float32), False)
        self._handle: Optional[_lib.Handle] = None
        self._synced_version = None

    # -- weights ---------------------------------------------------------------------------------
    def _weights_version(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def _ensure_handle(self, finalize: bool = True):
        """The library handle with this module's current weights.  finalize=False (training iterations only): changed weights
        are uploaded but the sampling path's derived tensors (folded / packed weights, bias tables) are not rebuilt -- the
        training-form kernels read the raw weights; the next ordinary call rebuilds them."""
        if self._handle is None:
            finalize = True
        if self._handle is None:
            if not torch.cuda.is_available():
                raise _lib.DsgError("DiffuseSGHip needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
            # one process per GPU: the library allocates and launches on the CURRENT HIP device; a model placed on another one
            # would mix devices silently, so insist (torchrun ranks call torch.cuda.set_device(LOCAL_RANK) first, as bench.py does)
            idx = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
            if idx != torch.cuda.current_device():
                raise _lib.DsgError(f"model is on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: "
                                    f"call torch.cuda.set_device({idx}) before the first forward (one process per GPU)")
            self._handle = _lib.Handle(self.config)
        ver = self._weights_version()
        if ver != self._synced_version:
            sd = self.state_dict()
            torch.cuda.synchronize()
            for k in self._handle.weight_keys():
                t = sd[k]
                if t.dtype == torch.float32:
                    t = t.detach().contiguous()
                    self._handle.set_weight(k, t.data_ptr(), tuple(t.shape), t.is_cuda)
                else:  # relative_position_index (int64 constant)
                    t = t.detach().contiguous().cpu()
                    self._handle.set_weight(k, t.data_ptr(), tuple(t.shape), False)
            self._synced_version = ver
            self._finalized_ok = False
        if finalize and not getattr(self, "_finalized_ok", False):
            self._handle.finalize()
            self._finalized_ok = True
        return self._handle

    def load_numpy_state_dict(self, sd):
        own = self.state_dict()
        missing = set(own) - set(k[6:] if k.startswith("model.") else k for k in sd)
        if missing:
            raise KeyError(f"missing keys: {sorted(missing)[:5]} ...")
        self.load_state_dict({(k[6:] if k.startswith("model.") else k): torch.from_numpy(np.ascontiguousarray(v))
                              for k, v in sd.items()}, strict=True)
        return self

    # -- forward ---------------------------------------------------------------------------------
    def _canon(self, adj, node, node_flags, sc_adj, sc_node):
        cfg = self.config
        n, B = cfg.max_node_num, node_flags.shape[0]
        if node_flags.dim() != 2:
            raise NotImplementedError("node-only ablation ([B,N,N] node_flags) is out of scope (SURVEY §8)")
        dev = self._dev

        def prep(x, shape, what):
            if x is None:
                return None
            x = x.to(device=dev, dtype=torch.float32)
            assert x.numel() == int(np.prod(shape)), f"{what} has wrong size {tuple(x.shape)} for {shape}"
            return x.reshape(shape).contiguous()
        a = prep(adj, (B, cfg.c_adj, n, n), "adj")
        assert adj.shape[-1] == n and adj.shape[-2] == n, \
            f"Input image size ({adj.shape[-2]}*{adj.shape[-1]}) doesn't match model ({n}*{n})."
        x = prep(node, (B, n, cfg.c_node), "node")
        sa = prep(sc_adj, (B, cfg.c_adj, n, n), "self_cond_x")
        sx = prep(sc_node, (B, n, cfg.c_node), "self_cond_feat")
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
        return B, a, x, fl, sa, sx

    def _shape_out(self, oa, on):
        # the reference squeezes single-channel outputs (diffusesg.py:806-818)
        if self.config.c_adj == 1:
            oa = oa[:, 0]
        if self.config.c_node == 1:
            on = on[..., 0]
        return oa, on

    @torch.no_grad()
    def forward(self, adj, node, node_flags, noise_labels, self_cond_x=None, self_cond_feat=None):
        h = self._ensure_handle()
        B, a, x, fl, sa, sx = self._canon(adj, node, node_flags, self_cond_x, self_cond_feat)
        nl = noise_labels.to(device=self._dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        oa, on = torch.empty_like(a), torch.empty_like(x)
        st = torch.cuda.current_stream(self._dev).cuda_stream
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        h.check(h.L.dsg_denoise(h.raw, B, p(a), p(x), p(fl), p(nl), p(sa), p(sx), p(oa), p(on), C.c_void_p(st)), "dsg_denoise")
        return self._shape_out(oa, on)

    def debug_forward(self, adj, node, node_flags, noise_labels, self_cond_x=None, self_cond_feat=None, taps=None):
        """forward + named intermediate activations (token-major [B,T,C]); test/debug only."""
        h = self._ensure_handle()
        bufs = {}
        B = node_flags.shape[0]
        for name, numel in (taps or {}).items():
            bufs[name] = torch.zeros(B * numel, dtype=torch.float32, device=self._dev)
            h.check(h.L.dsg_debug_tap(h.raw, name.encode(), C.c_void_p(bufs[name].data_ptr()), bufs[name].numel()), "tap")
        try:
            out = self.forward(adj, node, node_flags, noise_labels, self_cond_x, self_cond_feat)
            torch.cuda.synchronize()
        finally:
            h.L.dsg_debug_clear_taps(h.raw)
        return out, bufs


class NodeAdjPrecondHip(nn.Module):
    """`NodeAdjPrecond(precond='edm', ...)` on MI355X (R/model/precond/precond.py:60-114).

    Like the reference, every call in a self-conditioning model draws one `np.random.rand()` from NumPy's
    global generator and, when it is < 0.5, runs the extra self-conditioning pass -- also in eval mode
    (precond.py:90)."""

    def __init__(self, precond, model: DiffuseSGHip, self_condition, symmetric_noise=False):
        super().__init__()
        assert precond in ["vp", "ve", "edm"]
        if precond != "edm":
            raise NotImplementedError("only precond='edm' is built (both reference YAMLs: mcmc.precond = edm)")
        if symmetric_noise:
            raise NotImplementedError("symmetric_noise=True is not used for scene graphs")
        self.precond = precond
        self.model = model
        self.self_condition = self_condition
        self.symmetric_noise = symmetric_noise

    @torch.no_grad()
    def forward(self, adjs, nodes=None, node_flags=None, sigmas=None, self_cond_adjs=None, self_cond_nodes=None,
                *args, **model_kwargs):
        m = self.model
        h = m._ensure_handle()
        B, a, x, fl, sa, sx = m._canon(adjs, nodes, node_flags, self_cond_adjs, self_cond_nodes)
        sg = torch.as_tensor(sigmas).to(device=m._dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        coin = bool(self.self_condition and np.random.rand() < 0.5)
        oa, on = torch.empty_like(a), torch.empty_like(x)
        st = torch.cuda.current_stream(m._dev).cuda_stream
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        h.check(h.L.dsg_precond(h.raw, B, p(a), p(x), p(fl), p(sg), p(sa), p(sx), int(coin), p(oa), p(on), C.c_void_p(st)),
                "dsg_precond")
        return m._shape_out(oa, on)

    @staticmethod
    def round_sigma(sigma):
        return torch.as_tensor(sigma)


def build_network(cfg: ModelConfig, state_dict=None, device="cuda") -> NodeAdjPrecondHip:
    """DiffuseSGHip wrapped in NodeAdjPrecondHip, the way `get_network` does (learning_utils.py:47-79)."""
    net = DiffuseSGHip(img_size=cfg.max_node_num, patch_size=cfg.patch_size, in_chans=cfg.c_adj + 2 * cfg.c_node,
                       embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads, window_size=cfg.window_size,
                       mlp_ratio=cfg.mlp_ratio, out_chans_adj=cfg.c_adj, out_chans_node=cfg.c_node,
                       self_condition=cfg.self_condition, symmetric_noise=False, device=device)
    if state_dict is not None:
        net.load_numpy_state_dict(state_dict)
    return NodeAdjPrecondHip("edm", net, cfg.self_condition, symmetric_noise=False).eval()
