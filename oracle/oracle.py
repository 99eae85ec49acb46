"""ctypes binding of the ORACLE (oracle/libdsgref.so).  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by diffusesg_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdsgref.so")


class _SamplerCfg(C.Structure):
    _fields_ = [("num_steps", C.c_int32), ("heun", C.c_int32),
                ("S_churn", C.c_float), ("S_min", C.c_float), ("S_max", C.c_float), ("S_noise", C.c_float),
                ("sigma_min", C.c_double), ("sigma_max", C.c_double), ("rho", C.c_double),
                ("max_steps", C.c_int32)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dsg_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "libdsgref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("DSGREF_LIB", _LIB_PATH)   # tests/test_oracle_asan.py points this at the AddressSanitizer build
        if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(path)
        L.dsgref_create.restype = C.c_void_p
        L.dsgref_create.argtypes = [C.c_void_p]
        L.dsgref_destroy.argtypes = [C.c_void_p]
        L.dsgref_set_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.dsgref_tap.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.dsgref_clear_taps.argtypes = [C.c_void_p]
        L.dsgref_nfe.restype = C.c_long
        L.dsgref_nfe.argtypes = [C.c_void_p]
        L.dsgref_forward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.dsgref_precond.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 2
        L.dsgref_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10
        L.dsgref_sigma_steps.argtypes = [C.c_void_p, C.c_void_p]
        L.dsgref_train_inputs.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 10
        L.dsgref_rainbow_loss.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float] * 3 + [C.c_int] + [C.c_void_p] * 2
        L.dsgref_rainbow_loss_backward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float] * 3 + [C.c_int] + [C.c_void_p] * 5
        L.dsgref_noise_embed.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        L.dsgref_decode_bits.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 3
        L.dsgref_set_threads.argtypes = [C.c_int]
        L.dsgref_set_threads.restype = C.c_int
        L.dsgref_decode.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 3
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """CPU oracle for one ModelConfig + weight set (numpy state dict keyed like DiffuseSG.state_dict())."""

    def __init__(self, cfg, state_dict: Dict[str, np.ndarray]):
        self.cfg = cfg
        ints = np.zeros(24, dtype=np.int32)
        ints[0:5] = [cfg.max_node_num, cfg.c_adj, cfg.c_node, cfg.embed_dim, cfg.num_layers]
        ints[5:5 + cfg.num_layers] = cfg.depths
        ints[13:13 + cfg.num_layers] = cfg.num_heads[:cfg.num_layers]
        ints[21:24] = [cfg.window_size, cfg.mlp_ratio, int(cfg.self_condition)]
        self._h = lib().dsgref_create(_p(ints))
        self._keep = []
        for k, v in state_dict.items():
            if k.startswith("model."):
                k = k[len("model."):]
            if "relative_position_index" in k or "attn_mask" in k:
                continue  # constants, re-derived inside the oracle
            a = _f32(v)
            lib().dsgref_set_weight(self._h, k.encode(), _p(a), a.size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().dsgref_destroy(self._h)
            self._h = None

    @property
    def nfe(self) -> int:
        return int(lib().dsgref_nfe(self._h))

    def _shapes(self, B):
        c = self.cfg
        return (B, c.c_adj, c.max_node_num, c.max_node_num), (B, c.max_node_num, c.c_node)

    def forward(self, adj, node, flags, c_noise, sc_adj=None, sc_node=None, taps: Optional[dict] = None):
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        adj, node, sc_adj, sc_node = (_f32(x) if x is None else _f32(x).reshape(s) for x, s in
                                      ((adj, sa), (node, sn), (sc_adj, sa), (sc_node, sn)))
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        cn = _f32(c_noise)
        oa, on = np.empty(sa, np.float32), np.empty(sn, np.float32)
        bufs = {}
        if taps is not None:
            for name, numel in taps.items():
                bufs[name] = np.zeros(B * numel, np.float32)
                lib().dsgref_tap(self._h, name.encode(), _p(bufs[name]), bufs[name].size)
        lib().dsgref_forward(self._h, B, _p(adj), _p(node), _p(fl), _p(cn), _p(sc_adj), _p(sc_node), _p(oa), _p(on))
        if taps is not None:
            lib().dsgref_clear_taps(self._h)
            return oa, on, bufs
        return oa, on

    def precond(self, adj, node, flags, sigmas, sc_adj=None, sc_node=None, coin=False):
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        adj, node, sc_adj, sc_node = (_f32(x) if x is None else _f32(x).reshape(s) for x, s in
                                      ((adj, sa), (node, sn), (sc_adj, sa), (sc_node, sn)))
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        sg = _f32(sigmas)
        oa, on = np.empty(sa, np.float32), np.empty(sn, np.float32)
        lib().dsgref_precond(self._h, B, _p(adj), _p(node), _p(fl), _p(sg), _p(sc_adj), _p(sc_node), int(bool(coin)),
                             _p(oa), _p(on))
        return oa, on

    @staticmethod
    def sampler_cfg(num_steps, solver="heun", S_churn=40.0, S_min=0.05, S_max=50.0, S_noise=1.003,
                    sigma_min=0.002, sigma_max=80.0, rho=7.0, max_steps=0):
        return _SamplerCfg(num_steps, 1 if solver == "heun" else 0, S_churn, S_min, S_max, S_noise,
                           sigma_min, sigma_max, rho, max_steps)

    @staticmethod
    def sigma_steps(num_steps, **kw):
        c = Oracle.sampler_cfg(num_steps, **kw)
        out = np.empty(num_steps, np.float64)
        lib().dsgref_sigma_steps(C.byref(c), _p(out))
        return out

    def sample(self, flags, init_adj, init_node, noise_adj=None, noise_node=None, coins=None,
               gt_adj=None, gt_node=None, **cfg_kw):
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        c = self.sampler_cfg(**cfg_kw)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        init_adj, init_node = _f32(init_adj).reshape(sa), _f32(init_node).reshape(sn)
        noise_adj, noise_node, gt_adj, gt_node = map(_f32, (noise_adj, noise_node, gt_adj, gt_node))
        co = None if coins is None else np.ascontiguousarray(coins, dtype=np.uint8)
        oa, on = np.empty(sa, np.float32), np.empty(sn, np.float32)
        lib().dsgref_sample(self._h, C.byref(c), B, _p(fl), _p(init_adj), _p(init_node), _p(noise_adj), _p(noise_node),
                            _p(co), _p(gt_adj), _p(gt_node), _p(oa), _p(on))
        return oa, on

    def decode_bits(self, adj, node, flags, n_adj_type, n_node_type, bbox=True):
        """'bits' samples -> (q_adj [B,N,N] int32, q_node [B,N] int32, bbox [B,N,4] | None); sampler_node_adj.py:222-285"""
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        c = self.cfg
        adj, node = _f32(adj).reshape(sa), _f32(node).reshape(sn)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        qa = np.empty((B, c.max_node_num, c.max_node_num), np.int32)
        qn = np.empty((B, c.max_node_num), np.int32)
        bb = np.empty((B, c.max_node_num, 4), np.float32) if bbox else None
        lib().dsgref_decode_bits(self._h, B, _p(adj), _p(node), _p(fl), int(n_adj_type), int(n_node_type),
                                 c.c_node - 4 if bbox else c.c_node, _p(qa), _p(qn), _p(bb))
        return qa, qn, bb

    @staticmethod
    def set_threads(n: int) -> int:
        """OpenMP threads of the oracle's loops from now on (returns the number in effect)"""
        return int(lib().dsgref_set_threads(int(n)))

    def decode(self, adj, node, flags, edge_encoding, node_encoding, n_adj_type, n_node_type, bbox=True):
        """samples in any of the reference's encodings ('bits' | 'one_hot' | 'ddpm') -> (q_adj, q_node, bbox | None);
        sampler_node_adj.py:222-285 with attribute_code.py:13 (attribute_converter(..., out_encoding='int'))"""
        enc = {"bits": 0, "one_hot": 1, "ddpm": 2}
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        c = self.cfg
        adj, node = _f32(adj).reshape(sa), _f32(node).reshape(sn)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        qa = np.empty((B, c.max_node_num, c.max_node_num), np.int32)
        qn = np.empty((B, c.max_node_num), np.int32)
        bb = np.empty((B, c.max_node_num, 4), np.float32) if bbox else None
        rc = lib().dsgref_decode(self._h, B, _p(adj), _p(node), _p(fl), enc[edge_encoding], enc[node_encoding], int(n_adj_type),
                                 int(n_node_type), c.c_node - 4 if bbox else c.c_node, _p(qa), _p(qn), _p(bb))
        assert rc == 0
        return qa, qn, bb

    def train_inputs(self, clean_adj, clean_node, flags, rnd, eps_adj, eps_node):
        """objectives/edm.py:239-281 with replayed draws -> (sigmas [B], weights [B], noisy_adj, noisy_node)"""
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        ca, cn, ea, en = (_f32(x).reshape(s) for x, s in ((clean_adj, sa), (clean_node, sn), (eps_adj, sa), (eps_node, sn)))
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        rn = _f32(rnd)
        sig, wts = np.empty(B, np.float32), np.empty(B, np.float32)
        na, nn = np.empty(sa, np.float32), np.empty(sn, np.float32)
        lib().dsgref_train_inputs(self._h, B, _p(ca), _p(cn), _p(fl), _p(rn), _p(ea), _p(en), _p(sig), _p(wts), _p(na), _p(nn))
        return sig, wts, na, nn

    def noise_embed(self, c_noise):
        """PositionalEmbedding + map_layer0/1 (diffusesg.py:507-513, :768-771) -> (pe [rows,E], emb [rows,512])"""
        c = _f32(c_noise).reshape(-1)
        pe, emb = np.empty((c.size, self.cfg.embed_dim), np.float32), np.empty((c.size, 512), np.float32)
        lib().dsgref_noise_embed(self._h, c.size, _p(c), _p(pe), _p(emb))
        return pe, emb

    IOU_TYPES = {"iou": 0, "giou": 1, "giou_squared": 2, "diou": 3, "ciou": 4}   # trainer_node_adj.py:138-153

    def rainbow_loss_backward(self, pred_adj, pred_node, tgt_adj, tgt_node, flags, loss_weight=None, edge_w=1.0, node_w=1.0, iou_w=0.0,
                              sigmas=None, iou_type="iou"):
        """d(loss_adj.mean() + loss_node.mean()) / d(preconditioned outputs) and, with sigmas, / d(raw network outputs)
        -> (grad_adj, grad_node, grad_F_adj | None, grad_F_node | None)"""
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        pa, pn, ta, tn = (_f32(x).reshape(s) for x, s in ((pred_adj, sa), (pred_node, sn), (tgt_adj, sa), (tgt_node, sn)))
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        w, sg = _f32(loss_weight), _f32(sigmas)
        ga, gn = np.empty(sa, np.float32), np.empty(sn, np.float32)
        fa, fn = (np.empty(sa, np.float32), np.empty(sn, np.float32)) if sg is not None else (None, None)
        lib().dsgref_rainbow_loss_backward(self._h, B, _p(pa), _p(pn), _p(ta), _p(tn), _p(fl), _p(w), edge_w, node_w, iou_w,
                                           self.IOU_TYPES[iou_type], _p(sg), _p(ga), _p(gn), _p(fa), _p(fn))
        return ga, gn, fa, fn

    def rainbow_loss(self, pred_adj, pred_node, tgt_adj, tgt_node, flags, loss_weight=None, edge_w=1.0, node_w=1.0, iou_w=0.0,
                     iou_type="iou"):
        """loss/rainbow_loss.py:37-101 (reduction='none') + the trainer's IoU term -> (loss_adj [B], loss_node [B])"""
        B = flags.shape[0]
        sa, sn = self._shapes(B)
        pa, pn, ta, tn = (_f32(x).reshape(s) for x, s in ((pred_adj, sa), (pred_node, sn), (tgt_adj, sa), (tgt_node, sn)))
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        w = _f32(loss_weight)
        la, ln = np.empty(B, np.float32), np.empty(B, np.float32)
        lib().dsgref_rainbow_loss(self._h, B, _p(pa), _p(pn), _p(ta), _p(tn), _p(fl), _p(w), edge_w, node_w, iou_w, self.IOU_TYPES[iou_type],
                                  _p(la), _p(ln))
        return la, ln
