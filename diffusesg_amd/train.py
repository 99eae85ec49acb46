"""Training-time forward path (first half): what a test-loss / training step computes BEFORE backward.

  NodeAdjEDMObjectiveGeneratorHip <-> runner.objectives.edm.NodeAdjEDMObjectiveGenerator   (R/runner/objectives/edm.py:215-281)
  NodeAdjRainbowLossHip           <-> loss.rainbow_loss.NodeAdjRainbowLoss                  (R/loss/rainbow_loss.py:6-101)
  eval_loss_step                  <-> node_adj_move_forward_one_epoch(mode='test') body     (R/runner/trainer/trainer_node_adj.py:96-167)

Same constructor kwargs, call signatures and return conventions as the reference classes; the arithmetic runs in libdsg.so
(`dsg_train_inputs`, `dsg_rainbow_loss`, and the preconditioned network through `NodeAdjPrecondHip`).  The backward:
`NodeAdjRainbowLossHip.backward` -> `dsg_rainbow_loss_backward` (loss -> preconditioned outputs -> raw network outputs, every
`iou_loss_type` of the trainer) and the whole training iteration --
`train_step_grads` (network in training form, loss, backward to every parameter: `dsg_train_step_grads`), `AdamHip` (clip + Adam:
`dsg_adam_step`), `ExponentialLRHip` (the scheduler `get_optimizer` returns beside it), `EMAHip` (`dsg_ema_update`; parity unpinned),
`train_one_iteration`; gradients all-reduce through
`diffusesg_amd.dist.all_reduce_mean`.  The products of the training form run on the matrix pipe since round 3 (csrc/train_kernels.hip: y = x W^T and dx = dy W on the
sampling path's gemm4_f32_kernel, weight gradients on gemm_tn_f32_kernel, attention forward / backward on t_attn_mfma_kernel -- all
v_mfma_f32_32x32x2_f32); LayerNorm / modulate / optimiser are HBM-bound row and multi-tensor kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib

P_MEAN, P_STD, SIGMA_DATA = -1.2, 1.2, 0.5   # get_edm_params(), objectives/edm.py:60-63


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class NodeAdjEDMObjectiveGeneratorHip(object):
    """`get_training_objective_generator` builds it with precond = sigma_dist = 'edm', symmetric_noise=False
    (learning_utils.py:20-30)."""

    def __init__(self, precond, sigma_dist, *, other_params=None, dev="cuda", objective="edm", symmetric_noise=False):
        assert objective in ["diffusion", "score", "edm"]
        assert precond in ["vp", "ve", "edm"] and sigma_dist in ["vp", "ve", "edm"]
        if precond != "edm" or sigma_dist != "edm":
            raise NotImplementedError("only precond = sigma_dist = 'edm' (both reference YAMLs)")
        if symmetric_noise:
            raise NotImplementedError("symmetric_noise=True is not used for scene graphs (learning_utils.py:29)")
        self.precond, self.sigma_dist, self.other_params = precond, sigma_dist, other_params
        self.objective, self.dev, self.symmetric_noise = objective, torch.device(dev), False
        self.seed = 1234
        self._calls = 0

    @torch.no_grad()
    def get_input_output(self, clean_adjs, clean_x=None, node_flags=None, *args, rnd_sigma=None, noise=None, seed=None, **kwargs):
        """-> (net_input_a, net_input_x, net_cond, net_target_a, net_target_x, (c_skip, c_out, c_in, c_noise, sigmas, weights)),
        edm.py:258-281.  Extra keyword-only knobs: `rnd_sigma` [B] and `noise=(eps_adj, eps_node)` replay recorded N(0,1)
        draws (parity tests); otherwise the library's Philox streams of `seed` (default: self.seed + call count)."""
        if node_flags.dim() != 2:
            raise NotImplementedError("node-only ablation ([B,N,N] node_flags) is out of scope")
        L = _lib.load()
        dev = self.dev
        B, n = node_flags.shape
        adj = clean_adjs.to(device=dev, dtype=torch.float32)
        adj4 = (adj.unsqueeze(1) if adj.dim() == 3 else adj).contiguous()
        x = clean_x.to(device=dev, dtype=torch.float32)
        x3 = (x.unsqueeze(-1) if x.dim() == 2 else x).contiguous()
        ca, cn = adj4.shape[1], x3.shape[2]
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
        rnd = f32(rnd_sigma)
        ea, en = (None, None) if noise is None else (f32(noise[0]).reshape(adj4.shape), f32(noise[1]).reshape(x3.shape))
        sig, wts = torch.empty(B, device=dev), torch.empty(B, device=dev)
        na, nx = torch.empty_like(adj4), torch.empty_like(x3)
        if seed is None:
            seed = self.seed + self._calls
        self._calls += 1
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = L.dsg_train_inputs(B, n, ca, cn, _p(adj4), _p(x3), _p(fl), _p(rnd), _p(ea), _p(en), C.c_uint64(int(seed)), _p(sig), _p(wts),
                                _p(na), _p(nx), C.c_void_p(st))
        if rc != 0:
            raise _lib.DsgError(f"dsg_train_inputs: status {rc}")
        # get_preconditioning_params, edm branch (objectives/edm.py:122-126): returned for the caller's bookkeeping only
        s2 = sig ** 2 + SIGMA_DATA ** 2
        c_skip, c_out, c_in, c_noise = SIGMA_DATA ** 2 / s2, sig * SIGMA_DATA / s2.sqrt(), 1 / s2.sqrt(), sig.log() / 4
        na = na[:, 0] if adj.dim() == 3 else na
        nx = nx[..., 0] if x.dim() == 2 else nx
        return na, nx, sig, clean_adjs, clean_x, (c_skip, c_out, c_in, c_noise, sig, wts)


class NodeAdjRainbowLossHip(torch.nn.Module):
    def __init__(self, edge_loss_weight, node_loss_weight, objective, flag_reweight=False):
        super().__init__()
        assert objective in ["score", "diffusion", "edm"], "Loss mode {:s} is not supported!".format(objective)
        if objective == "score":
            raise NotImplementedError
        self.edge_loss_weight, self.node_loss_weight = edge_loss_weight, node_loss_weight
        self.flag_reweight, self.objective = flag_reweight, objective

    @torch.no_grad()
    def forward(self, net_pred_a, net_pred_x, net_target_a, net_target_x, net_cond, adjs_perturbed=None, adjs_gt=None,
                x_perturbed=None, x_gt=None, node_flags=None, loss_weight=None, cond_val=None, flag_matching=False,
                reduction="mean", iou_loss_weight=0.0, iou_loss_type="iou"):
        """rainbow_loss.py:24-35.  reduction 'none'/None -> per-sample (loss_adj [B], loss_node [B]); 'mean' follows the
        reference's expression literally (:84-86: a [B] tensor, both terms scaled by edge_loss_weight).
        `iou_loss_weight` / `iou_loss_type` (not reference kwargs of this class: the trainer's own arguments) add the trainer's bbox
        term on the device (trainer_node_adj.py:130-159; 'iou', 'giou', 'giou_squared', 'diou', 'ciou')."""
        if flag_matching:
            raise ValueError("Graph matching is not supported for node-adj loss!")
        if node_flags.dim() != 2:
            raise NotImplementedError("node-only ablation ([B,N,N] node_flags) is out of scope")
        L = _lib.load()
        dev = net_pred_a.device
        B, n = node_flags.shape
        f32 = lambda t: t.to(device=dev, dtype=torch.float32)
        a4 = lambda t: (f32(t).unsqueeze(1) if t.dim() == 3 else f32(t)).contiguous()
        x3 = lambda t: (f32(t).unsqueeze(-1) if t.dim() == 2 else f32(t)).contiguous()
        pa, ta, px, tx = a4(net_pred_a), a4(net_target_a), x3(net_pred_x), x3(net_target_x)
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
        w = None if loss_weight is None else f32(loss_weight).reshape(-1).contiguous()
        la, ln = torch.empty(B, device=dev), torch.empty(B, device=dev)
        none = reduction is None or reduction == "none"
        if not none and reduction != "mean":
            raise NotImplementedError(reduction)
        # 'mean' (:84-86) sums over the whole batch, divides by the per-sample entry counts and applies edge_loss_weight to
        # both terms; it is recovered below from unit-weight per-sample sums
        ew, nw = (self.edge_loss_weight, self.node_loss_weight) if none else (1.0, 1.0)
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = L.dsg_rainbow_loss(B, n, pa.shape[1], px.shape[2], _p(pa), _p(px), _p(ta), _p(tx), _p(fl), _p(w), float(ew), float(nw),
                                float(iou_loss_weight if none else 0.0), _lib.iou_loss_type_code(iou_loss_type), _p(la), _p(ln),
                                C.c_void_p(st))
        if rc != 0:
            raise _lib.DsgError(f"dsg_rainbow_loss: status {rc}")
        if none:
            return la, ln
        cnt = node_flags.to(device=dev).sum(dim=-1).to(torch.float32)
        tot_a = (la * cnt ** 2 * pa.shape[1]).sum()          # undo the per-sample normalisation: plain masked weighted sums
        tot_x = (ln * cnt * px.shape[2]).sum()
        return tot_a / cnt ** 2 * self.edge_loss_weight, tot_x / cnt * self.edge_loss_weight


    @torch.no_grad()
    def backward(self, net_pred_a, net_pred_x, net_target_a, net_target_x, node_flags, loss_weight=None, sigmas=None,
                 iou_loss_weight=0.0, iou_loss_type="iou"):
        """First stage of `loss.backward()` of a training step (trainer_node_adj.py:163-170), loss = loss_adj.mean() + loss_node.mean()
        with reduction='none' terms: -> (dL/d net_pred_a, dL/d net_pred_x, dL/dF_a | None, dL/dF_x | None); the last two (with `sigmas`)
        are the gradients at the raw network outputs, c_out(sigma) * the first two (precond.py:101-104).  (`train_step_grads` chains
        this with the network's own backward.)"""
        if node_flags.dim() != 2:
            raise NotImplementedError("node-only ablation ([B,N,N] node_flags) is out of scope")
        L = _lib.load()
        dev = net_pred_a.device
        B, n = node_flags.shape
        f32 = lambda t: t.to(device=dev, dtype=torch.float32)
        a4 = lambda t: (f32(t).unsqueeze(1) if t.dim() == 3 else f32(t)).contiguous()
        x3 = lambda t: (f32(t).unsqueeze(-1) if t.dim() == 2 else f32(t)).contiguous()
        pa, ta, px, tx = a4(net_pred_a), a4(net_target_a), x3(net_pred_x), x3(net_target_x)
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
        w = None if loss_weight is None else f32(loss_weight).reshape(-1).contiguous()
        sg = None if sigmas is None else f32(sigmas).reshape(-1).contiguous()
        ga, gx = torch.empty_like(pa), torch.empty_like(px)
        fa, fx = (torch.empty_like(pa), torch.empty_like(px)) if sg is not None else (None, None)
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = L.dsg_rainbow_loss_backward(B, n, pa.shape[1], px.shape[2], _p(pa), _p(px), _p(ta), _p(tx), _p(fl), _p(w),
                                         float(self.edge_loss_weight), float(self.node_loss_weight), float(iou_loss_weight),
                                         _lib.iou_loss_type_code(iou_loss_type), _p(sg),
                                         _p(ga), _p(gx), _p(fa), _p(fx), C.c_void_p(st))
        if rc != 0:
            raise _lib.DsgError(f"dsg_rainbow_loss_backward: status {rc}")
        sq_a = (lambda t: None if t is None else (t[:, 0] if net_pred_a.dim() == 3 else t))
        sq_x = (lambda t: None if t is None else (t[..., 0] if net_pred_x.dim() == 2 else t))
        return sq_a(ga), sq_x(gx), sq_a(fa), sq_x(fx)


def eval_loss_step(model, train_obj_gen, loss_func, adjs_gt, nodes_gt, node_flags, mode="test", iou_loss_type="iou",
                   iou_loss_weight=0.0, **replay):
    """One iteration of node_adj_move_forward_one_epoch in 'test' mode (trainer_node_adj.py:96-167): objective -> model pass
    under no_grad -> per-sample losses -> loss = adj.mean() + node.mean().  Returns (loss, reg_loss_adj, reg_loss_node, sigmas)."""
    if mode == "train":
        if replay.get("optimizer") is None:
            raise ValueError("mode='train' needs optimizer=AdamHip(model) (and optionally ema_helper=[EMAHip(model, beta), ...])")
        optimizer, ema_helper = replay.pop("optimizer"), replay.pop("ema_helper", None)
        loss, ra, rn, sg, _ = train_one_iteration(model, train_obj_gen, loss_func, optimizer, ema_helper, adjs_gt, nodes_gt, node_flags,
                                                  iou_loss_weight=iou_loss_weight, iou_loss_type=iou_loss_type, **replay)
        return loss, ra, rn, sg
    if mode != "test":
        raise NotImplementedError(mode)
    _lib.iou_loss_type_code(iou_loss_type)   # raises NotImplementedError for an unknown type, like trainer_node_adj.py:153-154
    net_input_a, net_input_x, net_cond, net_target_a, net_target_x, (c_skip, c_out, c_in, c_noise, sigmas, weights) = \
        train_obj_gen.get_input_output(adjs_gt, nodes_gt, node_flags, **replay)
    with torch.no_grad():
        net_output_a, net_output_x = model(adjs=net_input_a, nodes=net_input_x, node_flags=node_flags, sigmas=sigmas)
    reg_loss_adj, reg_loss_node = loss_func(net_pred_a=net_output_a, net_pred_x=net_output_x, net_target_a=net_target_a,
                                            net_target_x=net_target_x, net_cond=net_cond, adjs_perturbed=net_input_a, adjs_gt=adjs_gt,
                                            x_perturbed=net_input_x, x_gt=nodes_gt, node_flags=node_flags, loss_weight=weights,
                                            reduction="none", iou_loss_weight=iou_loss_weight, iou_loss_type=iou_loss_type)
    loss = reg_loss_adj.mean() + reg_loss_node.mean()
    return loss, reg_loss_adj, reg_loss_node, sigmas


BLOCK_PARAM_NAMES = ("affine.weight", "affine.bias", "norm1.weight", "norm1.bias", "attn.relative_position_bias_table",
                     "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                     "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


@torch.no_grad()
def swin_block_train(net, block: str, x, emb, grad_out=None):
    """One SwinTransformerBlock of `net` (a DiffuseSGHip) in training form: forward `block(x, emb)` (diffusesg.py:232-277) and, with
    `grad_out`, its backward -> (x_out, grad_x, grad_emb, {parameter name: gradient}) -- the first piece of the network backward
    (`dsg_block_train`; correctness-first kernels pinned to the reference's autograd, not the MFMA kernels of the sampling path).
    x, grad_out: [B, T, C]; emb: [B, 512] (the mapped noise embedding); `block` e.g. "down_layers.0.blocks.1"."""
    h = net._ensure_handle()
    dev = net._dev
    f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
    x, emb, gy = f32(x), f32(emb), f32(grad_out)
    B = x.shape[0]
    sd = net.state_dict()
    x_out = torch.empty_like(x)
    gx, ge = (torch.empty_like(x), torch.empty_like(emb)) if gy is not None else (None, None)
    grads = {k: torch.zeros(tuple(sd[f"{block}.{k}"].shape), device=dev, dtype=torch.float32) for k in BLOCK_PARAM_NAMES} if gy is not None else {}
    names = (C.c_char_p * len(BLOCK_PARAM_NAMES))(*[k.encode() for k in BLOCK_PARAM_NAMES])
    ptrs = (C.c_void_p * len(BLOCK_PARAM_NAMES))(*[grads[k].data_ptr() if grads else None for k in BLOCK_PARAM_NAMES])
    st = torch.cuda.current_stream(dev).cuda_stream
    h.check(h.L.dsg_block_train(h._h, block.encode(), B, _p(x), _p(emb), _p(gy), _p(x_out), _p(gx), _p(ge), len(BLOCK_PARAM_NAMES) if grads else 0,
                                names, ptrs, C.c_void_p(st)), "dsg_block_train")
    return x_out, gx, ge, grads


class GradDict(dict):
    """{state-dict key: gradient}: every tensor is a view into ONE flat buffer (`.flat`; 64-float aligned segments), so that zeroing,
    the data-parallel all-reduce and the optimiser each touch one tensor instead of 233."""
    flat = None


def _flat_layout(m):
    """(keys, offsets, shapes, total) of the module's parameters in a flat fp32 buffer; cached on the module"""
    lay = getattr(m, "_grad_layout", None)
    if lay is None:
        keys, offs, shapes, off = [], [], [], 0
        for k, p_ in m.named_parameters():
            keys.append(k); offs.append(off); shapes.append(tuple(p_.shape))
            off += (p_.numel() + 63) // 64 * 64
        lay = m._grad_layout = (keys, offs, shapes, off)
    return lay


def _train_handle(m):
    """The handle for the training-form entries: the parameters are read IN PLACE (dsg_train_bind_params) when they live on the
    device -- no upload per iteration; only the very first call pays for dsg_finalize_weights (block plans)."""
    if m._handle is None:
        m._ensure_handle()
    h = m._handle
    params = [(k, p_) for k, p_ in m.named_parameters()]
    if all(p_.is_cuda and p_.dtype == torch.float32 and p_.is_contiguous() for _, p_ in params):
        sig = tuple(p_.data_ptr() for _, p_ in params)
        if getattr(m, "_bound_sig", None) != sig:
            names = (C.c_char_p * len(params))(*[k.encode() for k, _ in params])
            ptrs = (C.c_void_p * len(params))(*sig)
            h.check(h.L.dsg_train_bind_params(h.raw, len(params), names, ptrs), "dsg_train_bind_params")
            m._bound_sig = sig
        return h
    if getattr(m, "_bound_sig", None) is not None:
        h.check(h.L.dsg_train_bind_params(h.raw, 0, None, None), "dsg_train_bind_params")
        m._bound_sig = None
    return m._ensure_handle(finalize=False)   # host parameters: uploaded when they changed


@torch.no_grad()
def train_step_grads(model, loss_func, net_input_a, net_input_x, node_flags, sigmas, net_target_a, net_target_x, loss_weight,
                     iou_loss_weight=0.0, want_grads=True, iou_loss_type="iou"):
    """One training iteration up to and including `loss.backward()` (trainer_node_adj.py:96-170) on the device:
    `model(adjs=net_input_a, nodes=net_input_x, node_flags=..., sigmas=...)` with the network in training form (the self-conditioning
    coin is drawn from NumPy's global generator like precond.py:90; the detached self-conditioning pass runs in training form too --
    `dsg_train_self_cond` -- so an iteration never rebuilds the sampling path's packed weights), per-sample losses, and the gradient of
    `loss_adj.mean() + loss_node.mean()` for every parameter.
    -> (net_output_a, net_output_x, reg_loss_adj [B], reg_loss_node [B], GradDict {state-dict key: gradient})."""
    import numpy as np
    m = model.model
    h = _train_handle(m)
    B, a, x, fl, _, _ = m._canon(net_input_a, net_input_x, node_flags, None, None)
    _, ta, tx, _, _, _ = m._canon(net_target_a, net_target_x, node_flags, None, None)
    dev = m._dev
    sg = torch.as_tensor(sigmas).to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
    w = None if loss_weight is None else loss_weight.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    sa = sx = None
    if model.self_condition and np.random.rand() < 0.5:   # precond.py:90-98: D of a no-grad pass becomes the (detached) self-cond input
        sa, sx = torch.empty_like(a), torch.empty_like(x)
        h.check(h.L.dsg_train_self_cond(h.raw, B, _p(a), _p(x), _p(fl), _p(sg), _p(sa), _p(sx), st), "dsg_train_self_cond")
    grads = GradDict()
    names = ptrs = None
    n_keys = 0
    if want_grads:
        keys, offs, shapes, total = _flat_layout(m)
        grads.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for k, o, shp in zip(keys, offs, shapes):
            n = 1
            for d_ in shp:
                n *= d_
            grads[k] = grads.flat[o:o + n].view(shp)
        cache = getattr(m, "_grad_names", None)
        if cache is None:
            cache = m._grad_names = (C.c_char_p * len(keys))(*[k.encode() for k in keys])
        names, n_keys = cache, len(keys)
        base = grads.flat.data_ptr()
        ptrs = (C.c_void_p * len(keys))(*[base + 4 * o for o in offs])
    da, dx_ = torch.empty_like(a), torch.empty_like(x)
    la, ln = torch.empty(B, device=dev), torch.empty(B, device=dev)
    h.check(h.L.dsg_train_step_grads(h.raw, B, _p(a), _p(x), _p(fl), _p(sg), _p(sa), _p(sx), _p(ta), _p(tx), _p(w),
                                     float(loss_func.edge_loss_weight), float(loss_func.node_loss_weight), float(iou_loss_weight),
                                     _lib.iou_loss_type_code(iou_loss_type), _p(da), _p(dx_), _p(la), _p(ln), n_keys, names, ptrs, st),
            "dsg_train_step_grads")
    oa, on = m._shape_out(da, dx_)
    return oa, on, la, ln, grads


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class AdamHip(object):
    """`torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999), eps=1e-8, weight_decay)` (utils/learning_utils.py:137-140) with
    `nn.utils.clip_grad_norm_(model.parameters(), max_norm)` folded in front of the step (trainer_node_adj.py:170) -- `dsg_adam_step`.
    Operates on the DiffuseSGHip's parameters (moved to the model's device) and on the gradient dict `train_step_grads` returns."""

    def __init__(self, model, lr=2.0e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.net = model.model if hasattr(model, "model") else model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        dev = self.net._dev
        for p_ in self.net.parameters():
            p_.data = p_.data.to(dev)
        self.keys = [k for k, _ in self.net.named_parameters()]
        self.params = [p_.data for _, p_ in self.net.named_parameters()]
        self.exp_avg = [torch.zeros_like(p_) for p_ in self.params]
        self.exp_avg_sq = [torch.zeros_like(p_) for p_ in self.params]
        self.step_count = 0
        self.last_total_norm = None

    @torch.no_grad()
    def step(self, grads, max_grad_norm=10.0):
        """-> total gradient norm before clipping (what clip_grad_norm_ returns)"""
        L = _lib.load()
        self.step_count += 1
        g = [grads[k] for k in self.keys]
        numel = (C.c_int64 * len(g))(*[int(t.numel()) for t in g])
        out = C.c_float(0.0)
        st = C.c_void_p(torch.cuda.current_stream(self.net._dev).cuda_stream)
        rc = L.dsg_adam_step(len(g), _ptr_array(self.params), _ptr_array(g), _ptr_array(self.exp_avg), _ptr_array(self.exp_avg_sq), numel,
                             self.step_count, self.lr, float(self.betas[0]), float(self.betas[1]), self.eps, self.weight_decay,
                             float(max_grad_norm if max_grad_norm else 0.0), C.byref(out), st)
        if rc != 0:
            raise _lib.DsgError(f"dsg_adam_step: status {rc}")
        self.net._synced_version = None   # the library re-reads (and re-packs) the weights before the next forward
        self.last_total_norm = float(out.value)
        return self.last_total_norm

    def zero_grad(self, set_to_none=True):
        pass   # gradients are returned fresh by every train_step_grads call


class ExponentialLRHip(object):
    """`torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=config.train.lr_dacey)` as `get_optimizer` pairs it with the optimiser
    (utils/learning_utils.py:142) and the epoch loop steps it once per epoch (trainer_node_adj.py:233): after `k` calls of `step()` the
    learning rate is lr_init * gamma**k (torch's closed form; `get_last_lr()` as torch returns it)."""

    def __init__(self, optimizer, gamma):
        self.optimizer, self.gamma = optimizer, float(gamma)
        self.base_lr, self.last_epoch = float(optimizer.lr), 0

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.base_lr * self.gamma ** self.last_epoch

    def get_last_lr(self):
        return [self.optimizer.lr]

    def state_dict(self):
        return {"gamma": self.gamma, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.gamma, self.base_lr, self.last_epoch = float(sd["gamma"]), float(sd["base_lrs"][0]), int(sd["last_epoch"])
        self.optimizer.lr = self.base_lr * self.gamma ** self.last_epoch


class EMAHip(object):
    """`ema_pytorch.EMA(model, beta=coef, update_every=1, update_after_step=0, inv_gamma=1, power=1)` as `get_ema_helper` builds it
    (utils/learning_utils.py:145-165).  ema_pytorch is absent here (un-vendored, unpinned dependency): PARITY UNPINNED; restated from
    its published algorithm -- the first `update()` copies the online weights, afterwards
    decay = clamp(1 - (1 + epoch / inv_gamma)^-power, 0, beta) with epoch = step - update_after_step - 1 (0 for epoch <= 0) and
    ema <- ema + (online - ema)(1 - decay)."""

    def __init__(self, model, beta=0.9999, update_every=1, update_after_step=0, inv_gamma=1.0, power=1.0, min_value=0.0):
        self.net = model.model if hasattr(model, "model") else model
        self.beta, self.update_every, self.update_after_step = float(beta), int(update_every), int(update_after_step)
        self.inv_gamma, self.power, self.min_value = float(inv_gamma), float(power), float(min_value)
        self.step, self.initted = 0, False
        self.keys = [k for k, _ in self.net.named_parameters()]
        self.shadow = {k: p_.data.detach().clone().to(self.net._dev) for k, p_ in self.net.named_parameters()}

    def get_current_decay(self):
        epoch = max(self.step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        return min(max(1.0 - (1.0 + epoch / self.inv_gamma) ** (-self.power), self.min_value), self.beta)

    @torch.no_grad()
    def update(self):
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return
        online = dict(self.net.named_parameters())
        if step <= self.update_after_step or not self.initted:
            for k in self.keys:
                self.shadow[k].copy_(online[k].data.to(self.shadow[k].device))
            self.initted = True
            if step <= self.update_after_step:
                return
        L = _lib.load()
        ema = [self.shadow[k] for k in self.keys]
        cur = [online[k].data for k in self.keys]
        numel = (C.c_int64 * len(ema))(*[int(t.numel()) for t in ema])
        rc = L.dsg_ema_update(len(ema), _ptr_array(ema), _ptr_array(cur), numel, float(self.get_current_decay()),
                              C.c_void_p(torch.cuda.current_stream(self.net._dev).cuda_stream))
        if rc != 0:
            raise _lib.DsgError(f"dsg_ema_update: status {rc}")

    def state_dict(self):
        return {k: v.clone() for k, v in self.shadow.items()}

    @property
    def ema_model(self):
        """ema_pytorch's `.ema_model`: the averaged network the trainer evaluates and samples with (trainer_node_adj.py:238, :263) -- a
        `NodeAdjPrecondHip` around a `DiffuseSGHip` whose parameters ALIAS the shadow tensors (no copy; its own library handle picks the
        values up at its next forward, every time `update()` has run in between)."""
        from .model import build_network
        if getattr(self, "_ema_model", None) is None:
            self._ema_model = build_network(self.net.config, None, device=self.net._dev)
            own = dict(self._ema_model.model.named_parameters())
            for k in self.keys:
                own[k].data = self.shadow[k]
        self._ema_model.model._synced_version = None   # the shadow tensors are updated in place, behind torch's version counters
        return self._ema_model


def train_one_iteration(model, train_obj_gen, loss_func, optimizer, ema_helper, adjs_gt, nodes_gt, node_flags, iou_loss_weight=0.0,
                        max_grad_norm=10.0, iou_loss_type="iou", **replay):
    """One iteration of node_adj_move_forward_one_epoch in 'train' mode (trainer_node_adj.py:96-175): objective -> model pass ->
    per-sample losses -> loss.backward() -> clip_grad_norm_(10) -> optimizer.step() -> EMA updates.
    -> (loss, reg_loss_adj, reg_loss_node, sigmas, total_grad_norm)"""
    net_input_a, net_input_x, net_cond, net_target_a, net_target_x, (c_skip, c_out, c_in, c_noise, sigmas, weights) = \
        train_obj_gen.get_input_output(adjs_gt, nodes_gt, node_flags, **replay)
    optimizer.zero_grad(set_to_none=True)
    oa, on, reg_loss_adj, reg_loss_node, grads = train_step_grads(model, loss_func, net_input_a, net_input_x, node_flags, sigmas,
                                                                  net_target_a, net_target_x, weights, iou_loss_weight=iou_loss_weight,
                                                                  iou_loss_type=iou_loss_type)
    loss = reg_loss_adj.mean() + reg_loss_node.mean()
    from . import dist as _dist
    _dist.all_reduce_mean(grads)   # data parallel: the mean over ranks, what DDP's backward leaves (identity on one rank)
    total_norm = optimizer.step(grads, max_grad_norm=max_grad_norm)
    if ema_helper is not None:
        [ema.update() for ema in ema_helper]
    return loss, reg_loss_adj, reg_loss_node, sigmas, total_norm
