#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own Python modules.

DEV-CONTAINER ONLY.  /root/reference does not exist on the GPU box; nothing at run time
(tests, smoke, bench) imports this script or the reference.  Only the small output
vectors it writes are committed; every *input* is regenerated on demand from the portable
generator in diffusesg_amd/weights.py (same seed / stream names as below).

How the reference is imported (disclosed in DESIGN.md §3):
  * `timm` is not installed in this image and the reference imports three helpers from it
    (`DropPath`, `to_2tuple`, `trunc_normal_`; R/model/diffusesg/diffusesg.py:5).  None of
    them touches the arithmetic of a forward pass here: weights are overwritten with the
    portable generator's, `DropPath` is never instantiated (drop_path_rate=0,
    R/utils/learning_utils.py:59) and `to_2tuple` only duplicates an int.  A three-symbol
    in-process module object is registered so the import statement resolves.
  * Randomness the reference draws internally is pinned, not changed: `torch.randn_like`
    (churn noise, R/runner/mcmc_sampler/edm.py:361-364) and `numpy.random.rand` (the
    self-conditioning coin, R/model/precond/precond.py:90) are temporarily replaced by
    functions that replay portable-generator streams.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DiffuseSG"
sys.path.insert(0, REPO)

from diffusesg_amd import spec as S          # noqa: E402
from diffusesg_amd import weights as W       # noqa: E402
from diffusesg_amd import synth as Y         # noqa: E402


def _import_reference():
    layers = types.ModuleType("timm.models.layers")
    layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.DropPath = torch.nn.Identity
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    timm.models, models.layers = models, layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    sys.path.insert(0, REF)
    from model.diffusesg.diffusesg import DiffuseSG
    from model.precond.precond import NodeAdjPrecond
    from runner.mcmc_sampler.edm import NodeAdjEDMSampler
    return DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler


CONFIGS = Y.CONFIGS


def build_ref_net(DiffuseSG, cfg, seed=0):
    net = DiffuseSG(img_size=cfg.max_node_num, in_chans=cfg.c_adj + 2 * cfg.c_node,
                    patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths),
                    num_heads=[3, 6, 12, 24], window_size=cfg.window_size, mlp_ratio=4.,
                    drop_rate=0., attn_drop_rate=0., drop_path_rate=0.0,
                    self_condition=cfg.self_condition, symmetric_noise=False,
                    out_chans_adj=cfg.c_adj, out_chans_node=cfg.c_node)
    ref_sd = net.state_dict()
    mine = W.synth_state_dict(cfg, seed)
    # the layout in diffusesg_amd/spec.py must equal the reference's state_dict exactly
    assert set(ref_sd.keys()) == set(mine.keys()), (set(ref_sd) ^ set(mine))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(mine[k].shape), (k, v.shape, mine[k].shape)
        if "relative_position_index" in k or "attn_mask" in k:
            assert np.array_equal(v.numpy(), mine[k]), k   # constant buffers re-derived in spec.py
    net.load_state_dict({k: torch.from_numpy(v) for k, v in mine.items()}, strict=True)
    net.eval()
    return net


case_inputs = Y.case_inputs


def squeeze_like_ref(cfg, adj, node):
    """The reference drops singleton channel dims (edm.py:281-288)."""
    if cfg.c_adj == 1:
        adj = adj[:, 0]
    if cfg.c_node == 1:
        node = node[..., 0]
    return adj, node


def t(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x))


VALID = Y.VALID


def gen_forward(DiffuseSG, out):
    """G1/G2: full DiffuseSG.forward, with per-module intermediates for the two smallest nets."""
    for name, mk in CONFIGS.items():
        cfg = mk()
        net = build_ref_net(DiffuseSG, cfg)
        B = 2
        flags, adj, node, sc_adj, sc_node = case_inputs(cfg, B, VALID[name], 1, f"fwd/{name}")
        c_noise = Y.FWD_C_NOISE[:B].copy()
        res = {}
        inter = {}
        hooks = []
        if name in ("tiny", "small", "vg", "coco"):
            def mk_hook(key):
                def h(_m, _i, o):
                    a = o.detach().numpy().copy()
                    if name in ("vg", "coco"):
                        # full-size nets: a fixed pseudo-random sample of Y.INTER_ROWS token rows per tap (every block of every
                        # (T, C, shift) class, merges, breakups) keeps the fixture small; rows of [B*T, C] in token order
                        if key == "read_out":
                            a = a.transpose(0, 2, 3, 1)
                        a = a.reshape(-1, a.shape[-1])
                        a = a[Y.inter_rows(a.shape[0])]
                    inter[key] = a
                return h
            hooks.append(net.patch_embed.register_forward_hook(mk_hook("patch_embed")))
            for i, l in enumerate(net.down_layers):
                for j, b in enumerate(l.blocks):
                    hooks.append(b.register_forward_hook(mk_hook(f"down{i}.block{j}")))
                hooks.append(l.register_forward_hook(mk_hook(f"down{i}")))
            for i, l in enumerate(net.up_layers):
                if l.upsample is not None:
                    hooks.append(l.upsample.register_forward_hook(mk_hook(f"up{i}.upsample")))
                for j, b in enumerate(l.blocks):
                    hooks.append(b.register_forward_hook(mk_hook(f"up{i}.block{j}")))
            hooks.append(net.read_out.register_forward_hook(mk_hook("read_out")))
        a_in, n_in = squeeze_like_ref(cfg, adj, node)
        sa_in, sn_in = squeeze_like_ref(cfg, sc_adj, sc_node)
        with torch.no_grad():
            if cfg.self_condition:
                oa, on = net(t(a_in.copy()), t(n_in.copy()), t(flags), t(c_noise), t(sa_in.copy()), t(sn_in.copy()))
                res["sc_adj_out"], res["sc_node_out"] = oa.numpy(), on.numpy()
                for k, v in inter.items():
                    res[("rows/" if name in ("vg", "coco") else "inter/") + k] = v
                inter.clear()
            for h in hooks:
                h.remove()
            oa, on = net(t(a_in.copy()), t(n_in.copy()), t(flags), t(c_noise), None, None)
            res["nosc_adj_out"], res["nosc_node_out"] = oa.numpy(), on.numpy()
        res["c_noise"] = c_noise
        res["valid"] = np.array(VALID[name])
        np.savez_compressed(os.path.join(out, f"fwd_{name}.npz"), **res)
        print(f"fwd_{name}: adj {res['nosc_adj_out'].shape} max|adj| "
              f"{np.abs(res['nosc_adj_out']).max():.3f} params {S.num_parameters(cfg):,}")


class _Replay:
    def __init__(self, items):
        self.items, self.i = list(items), 0

    def pop(self):
        v = self.items[self.i]
        self.i += 1
        return v


def gen_precond(DiffuseSG, NodeAdjPrecond, out):
    """G3: NodeAdjPrecond.forward at three sigmas, coin forced both ways."""
    import model.precond.precond as P
    for name in ("tiny", "small", "nosc"):
        cfg = CONFIGS[name]()
        net = build_ref_net(DiffuseSG, cfg)
        pre = NodeAdjPrecond("edm", net, cfg.self_condition, symmetric_noise=False).eval()
        res = {}
        for si, sigma in enumerate((80.0, 1.5, 0.002)):
            B = 2
            flags, adj, node, sc_adj, sc_node = case_inputs(cfg, B, VALID[name], 2, f"pre/{name}/{si}",
                                                            sigma_scale=float(np.sqrt(sigma ** 2 + 0.25)))
            a_in, n_in = squeeze_like_ref(cfg, adj, node)
            sa_in, sn_in = squeeze_like_ref(cfg, sc_adj, sc_node)
            sig = np.full((B,), sigma, dtype=np.float32)
            for coin in (0.9, 0.1):
                for with_sc in (False, True):
                    real = np.random.rand
                    P.np.random.rand = lambda: coin
                    try:
                        with torch.no_grad():
                            da, dn = pre(t(a_in.copy()), t(n_in.copy()), t(flags), t(sig),
                                         t(sa_in.copy()) if with_sc else None,
                                         t(sn_in.copy()) if with_sc else None)
                    finally:
                        P.np.random.rand = real
                    key = f"s{si}_coin{int(coin < 0.5)}_sc{int(with_sc)}"
                    res[key + "_adj"], res[key + "_node"] = da.numpy(), dn.numpy()
        np.savez_compressed(os.path.join(out, f"precond_{name}.npz"), **res)
        print(f"precond_{name}: {len(res)} arrays")


def run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg, *, T, B, valid, seed, tag,
                    solver="heun", S_churn=40, gt=None, weight_seed=0, coins=None):
    import model.precond.precond as P
    n = cfg.max_node_num
    net = build_ref_net(DiffuseSG, cfg, weight_seed)
    pre = NodeAdjPrecond("edm", net, cfg.self_condition, symmetric_noise=False).eval()
    smp = NodeAdjEDMSampler(num_steps=T, solver=solver, S_churn=S_churn,
                            clip_samples=True, clip_samples_min=-1.0, clip_samples_max=1.0,
                            clip_samples_scope="x_0", dev="cpu", objective="edm",
                            self_condition=cfg.self_condition, symmetric_noise=False)
    flags, init_adj, init_node, noise_adj, noise_node, coin_vals = Y.sampler_case(cfg, T, B, valid, seed, tag, solver)
    if coins is not None:   # an explicit 0/1 coin sequence: the reference fires the coin when its draw is < 0.5
        coin_vals = np.where(np.asarray(coins) != 0, 0.25, 0.75)
    noise = []
    for i in range(T):   # draw order inside one step: adjacency first, then nodes (edm.py:361-364)
        noise.append(noise_adj[i])
        noise.append(noise_node[i])
    rn = _Replay(noise)
    rc = _Replay(coin_vals)
    ia, inn = squeeze_like_ref(cfg, init_adj, init_node)

    def fake_randn_like(x, **kw):
        v = rn.pop()
        v = v.reshape(tuple(x.shape))
        return torch.from_numpy(v.copy()).to(x.dtype)

    real_rl, real_rand = torch.randn_like, P.np.random.rand
    torch.randn_like = fake_randn_like
    P.np.random.rand = lambda: float(rc.pop())
    try:
        kw = {}
        if gt is not None:
            ga, gn = squeeze_like_ref(cfg, gt[0], gt[1])
            kw = dict(sanity_check_gt_adjs=t(ga.copy()), sanity_check_gt_nodes=t(gn.copy()))
        adjs, nodes = smp.sample(pre, t(flags), init_adjs=t(ia.copy()), init_nodes=t(inn.copy()),
                                 flag_node_multi_channel=True, flag_adj_multi_channel=True,
                                 num_node_chan=cfg.c_node, num_edge_chan=cfg.c_adj, **kw)
    finally:
        torch.randn_like, P.np.random.rand = real_rl, real_rand
    sigma_steps = smp.sigma_steps.numpy().copy()
    return adjs.numpy(), nodes.numpy(), sigma_steps, rc.i


def gen_sampler(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, out):
    """G4/G5: short trajectories with replayed noise + coins, the known-answer run, sigma tables."""
    res = {}
    cfg = S.tiny_config()
    for (tag, T, solver, churn) in Y.SAMPLER_RUNS:
        a, nd, sig, used = run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg, T=T, B=4,
                                           valid=Y.SAMPLER_VALID, seed=3, tag=f"smp/{tag}", solver=solver,
                                           S_churn=churn)
        res[f"{tag}_adj"], res[f"{tag}_node"], res[f"{tag}_sigma_steps"] = a, nd, sig
        res[f"{tag}_coins_used"] = np.array(used)
        print(f"sampler {tag}: max|adj| {np.abs(a).max():.3f} coins used {used}")
    # nosc config (no self-conditioning => no coin, no extra forward)
    cfg2 = Y.nosc_config()
    a, nd, sig, used = run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg2, T=8, B=2,
                                       valid=[8, 3], seed=3, tag="smp/nosc_t8")
    res["nosc_t8_adj"], res["nosc_t8_node"] = a, nd
    # known-answer test (R/runner/mcmc_sampler/edm.py:372-377): GT replaces the denoiser
    B, n = 4, cfg.max_node_num
    gt_adj, gt_node = Y.gt_case(cfg, B, Y.SAMPLER_VALID)
    a, nd, _, _ = run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg, T=8, B=B,
                                  valid=Y.SAMPLER_VALID, seed=3, tag="smp/gt", gt=(gt_adj, gt_node))
    res["gt_adj"], res["gt_node"] = a, nd
    print("sanity-check run: max|out-GT| adj %.3e node %.3e" % (np.abs(a - gt_adj).max(), np.abs(nd - gt_node).max()))
    for T in (50, 100, 256, 1000):
        smp = NodeAdjEDMSampler(num_steps=T, clip_samples=True, clip_samples_min=-1.0, clip_samples_max=1.0,
                                clip_samples_scope="x_0", dev="cpu", self_condition=True, symmetric_noise=False)
        res[f"sigma_steps_{T}"] = smp.sigma_steps.numpy().copy()
    np.savez_compressed(os.path.join(out, "sampler.npz"), **res)


def gen_big_trajectories(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, out):
    """Short trajectories of the full-size networks through the REFERENCE's sampler (replayed noise and coins): the headline VG
    configuration (6 Heun+churn steps; 6 Euler steps without churn = BASELINE configs[2]'s variant) and the COCO-Stuff one.
    The GPU tests used to recompute these with the CPU oracle on every run (~100 s); as fixtures they are pinned to the
    reference itself.  Cases (inputs from Y.sampler_case): Y.BIG_TRAJ."""
    res = {}
    for tag, (name, T, solver, churn, valid, seed, stream, coins) in Y.BIG_TRAJ.items():
        cfg = CONFIGS[name]()
        a, nd, _sig, used = run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg, T=T, B=len(valid), valid=valid, seed=seed,
                                            tag=stream, solver=solver, S_churn=churn, coins=coins)
        res[f"{tag}_adj"], res[f"{tag}_node"], res[f"{tag}_coins_used"] = a, nd, np.array(used)
        print(f"big trajectory {tag}: max|adj| {np.abs(a).max():.3f} max|node| {np.abs(nd).max():.3f} coins used {used}", flush=True)
    np.savez_compressed(os.path.join(out, "traj_big.npz"), **res)


def ref_decode_bits(adj, node, flags, raw_num_adj_type, raw_num_node_type):
    """`_decode_node` / `_decode_adj`, 'bits' branch (sampler_node_adj.py:194-209, :222-233, :242-247, :265-275, :279-283), restated
    statement by statement around the reference's OWN bin2dec / mask_nodes / mask_adjs -> (q_adj, q_node, bbox) as numpy"""
    from utils.attribute_code import bin2dec
    from utils.graph_utils import mask_adjs, mask_nodes
    node_flags = t(flags)
    final_samples_adjs, final_samples_nodes = t(adj.copy()), t(node.copy())
    final_samples_nodes, final_samples_nodes_bbox = final_samples_nodes[..., :-4], final_samples_nodes[..., -4:]
    final_samples_nodes_bbox = final_samples_nodes_bbox * 0.5 + 0.5
    final_samples_nodes_bbox = mask_nodes(final_samples_nodes_bbox.cpu(), node_flags.cpu())
    node_samples = final_samples_nodes.clamp(-1.0, 1.0)
    node_samples = torch.where(node_samples > 0.0, torch.ones_like(node_samples), -torch.ones_like(node_samples))
    node_samples = mask_nodes(node_samples, node_flags)
    _q_binary_node = node_samples.gt(0.0).cpu().float()
    _q_binary_node = mask_nodes(_q_binary_node, node_flags.cpu())
    _q_node = bin2dec(_q_binary_node, num_bits=np.ceil(np.log2(raw_num_node_type)).astype(int))
    _q_node = mask_nodes(_q_node, node_flags.cpu()).clamp(min=0, max=raw_num_node_type - 1)
    adj_samples = final_samples_adjs.clamp(-1.0, 1.0)
    adj_samples = torch.where(adj_samples > 0.0, torch.ones_like(adj_samples), -torch.ones_like(adj_samples))
    adj_samples = mask_adjs(adj_samples, node_flags)
    _q_binary_adj = adj_samples.gt(0.0).cpu().float()
    _q_binary_adj = mask_adjs(_q_binary_adj, node_flags.cpu())
    _q_binary_adj = _q_binary_adj.permute(0, 2, 3, 1)
    _q_adj = bin2dec(_q_binary_adj, num_bits=np.ceil(np.log2(raw_num_adj_type)).astype(int))
    _q_adj = mask_adjs(_q_adj, node_flags.cpu()).clamp(min=0, max=raw_num_adj_type - 1)
    b, n = node_flags.shape[:2]
    _q_adj[:, torch.eye(n, device=_q_adj.device).bool()] = 0.0
    return _q_adj.contiguous().numpy(), _q_node.numpy(), final_samples_nodes_bbox.numpy().astype(np.float32)


def gen_long_trajectories(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, out):
    """SURVEY §8c G4 at realistic length on the full-size networks: the VG and COCO-Stuff nets, B = 2, T = 50 Heun + churn through the
    REFERENCE's own sampler with replayed noise and coins (Y.LONG_TRAJ; ~150 network forwards of 2 graphs each), and the reference's
    'bits' decode of the result (ref_decode_bits).  Stored: the final raw (adj, node) in full (padded rows / columns are zeros and
    compress away) and the decoded integer graphs + bbox.  The GPU tests compare the fp32 and the bf16 HIP paths with these: a stated
    continuous bound and the decoded agreement as a RATE against the reference's decode."""
    res = {}
    for tag, (name, T, solver, churn, valid, seed, stream, coins, (dataset, n_adj, n_node)) in Y.LONG_TRAJ.items():
        cfg = CONFIGS[name]()
        a, nd, _sig, used = run_ref_sampler(NodeAdjEDMSampler, NodeAdjPrecond, DiffuseSG, cfg, T=T, B=len(valid), valid=valid, seed=seed,
                                            tag=stream, solver=solver, S_churn=churn, coins=coins)
        flags = W.synth_flags(len(valid), cfg.max_node_num, valid)
        qa, qn, bb = ref_decode_bits(a, nd, flags, n_adj, n_node)
        res[f"{tag}_adj"], res[f"{tag}_node"], res[f"{tag}_coins_used"] = a, nd, np.array(used)
        res[f"{tag}_q_adj"], res[f"{tag}_q_node"], res[f"{tag}_bbox"] = qa.astype(np.int16), qn.astype(np.int16), bb
        near0 = float((np.abs(a[a != 0]) < 1e-3).mean())
        print(f"long trajectory {tag}: max|adj| {np.abs(a).max():.3f} max|node| {np.abs(nd).max():.3f} coins used {used}; "
              f"|adj| < 1e-3 on {near0:.2%} of the valid entries; edge classes {len(np.unique(qa))}, node classes {len(np.unique(qn))}", flush=True)
    np.savez_compressed(os.path.join(out, "traj_long.npz"), **res)


def gen_decode(out):
    """G6: the post-decode of 'bits' samples.  `_decode_node` / `_decode_adj` are closures inside sg_go_sampling
    (R/runner/sampler/sampler_node_adj.py:222-285; that module itself needs torchvision/pyemd and cannot be imported), so
    their 'bits' branch is restated here statement by statement around the reference's OWN `bin2dec`
    (R/utils/attribute_code.py:319-328) and `mask_nodes` / `mask_adjs` (R/utils/graph_utils.py), which are imported.
    bbox: R/runner/sampler/sampler_node_adj.py:194-209 (last four node channels * 0.5 + 0.5, masked)."""
    from utils.attribute_code import bin2dec
    from utils.graph_utils import mask_adjs, mask_nodes
    from utils.sg_utils import get_node_adj_num_type
    res = {}
    for name, (dataset, raw_adj, raw_node, _valid) in Y.DECODE_CASES.items():
        cfg, flags, adj, node = Y.decode_case(name)
        info = get_node_adj_num_type(dataset, flag_sg=True, encoding="bits", flag_node_only=False, flag_node_bbox=True)
        raw_num_node_type, raw_num_adj_type = info["raw_num_node_type"], info["raw_num_adj_type"]
        assert (raw_num_adj_type, raw_num_node_type) == (raw_adj, raw_node)
        node_flags = t(flags)
        final_samples_adjs, final_samples_nodes = t(adj.copy()), t(node.copy())
        # :201-209 (flag_bbox, not node-only)
        final_samples_nodes, final_samples_nodes_bbox = final_samples_nodes[..., :-4], final_samples_nodes[..., -4:]
        final_samples_nodes_bbox = final_samples_nodes_bbox * 0.5 + 0.5
        final_samples_nodes_bbox = mask_nodes(final_samples_nodes_bbox.cpu(), node_flags.cpu())
        # _decode_node, 'bits' (:222-233)
        node_samples = final_samples_nodes.clamp(-1.0, 1.0)
        node_samples = torch.where(node_samples > 0.0, torch.ones_like(node_samples), -torch.ones_like(node_samples))
        node_samples = mask_nodes(node_samples, node_flags)
        _q_binary_node = node_samples.gt(0.0).cpu().float()
        _q_binary_node = mask_nodes(_q_binary_node, node_flags.cpu())
        _q_node = bin2dec(_q_binary_node, num_bits=np.ceil(np.log2(raw_num_node_type)).astype(int))
        _q_node = mask_nodes(_q_node, node_flags.cpu()).clamp(min=0, max=raw_num_node_type - 1)
        # _decode_adj, 'bits' (:242-247, :265-275, :279-283)
        adj_samples = final_samples_adjs.clamp(-1.0, 1.0)
        adj_samples = torch.where(adj_samples > 0.0, torch.ones_like(adj_samples), -torch.ones_like(adj_samples))
        adj_samples = mask_adjs(adj_samples, node_flags)
        _q_binary_adj = adj_samples.gt(0.0).cpu().float()
        _q_binary_adj = mask_adjs(_q_binary_adj, node_flags.cpu())
        _q_binary_adj = _q_binary_adj.permute(0, 2, 3, 1)
        _q_adj = bin2dec(_q_binary_adj, num_bits=np.ceil(np.log2(raw_num_adj_type)).astype(int))
        _q_adj = mask_adjs(_q_adj, node_flags.cpu()).clamp(min=0, max=raw_num_adj_type - 1)
        b, n = node_flags.shape[:2]
        _q_adj[:, torch.eye(n, device=_q_adj.device).bool()] = 0.0
        qa, qn = _q_adj.contiguous().numpy(), _q_node.numpy()
        assert np.array_equal(qa, np.round(qa)) and np.array_equal(qn, np.round(qn))
        res[f"{name}_q_adj"], res[f"{name}_q_node"] = qa.astype(np.int16), qn.astype(np.int16)
        res[f"{name}_bbox"] = final_samples_nodes_bbox.numpy().astype(np.float32)
        print(f"decode {name}: q_adj max {qa.max():.0f} (clamped at {raw_num_adj_type - 1}: {(qa == raw_num_adj_type - 1).sum()}), "
              f"q_node max {qn.max():.0f} (clamped at {raw_num_node_type - 1}: {(qn == raw_num_node_type - 1).sum()})")
    np.savez_compressed(os.path.join(out, "decode.npz"), **res)


def gen_decode_enc(out):
    """The post-decode for `--edge_encoding` / `--node_encoding` in {'one_hot', 'ddpm'} (and mixed with 'bits'): the closures
    `_decode_node` / `_decode_adj` of sg_go_sampling (R/runner/sampler/sampler_node_adj.py:222-285; the module needs torchvision /
    pyemd and cannot be imported) restated statement by statement -- every branch, with flag_node_only = flag_binary_edge = False as
    for scene graphs -- around the reference's OWN `attribute_converter` and `bin2dec` (R/utils/attribute_code.py:13, :319) and
    `mask_nodes` / `mask_adjs` (R/utils/graph_utils.py), which are imported and do the arithmetic."""
    from utils.attribute_code import attribute_converter, bin2dec
    from utils.graph_utils import mask_adjs, mask_nodes
    from utils.sg_utils import get_node_adj_num_type
    flag_node_only = flag_binary_edge = False
    res = {}
    for name, (dataset, edge_encoding, node_encoding, _n, _valid) in Y.DECODE_ENC_CASES.items():
        cfg, flags, adj, node, e_adj, e_node, n_adj_type, n_node_type = Y.decode_enc_case(name)
        for enc, c_have, key in ((edge_encoding, cfg.c_adj, "out_chans_adj"), (node_encoding, cfg.c_node, "out_chans_node")):
            info = get_node_adj_num_type(dataset, flag_sg=True, encoding=enc, flag_node_only=False, flag_node_bbox=True)
            assert info[key] == c_have, (name, key, info[key], c_have)
        raw_num_node_type, raw_num_adj_type = info["raw_num_node_type"], info["raw_num_adj_type"]
        assert (raw_num_adj_type, raw_num_node_type) == (n_adj_type, n_node_type)

        def _decode_node(node_samples, node_flags, encoding_method):   # sampler_node_adj.py:222-240
            node_samples = node_samples.clamp(-1.0, 1.0)
            if encoding_method in ['bits', 'one_hot']:
                node_samples = torch.where(node_samples > 0.0, torch.ones_like(node_samples), -torch.ones_like(node_samples))
                node_samples = mask_nodes(node_samples, node_flags)
            if encoding_method == 'bits':
                _q_binary_node = node_samples.gt(0.0).cpu().float()
                _q_binary_node = mask_nodes(_q_binary_node, node_flags.cpu())
                _q_node = bin2dec(_q_binary_node, num_bits=np.ceil(np.log2(raw_num_node_type)).astype(int))
                _q_node = mask_nodes(_q_node, node_flags.cpu()).clamp(min=0, max=raw_num_node_type - 1)
            else:
                if len(node_samples.shape) == 3 and node_samples.shape[-1] == 1:
                    node_samples = node_samples.squeeze(-1)
                _q_node = attribute_converter(in_attr=node_samples, attr_flags=node_flags.cpu(),
                                              in_encoding=encoding_method, out_encoding='int', num_attr_type=raw_num_node_type,
                                              flag_nodes=True, flag_adjs=False,
                                              flag_in_ddpm_range=True, flag_out_ddpm_range=False)
            return _q_node

        def _decode_adj(adj_samples, node_flags, encoding_method):     # sampler_node_adj.py:242-285
            adj_samples = adj_samples.clamp(-1.0, 1.0)
            if encoding_method in ['bits', 'one_hot']:
                adj_samples = torch.where(adj_samples > 0.0, torch.ones_like(adj_samples), -torch.ones_like(adj_samples))
                adj_samples = mask_adjs(adj_samples, node_flags)
            if encoding_method in ['ddpm', 'one_hot']:
                if encoding_method == 'ddpm':
                    _num_attr_type = raw_num_adj_type
                    if flag_node_only:
                        _num_attr_type = raw_num_node_type
                    if flag_binary_edge:
                        _num_attr_type = 2
                elif encoding_method == 'one_hot':
                    _num_attr_type = raw_num_adj_type
                else:
                    raise NotImplementedError
                _q_adj = attribute_converter(in_attr=adj_samples, attr_flags=node_flags.cpu(),
                                             in_encoding=encoding_method, out_encoding='int',
                                             num_attr_type=_num_attr_type,
                                             flag_nodes=True, flag_adjs=False,
                                             flag_in_ddpm_range=True, flag_out_ddpm_range=False)
            elif encoding_method == 'bits':
                if flag_binary_edge:
                    adj_samples = adj_samples.unsqueeze(1)
                _q_binary_adj = adj_samples.gt(0.0).cpu().float()
                _q_binary_adj = mask_adjs(_q_binary_adj, node_flags.cpu())
                _q_binary_adj = _q_binary_adj.permute(0, 2, 3, 1)
                _q_adj = bin2dec(_q_binary_adj, num_bits=np.ceil(np.log2(raw_num_adj_type)).astype(int))
                _q_adj = mask_adjs(_q_adj, node_flags.cpu()).clamp(min=0, max=raw_num_adj_type - 1)
            else:
                raise NotImplementedError
            b, n = node_flags.shape[:2]
            if not flag_node_only:
                _q_adj[:, torch.eye(n, device=_q_adj.device).bool()] = 0.0
            return _q_adj.contiguous()

        node_flags = t(flags)
        # the sampler hands back squeezed singleton channels (edm.py:437-443 via flag_*_multi_channel): 'ddpm' adjacency is [B, N, N]
        final_samples_adjs = t(adj.copy())
        if final_samples_adjs.shape[1] == 1:
            final_samples_adjs = final_samples_adjs[:, 0]
        final_samples_nodes = t(node.copy())
        # :194-209 (flag_bbox, not node-only)
        final_samples_nodes, final_samples_nodes_bbox = final_samples_nodes[..., :-4], final_samples_nodes[..., -4:]
        final_samples_nodes_bbox = final_samples_nodes_bbox * 0.5 + 0.5
        final_samples_nodes_bbox = mask_nodes(final_samples_nodes_bbox.cpu(), node_flags.cpu())
        q_node = _decode_node(final_samples_nodes.cpu(), node_flags.cpu(), node_encoding)      # :288
        q_adj = _decode_adj(final_samples_adjs.cpu(), node_flags.cpu(), edge_encoding)        # :290
        qa, qn = q_adj.numpy().astype(np.float64), q_node.numpy().astype(np.float64)
        assert np.array_equal(qa, np.round(qa)) and np.array_equal(qn, np.round(qn)) and qa.min() >= 0 and qn.min() >= 0
        res[f"{name}_q_adj"], res[f"{name}_q_node"] = qa.astype(np.int16), qn.astype(np.int16)
        res[f"{name}_bbox"] = final_samples_nodes_bbox.numpy().astype(np.float32)
        print(f"decode {name} ({edge_encoding}/{node_encoding}): q_adj classes used {len(np.unique(qa))} of {raw_num_adj_type} (max {qa.max():.0f}), "
              f"q_node classes used {len(np.unique(qn))} of {raw_num_node_type} (max {qn.max():.0f})")
    np.savez_compressed(os.path.join(out, "decode_enc.npz"), **res)


def gen_train_forward(DiffuseSG, NodeAdjPrecond, out):
    """G7: the forward half of a training / test-loss step (R/runner/trainer/trainer_node_adj.py:96-163, mode 'test'):
    NodeAdjEDMObjectiveGenerator.get_input_output (imported; its torch.randn / randn_like draws replayed from the portable
    streams), the preconditioned model called the way the trainer calls it (kwargs, per-sample sigmas; the self-conditioning
    coin replayed), NodeAdjRainbowLoss(reduction='none') (imported) and the trainer's bounding-box IoU term.  torchvision is
    not installed here, so its two helpers used by that term are restated in torch: box_convert('cxcywh'->'xyxy') and the
    diagonal of box_iou = inter / (area_a + area_b - inter)."""
    import model.precond.precond as P
    from loss.rainbow_loss import NodeAdjRainbowLoss
    from runner.objectives.edm import NodeAdjEDMObjectiveGenerator
    res = {}
    for name in ("tiny",):
        cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin = Y.train_case(name)
        net = build_ref_net(DiffuseSG, cfg)
        model = NodeAdjPrecond("edm", net, cfg.self_condition, symmetric_noise=False).eval()
        gen = NodeAdjEDMObjectiveGenerator(precond="edm", sigma_dist="edm", other_params=None, dev="cpu", symmetric_noise=False)
        draws = _Replay([rnd, eps_adj, eps_node])   # draw order: sigmas (:176), adjacency noise (graph_utils:139), node noise (:249)

        def fake_randn(*size, **kw):
            v = draws.pop()
            return torch.from_numpy(np.ascontiguousarray(v).copy())

        def fake_randn_like(x, **kw):
            v = draws.pop()
            return torch.from_numpy(np.ascontiguousarray(v).reshape(tuple(x.shape)).copy()).to(x.dtype)

        real_r, real_rl, real_rand = torch.randn, torch.randn_like, P.np.random.rand
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        P.np.random.rand = lambda: coin
        try:
            adjs_gt, nodes_gt, node_flags = t(clean_adj.copy()), t(clean_node.copy()), t(flags)
            net_input_a, net_input_x, net_cond, net_target_a, net_target_x, (c_skip, c_out, c_in, c_noise, sigmas, weights) = \
                gen.get_input_output(adjs_gt, nodes_gt, node_flags)
            assert draws.i == 3
            with torch.no_grad():   # mode == 'test' (:109-111)
                net_output_a, net_output_x = model(adjs=net_input_a, nodes=net_input_x, node_flags=node_flags, sigmas=sigmas)
        finally:
            torch.randn, torch.randn_like, P.np.random.rand = real_r, real_rl, real_rand
        loss_func = NodeAdjRainbowLoss(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
        reg_loss_adj, reg_loss_node = loss_func(net_pred_a=net_output_a, net_pred_x=net_output_x, net_target_a=net_target_a,
                                                net_target_x=net_target_x, net_cond=net_cond, adjs_perturbed=net_input_a,
                                                adjs_gt=adjs_gt, x_perturbed=net_input_x, x_gt=nodes_gt, node_flags=node_flags,
                                                loss_weight=weights, reduction='none')
        res[f"{name}_loss_adj_noiou"], res[f"{name}_loss_node_noiou"] = reg_loss_adj.numpy().copy(), reg_loss_node.numpy().copy()
        # IoU term (:130-159), iou_loss_type == 'iou', iou_loss_weight == 1.0 (both reference YAMLs)
        iou_loss_weight = 1.0

        def box_convert_cxcywh_xyxy(b):
            cx, cy, w, h = b.unbind(-1)
            return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

        def box_iou_diag(a, b):
            area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
            area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            lt, rb = torch.max(a[:, :2], b[:, :2]), torch.min(a[:, 2:], b[:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            return inter / (area_a + area_b - inter)
        net_output_x_bbox = (net_output_x[..., -4:] + 1.0) / 2.0
        net_target_x_bbox = (net_target_x[..., -4:] + 1.0) / 2.0
        net_output_x_bbox = box_convert_cxcywh_xyxy(net_output_x_bbox).clamp(min=0.0, max=1.0)
        net_target_x_bbox = box_convert_cxcywh_xyxy(net_target_x_bbox).clamp(min=0.0, max=1.0)
        bbox_iou = box_iou_diag(net_output_x_bbox.view(-1, 4), net_target_x_bbox.view(-1, 4))
        node_iou_loss = -(bbox_iou.view(-1)) ** 2.0
        node_flags_t = node_flags.view(-1)
        node_iou_loss = node_iou_loss * node_flags_t.to(torch.float32)
        node_iou_loss = node_iou_loss.view(-1, node_flags.shape[1])
        # (the reference divides by node_flags_t.sum(dim=-1), the TOTAL number of valid nodes of the batch -- :158)
        node_iou_loss = node_iou_loss.sum(dim=-1) / node_flags_t.sum(dim=-1).to(torch.float32)
        reg_loss_node_iou = reg_loss_node + iou_loss_weight * node_iou_loss * weights
        loss = reg_loss_adj.mean() + reg_loss_node_iou.mean()
        for k, v in dict(sigmas=sigmas, weights=weights, noisy_adj=net_input_a, noisy_node=net_input_x, pred_adj=net_output_a,
                         pred_node=net_output_x, loss_adj=reg_loss_adj, loss_node=reg_loss_node_iou).items():
            res[f"{name}_{k}"] = v.numpy().copy()
        res[f"{name}_loss"] = np.array(float(loss))
        print(f"train forward {name}: sigmas {sigmas.numpy()}, loss_adj {reg_loss_adj.numpy()}, loss_node {reg_loss_node_iou.numpy()}, loss {float(loss):.6f}")
    np.savez_compressed(os.path.join(out, "train_forward.npz"), **res)


# ---- torchvision.ops box losses, restated (torchvision is not installed in this image; it is an un-vendored dependency of the
# reference).  Statement-for-statement after torchvision/ops/_utils.py::_loss_inter_union, giou_loss.py, diou_loss.py, ciou_loss.py
# (0.18, the companion of the pinned torch 2.3.0), reduction='none', so that the reference's own autograd engine differentiates them.
def _tv_loss_inter_union(boxes1, boxes2):
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xkis1, ykis1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xkis2, ykis2 = torch.min(x2, x2g), torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    return intsctk, unionk


def _tv_generalized_box_iou_loss(boxes1, boxes2, eps=1e-7):
    intsctk, unionk = _tv_loss_inter_union(boxes1, boxes2)
    iouk = intsctk / (unionk + eps)
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    miouk = iouk - ((area_c - unionk) / (area_c + eps))
    return 1 - miouk


def _tv_diou_iou_loss(boxes1, boxes2, eps=1e-7):
    intsct, union = _tv_loss_inter_union(boxes1, boxes2)
    iou = intsct / (union + eps)
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    diagonal_distance_squared = ((xc2 - xc1) ** 2) + ((yc2 - yc1) ** 2) + eps
    x_p, y_p = (x2 + x1) / 2, (y2 + y1) / 2
    x_g, y_g = (x1g + x2g) / 2, (y1g + y2g) / 2
    centers_distance_squared = ((x_p - x_g) ** 2) + ((y_p - y_g) ** 2)
    loss = 1 - iou + (centers_distance_squared / diagonal_distance_squared)
    return loss, iou


def _tv_complete_box_iou_loss(boxes1, boxes2, eps=1e-7):
    import math
    diou_loss, iou = _tv_diou_iou_loss(boxes1, boxes2)
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    w_pred, h_pred = x2 - x1, y2 - y1
    w_gt, h_gt = x2g - x1g, y2g - y1g
    v = (4 / (math.pi ** 2)) * torch.pow((torch.atan(w_pred / h_pred) - torch.atan(w_gt / h_gt)), 2)
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return diou_loss + alpha * v


def gen_iou_losses(out):
    """G10: the trainer's bounding-box term for every iou_loss_type (R/runner/trainer/trainer_node_adj.py:130-159, the block restated
    literally below) on top of the imported NodeAdjRainbowLoss(reduction='none'), and the gradient of
    loss_adj.mean() + loss_node.mean() with respect to the model outputs through the reference's autograd.  `box_convert` /
    `box_iou` as in G7; the three torchvision losses restated above.  Inputs: diffusesg_amd.synth.iou_case."""
    from loss.rainbow_loss import NodeAdjRainbowLoss
    cfg, flags, pred_adj, pred_node, tgt_adj, tgt_node, wts, sigmas = Y.iou_case()
    res = {}
    edge_w, node_w, iou_loss_weight = 2.0, 0.5, 1.5

    def box_convert(b, in_fmt, out_fmt):
        assert (in_fmt, out_fmt) == ("cxcywh", "xyxy")
        cx, cy, w, h = b.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

    def box_iou(a, b):   # torchvision.ops.boxes.box_iou, full [M, M] matrix like the reference's call
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt, rb = torch.max(a[:, None, :2], b[:, :2]), torch.min(a[:, None, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        return inter / (area_a[:, None] + area_b - inter)
    generalized_box_iou_loss = lambda a, b, reduction: _tv_generalized_box_iou_loss(a, b)
    distance_box_iou_loss = lambda a, b, reduction: _tv_diou_iou_loss(a, b)[0]
    complete_box_iou_loss = lambda a, b, reduction: _tv_complete_box_iou_loss(a, b)
    loss_func = NodeAdjRainbowLoss(edge_loss_weight=edge_w, node_loss_weight=node_w, flag_reweight=False, objective="edm")
    torch.set_num_threads(1)
    for iou_loss_type in Y.IOU_TYPES:
        net_output_a, net_output_x = t(pred_adj.copy()).requires_grad_(True), t(pred_node.copy()).requires_grad_(True)
        net_target_a, net_target_x, node_flags, weights = t(tgt_adj.copy()), t(tgt_node.copy()), t(flags), t(wts.copy())
        reg_loss_adj, reg_loss_node = loss_func(net_pred_a=net_output_a, net_pred_x=net_output_x, net_target_a=net_target_a,
                                                net_target_x=net_target_x, net_cond=None, adjs_perturbed=None, adjs_gt=net_target_a,
                                                x_perturbed=None, x_gt=net_target_x, node_flags=node_flags, loss_weight=weights,
                                                reduction='none')
        # ---- trainer_node_adj.py:130-159 ----
        net_output_x_bbox = (net_output_x[..., -4:] + 1.0) / 2.0
        net_target_x_bbox = (net_target_x[..., -4:] + 1.0) / 2.0
        net_output_x_bbox = box_convert(net_output_x_bbox, in_fmt='cxcywh', out_fmt='xyxy').clamp(min=0.0, max=1.0)
        net_target_x_bbox = box_convert(net_target_x_bbox, in_fmt='cxcywh', out_fmt='xyxy').clamp(min=0.0, max=1.0)
        if iou_loss_type == 'iou':
            bbox_iou_loss = box_iou(net_output_x_bbox.view(-1, 4), net_target_x_bbox.view(-1, 4))
            node_iou_loss = - (bbox_iou_loss.diag().view(-1)) ** 2.0
        elif iou_loss_type == 'ciou':
            node_iou_loss = complete_box_iou_loss(net_output_x_bbox.view(-1, 4), net_target_x_bbox.view(-1, 4), reduction='none')
        elif iou_loss_type == 'diou':
            node_iou_loss = distance_box_iou_loss(net_output_x_bbox.view(-1, 4), net_target_x_bbox.view(-1, 4), reduction='none')
        elif iou_loss_type == 'giou' or iou_loss_type == 'giou_squared':
            node_iou_loss = generalized_box_iou_loss(net_output_x_bbox.view(-1, 4), net_target_x_bbox.view(-1, 4), reduction='none')
            if iou_loss_type == 'giou_squared':
                node_iou_loss = node_iou_loss ** 2.0
        else:
            raise NotImplementedError
        node_flags_t = node_flags.view(-1)
        per_node = node_iou_loss.detach().clone()
        node_iou_loss = node_iou_loss * node_flags_t.to(torch.float32)
        node_iou_loss = node_iou_loss.view(-1, node_flags.shape[1])
        node_iou_loss = node_iou_loss.sum(dim=-1) / node_flags_t.sum(dim=-1).to(torch.float32)
        reg_loss_node = reg_loss_node + iou_loss_weight * node_iou_loss * weights
        # ---- :163-170 ----
        loss = reg_loss_adj.mean() + reg_loss_node.mean()
        loss.backward()
        assert torch.isfinite(net_output_x.grad).all() and torch.isfinite(per_node[node_flags_t]).all(), iou_loss_type
        res[f"{iou_loss_type}_loss_adj"], res[f"{iou_loss_type}_loss_node"] = reg_loss_adj.detach().numpy().copy(), reg_loss_node.detach().numpy().copy()
        res[f"{iou_loss_type}_grad_adj"], res[f"{iou_loss_type}_grad_node"] = net_output_a.grad.numpy().copy(), net_output_x.grad.numpy().copy()
        res[f"{iou_loss_type}_per_node"] = per_node.numpy().copy()
        f_ = node_flags_t.numpy().astype(bool)
        raw = box_convert((net_output_x.detach()[..., -4:] + 1.0) / 2.0, 'cxcywh', 'xyxy').view(-1, 4).numpy()[f_]
        print(f"iou losses {iou_loss_type}: loss_node {reg_loss_node.detach().numpy()}, valid boxes {f_.sum()}, clamped corners "
              f"{int(((raw < 0) | (raw > 1)).sum())}, |grad bbox| {float(net_output_x.grad[..., -4:].abs().max()):.3e}")
    res["edge_w"], res["node_w"], res["iou_w"] = np.array(edge_w), np.array(node_w), np.array(iou_loss_weight)
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(out, "iou_losses.npz"), **res)


def gen_train_backward(DiffuseSG, NodeAdjPrecond, out):
    """G8: gradients of one training step from the reference's own autograd (R/runner/trainer/trainer_node_adj.py:96-170,
    mode 'train' up to and including loss.backward(), the clip-norm it computes and -- tiny case -- the Adam step of
    utils/learning_utils.py:137-140 with the YAMLs' lr 2e-4 / weight_decay 0).  Cases: `tiny` (inputs, draws and coin of G7: the coin
    does not fire), `tinysc` (same, coin forced to 0.3: the detached self-conditioning pass feeds the differentiated one), `vg` (the
    Visual Genome network, B = 2, coin 0.3; 64-token shifted windows on four levels), `coco` (the COCO-Stuff network, B = 1, coin 0.3;
    100-token windows, three levels).  Saved per case: loss, dL/d(preconditioned
    outputs), dL/d(raw network outputs), per parameter the gradient's L2 norm and every stride-th element (<= 1024, tinysc: <= 256, vg: <= 64, coco: <= 32), the
    total gradient norm nn.utils.clip_grad_norm_ reports; tiny: the parameters after the optimiser step (<= 256 values each).
    torchvision's box helpers are restated as in G7."""
    import model.precond.precond as P
    from loss.rainbow_loss import NodeAdjRainbowLoss
    from runner.objectives.edm import NodeAdjEDMObjectiveGenerator
    res = {}
    for name, cfg_name, B, forced_coin, max_sample in (("tiny", "tiny", 4, None, 1024), ("tinysc", "tiny", 4, 0.3, 256), ("vg", "vg", 2, 0.3, 64),
                                                          ("coco", "coco", 1, 0.3, 32)):
        cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin = Y.train_case(cfg_name, B=B)
        if forced_coin is not None:
            coin = forced_coin
        net = build_ref_net(DiffuseSG, cfg)
        model = NodeAdjPrecond("edm", net, cfg.self_condition, symmetric_noise=False)
        model.train()
        gen = NodeAdjEDMObjectiveGenerator(precond="edm", sigma_dist="edm", other_params=None, dev="cpu", symmetric_noise=False)
        draws = _Replay([rnd, eps_adj, eps_node])

        def fake_randn(*size, **kw):
            return torch.from_numpy(np.ascontiguousarray(draws.pop()).copy())

        def fake_randn_like(x, **kw):
            return torch.from_numpy(np.ascontiguousarray(draws.pop()).reshape(tuple(x.shape)).copy()).to(x.dtype)

        raw = {}

        def keep_raw(mod, inp, outp):   # the LAST call of the network is the differentiated one (the coin's extra call runs under no_grad)
            if outp[0].requires_grad:
                outp[0].retain_grad(); outp[1].retain_grad()
                raw["a"], raw["x"] = outp
        hook = net.register_forward_hook(keep_raw)
        real_r, real_rl, real_rand = torch.randn, torch.randn_like, P.np.random.rand
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        P.np.random.rand = lambda: coin
        try:
            adjs_gt, nodes_gt, node_flags = t(clean_adj.copy()), t(clean_node.copy()), t(flags)
            net_input_a, net_input_x, net_cond, net_target_a, net_target_x, (c_skip, c_out, c_in, c_noise, sigmas, weights) = \
                gen.get_input_output(adjs_gt, nodes_gt, node_flags)
            optimizer = torch.optim.Adam(model.parameters(), lr=2.0e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
            optimizer.zero_grad(set_to_none=True)
            net_output_a, net_output_x = model(adjs=net_input_a, nodes=net_input_x, node_flags=node_flags, sigmas=sigmas)
        finally:
            torch.randn, torch.randn_like, P.np.random.rand = real_r, real_rl, real_rand
            hook.remove()
        net_output_a.retain_grad(); net_output_x.retain_grad()
        loss_func = NodeAdjRainbowLoss(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
        reg_loss_adj, reg_loss_node = loss_func(net_pred_a=net_output_a, net_pred_x=net_output_x, net_target_a=net_target_a,
                                                net_target_x=net_target_x, net_cond=net_cond, adjs_perturbed=net_input_a,
                                                adjs_gt=adjs_gt, x_perturbed=net_input_x, x_gt=nodes_gt, node_flags=node_flags,
                                                loss_weight=weights, reduction='none')
        iou_loss_weight = 1.0

        def box_convert_cxcywh_xyxy(b):
            cx, cy, w, h = b.unbind(-1)
            return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

        def box_iou_diag(a, b):
            area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
            area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            lt, rb = torch.max(a[:, :2], b[:, :2]), torch.min(a[:, 2:], b[:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            return inter / (area_a + area_b - inter)
        ob = box_convert_cxcywh_xyxy((net_output_x[..., -4:] + 1.0) / 2.0).clamp(min=0.0, max=1.0)
        tb = box_convert_cxcywh_xyxy((net_target_x[..., -4:] + 1.0) / 2.0).clamp(min=0.0, max=1.0)
        node_iou_loss = -(box_iou_diag(ob.view(-1, 4), tb.view(-1, 4)).view(-1)) ** 2.0
        node_flags_t = node_flags.view(-1)
        node_iou_loss = (node_iou_loss * node_flags_t.to(torch.float32)).view(-1, node_flags.shape[1])
        node_iou_loss = node_iou_loss.sum(dim=-1) / node_flags_t.sum(dim=-1).to(torch.float32)
        reg_loss_node = reg_loss_node + iou_loss_weight * node_iou_loss * weights
        loss = reg_loss_adj.mean() + reg_loss_node.mean()
        torch.set_num_threads(1)   # index_add / scatter accumulations of autograd are order-dependent across threads: one thread -> bit-stable fixtures
        loss.backward()
        torch.set_num_threads(8)
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0, norm_type=2)   # returns the norm BEFORE clipping
        res[f"{name}_loss"] = np.array(float(loss.detach()))
        res[f"{name}_coin"] = np.array(coin)
        res[f"{name}_sigmas"], res[f"{name}_weights"] = sigmas.detach().numpy().copy(), weights.detach().numpy().copy()
        res[f"{name}_total_grad_norm"] = np.array(float(total_norm))
        if name not in ("vg", "coco"):
            res[f"{name}_pred_adj"], res[f"{name}_pred_node"] = net_output_a.detach().numpy().copy(), net_output_x.detach().numpy().copy()
        if name == "tiny":
            res[f"{name}_grad_pred_adj"], res[f"{name}_grad_pred_node"] = net_output_a.grad.numpy().copy(), net_output_x.grad.numpy().copy()
            res[f"{name}_grad_F_adj"], res[f"{name}_grad_F_node"] = raw["a"].grad.numpy().copy(), raw["x"].grad.numpy().copy()
        scale = min(1.0, 10.0 / (float(total_norm) + 1e-6))
        n_par = 0
        names, norms = [], []
        for k, p_ in model.named_parameters():   # un-clipped gradients: small tensors whole, large ones as every stride-th element + norm
            assert p_.grad is not None, k
            g_ = (p_.grad / scale).numpy().reshape(-1)
            stride = max(1, -(-g_.size // max_sample))
            res[f"{name}_gparam/{k}"] = g_[::stride].copy()
            names.append(k); norms.append(float(np.sqrt((g_.astype(np.float64) ** 2).sum())))
            n_par += p_.numel()
        res[f"{name}_gparam_names"] = np.array(names)
        res[f"{name}_gparam_norms"] = np.array(norms)
        if name == "tiny":   # the optimiser step on the clipped gradients (trainer_node_adj.py:170-171)
            optimizer.step()
            for k, p_ in model.named_parameters():
                v_ = p_.detach().numpy().reshape(-1)
                res[f"{name}_param_after/{k}"] = v_[::max(1, -(-v_.size // 256))].copy()
        print(f"train backward {name}: loss {float(loss.detach()):.6f}, |grad| {float(total_norm):.5f} over {n_par} parameters ({len(names)} tensors), "
              f"coin {coin:.3f}, |dL/dF_adj| {float(raw['a'].grad.norm()):.4e}")
    np.savez_compressed(os.path.join(out, "train_backward.npz"), **res)


def gen_noise_embed(DiffuseSG, out):
    """G1 (stand-alone): the noise-conditioning path on its own -- PositionalEmbedding (`map_noise`), map_layer0/1 with SiLU
    (diffusesg.py:768-771) and every `affine` linear applied to the embedding (PatchEmbed :574, the Swin blocks :238), concatenated
    in module order: patch_embed, down_layers[l].blocks[j], up_layers[i].blocks[j]."""
    from torch.nn.functional import silu
    res = {}
    c_noise = np.array([-2.3, -0.35, 0.4, 1.0955], np.float32)   # ln(sigma)/4 for sigma in [1e-4, 80]
    res["c_noise"] = c_noise
    for name in ("tiny", "vg", "coco"):
        cfg = CONFIGS[name]()
        net = build_ref_net(DiffuseSG, cfg)
        with torch.no_grad():
            pe = net.map_noise(t(c_noise))
            emb = silu(net.map_layer1(silu(net.map_layer0(pe))))
            aff = [net.patch_embed.affine(emb)]
            for l in net.down_layers:
                aff += [b.affine(emb) for b in l.blocks]
            for l in net.up_layers:
                aff += [b.affine(emb) for b in l.blocks]
        res[f"{name}_pe"], res[f"{name}_emb"] = pe.numpy(), emb.numpy()
        if name != "coco":   # (kept small: COCO's 11328 affine outputs add nothing the VG table does not exercise)
            res[f"{name}_aff"] = torch.cat(aff, dim=1).numpy()
        print(f"noise embed {name}: pe {tuple(pe.shape)} emb {tuple(emb.shape)} affine outputs {sum(a.shape[1] for a in aff)}")
    np.savez_compressed(os.path.join(out, "noise_embed.npz"), **res)


BLOCK_CASES = [("small", "down_layers.0.blocks.1", 2),    # 16x16 tokens, 4x4 windows, shift 2: the -100 region mask is active
               ("tiny", "down_layers.1.blocks.0", 3),     # 4x4 tokens, window = whole map (no partition), C = 192, 6 heads
               ("coco", "down_layers.1.blocks.1", 1)]     # 20x20 tokens, 10x10 windows (100 tokens), shift 5, C = 192


def gen_block_backward(DiffuseSG, out):
    """G9: one SwinTransformerBlock on its own (diffusesg.py:232-277), forward and the reference's autograd backward:
    x_out, dL/dx_in, dL/demb and the gradient of each of the block's 15 parameters (L2 norm + every stride-th element, <= 512
    values; x_out / dL/dx_in as every 4th token row except for the tiny case) for L = sum(x_out * dY) with a random dY.  Inputs from the portable generator (diffusesg_amd.synth.block_case)."""
    res = {}
    for name, prefix, B in BLOCK_CASES:
        cfg = CONFIGS[name]()
        net = build_ref_net(DiffuseSG, cfg)
        mod = net
        for part in prefix.split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        x, emb, dy = Y.block_case(cfg, prefix, B)
        xt, et = t(x.copy()).requires_grad_(True), t(emb.copy()).requires_grad_(True)
        for p_ in mod.parameters():
            p_.grad = None
        y = mod(xt, et)
        torch.set_num_threads(1)   # bit-stable accumulation order (see gen_train_backward)
        (y * t(dy)).sum().backward()
        torch.set_num_threads(8)
        key = f"{name}/{prefix}"
        rs = 1 if name == "tiny" else 4   # token rows kept (every rs-th): the fixture stays small
        res[f"{key}/row_stride"] = np.array(rs)
        res[f"{key}/x_out"], res[f"{key}/grad_in"] = y.detach().numpy()[:, ::rs].copy(), xt.grad.numpy()[:, ::rs].copy()
        res[f"{key}/grad_emb"] = et.grad.numpy().copy()
        for k, p_ in mod.named_parameters():
            g_ = p_.grad.numpy().reshape(-1)
            stride = max(1, -(-g_.size // 512))
            res[f"{key}/gparam/{k}"] = g_[::stride].copy()
            res[f"{key}/gnorm/{k}"] = np.array(float(np.sqrt((g_.astype(np.float64) ** 2).sum())))
        print(f"block backward {key}: |x_out| {float(y.abs().max()):.3f} |grad_in| {float(xt.grad.norm()):.4f} "
              f"|d table| {float(mod.attn.relative_position_bias_table.grad.norm()):.4e}")
    np.savez_compressed(os.path.join(out, "block_backward.npz"), **res)


def check_channel_table():
    """diffusesg_amd.spec.sg_channels (data restated from sg_utils.py:348-409) against the imported reference function."""
    from utils.sg_utils import get_node_adj_num_type
    for dataset in ("visual_genome", "coco_stuff"):
        for enc in ("bits", "ddpm", "one_hot"):
            info = get_node_adj_num_type(dataset, flag_sg=True, encoding=enc, flag_node_only=False, flag_node_bbox=True)
            mine = S.sg_channels(dataset, enc)
            assert mine["c_adj"] == int(info["num_adj_type"]) == int(info["out_chans_adj"]) == int(info["in_chans_adj"]), (dataset, enc)
            assert mine["c_node"] == int(info["num_node_type"]) == int(info["out_chans_node"]), (dataset, enc)
            assert mine["in_chans"] == int(info["in_chans_node"]) + int(info["in_chans_adj"]), (dataset, enc)
            assert (mine["raw_num_node_type"], mine["raw_num_adj_type"]) == (info["raw_num_node_type"], info["raw_num_adj_type"])
    print("channel table: spec.sg_channels == get_node_adj_num_type for {visual_genome, coco_stuff} x {bits, ddpm, one_hot}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler = _import_reference()
    check_channel_table()
    if args.only in ("", "decode"):
        gen_decode(args.out)
    if args.only in ("", "decode_enc"):
        gen_decode_enc(args.out)
    if args.only in ("", "train"):
        gen_train_forward(DiffuseSG, NodeAdjPrecond, args.out)
    if args.only in ("", "train", "train_bwd"):
        gen_train_backward(DiffuseSG, NodeAdjPrecond, args.out)
    if args.only in ("", "iou"):
        gen_iou_losses(args.out)
    if args.only in ("", "noise"):
        gen_noise_embed(DiffuseSG, args.out)
    if args.only in ("", "block_bwd"):
        gen_block_backward(DiffuseSG, args.out)
    if args.only in ("", "fwd"):
        gen_forward(DiffuseSG, args.out)
    if args.only in ("", "precond"):
        gen_precond(DiffuseSG, NodeAdjPrecond, args.out)
    if args.only in ("", "sampler"):
        gen_sampler(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, args.out)
    if args.only in ("", "big"):
        gen_big_trajectories(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, args.out)
    if args.only in ("", "long"):
        gen_long_trajectories(DiffuseSG, NodeAdjPrecond, NodeAdjEDMSampler, args.out)


if __name__ == "__main__":
    main()
