"""Drop-in counterpart of `runner.mcmc_sampler.edm.NodeAdjEDMSampler` (R/runner/mcmc_sampler/edm.py:231-445).

Same constructor kwargs as `get_mc_sampler` passes (R/utils/sampling_utils.py:15-23), same `sample(...)`
signature and return convention (CPU tensors; 4-tuple with `[None]` in slot 3 when interim snapshots of a
multi-channel adjacency are requested, edm.py:439-443).  The T-step loop itself runs inside libdsg.so
(`dsg_sample`): per-step scalars are computed on the host up front, no device->host sync happens inside the
loop (the reference's per-step `.item()` logging, edm.py:433-434, is dropped), and the network forward is
replayed from a captured hipGraph.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Optional

import numpy as np
import torch

from . import lib as _lib
from .model import NodeAdjPrecondHip


class NodeAdjEDMSamplerHip(object):
    def __init__(self, *, sigma_min=None, sigma_max=None, solver="heun", discretization="edm", schedule="linear",
                 scaling="none", C_1=0.001, C_2=0.008, M=1000, alpha=1,
                 num_steps=256, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003,
                 clip_samples=False, clip_samples_min=None, clip_samples_max=None, clip_samples_scope="x_0",
                 self_condition=True, dev="cuda", objective="edm", symmetric_noise=False, use_graph=True):
        assert clip_samples_scope == "x_0"
        assert solver in ["euler", "heun"]
        assert objective in ["diffusion", "score", "edm"]
        if discretization != "edm" or schedule != "linear" or scaling != "none" or alpha != 1:
            raise NotImplementedError("only discretization='edm', schedule='linear', scaling='none', alpha=1 "
                                      "(what get_mc_sampler builds, sampling_utils.py:15-23)")
        if symmetric_noise:
            raise NotImplementedError("symmetric_noise=True is not used for scene graphs (sampling_utils.py:23)")
        self.solver, self.num_steps = solver, int(num_steps)
        self.S_churn, self.S_min, self.S_max, self.S_noise = S_churn, S_min, S_max, S_noise
        self.sigma_min = 0.002 if sigma_min is None else sigma_min   # edm_params.sigma_min_sampling
        self.sigma_max = 80.0 if sigma_max is None else sigma_max
        self.self_condition = self_condition
        self.dev = dev
        # stored but never applied by the reference loop either (mcmc_sampler/__init__.py:24-26)
        self.clip_samples, self.clip_samples_min, self.clip_samples_max = clip_samples, clip_samples_min, clip_samples_max
        self.symmetric_noise = False
        self.use_graph = use_graph
        self.seed = 1234
        self.last_stats = None
        self.sigma_steps = torch.from_numpy(_lib.sigma_schedule(self._cfg())[0])

    def _cfg(self) -> _lib.DsgSamplerCfg:
        return _lib.make_sampler_cfg(self.num_steps, self.solver, float(self.S_churn), float(self.S_min), float(self.S_max),
                                     float(self.S_noise), float(self.sigma_min), float(self.sigma_max), 7.0, self.use_graph)

    @staticmethod
    def adj_to_int(adjs_cont, node_flags, threshold):
        f = node_flags.to(adjs_cont.dtype)
        m = f.unsqueeze(-1) * f.unsqueeze(-2)
        if adjs_cont.dim() == 4:
            m = m.unsqueeze(1)
        return (adjs_cont >= threshold).to(adjs_cont.dtype) * m

    @staticmethod
    def get_num_edges(adjs_cont, node_flags, threshold):
        return (NodeAdjEDMSamplerHip.adj_to_int(adjs_cont, node_flags, threshold) > 0.0).sum([-1, -2]).float() / 2.0

    def draw_coins(self, n_calls: int) -> np.ndarray:
        """One `np.random.rand() < 0.5` per preconditioned call, in call order -- the same draws, from the same
        global NumPy generator, the reference makes inside NodeAdjPrecond.forward (precond.py:90)."""
        if not self.self_condition:
            return np.zeros(n_calls, dtype=np.uint8)
        return np.array([np.random.rand() < 0.5 for _ in range(n_calls)], dtype=np.uint8)

    @torch.no_grad()
    def device_noise(self, model, node_flags, stream: int = 0, seed=None):
        """The library's Philox stream `stream` for this batch, on the device: stream 0 is what `sample()` uses as
        gen_init_sample (edm.py:257-289: masked, unscaled), stream i+1 its churn noise of step i (edm.py:361-364)."""
        net = getattr(model, "module", model).model
        h, cfg, dev = net._ensure_handle(), net.config, net._dev
        B, n = node_flags.shape[0], cfg.max_node_num
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()
        a = torch.empty((B, cfg.c_adj, n, n), dtype=torch.float32, device=dev)
        x = torch.empty((B, n, cfg.c_node), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        h.check(h.L.dsg_gen_noise(h.raw, B, C.c_void_p(fl.data_ptr()), C.c_uint64(self.seed if seed is None else int(seed)),
                                  int(stream), C.c_void_p(a.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(st)), "dsg_gen_noise")
        return a, x

    @torch.no_grad()
    def _sample_sanity_double(self, model, node_flags, init_adjs, init_nodes, gt_adjs, gt_nodes, flag_interim_adjs, max_num_interim_adjs,
                              flag_adj_multi_channel, churn_noise, seed):
        """The float64 known-answer run (flag_use_double=True with sanity_check_gt_*; edm.py:318-445 with the denoiser bypassed): pure
        elementwise loop algebra on the device in torch float64 -- no network, hence no kernel of libdsg.so, is involved.  Returns
        float64 CPU tensors like the reference."""
        net = getattr(model, "module", model).model
        cfg, dev = net.config, net._dev
        B, n, T = node_flags.shape[0], cfg.max_node_num, self.num_steps
        f = node_flags.to(device=dev).bool()
        fa, fn = (f[:, None, :, None] & f[:, None, None, :]), f[:, :, None]
        sa, sn = (B, cfg.c_adj, n, n), (B, n, cfg.c_node)
        D = lambda t, shp: t.to(device=dev, dtype=torch.float64).reshape(shp)
        ga, gn = D(gt_adjs, sa) * fa, D(gt_nodes, sn) * fn                                   # edm.py:381-382 (masked in place)
        if init_adjs is None or init_nodes is None:                                          # gen_init_sample, edm.py:257-289
            gen = torch.Generator(device=dev).manual_seed(int(self.seed if seed is None else seed))
            init_adjs = torch.randn(sa, generator=gen, device=dev) * fa
            init_nodes = torch.randn(sn, generator=gen, device=dev) * fn
        ia, inn = D(init_adjs, sa), D(init_nodes, sn)
        t_steps = torch.cat([self.sigma_steps.to(torch.float64), torch.zeros(1, dtype=torch.float64)])   # sigma(t) = t, t_N = 0
        xa, xn = ia * t_steps[0], inn * t_steps[0]
        snaps_a, snaps_n = [ia.cpu()], [inn.cpu()]
        ts_snap = np.arange(T) if max_num_interim_adjs is None else np.linspace(0, T, max_num_interim_adjs).astype(int).clip(max=T - 1)
        for i in range(T):
            t_cur, t_next = float(t_steps[i]), float(t_steps[i + 1])
            gamma = min(self.S_churn / T, np.sqrt(2) - 1) if self.S_min <= t_cur <= self.S_max else 0
            t_hat = t_cur + gamma * t_cur
            coef = max(t_hat ** 2 - t_cur ** 2, 0.0) ** 0.5 * self.S_noise
            if churn_noise is not None:
                ea, en = D(churn_noise[0][i], sa), D(churn_noise[1][i], sn)
            else:
                ea, en = torch.randn_like(xa), torch.randn_like(xn)
            xha, xhn = (xa + coef * ea) * fa, (xn + coef * en) * fn                            # edm.py:356-366
            h = t_next - t_hat
            da, dn = ((xha - ga) / t_hat) * fa, ((xhn - gn) / t_hat) * fn                     # edm.py:384-387
            if self.solver == "euler" or i == T - 1:
                xa, xn = xha + h * da, xhn + h * dn                                           # edm.py:394-396
            else:
                t_prime = t_hat + h
                dpa, dpn = ((xha + h * da) - ga) / t_prime, ((xhn + h * dn) - gn) / t_prime   # edm.py:414-417 (stage 2 returns the GT again)
                xa, xn = xha + h * (0.5 * da + 0.5 * dpa), xhn + h * (0.5 * dn + 0.5 * dpn)
            xa, xn = xa * fa, xn * fn
            if flag_interim_adjs and i in ts_snap:
                snaps_a.append(xa.cpu()); snaps_n.append(xn.cpu())
        self.last_stats = {"precond_calls": 0, "net_forwards": 0, "graph_replays": 0}
        sq_a = (lambda t: t[:, 0]) if cfg.c_adj == 1 else (lambda t: t)
        sq_n = (lambda t: t[..., 0]) if cfg.c_node == 1 else (lambda t: t)
        adjs, nodes = sq_a(xa.cpu()), sq_n(xn.cpu())
        if flag_interim_adjs:
            nodes_ls = torch.stack([sq_n(t) for t in snaps_n])
            if flag_adj_multi_channel:
                return adjs, nodes, [None], nodes_ls
            return adjs, nodes, torch.stack([sq_a(t) for t in snaps_a]), nodes_ls
        return adjs, nodes

    @torch.no_grad()
    def sample(self, model, node_flags, init_adjs=None, init_nodes=None,
               sanity_check_gt_adjs=None, sanity_check_gt_nodes=None,
               flag_interim_adjs=False, max_num_interim_adjs=None, flag_use_double=False,
               flag_node_multi_channel=False, flag_adj_multi_channel=False,
               num_node_chan=150, num_edge_chan=51, churn_noise=None, coins=None, seed=None, return_device=False):
        """See NodeAdjEDMSampler.sample (edm.py:291).  Extra keyword-only knobs (not in the reference):
        `churn_noise=(adj [T,B,..], node [T,B,..])` and `coins` replay recorded randomness (parity tests);
        `seed` seeds the on-device Philox streams (default self.seed; the reference seeds torch per rank,
        arg_parser.py:293-294 -- set `sampler.seed = base_seed + rank`); `return_device=True` skips the final `.cpu()`."""
        if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
            model = model.module
        if flag_use_double:
            # What the reference does with this kwarg (edm.py:320-323, :342-344, :378-380): the loop's state and time steps become float64
            # and the denoiser's output is cast up.  With a real (fp32) network the reference FAILS in its first preconditioned call --
            # the float64 state meets float32 weights ("mat1 and mat2 must have the same dtype, but got Double and Float"; checked by
            # running the reference) -- so the only form that works there is the sanity check, where the denoiser is
            # bypassed (edm.py:372-377).  Mirrored: the same RuntimeError without the ground truth, the float64 loop with it.
            if sanity_check_gt_adjs is None or sanity_check_gt_nodes is None:
                raise RuntimeError("flag_use_double=True: mat1 and mat2 must have the same dtype, but got Double and Float "
                                   "(the reference's float64 sampler state cannot be fed to its float32 network either; "
                                   "only the sanity-check form, which bypasses the network, runs in float64)")
            return self._sample_sanity_double(model, node_flags, init_adjs, init_nodes, sanity_check_gt_adjs, sanity_check_gt_nodes,
                                              flag_interim_adjs, max_num_interim_adjs, flag_adj_multi_channel, churn_noise, seed)
        if not isinstance(model, NodeAdjPrecondHip):
            raise TypeError("NodeAdjEDMSamplerHip needs the NodeAdjPrecondHip network returned by build_network()")
        net = model.model
        h = net._ensure_handle()
        cfg = net.config
        assert num_node_chan == cfg.c_node and num_edge_chan == cfg.c_adj, "channel counts do not match the network"
        B, n, T = node_flags.shape[0], cfg.max_node_num, self.num_steps
        dev = net._dev
        fl = node_flags.to(device=dev).to(torch.uint8).contiguous()

        def prep(x, shape):
            return None if x is None else x.to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
        sa, sn = (B, cfg.c_adj, n, n), (B, n, cfg.c_node)
        if init_adjs is None or init_nodes is None:
            init_adjs = init_nodes = None  # both are redrawn together (edm.py:325-329)
        ia, inn = prep(init_adjs, sa), prep(init_nodes, sn)
        ga, gn = prep(sanity_check_gt_adjs, sa), prep(sanity_check_gt_nodes, sn)
        na = nn_ = None
        if churn_noise is not None:
            na, nn_ = prep(churn_noise[0], (T,) + sa), prep(churn_noise[1], (T,) + sn)
        n_calls = T if self.solver == "euler" else 2 * T - 1
        if coins is None:
            coins = self.draw_coins(n_calls) if ga is None else np.zeros(n_calls, np.uint8)
        coins = np.ascontiguousarray(coins, dtype=np.uint8)
        assert coins.size >= n_calls
        # snapshot schedule (edm.py:333-337, :429-432)
        snap_steps = None
        snap_a = snap_n = None
        if flag_interim_adjs:
            if max_num_interim_adjs is None:
                ts = np.arange(T)
            else:
                ts = np.linspace(0, T, max_num_interim_adjs).astype(int).clip(max=T - 1)
            snap_steps = np.ascontiguousarray(np.unique(ts), dtype=np.int32)
            snap_n = torch.empty((len(snap_steps),) + sn, dtype=torch.float32, device=dev)
            if not flag_adj_multi_channel:
                snap_a = torch.empty((len(snap_steps),) + sa, dtype=torch.float32, device=dev)
        oa = torch.empty(sa, dtype=torch.float32, device=dev)
        on = torch.empty(sn, dtype=torch.float32, device=dev)
        stats = _lib.DsgSampleStats()
        scfg = self._cfg()
        st = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        seed_v = self.seed if seed is None else int(seed)
        if flag_interim_adjs and ia is None:
            # The reference's own call (sampler_node_adj.py:166-177) passes init_adjs=None with flag_interim_adjs=True and
            # reads slot 0 of the snapshot list as the UNSCALED initial sample (edm.py:326-337).  Draw the library's
            # init stream (noise stream 0) into caller-visible buffers first; handing them back in is bit-identical
            # to letting dsg_sample draw them itself.
            ia = torch.empty(sa, dtype=torch.float32, device=dev)
            inn = torch.empty(sn, dtype=torch.float32, device=dev)
            h.check(h.L.dsg_gen_noise(h.raw, B, p(fl), C.c_uint64(seed_v), 0, p(ia), p(inn), C.c_void_p(st)), "dsg_gen_noise")
        h.check(h.L.dsg_sample(h.raw, C.byref(scfg), B, p(fl), p(ia), p(inn), p(na), p(nn_),
                               C.c_void_p(coins.ctypes.data), C.c_uint64(seed_v),
                               p(ga), p(gn),
                               C.c_void_p(0 if snap_steps is None else snap_steps.ctypes.data),
                               0 if snap_steps is None else len(snap_steps), p(snap_a), p(snap_n),
                               p(oa), p(on), C.byref(stats), C.c_void_p(st)), "dsg_sample")
        self.last_stats = {"precond_calls": stats.precond_calls, "net_forwards": stats.net_forwards,
                           "graph_replays": stats.graph_replays}
        logging.info("Done with EDM-NodeAdj MCMC (HIP).")
        if cfg.c_adj == 1:
            oa = oa[:, 0]
        if cfg.c_node == 1:
            on = on[..., 0]
        if return_device:
            return oa, on
        adjs, nodes = oa.cpu(), on.cpu()
        if flag_interim_adjs:
            # torch.stack(nodes_ls) of edm.py:441-443: slot 0 = unscaled init, then one entry per snapshot step
            sq_n = (lambda t: t[..., 0]) if cfg.c_node == 1 else (lambda t: t)
            sq_a = (lambda t: t[:, :, 0]) if cfg.c_adj == 1 else (lambda t: t)
            nodes_ls = sq_n(torch.cat([inn.cpu().reshape((1,) + sn), snap_n.cpu()]))
            if flag_adj_multi_channel:
                return adjs, nodes, [None], nodes_ls
            adjs_ls = sq_a(torch.cat([ia.cpu().reshape((1,) + sa), snap_a.cpu()]))
            return adjs, nodes, adjs_ls, nodes_ls
        return adjs, nodes
