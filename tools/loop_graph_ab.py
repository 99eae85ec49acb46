#!/usr/bin/env python3
"""A/B of the reverse-loop capture on the launch-bound shape (BASELINE configs[0]'s network on the GPU: tiny, B=4, T=1000) and
on the headline shape: option loop_graph = 1 (whole step bodies replayed) vs 0 (round-1 scheme: forward-only graph) vs eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffusesg_amd import synth, weights
from diffusesg_amd.model import build_network
from diffusesg_amd.sampler import NodeAdjEDMSamplerHip

def run(name, B, T, valid, reps):
    cfg = synth.CONFIGS[name]()
    net = build_network(cfg, weights.synth_state_dict(cfg, 0), device="cuda")
    h = net.model._ensure_handle()
    flags = torch.from_numpy(weights.synth_flags(B, cfg.max_node_num, valid)).cuda()
    res = {}
    for label, use_graph, loop in (("eager", False, 0), ("forward-graph (round 1)", True, 0), ("step-graph", True, 1)):
        h.set_option("loop_graph", loop)
        smp = NodeAdjEDMSamplerHip(num_steps=T, self_condition=True, dev="cuda", use_graph=use_graph)
        np.random.seed(1); smp.sample(net, flags, num_node_chan=cfg.c_node, num_edge_chan=cfg.c_adj, seed=1, return_device=True)
        torch.cuda.synchronize(); ts = []
        for r in range(reps):
            np.random.seed(2); t0 = time.perf_counter()
            oa, on = smp.sample(net, flags, num_node_chan=cfg.c_node, num_edge_chan=cfg.c_adj, seed=2, return_device=True)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[label] = (min(ts), oa.clone())
        print(f"{name} B={B} T={T} {label:26s}: {min(ts)*1e3:9.1f} ms per sample()  = {B/min(ts):8.2f} graphs/s  ({smp.last_stats})", flush=True)
    ref = res["eager"][1]
    for k, (_, o) in res.items():
        print(f"   {k}: max |diff| vs eager {float((o - ref).abs().max()):.3e} (max |out| {float(ref.abs().max()):.3f})", flush=True)
    for k, (_, o) in res.items():
        assert torch.equal(o, ref), f"{name} B={B} T={T}: {k} differs from eager"
    print(f"   {name} B={B} T={T}: all three modes bit-identical", flush=True)

if __name__ == "__main__":
    run("tiny", 4, 1000, 8, 3)
    run("vg", 64, 40, 30, 2)
