"""A few hundred training iterations on a fixed synthetic batch (tiny network): the loss of the device training iteration must fall.
Not a benchmark and not a convergence claim -- an end-to-end sanity run of objective -> forward -> backward -> clip + Adam -> EMA."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from diffusesg_amd import synth as Y, weights as W
from diffusesg_amd.model import build_network
from diffusesg_amd.train import NodeAdjEDMObjectiveGeneratorHip, NodeAdjRainbowLossHip, AdamHip, EMAHip, train_one_iteration, eval_loss_step
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
name, B, iters = (sys.argv[1] if len(sys.argv) > 1 else "tiny"), 16, int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg, flags, ca, cn, *_ = Y.train_case(name, B=B)
model = build_network(cfg, W.synth_state_dict(cfg, 0), device="cuda")
gen = NodeAdjEDMObjectiveGeneratorHip(precond="edm", sigma_dist="edm", other_params=None, dev="cuda", symmetric_noise=False)
lf = NodeAdjRainbowLossHip(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
opt, emas = AdamHip(model, lr=5e-4), [EMAHip(model, beta=0.99)]
np.random.seed(0)
def held_out():   # fixed sigmas / noise: the same test-loss batch before and after
    np.random.seed(123)
    l, *_ = eval_loss_step(model, gen, lf, T(ca), T(cn), T(flags), mode="test", seed=777)
    return float(l)
print(f"{name}: B={B}, {iters} iterations, lr 5e-4; test loss (fixed draws) before: {held_out():.4f}", flush=True)
win, t0 = [], time.time()
for it in range(iters):
    loss, *_ = train_one_iteration(model, gen, lf, opt, emas, T(ca), T(cn), T(flags), iou_loss_weight=0.0)
    win.append(float(loss))
    if (it + 1) % 50 == 0:
        print(f"  iterations {it - 48:4d}-{it + 1:4d}: mean train loss {np.mean(win):.4f}  (|grad| {opt.last_total_norm:.3f})", flush=True)
        win = []
print(f"test loss (same fixed draws) after: {held_out():.4f};  {iters / (time.time() - t0):.1f} iterations/s", flush=True)
