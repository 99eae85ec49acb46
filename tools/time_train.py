"""Wall time of one training iteration (correctness-first kernels) at the VG shape -- for the record in DESIGN.md."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusesg_amd import synth as Y, weights as W
from diffusesg_amd.model import build_network
from diffusesg_amd.train import NodeAdjEDMObjectiveGeneratorHip, NodeAdjRainbowLossHip, AdamHip, EMAHip, train_one_iteration
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for B in [int(b) for b in os.environ.get("TT_BATCHES", "8,32,64").split(",")]:
    cfg, flags, ca, cn, rnd, ea, en, coin = Y.train_case("vg", B=B)
    model = build_network(cfg, W.synth_state_dict(cfg, 0), device="cuda")
    gen = NodeAdjEDMObjectiveGeneratorHip(precond="edm", sigma_dist="edm", other_params=None, dev="cuda", symmetric_noise=False)
    lf = NodeAdjRainbowLossHip(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
    opt, emas = AdamHip(model), [EMAHip(model, beta=0.9999)]
    ts = []
    for it in range(int(os.environ.get("TT_ITERS", "4"))):
        torch.cuda.synchronize(); t0 = time.time()
        loss, *_ = train_one_iteration(model, gen, lf, opt, emas, T(ca), T(cn), T(flags), iou_loss_weight=1.0)
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    print(f"VG B={B}: iteration {min(ts):.3f} s (loss {float(loss):.3f}); forward+backward FLOPs ~ {3 * 13.3e9 * B / min(ts) / 1e12:.2f} TFLOP/s", flush=True)
