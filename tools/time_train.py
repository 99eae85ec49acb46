"""Wall time of training iterations at the VG shape -- for the record in DESIGN.md.  The self-conditioning coin (precond.py:90) decides
whether an iteration carries the extra no-grad pass, so the two kinds are timed separately (coin forced) and together (NumPy's draw).
TT_BATCHES=8,32,64  TT_ITERS=12  TT_PHASES=1 (per-phase host timers with a synchronize after each phase)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusesg_amd import synth as Y, weights as W, train as TR
from diffusesg_amd.model import build_network
from diffusesg_amd.train import NodeAdjEDMObjectiveGeneratorHip, NodeAdjRainbowLossHip, AdamHip, EMAHip, train_one_iteration
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
FLOP_FWD = 13.3e9   # per graph, VG network


class _Coin:
    """np.random.rand() replaced by a fixed draw for the duration of a block (the trainer's coin is NumPy's global generator)"""
    def __init__(self, v): self.v = v
    def __enter__(self):
        self.orig = np.random.rand
        if self.v is not None: np.random.rand = lambda *a: self.v
    def __exit__(self, *a): np.random.rand = self.orig


for B in [int(b) for b in os.environ.get("TT_BATCHES", "8,32,64").split(",")]:
    cfg, flags, ca, cn, rnd, ea, en, coin = Y.train_case("vg", B=B)
    model = build_network(cfg, W.synth_state_dict(cfg, 0), device="cuda")
    gen = NodeAdjEDMObjectiveGeneratorHip(precond="edm", sigma_dist="edm", other_params=None, dev="cuda", symmetric_noise=False)
    lf = NodeAdjRainbowLossHip(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
    opt, emas = AdamHip(model), [EMAHip(model, beta=0.9999)]
    a, x, f = T(ca), T(cn), T(flags)
    n_it = int(os.environ.get("TT_ITERS", "12"))
    for label, forced in (("no self-cond pass", 0.9), ("with self-cond pass", 0.1), ("coin drawn", None)):
        ts = []
        with _Coin(forced):
            for it in range(n_it + 2):
                torch.cuda.synchronize(); t0 = time.time()
                loss, *_ = train_one_iteration(model, gen, lf, opt, emas, a, x, f, iou_loss_weight=1.0, iou_loss_type="giou")
                torch.cuda.synchronize(); ts.append(time.time() - t0)
        ts = ts[2:]
        k = 3 if forced == 0.9 else (4 if forced == 0.1 else 3.5)
        print(f"VG B={B} [{label}]: iteration mean {np.mean(ts):.4f} s, min {min(ts):.4f} s (loss {float(loss):.3f}); "
              f"model FLOPs (x{k} forward) ~ {k * FLOP_FWD * B / np.mean(ts) / 1e12:.1f} TFLOP/s", flush=True)
    if os.environ.get("TT_PHASES"):
        import diffusesg_amd.train as tr
        names = ["get_input_output", "train_step_grads", "all_reduce", "adam", "ema"]
        acc = dict.fromkeys(names, 0.0)
        with _Coin(0.9):
            for it in range(n_it):
                def tick(name, t0):
                    torch.cuda.synchronize(); acc[name] += time.time() - t0; return time.time()
                t0 = time.time()
                io = gen.get_input_output(a, x, f)
                t0 = tick("get_input_output", t0)
                na, nx, _, ta, tx, (c_skip, c_out, c_in, c_noise, sigmas, weights) = io
                out = tr.train_step_grads(model, lf, na, nx, f, sigmas, ta, tx, weights, iou_loss_weight=1.0, iou_loss_type="giou")
                t0 = tick("train_step_grads", t0)
                from diffusesg_amd import dist as _d
                _d.all_reduce_mean(out[4]); t0 = tick("all_reduce", t0)
                opt.step(out[4]); t0 = tick("adam", t0)
                [e.update() for e in emas]; t0 = tick("ema", t0)
        print("  phases (ms / iteration, synchronised): " + ", ".join(f"{k} {1e3 * v / n_it:.2f}" for k, v in acc.items()), flush=True)
