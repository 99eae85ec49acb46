import numpy as np, torch, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from util import load
from diffusesg_amd import synth as Y, weights as W
from diffusesg_amd.model import build_network
from diffusesg_amd.train import NodeAdjEDMObjectiveGeneratorHip, NodeAdjRainbowLossHip, train_step_grads
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = load("train_backward.npz")
cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin = Y.train_case("tiny")
model = build_network(cfg, W.synth_state_dict(cfg, 0), device="cuda")
gen = NodeAdjEDMObjectiveGeneratorHip(precond="edm", sigma_dist="edm", other_params=None, dev="cuda", symmetric_noise=False)
lf = NodeAdjRainbowLossHip(edge_loss_weight=1.0, node_loss_weight=1.0, flag_reweight=False, objective="edm")
na, nx, cond, ta, tx, (c_skip, c_out, c_in, c_noise, sigmas, weights) = gen.get_input_output(T(clean_adj), T(clean_node), T(flags), rnd_sigma=T(rnd), noise=(T(eps_adj), T(eps_node)))
real=np.random.rand; np.random.rand=lambda: coin
import time; t0=time.time()
oa,on,la,ln,grads = train_step_grads(model, lf, na, nx, T(flags), sigmas, T(clean_adj), T(clean_node), weights, iou_loss_weight=1.0)
torch.cuda.synchronize(); print('step time %.3f s'%(time.time()-t0), 'coin', coin)
np.random.rand=real
errs=[]
for k,rn in zip([str(x) for x in g["tiny_gparam_names"]], g["tiny_gparam_norms"]):
    key=k[6:]; mine=grads[key].cpu().numpy().reshape(-1); ref=g[f"tiny_gparam/{k}"]; st=max(1,-(-mine.size//1024))
    e=np.abs(mine[::st]-ref).max()/max(np.abs(ref).max(),1e-12); n=np.sqrt((mine.astype(np.float64)**2).sum())
    errs.append((e,abs(n-rn)/rn,key))
errs.sort(reverse=True)
for e in errs[:6]: print('%.2e norm rel %.2e %s'%e)
print('median sample err %.2e'%np.median([e[0] for e in errs]), 'n', len(errs))
print('loss', float(la.mean()+ln.mean()), float(g['tiny_loss']))
