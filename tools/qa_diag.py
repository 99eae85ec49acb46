"""GPU box: error pattern of the fused qkv + attention kernel against the fp64 reference (by head, position tile, head dim, sample)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffusesg_amd import lib as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_bx_kernels import _window_tokens, _bf, _p
lib = L.load()
B, res, ws, shift, heads = [int(v) for v in os.environ.get("QA_CASE", "3,10,10,0,12").split(",")]
Cc, T, Wt = 32 * heads, res * res, ws * ws
Wp = (Wt + 31) // 32 * 32
nW = (res // ws) ** 2
nWt = nW if shift > 0 else 1
gen = torch.Generator(device="cuda").manual_seed(5)
xn = torch.randn(B * T, Cc, device="cuda", generator=gen)
W = torch.randn(3 * Cc, Cc, device="cuda", generator=gen) / Cc ** 0.5
W[:Cc] *= 0.25
bqkv = torch.randn(3 * Cc, device="cuda", generator=gen) * 0.2
bias = torch.randn(nWt, heads, Wp, Wp, device="cuda", generator=gen) * float(os.environ.get("QA_BIAS", "1.5"))
bias[:, :, Wt:, :] = -1.0e30
out = torch.full((B * T, Cc), float("nan"), device="cuda")
rc = lib.dsg_debug_qkv_attn_bx(B, res, ws, shift, heads, _p(xn), _p(W), _p(bqkv), _p(bias.contiguous()), _p(out), 0, None, None)
assert rc == 0
qkv = _bf((_bf(xn).double() @ _bf(W).double().t() + bqkv.double()).float())
tok = torch.from_numpy(_window_tokens(res, ws, shift)).cuda()
x = qkv.double().view(B, T, 3, heads, 32)
xw = x[:, tok]
q, k, v = [xw[..., i, :, :].permute(0, 1, 3, 2, 4) for i in range(3)]
bt = bias[:, :, :Wt, :Wt].clamp(min=-60000.0).half().double()
def ref_of(q, k, v, bt):
    s = torch.einsum("bwhkd,bwhqd->bwhkq", k, q) + (bt[None] if shift > 0 else bt[None].expand(1, nW, -1, -1, -1))
    e = torch.exp2(s - s.max(dim=3, keepdim=True).values)
    return torch.einsum("bwhkq,bwhkd->bwhqd", _bf(e.float()).double(), v) / e.sum(dim=3)[..., None]   # [B, nW, heads, Wt, 32]
o = ref_of(q, k, v, bt)
got = out.double().view(B, T, heads, 32)[:, tok].permute(0, 1, 3, 2, 4)   # [B, nW, heads, Wt, 32]
err = (got - o).abs()
print("scale", float(o.abs().max()), "max err", float(err.max()))
print("by sample", [f"{float(err[b].max()):.3f}" for b in range(B)])
print("by head", [f"{float(err[:, :, h].max()):.3f}" for h in range(heads)])
print("by pos tile", [f"{float(err[:, :, :, 32 * t:32 * t + 32].max()):.3f}" for t in range((Wt + 31) // 32)])
print("by d", [f"{float(err[..., d].max()):.2f}" for d in range(32)])
# hypotheses
for name, r in (("no bias", ref_of(q, k, v, bt * 0)), ("bias transposed", ref_of(q, k, v, bt.transpose(-1, -2))), ("v = mean", None)):
    if r is None: continue
    print(name, float((got - r).abs().max()))
print("uniform attention (mean of v)", float((got - v.mean(dim=3, keepdim=True)).abs().max()))
print("got[0,0,0,:4,:8]\n", got[0, 0, 0, :4, :8].cpu().numpy().round(3), "\nref\n", o[0, 0, 0, :4, :8].cpu().numpy().round(3))
# one-hot attention: bias[key = (q + 7) % Wt][q] = +60 -> out[q] must be v[(q + 7) % Wt]; which key does the kernel pick?
bias2 = torch.zeros(nWt, heads, Wp, Wp, device="cuda")
for qq in range(Wt):
    bias2[:, :, (qq + 7) % Wt, qq] = 60.0
bias2[:, :, Wt:, :] = -1.0e30
out2 = torch.full((B * T, Cc), float("nan"), device="cuda")
rc = lib.dsg_debug_qkv_attn_bx(B, res, ws, shift, heads, _p(xn), _p(W), _p(bqkv), _p(bias2.contiguous()), _p(out2), 0, None, None)
got2 = out2.double().view(B, T, heads, 32)[:, tok].permute(0, 1, 3, 2, 4)[0, 0, 0]    # [Wt, 32]
v0 = v[0, 0, 0]                                                                       # [Wt, 32]
d2 = (got2[:, None, :] - v0[None, :, :]).abs().max(dim=2).values                       # [query, key]
best = d2.argmin(dim=1).cpu().numpy()
print("picked key - expected key per query (0 = right):", ((best - (np.arange(Wt) + 7) % Wt)).tolist())
print("residual of the best match:", [round(float(d2[i, best[i]]), 3) for i in range(0, Wt, 9)])
