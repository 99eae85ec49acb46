"""GPU: the kernels of the bf16 block pipeline (csrc/kernels_bx.hip, "gemm_bf16" mode) on their own, against plain PyTorch fp32
references evaluated on the SAME bf16-rounded operands -- so that only the summation order and the final bf16 rounding differ and the
bars can be tight: EVERY element of the output is compared (a wrong lane group or tile row cannot hide behind a sampled check or
behind the end-to-end bf16 tolerance)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


# (M, N, K, act, res, mod, ln): every tile geometry (256x192, 128x384 full row, 512x96), ragged M, half-filled last column tile,
# K that ends inside a chunk, the epilogue pieces in the combinations the forward uses
GEMM_CASES = [
    (51200, 1536, 384, 1, 0, 0, 0),     # fc1 at COCO level 2: GELU, bf16 store
    (51200, 384, 1536, 0, 1, 1, 1),     # fc2 at C = 384: residual, next block's modulate+SiLU, LayerNorm of the row (128x384 tile)
    (20037, 192, 768, 0, 1, 1, 1),      # C = 192 full row on the 256x192 tile, ragged M
    (65541, 96, 384, 0, 1, 1, 1),       # C = 96 full row on the 512x96 tile, ragged M
    (8192, 288, 96, 0, 0, 0, 0),        # qkv at C = 96: three 96-wide tiles
    (4096, 768, 3072, 0, 1, 1, 0),      # fc2 at C = 768: modulate without LayerNorm (row wider than a tile)
    (300, 576, 192, 0, 0, 0, 0),        # fewer rows than one tile
    (2048, 192, 96, 0, 1, 0, 1),        # K = 96 on 64-deep chunks (the last chunk is half valid), LayerNorm without modulate
    (1000, 384, 384, 0, 0, 0, 0),       # plain product, fp32 + bf16 stores
]


@pytest.mark.parametrize("M,N,K,act,res,mod,ln", GEMM_CASES)
def test_gemm_bx_whole_matrix(M, N, K, act, res, mod, ln):
    from diffusesg_amd import lib as L
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(99 + M + N + K)
    A = torch.randn(M, K, device="cuda", generator=gen)
    W = torch.randn(N, K, device="cuda", generator=gen) * (1.0 / K ** 0.5)
    bias = torch.randn(N, device="cuda", generator=gen) * 0.3
    R = torch.randn(M, N, device="cuda", generator=gen) if res else None
    aff = (torch.randn(2 * N, device="cuda", generator=gen) * 0.5) if mod else None
    out_C = torch.full((M, N), float("nan"), device="cuda")
    out_Cb = torch.full((M, N), float("nan"), device="cuda")
    out_C2 = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.dsg_debug_gemm_bx(M, N, K, _p(A), _p(W), _p(bias), _p(R), act, _p(aff), ln, _p(out_C), _p(out_Cb), _p(out_C2), 0, None, None)
    assert rc == 0
    v = _bf(A).double() @ _bf(W).double().t() + bias.double()
    if act:
        v = torch.nn.functional.gelu(v)
    if R is not None:
        v = v + R.double()
    pre = v.clone()
    if aff is not None:
        v = torch.nn.functional.silu(aff[N:].double() + v * (1.0 + aff[:N].double()))
    scale = float(v.abs().max())
    d = float((out_C.double() - v).abs().max()) / scale
    assert d <= 1e-4, f"fp32 store: {d:.2e}"
    ref_b = torch.nn.functional.layer_norm(v, (N,), eps=1e-5) if ln else v

    def check_bf16(got, ref, what):
        assert torch.isfinite(got).all(), what
        err = (got.double() - ref).abs() - 2.0 ** -8 * ref.abs()       # half a bf16 ulp of rounding + slack below
        s = float(ref.abs().max())
        worst = float(err.max()) / s
        assert worst <= 3e-4, f"{what}: {worst:.2e} beyond bf16 rounding"
    check_bf16(out_Cb, ref_b, "bf16 store" + (" (LayerNorm)" if ln else ""))
    check_bf16(out_C2, pre, "bf16 pre-modulation copy")


def _window_tokens(res, ws, shift):
    """[nW, ws*ws] token index of every window position (partition after the cyclic shift; diffusesg.py:28-57, :246-256)"""
    nwr = res // ws
    idx = np.zeros((nwr * nwr, ws * ws), np.int64)
    for wi in range(nwr):
        for wj in range(nwr):
            for p in range(ws * ws):
                ti = (wi * ws + p // ws + shift) % res
                tj = (wj * ws + p % ws + shift) % res
                idx[wi * nwr + wj, p] = ti * res + tj
    return idx


@pytest.mark.parametrize("B,res,ws,shift,heads", [(3, 10, 10, 0, 12), (5, 20, 10, 5, 6), (2, 40, 10, 0, 3), (4, 16, 8, 4, 12), (3, 8, 8, 0, 24),
                                                  (2, 16, 4, 2, 3), (7, 4, 4, 0, 6)])
def test_attn_bx_vs_torch(B, res, ws, shift, heads):
    """softmax(q k^T + bias) v per (window, head) on bf16 q, k, v: 100-token windows (padded to 128 positions, -1e30 bias in the padded
    key slots), 64- and 16-token windows, cyclic shift with a per-window bias table, against fp64 on the bf16-rounded inputs (P is
    rounded to bf16 before the second product, as the kernel does; the row sum is taken before that rounding)."""
    from diffusesg_amd import lib as L
    lib = L.load()
    Cc, T, Wt = 32 * heads, res * res, ws * ws
    Wp = (Wt + 31) // 32 * 32
    nW = (res // ws) ** 2
    nWt = nW if shift > 0 else 1
    gen = torch.Generator(device="cuda").manual_seed(7 + B + res + ws)
    qkv = torch.randn(B * T, 3 * Cc, device="cuda", generator=gen)
    qkv[:, :Cc] *= 0.25 * 1.4426950408889634          # q arrives pre-scaled by d^-1/2 log2(e)
    bias = torch.randn(nWt, heads, Wp, Wp, device="cuda", generator=gen) * 1.5      # key-major [key][query], log2(e)-scaled
    if shift > 0:
        bias[torch.rand(nWt, 1, Wp, Wp, device="cuda", generator=gen).expand(-1, heads, -1, -1) < 0.2] -= 144.0   # the shift mask (-100 log2 e)
    bias[:, :, Wt:, :] = -1.0e30
    out = torch.full((B * T, Cc), float("nan"), device="cuda")
    rc = lib.dsg_debug_attn_bx(B, res, ws, shift, heads, _p(qkv), _p(bias.contiguous()), _p(out), 0, None, None)
    assert rc == 0
    tok = torch.from_numpy(_window_tokens(res, ws, shift)).cuda()                   # [nW, Wt]
    x = _bf(qkv).double().view(B, T, 3, heads, 32)
    xw = x[:, tok]                                                                   # [B, nW, Wt, 3, heads, 32]
    q, k, v = xw[..., 0, :, :].permute(0, 1, 3, 2, 4), xw[..., 1, :, :].permute(0, 1, 3, 2, 4), xw[..., 2, :, :].permute(0, 1, 3, 2, 4)
    bt = bias[:, :, :Wt, :Wt].double()                                               # [nWt, heads, key, query]
    s = torch.einsum("bwhkd,bwhqd->bwhkq", k, q) + (bt[None] if shift > 0 else bt[None].expand(1, nW, -1, -1, -1))
    e = torch.exp2(s - s.max(dim=3, keepdim=True).values)
    o = torch.einsum("bwhkq,bwhkd->bwhqd", _bf(e.float()).double(), v) / e.sum(dim=3)[..., None]      # [B, nW, heads, Wt, 32]
    ref = torch.zeros(B, T, heads, 32, device="cuda", dtype=torch.float64)
    ref[:, tok] = o.permute(0, 1, 3, 2, 4)
    ref = ref.view(B * T, Cc)
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    err = (out.double() - ref).abs() - 2.0 ** -8 * ref.abs()
    worst = float(err.max()) / scale
    assert worst <= 2e-3, f"attention: {worst:.2e} beyond bf16 rounding of the output"


@pytest.mark.parametrize("B,res,ws,shift,heads", [(3, 10, 10, 0, 12), (5, 20, 10, 5, 6), (2, 40, 10, 0, 3), (4, 16, 8, 4, 12), (3, 8, 8, 0, 24),
                                                  (5, 16, 8, 0, 3), (2, 16, 4, 2, 3), (7, 4, 4, 0, 6), (3, 10, 5, 2, 3),
                                                  (48, 20, 10, 5, 6), (41, 16, 8, 4, 12),    # > 512 (window group, head) tiles
                                                  (3, 10, 10, 1000, 12), (5, 20, 10, 1005, 6), (2, 40, 10, 1000, 3),   # shift + 1000: the block-per-head kernel at 10 x 10
                                                  (1, 10, 10, 0, 3), (7, 10, 10, 0, 6), (3, 20, 10, 5, 3)])            # units not a multiple of 4, blocks across two windows
def test_qkv_attn_bx_vs_torch(B, res, ws, shift, heads):
    """QKV projection + window attention in one kernel against fp64 on the bf16-rounded operands: q, k, v are formed in fp32, rounded
    to bf16 (as the kernel hands them to the second and third product) and then follow test_attn_bx_vs_torch's reference.  Odd unit
    counts (the last block is partly empty), two / four windows per block (64- / 16- / 25-token windows), shifted masks, C = 96 (K ends
    inside a 64-deep chunk) up to 768."""
    from diffusesg_amd import lib as L
    lib = L.load()
    variant, shift = divmod(shift, 1000)               # 10 x 10 windows: 0 the wave-per-unit kernel, 1 the block-per-head kernel
    Cc, T, Wt = 32 * heads, res * res, ws * ws
    Wp = (Wt + 31) // 32 * 32
    nW = (res // ws) ** 2
    nWt = nW if shift > 0 else 1
    gen = torch.Generator(device="cuda").manual_seed(11 + B + res + ws + heads)
    xn = torch.randn(B * T, Cc, device="cuda", generator=gen)
    W = torch.randn(3 * Cc, Cc, device="cuda", generator=gen) / Cc ** 0.5
    W[:Cc] *= 0.25 * 1.4426950408889634               # q rows pre-scaled by d^-1/2 log2(e)
    bqkv = torch.randn(3 * Cc, device="cuda", generator=gen) * 0.2
    bias = torch.randn(nWt, heads, Wp, Wp, device="cuda", generator=gen) * 1.5
    if shift > 0:
        bias[torch.rand(nWt, 1, Wp, Wp, device="cuda", generator=gen).expand(-1, heads, -1, -1) < 0.2] -= 144.0
    bias[:, :, Wt:, :] = -1.0e30
    out = torch.full((B * T, Cc), float("nan"), device="cuda")
    rc = lib.dsg_debug_qkv_attn_bx(B, res, ws, shift + 1000 * variant, heads, _p(xn), _p(W), _p(bqkv), _p(bias.contiguous()), _p(out), 0, None, None)
    assert rc == 0
    qkv = _bf((_bf(xn).double() @ _bf(W).double().t() + bqkv.double()).float())
    tok = torch.from_numpy(_window_tokens(res, ws, shift)).cuda()
    x = qkv.double().view(B, T, 3, heads, 32)
    xw = x[:, tok]
    q, k, v = xw[..., 0, :, :].permute(0, 1, 3, 2, 4), xw[..., 1, :, :].permute(0, 1, 3, 2, 4), xw[..., 2, :, :].permute(0, 1, 3, 2, 4)
    # the block-per-head kernel reads the bias tile as fp16, the wave-per-unit one (10 x 10 windows, variant 0) as fp32
    bt = bias[:, :, :Wt, :Wt].double() if (ws == 10 and variant == 0) else bias[:, :, :Wt, :Wt].clamp(min=-60000.0).half().double()
    s = torch.einsum("bwhkd,bwhqd->bwhkq", k, q) + (bt[None] if shift > 0 else bt[None].expand(1, nW, -1, -1, -1))
    e = torch.exp2(s - s.max(dim=3, keepdim=True).values)
    o = torch.einsum("bwhkq,bwhkd->bwhqd", _bf(e.float()).double(), v) / e.sum(dim=3)[..., None]
    ref = torch.zeros(B, T, heads, 32, device="cuda", dtype=torch.float64)
    ref[:, tok] = o.permute(0, 1, 3, 2, 4)
    ref = ref.view(B * T, Cc)
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    err = (out.double() - ref).abs() - 2.0 ** -8 * ref.abs()
    worst = float(err.max()) / scale
    # (a q / k / v value on a bf16 rounding boundary may round the other way than in fp64: 2^-9 relative on one of 32 + 100 terms)
    assert worst <= 4e-3, f"qkv + attention: {worst:.2e} beyond bf16 rounding of the output"


@pytest.mark.parametrize("M,C,mod,out_mode", [(51200, 384, 1, 1), (20037, 192, 1, 1), (65541, 96, 0, 2), (300, 96, 1, 1), (4096, 384, 0, 0),
                                               (1000, 192, 0, 2), (20037, 384, 0, 2), (51200, 384, 1, 1 + 16), (4100, 384, 0, 2 + 16),
                                               (51200, 384, 1, 1 + 32), (20037, 384, 0, 2 + 32), (130, 384, 1, 1),
                                               (51200, 384, 1, 1 + 64), (20037, 384, 0, 2 + 64), (4096, 384, 0, 0 + 64), (130, 384, 2, 1 + 64)])
def test_mlp_bx_whole_matrix(M, C, mod, out_mode):
    """the fused fc1 -> GELU -> fc2 -> + residual -> [modulate] -> [LayerNorm | copy] kernel at its three widths against fp64 on the
    bf16-rounded operands, with the hidden activations rounded to bf16 between the two products as the kernel does"""
    from diffusesg_amd import lib as L
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(5 + M + C)
    xn = torch.randn(M, C, device="cuda", generator=gen)
    x = torch.randn(M, C, device="cuda", generator=gen)
    W1 = torch.randn(4 * C, C, device="cuda", generator=gen) / C ** 0.5
    b1 = torch.randn(4 * C, device="cuda", generator=gen) * 0.3
    W2 = torch.randn(C, 4 * C, device="cuda", generator=gen) / (4 * C) ** 0.5
    b2 = torch.randn(C, device="cuda", generator=gen) * 0.3
    aff = (torch.randn(2 * C, device="cuda", generator=gen) * 0.5) if mod else None
    x_io = x.clone()
    out_xn = torch.full((M, C), float("nan"), device="cuda")
    # C = 384: the eight-wave LDS-DMA kernel on pre-arranged weight images (round 4); out_mode + 32 selects round 3's eight-wave kernel
    # (register-staged weights), + 16 the one-wave-per-SIMD kernel
    rc = lib.dsg_debug_mlp_bx(M, C, _p(xn), _p(x_io), _p(W1), _p(b1), _p(W2), _p(b2), _p(aff), out_mode, _p(out_xn) if out_mode else None, 0, None, None)
    assert rc == 0
    out_mode &= 15
    hid = torch.nn.functional.gelu(_bf(xn).double() @ _bf(W1).double().t() + b1.double())
    v = _bf(hid.float()).double() @ _bf(W2).double().t() + b2.double() + x.double()
    if aff is not None:
        v = torch.nn.functional.silu(aff[C:].double() + v * (1.0 + aff[:C].double()))
    scale = float(v.abs().max())
    d = float((x_io.double() - v).abs().max()) / scale
    # (a hidden value that sits on a bf16 rounding boundary may round the other way than in fp64: 2^-9 of one of 4C terms)
    assert d <= 5e-4, f"fp32 residual stream: {d:.2e}"
    if out_mode:
        ref = torch.nn.functional.layer_norm(v, (C,), eps=1e-5) if out_mode == 1 else v
        assert torch.isfinite(out_xn).all()
        err = (out_xn.double() - ref).abs() - 2.0 ** -8 * ref.abs()
        worst = float(err.max()) / float(ref.abs().max())
        assert worst <= 1e-3, f"bf16 output: {worst:.2e} beyond bf16 rounding"


@pytest.mark.parametrize("M,C,mod,out_mode", [(20037, 192, 1, 1), (65541, 96, 0, 2), (300, 96, 1, 1), (4096, 192, 0, 0), (51200, 96, 1, 1),
                                               (51200, 384, 1, 1), (20037, 384, 0, 2), (300, 384, 0, 0),
                                               (51200, 384, 1, 1 + 32), (20037, 384, 0, 2 + 32), (129, 384, 1, 1),
                                               (51200, 384, 1, 1 + 64), (20037, 384, 0, 2 + 64), (300, 384, 0, 0 + 64), (129, 384, 1, 1 + 64),
                                               (65541, 96, 0, 2 + 128), (4096, 96, 0, 0 + 128), (129, 96, 0, 1 + 128), (100003, 96, 0, 1 + 128), (31, 96, 0, 2 + 128),   # + 128: C = 96 on the LDS-resident kernel
                                               (4096, 96, 0, 0), (129, 96, 0, 1)])
def test_projmlp_bx_whole_matrix(M, C, mod, out_mode):
    """proj + residual + LayerNorm-2 + fc1 + GELU + fc2 + residual [+ modulate] [+ LayerNorm | copy] in one kernel against fp64 on the
    bf16-rounded operands: x1 = x + att Wp^T + bp stays in the accumulators (fp32), its LayerNorm is rounded to bf16 as fc1's operand,
    the hidden activations are rounded to bf16 between the two products"""
    from diffusesg_amd import lib as L
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(17 + M + C)
    att = torch.randn(M, C, device="cuda", generator=gen)
    x = torch.randn(M, C, device="cuda", generator=gen)
    Wp = torch.randn(C, C, device="cuda", generator=gen) / C ** 0.5
    bp = torch.randn(C, device="cuda", generator=gen) * 0.3
    W1 = torch.randn(4 * C, C, device="cuda", generator=gen) / C ** 0.5
    b1 = torch.randn(4 * C, device="cuda", generator=gen) * 0.3
    W2 = torch.randn(C, 4 * C, device="cuda", generator=gen) / (4 * C) ** 0.5
    b2 = torch.randn(C, device="cuda", generator=gen) * 0.3
    aff = (torch.randn(2 * C, device="cuda", generator=gen) * 0.5) if mod else None
    x_io = x.clone()
    out_xn = torch.full((M, C), float("nan"), device="cuda")
    # (C = 384: out_mode + 32 selects round 3's eight-wave kernel instead of the LDS-DMA one)
    rc = lib.dsg_debug_projmlp_bx(M, C, _p(att), _p(x_io), _p(Wp), _p(bp), _p(W1), _p(b1), _p(W2), _p(b2), _p(aff), out_mode,
                                  _p(out_xn) if out_mode else None, 0, None, None)
    assert rc == 0
    out_mode &= 15
    x1 = x.double() + _bf(att).double() @ _bf(Wp).double().t() + bp.double()
    xn = _bf(torch.nn.functional.layer_norm(x1, (C,), eps=1e-5).float()).double()
    hid = torch.nn.functional.gelu(xn @ _bf(W1).double().t() + b1.double())
    v = _bf(hid.float()).double() @ _bf(W2).double().t() + b2.double() + x1
    if aff is not None:
        v = torch.nn.functional.silu(aff[C:].double() + v * (1.0 + aff[:C].double()))
    scale = float(v.abs().max())
    d = float((x_io.double() - v).abs().max()) / scale
    # (LayerNorm values / hidden values on a bf16 rounding boundary may round the other way than in fp64)
    assert d <= 1e-3, f"fp32 residual stream: {d:.2e}"
    if out_mode:
        ref = torch.nn.functional.layer_norm(v, (C,), eps=1e-5) if out_mode == 1 else v
        assert torch.isfinite(out_xn).all()
        err = (out_xn.double() - ref).abs() - 2.0 ** -8 * ref.abs()
        worst = float(err.max()) / float(ref.abs().max())
        assert worst <= 2e-3, f"bf16 output: {worst:.2e} beyond bf16 rounding"
