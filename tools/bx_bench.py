#!/usr/bin/env python3
"""GPU box: per-shape timings of the bf16 block pipeline's kernels (csrc/kernels_bx.hip) at BASELINE configs[4]'s per-GPU share
(COCO-bits, B = 512): HIP-event mean over back-to-back launches on random data, with the algorithmic TFLOP/s and the HBM bytes each
launch has to move at least (-> the GB/s it would need at that time)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusesg_amd import lib as L

lib = L.load(os.environ.get("BX_LIB") or None)   # BX_LIB: a kernel-variant build (tools/bx_exp.sh)
p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
ITERS = int(os.environ.get("BX_ITERS", "20"))


def gemm(tag, M, N, K, act=0, res=0, mod=0, ln=0, c32=0, cb=1, c2=0):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g) if res else None
    aff = torch.randn(2 * N, device="cuda", generator=g) * 0.5 if mod else None
    oC = torch.empty(M, N, device="cuda") if (c32 or res) else None
    oCb = torch.empty(M, N, device="cuda") if cb else None
    oC2 = torch.empty(M, N, device="cuda") if c2 else None
    ms = C.c_float(0)
    rc = lib.dsg_debug_gemm_bx(M, N, K, p(A), p(W), p(bias), p(R), act, p(aff), ln, p(oC), p(oCb), p(oC2), ITERS, C.byref(ms), None)
    assert rc == 0, rc
    fl = 2.0 * M * N * K
    by = M * K * 2 + N * K * 2 + (M * N * 4 * 2 if res else (M * N * 4 if c32 else 0)) + (M * N * 2 if cb else 0) + (M * N * 2 if c2 else 0)
    print(f"{tag:34s} M={M:7d} N={N:5d} K={K:5d}: {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   min HBM {by / 1e6:7.1f} MB -> "
          f"{by / ms.value / 1e9:6.2f} TB/s", flush=True)


def mlp(tag, M, Cc, mod=1, out_mode=1):
    g = torch.Generator(device="cuda").manual_seed(3)
    xn = torch.randn(M, Cc, device="cuda", generator=g)
    x = torch.randn(M, Cc, device="cuda", generator=g)
    W1 = torch.randn(4 * Cc, Cc, device="cuda", generator=g) / Cc ** 0.5
    b1 = torch.randn(4 * Cc, device="cuda", generator=g)
    W2 = torch.randn(Cc, 4 * Cc, device="cuda", generator=g) / (4 * Cc) ** 0.5
    b2 = torch.randn(Cc, device="cuda", generator=g)
    aff = torch.randn(2 * Cc, device="cuda", generator=g) * 0.1 if mod else None
    o = torch.empty(M, Cc, device="cuda")
    ms = C.c_float(0)
    rc = lib.dsg_debug_mlp_bx(M, Cc, p(xn), p(x), p(W1), p(b1), p(W2), p(b2), p(aff), out_mode, p(o), ITERS, C.byref(ms), None)
    assert rc == 0, rc
    fl = 16.0 * M * Cc * Cc
    by = M * Cc * (2 + 4 + 4 + 2)
    print(f"{tag:34s} M={M:7d} C={Cc:5d}        : {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   min HBM {by / 1e6:7.1f} MB -> "
          f"{by / ms.value / 1e9:6.2f} TB/s", flush=True)


def projmlp(tag, M, Cc, mod=1, out_mode=1):
    g = torch.Generator(device="cuda").manual_seed(3)
    att = torch.randn(M, Cc, device="cuda", generator=g)
    x = torch.randn(M, Cc, device="cuda", generator=g)
    Wp = torch.randn(Cc, Cc, device="cuda", generator=g) / Cc ** 0.5
    bp = torch.randn(Cc, device="cuda", generator=g)
    W1 = torch.randn(4 * Cc, Cc, device="cuda", generator=g) / Cc ** 0.5
    b1 = torch.randn(4 * Cc, device="cuda", generator=g)
    W2 = torch.randn(Cc, 4 * Cc, device="cuda", generator=g) / (4 * Cc) ** 0.5
    b2 = torch.randn(Cc, device="cuda", generator=g)
    aff = torch.randn(2 * Cc, device="cuda", generator=g) * 0.1 if mod else None
    o = torch.empty(M, Cc, device="cuda")
    ms = C.c_float(0)
    rc = lib.dsg_debug_projmlp_bx(M, Cc, p(att), p(x), p(Wp), p(bp), p(W1), p(b1), p(W2), p(b2), p(aff), out_mode, p(o), ITERS, C.byref(ms), None)
    assert rc == 0, rc
    fl = 18.0 * M * Cc * Cc
    by = M * Cc * (2 + 4 + 4 + 2)
    print(f"{tag:34s} M={M:7d} C={Cc:5d}        : {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   min HBM {by / 1e6:7.1f} MB -> "
          f"{by / ms.value / 1e9:6.2f} TB/s", flush=True)


def attn(tag, B, res, ws, shift, heads):
    g = torch.Generator(device="cuda").manual_seed(2)
    Cc, T = 32 * heads, res * res
    Wp = (ws * ws + 31) // 32 * 32
    nWt = (res // ws) ** 2 if shift else 1
    qkv = torch.randn(B * T, 3 * Cc, device="cuda", generator=g)
    qkv[:, :Cc] *= 0.25
    bias = torch.randn(nWt, heads, Wp, Wp, device="cuda", generator=g)
    bias[:, :, ws * ws:, :] = -1e30
    out = torch.empty(B * T, Cc, device="cuda")
    ms = C.c_float(0)
    rc = lib.dsg_debug_attn_bx(B, res, ws, shift, heads, p(qkv), p(bias), p(out), ITERS, C.byref(ms), None)
    assert rc == 0, rc
    fl = 4.0 * B * T * ws * ws * Cc
    by = B * T * 4 * Cc * 2
    print(f"{tag:34s} B={B} res={res} ws={ws} heads={heads}: {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   min HBM {by / 1e6:7.1f} MB -> "
          f"{by / ms.value / 1e9:6.2f} TB/s", flush=True)


def qkv_attn(tag, B, res, ws, shift, heads):
    g = torch.Generator(device="cuda").manual_seed(4)
    Cc, T = 32 * heads, res * res
    Wp = (ws * ws + 31) // 32 * 32
    nWt = (res // ws) ** 2 if shift % 1000 else 1
    xn = torch.randn(B * T, Cc, device="cuda", generator=g)
    W = torch.randn(3 * Cc, Cc, device="cuda", generator=g) / Cc ** 0.5
    W[:Cc] *= 0.25
    bq = torch.randn(3 * Cc, device="cuda", generator=g) * 0.1
    bias = torch.randn(nWt, heads, Wp, Wp, device="cuda", generator=g)
    bias[:, :, ws * ws:, :] = -1e30
    out = torch.empty(B * T, Cc, device="cuda")
    ms = C.c_float(0)
    rc = lib.dsg_debug_qkv_attn_bx(B, res, ws, shift, heads, p(xn), p(W), p(bq), p(bias), p(out), ITERS, C.byref(ms), None)
    assert rc == 0, rc
    fl = 4.0 * B * T * ws * ws * Cc + 6.0 * B * T * Cc * Cc
    by = B * T * 2 * Cc * 2
    print(f"{tag:34s} B={B} res={res} ws={ws} heads={heads}: {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:8.1f} TFLOP/s   min HBM {by / 1e6:7.1f} MB -> "
          f"{by / ms.value / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    B = int(os.environ.get("BX_B", "512"))
    print(f"B={B}")
    M2, M1, M0 = B * 100, B * 400, B * 1600
    only = os.environ.get("BX_ONLY", "")
    if only == "mlp384":   # the LDS-DMA kernel alone (tools/m384_exp.sh)
        projmlp("L2 proj + MLP (mod, LN), LDS-DMA", M2, 384)
        sys.exit(0)
    if only == "mlp384s":  # the one-wave-per-SIMD LDS-DMA kernel alone
        projmlp("L2 proj + MLP (mod, LN), 4 waves solo", M2, 384, out_mode=1 + 64)
        sys.exit(0)
    if only == "mlpb":  # the level-0 / level-1 fused (proj +) MLP kernels alone (tools/mlpb_exp.sh)
        projmlp("L1 proj + MLP fused (mod, LN)", M1, 192)
        projmlp("L1 proj + MLP fused (copy)", M1, 192, mod=0, out_mode=2)
        projmlp("L0 proj + MLP fused (copy)", M0, 96, mod=0, out_mode=2)
        projmlp("L0 proj + MLP fused (copy), LDS-resident", M0, 96, mod=0, out_mode=2 + 128)
        sys.exit(0)
    if only == "mlp":   # the fused MLP kernels alone
        mlp("L2 fused MLP (mod, LN), 8 waves LDS-DMA", M2, 384)
        mlp("L2 fused MLP (mod, LN), 8 waves round 3", M2, 384, out_mode=1 + 32)
        mlp("L2 fused MLP (mod, LN), 4 waves solo", M2, 384, out_mode=1 + 64)
        projmlp("L2 proj + MLP (mod, LN), LDS-DMA", M2, 384)
        projmlp("L2 proj + MLP (mod, LN), round 3", M2, 384, out_mode=1 + 32)
        projmlp("L2 proj + MLP (mod, LN), 4 waves solo", M2, 384, out_mode=1 + 64)
        projmlp("L1 proj + MLP fused (mod, LN)", M1, 192)
        projmlp("L0 proj + MLP fused (copy)", M0, 96, mod=0, out_mode=2)
        sys.exit(0)
    if only == "wx":    # the wave-per-unit QKV + attention kernel alone (DSG_WX_CLK=1: phase clocks)
        qkv_attn("L2 qkv + attn fused", B, 10, 10, 0, 12)
        qkv_attn("L1 qkv + attn fused, shifted", B, 20, 10, 5, 6)
        qkv_attn("L0 qkv + attn fused", B, 40, 10, 0, 3)
        sys.exit(0)
    if only == "qa":    # the fused QKV + attention kernel alone
        qkv_attn("L2 qkv + attn fused", B, 10, 10, 0, 12)
        qkv_attn("L1 qkv + attn fused", B, 20, 10, 0, 6)
        qkv_attn("L1 qkv + attn fused, shifted", B, 20, 10, 5, 6)
        qkv_attn("L0 qkv + attn fused", B, 40, 10, 0, 3)
        qkv_attn("L2 qkv + attn, block per head", B, 10, 10, 1000, 12)
        qkv_attn("L1 qkv + attn, block per head", B, 20, 10, 1000, 6)
        qkv_attn("L1 qkv + attn, block per head, shifted", B, 20, 10, 1005, 6)
        qkv_attn("L0 qkv + attn, block per head", B, 40, 10, 1000, 3)
        sys.exit(0)
    gemm("L2 qkv", M2, 1152, 384)
    gemm("L2 proj (res, LN)", M2, 384, 384, res=1, ln=1)
    gemm("L2 fc1 (gelu)", M2, 1536, 384, act=1)
    gemm("L2 fc2 (res, mod, LN)", M2, 384, 1536, res=1, mod=1, ln=1)
    gemm("L1 qkv", M1, 576, 192)
    gemm("L1 proj (res, LN)", M1, 192, 192, res=1, ln=1)
    gemm("L1 fc1 (gelu)", M1, 768, 192, act=1)
    gemm("L1 fc2 (res, mod, LN)", M1, 192, 768, res=1, mod=1, ln=1)
    gemm("L0 qkv", M0, 288, 96)
    gemm("L0 proj (res, LN)", M0, 96, 96, res=1, ln=1)
    gemm("L0 fc1 (gelu)", M0, 384, 96, act=1)
    gemm("L0 fc2 (res, copy)", M0, 96, 384, res=1)
    gemm("merge L1->L2 (skip, mod, LN)", M2, 384, 768, c32=1, mod=1, ln=1, c2=1)
    gemm("pre_linear L2->L1 (fp32 out)", M2, 768, 768, c32=1, cb=0)
    gemm("post_linear L1 (mod, LN)", M1, 192, 192, c32=1, mod=1, ln=1)
    mlp("L2 fused MLP (mod, LN), 8 waves", M2, 384)
    mlp("L2 fused MLP (mod, LN), 4 waves", M2, 384, out_mode=1 + 16)
    mlp("L1 fused MLP (mod, LN)", M1, 192)
    mlp("L0 fused MLP (copy)", M0, 96, mod=0, out_mode=2)
    projmlp("L2 proj + MLP fused (mod, LN)", M2, 384)
    projmlp("L1 proj + MLP fused (mod, LN)", M1, 192)
    projmlp("L0 proj + MLP fused (copy)", M0, 96, mod=0, out_mode=2)
    if only == "gemm":
        sys.exit(0)
    qkv_attn("L2 qkv + attn fused", B, 10, 10, 0, 12)
    qkv_attn("L1 qkv + attn fused", B, 20, 10, 0, 6)
    qkv_attn("L1 qkv + attn fused, shifted", B, 20, 10, 5, 6)
    qkv_attn("L0 qkv + attn fused", B, 40, 10, 0, 3)
    attn("L2 attn", B, 10, 10, 0, 12)
    attn("L1 attn", B, 20, 10, 0, 6)
    attn("L1 attn shifted", B, 20, 10, 5, 6)
    attn("L0 attn", B, 40, 10, 0, 3)
