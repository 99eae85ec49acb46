// kernels_bx.hip -- the bf16 block pipeline of the opt-in "gemm_bf16" precision mode (BASELINE config 5: COCO-Stuff, bf16).
//
// Round 2's bf16 mode kept the f32 kernel's shape (128x96 tiles, fp32 activations converted on the way into LDS): at the mode's
// 16x matrix rate that kernel is bound by re-reading its fp32 A tile from L2 once per 96 output columns and by 2-byte stores.
// Here every tensor a GEMM reads is ALREADY bf16 in HBM and every GEMM leaves what its consumer reads:
//   * gemm_bx_kernel: bf16 A [M,K] x bf16 W [N,K]^T, 8 waves, block tile (64 WM) x (96 WN), wave tile 64 x 96 (six 32x32x16 MFMAs per
//     16-deep k-step from five ds_read_b128: ~107 B/clk of LDS reads per CU at the full matrix rate, against 171 for the old 32x96
//     wave tile).  The product is formed TRANSPOSED (D[n][m]: the W fragment is the MFMA's A operand), so a lane owns ONE output row
//     and four consecutive columns per accumulator quad: 16-byte fp32 / 8-byte bf16 stores, and LayerNorm statistics of a row are
//     lane-local sums.
//   * epilogue (all optional, in this order): + bias, GELU, + fp32 residual, bf16 copy of the value (skip connection / PatchBreakup
//     input), the NEXT block's modulate+SiLU  silu(shift + v (1 + scale))  (diffusesg.py:238-243), fp32 store (the residual stream
//     stays fp32), and -- when the tile spans the whole row (N = 96 WN: the proj / fc2 / PatchMerging / post_linear GEMMs) -- the
//     LayerNorm of the stored row, written as the bf16 tensor the next QKV / fc1 GEMM multiplies (gamma / beta are folded into that
//     GEMM's weights at pack time), so no GEMM of the pipeline has a prologue and no separate normalisation pass runs inside a level.
//   * attn_bx_kernel: window attention (diffusesg.py:108-139) on bf16 q, k, v with v_mfma_f32_32x32x16_bf16 for both products, fp32
//     softmax in the accumulators; P never leaves the register file (the S^T accumulator is the B operand of O^T = V^T P^T).
//   * ln_bx_kernel: the row pass where no GEMM produces a level's first tensor (after PatchEmbed; rows wider than a tile).
#include "kernels_common.hip.h"

namespace dsg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // RNE (v_cvt_pk_bf16_f32); a in the low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u32x2 pack_bf16x4(const f32x4 v) { u32x2 r; r[0] = pack_bf16(v[0], v[1]); r[1] = pack_bf16(v[2], v[3]); return r; }
__device__ __forceinline__ void buf_store2(u32x2 v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4(f32x4 v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ u32x4 buf_load_u4(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }

// -------------------------------------------------------------------------------------------------
// GEMM.  LDS: two stages of (BM + BN) rows x (KB + 8) bf16 (row stride 144 B / 80 B: conflict-free ds_read_b128); register-staged
// (global -> VGPR one chunk ahead -> LDS), one barrier per chunk.
// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int KB>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bx_kernel(BxGemm g, int tiles_m, int tiles_n) {
    constexpr int NT = 64 * WM * WN, BM = 64 * WM, BN = 96 * WN, LDP = KB + 8;
    constexpr int CPR = KB / 8;                       // 16-byte pieces per tile row
    constexpr int RPP = NT / CPR;                     // tile rows covered by one pass of the block
    constexpr int PA = BM / RPP, PW = (BN + RPP - 1) / RPP;
    static_assert(NT % CPR == 0 && BM % RPP == 0, "staging layout");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (BM + BN) * LDP];
    constexpr int STAGE = (BM + BN) * LDP;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3;
    const int tm = (seq / tiles_n) * 8 + xcd, tn = seq % tiles_n;
    if (tm >= tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int rows_m = min(BM, g.M - m0), rows_n = min(BN, g.N - n0);
    const __bf16 *Ap = static_cast<const __bf16 *>(g.A), *A2p = static_cast<const __bf16 *>(g.A2), *Wp = static_cast<const __bf16 *>(g.W);
    const rsrc_t rsA1 = make_rsrc(Ap + (size_t)m0 * g.lda, (unsigned)rows_m * g.lda * 2u);
    const rsrc_t rsA2 = make_rsrc(A2p ? A2p + (size_t)m0 * g.lda2 : Ap, A2p ? (unsigned)rows_m * g.lda2 * 2u : 0u);
    const rsrc_t rsW = make_rsrc(Wp + (size_t)n0 * g.K, (unsigned)rows_n * g.K * 2u);
    const int sc = tid % CPR, sr = tid / CPR;         // this thread's 16-byte piece / first tile row
    unsigned voffA1[PA], voffA2[PA], voffW[PW];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        voffA1[p] = ((unsigned)(sr + RPP * p) * g.lda + 8u * sc) * 2u;
        voffA2[p] = ((unsigned)(sr + RPP * p) * g.lda2 + 8u * sc) * 2u;
    }
#pragma unroll
    for (int p = 0; p < PW; p++) voffW[p] = (sr + RPP * p < BN) ? ((unsigned)(sr + RPP * p) * g.K + 8u * sc) * 2u : 0x7fffffffu;
    const int nk = (g.K + KB - 1) / KB;
    const int K1 = A2p ? g.K1 : g.K;

    u32x4 sa[PA], sw[PW];
    auto issue = [&](int kc) {
        const int k0 = kc * KB;
        const bool kvalid = k0 + 8 * sc < g.K;        // K may end inside a chunk (K = 96, KB = 64): the rest of the row is not zero
        const bool second = k0 >= K1;
        const unsigned soffA = (unsigned)(second ? k0 - K1 : k0) * 2u;
#pragma unroll
        for (int p = 0; p < PA; p++) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kvalid) v = second ? buf_load_u4(rsA2, voffA2[p], soffA) : buf_load_u4(rsA1, voffA1[p], soffA);
            sa[p] = v;
        }
#pragma unroll
        for (int p = 0; p < PW; p++) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kvalid) v = buf_load_u4(rsW, voffW[p], (unsigned)k0 * 2u);
            sw[p] = v;
        }
    };
    auto write = [&](int buf) {
        __bf16 *As = lds + buf * STAGE, *Ws = As + BM * LDP;
#pragma unroll
        for (int p = 0; p < PA; p++) *reinterpret_cast<u32x4 *>(As + (sr + RPP * p) * LDP + 8 * sc) = sa[p];
#pragma unroll
        for (int p = 0; p < PW; p++)
            if (BN % RPP == 0 || sr + RPP * p < BN) *reinterpret_cast<u32x4 *>(Ws + (sr + RPP * p) * LDP + 8 * sc) = sw[p];
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    issue(0);
    write(0);
    __syncthreads();
    for (int kc = 0; kc < nk; kc++) {
        const int cur = kc & 1;
        if (kc + 1 < nk) issue(kc + 1);
        const __bf16 *As = lds + cur * STAGE + (wm * 64 + lrow) * LDP + 8 * lhalf;
        const __bf16 *Ws = lds + cur * STAGE + (BM + wn * 96 + lrow) * LDP + 8 * lhalf;
#pragma unroll
        for (int s = 0; s < KB / 16; s++) {
            bf16x8 af[2], wf[3];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) af[mt] = *reinterpret_cast<const bf16x8 *>(As + 32 * mt * LDP + 16 * s);
#pragma unroll
            for (int nt = 0; nt < 3; nt++) wf[nt] = *reinterpret_cast<const bf16x8 *>(Ws + 32 * nt * LDP + 16 * s);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 3; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
        if (kc + 1 < nk) write(1 - cur);
        __syncthreads();
    }

    // ---- epilogue: lane (m = lrow, half) owns output rows m0 + 64 wm + 32 mt + lrow; accumulator quad q of tile nt holds its columns
    // n0 + 96 wn + 32 nt + 8 q + 4 half + {0..3}
    const unsigned OOB = 0x7fffffffu;
    const rsrc_t rsC = make_rsrc(g.C ? g.C + (size_t)m0 * g.ldc : nullptr, g.C ? (unsigned)rows_m * g.ldc * 4u : 0u);
    const rsrc_t rsR = make_rsrc(g.res ? g.res + (size_t)m0 * g.ldres : nullptr, g.res ? (unsigned)rows_m * g.ldres * 4u : 0u);
    __bf16 *Cbp = static_cast<__bf16 *>(g.Cb), *C2p = static_cast<__bf16 *>(g.C2b);
    const rsrc_t rsCb = make_rsrc(Cbp ? Cbp + (size_t)m0 * g.ldcb : nullptr, Cbp ? (unsigned)rows_m * g.ldcb * 2u : 0u);
    const rsrc_t rsC2 = make_rsrc(C2p ? C2p + (size_t)m0 * g.ldc2b : nullptr, C2p ? (unsigned)rows_m * g.ldc2b * 2u : 0u);
    f32x2 *part = reinterpret_cast<f32x2 *>(lds);   // [BM][WN] (sum, sumsq) of a row over one wave's 96 columns; the tiles are dead
    const bool ln = g.ln_out != 0;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        const unsigned mrow = (unsigned)(wm * 64 + 32 * mt + lrow);
        const float *aff_row = nullptr;
        if (g.mod_aff) aff_row = g.mod_aff + (size_t)(g.mod_ld ? min(m0 + (int)mrow, g.M - 1) / g.mod_T : 0) * g.mod_ld + g.mod_off;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = n0 + wn * 96 + 32 * nt + 8 * q + 4 * lhalf;
                const bool nok = n < g.N;
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = acc[mt][nt][4 * q + t];
                if (g.bias && nok) v += *reinterpret_cast<const f32x4 *>(g.bias + n);
                if (g.act == ACT_GELU) {
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = gelu_f(v[t]);
                }
                if (g.res) v += buf_load4(rsR, nok ? (mrow * g.ldres + (unsigned)n) * 4u : OOB, 0u);
                if (C2p) buf_store2(pack_bf16x4(v), rsC2, nok ? (mrow * g.ldc2b + (unsigned)n) * 2u : OOB, 0u);
                if (aff_row && nok) {
                    const f32x4 scl = *reinterpret_cast<const f32x4 *>(aff_row + n), sft = *reinterpret_cast<const f32x4 *>(aff_row + g.N + n);
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = silu_exact(fmaf(v[t], scl[t] + 1.0f, sft[t]));
                }
                if (g.C) buf_store4(v, rsC, nok ? (mrow * g.ldc + (unsigned)n) * 4u : OOB, 0u);
                if (ln) {
                    if (nok) {
#pragma unroll
                        for (int t = 0; t < 4; t++) { ssum += v[t]; ssq = fmaf(v[t], v[t], ssq); }
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[mt][nt][4 * q + t] = v[t];
                } else if (Cbp) {
                    buf_store2(pack_bf16x4(v), rsCb, nok ? (mrow * g.ldcb + (unsigned)n) * 2u : OOB, 0u);
                }
            }
        if (ln) {
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (lhalf == 0) part[mrow * WN + wn] = (f32x2){ssum, ssq};
        }
    }
    if (ln) {
        __syncthreads();
        const float invn = 1.0f / (float)g.N;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const unsigned mrow = (unsigned)(wm * 64 + 32 * mt + lrow);
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WN; w2++) { const f32x2 p2 = part[mrow * WN + w2]; sm += p2[0]; sq += p2[1]; }
            const float mean = sm * invn, rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * invn), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = n0 + wn * 96 + 32 * nt + 8 * q + 4 * lhalf;
                    f32x4 v;
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = fmaf(acc[mt][nt][4 * q + t], rstd, nmr);
                    buf_store2(pack_bf16x4(v), rsCb, n < g.N ? (mrow * g.ldcb + (unsigned)n) * 2u : OOB, 0u);
                }
        }
    }
}

static int round_up8(int x) { return (x + 7) / 8 * 8; }

bool launch_gemm_bx(const BxGemm &g, hipStream_t s) {
    if (!g.A || !g.W || g.M < 1 || g.N < 1 || g.K < 8 || g.K % 8 != 0 || g.N % 4 != 0 || (g.act != ACT_NONE && g.act != ACT_GELU)) return false;
    if (g.lda % 8 != 0 || (g.A2 && (g.lda2 % 8 != 0 || g.K1 <= 0))) return false;
    // geometry: the tile spans the whole row when a LayerNorm output is asked for; otherwise 256 x 192 where N splits into 192s,
    // 512 x 96 for the narrow / odd widths (N = 96, 288)
    int geo;   // 0: <4,2,64> 256x192   1: <2,4,64> 128x384   2: <8,1,32> 512x96
    if (g.ln_out) {
        if (!g.Cb) return false;
        if (g.N == 96) geo = 2; else if (g.N == 192) geo = 0; else if (g.N == 384) geo = 1; else return false;
    } else {
        geo = (g.N % 192 == 0) ? 0 : 2;
    }
    const int kb = geo == 2 ? 32 : 64;
    if (g.A2 && g.K1 % kb != 0) return false;
    if (geo == 2 && g.K % 32 != 0) return false;
    const int bm = geo == 0 ? 256 : (geo == 1 ? 128 : 512), bn = geo == 0 ? 192 : (geo == 1 ? 384 : 96);
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = (g.N + bn - 1) / bn;
    const dim3 grid(round_up8(tiles_m) * tiles_n), block(512);
    if (geo == 0) hipLaunchKernelGGL((gemm_bx_kernel<4, 2, 64>), grid, block, 0, s, g, tiles_m, tiles_n);
    else if (geo == 1) hipLaunchKernelGGL((gemm_bx_kernel<2, 4, 64>), grid, block, 0, s, g, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_bx_kernel<8, 1, 32>), grid, block, 0, s, g, tiles_m, tiles_n);
    return true;
}

// -------------------------------------------------------------------------------------------------
// Row pass: x fp32 [M, C] -> optional modulate+SiLU in place (diffusesg.py:238-243: the modulated tensor is also the shortcut) ->
// xn bf16 = LayerNorm of the row without affine (gamma / beta live in the consuming GEMM's weights), or, with ln == 0, the plain bf16
// copy.  One wave per row, 16 B per lane per access.
// -------------------------------------------------------------------------------------------------
constexpr int BXROW_MAXV4 = 6;   // rows up to 64 * 6 * 4 = 1536 channels
__global__ __launch_bounds__(256) void ln_bx_kernel(float *x, const float *aff, int aff_ld, int aff_off, __bf16 *xn, int T, int C, int M, int ln) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int C4 = C >> 2;
    f32x4 *xr = reinterpret_cast<f32x4 *>(x + (size_t)m * C);
    const f32x4 *scale = aff ? reinterpret_cast<const f32x4 *>(aff + (size_t)(aff_ld ? m / T : 0) * aff_ld + aff_off) : nullptr;
    const f32x4 *shift = aff ? scale + C4 : nullptr;
    f32x4 v[BXROW_MAXV4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < BXROW_MAXV4; i++) {
        const int c = lane + 64 * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C4) {
            v[i] = xr[c];
            if (aff) {
                const f32x4 sc = scale[c], sh = shift[c];
#pragma unroll
                for (int t = 0; t < 4; t++) v[i][t] = silu_exact(sh[t] + v[i][t] * (sc[t] + 1.0f));
                xr[c] = v[i];
            }
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (ln) {
        mean = wave_sum(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < BXROW_MAXV4; i++)
            if (lane + 64 * i < C4)
#pragma unroll
                for (int t = 0; t < 4; t++) { const float d = v[i][t] - mean; var += d * d; }
        rstd = fast_rsqrt(wave_sum(var) / (float)C + LN_EPS);
    }
    u32x2 *dst = reinterpret_cast<u32x2 *>(xn + (size_t)m * C);
#pragma unroll
    for (int i = 0; i < BXROW_MAXV4; i++) {
        const int c = lane + 64 * i;
        if (c < C4) dst[c] = pack_bf16x4((v[i] - mean) * rstd);
    }
}
void launch_ln_bx(float *x, const float *aff, int aff_ld, int aff_off, void *xn, int B, int T, int C, bool ln, hipStream_t s) {
    const int M = B * T;
    hipLaunchKernelGGL(ln_bx_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, aff, aff_ld, aff_off, (__bf16 *)xn, T, C, M, ln ? 1 : 0);
}

// -------------------------------------------------------------------------------------------------
// Window attention on bf16 q, k, v (diffusesg.py:108-139; window partition / cyclic shift / reverse folded into the token index as in
// window_attn_kernel).  One wave per (sample, window, head); Wp = 32 KT >= WS^2 positions.
//   S^T[key][query] = K Q^T + bias   A operand = K rows, B operand = Q rows (q pre-scaled by d^-1/2 log2 e in the QKV weights): a lane
//                                    owns one query column, so the softmax is lane-local (+ one exchange between the half-waves);
//   O^T[d][query]  = V^T P^T         B operand = the S^T accumulators themselves, converted pairwise to bf16 (an accumulator tile's
//                                    rows are exactly a k-step's k index, in the permuted order below); A operand = V^T, staged through
//                                    LDS once per wave: vt[d][pos(key)] with pos(16 s + o) = 16 s + 8 ((o >> 2) & 1) + 4 (o >> 3) + (o & 3),
//                                    so that one ds_read_b128 of lane (d, half) returns keys 16 s + 8 (j >> 2) + 4 half + (j & 3), j = 0..7
//                                    -- the k order in which registers 8 s' .. 8 s' + 7 of a 32x32 accumulator enumerate its rows.
// The result is a lane's 4 consecutive head dims per accumulator quad: 8-byte bf16 stores.
// -------------------------------------------------------------------------------------------------
template <int KT, int WS>
__global__ __launch_bounds__(256) void attn_bx_kernel(const __bf16 *__restrict__ qkv, const float *__restrict__ biasT, __bf16 *__restrict__ out,
                                                      int B, WinGeom g, int n_units) {
    constexpr int Wp = 32 * KT, Wt = WS * WS, VLD = Wp + 8;   // vt row stride in bf16 (16-B aligned, conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) __bf16 vt_lds[4][32 * VLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    int unit = blockIdx.x * 4 + wave;
    const bool active = unit < n_units;
    if (!active) unit = n_units - 1;
    const int heads = g.heads, C = g.C, res = g.res;
    const int nwr = res / WS, nW = nwr * nwr, T = res * res;
    const int head = unit % heads;
    const int bw = unit / heads;
    const int w = bw % nW, b = bw / nW;
    const int wi = w / nwr, wj = w % nwr;
    auto token_of = [&](int p) -> int {   // window position -> token of the sample (padded positions: token 0)
        if (p >= Wt) return 0;
        int ti = wi * WS + p / WS + g.shift, tj = wj * WS + p % WS + g.shift;
        if (ti >= res) ti -= res;
        if (tj >= res) tj -= res;
        return ti * res + tj;
    };
    const rsrc_t rsQ = make_rsrc(qkv + (size_t)b * T * 3 * C, (unsigned)T * 3u * C * 2u);
    const rsrc_t rsB = make_rsrc(biasT + ((size_t)(g.shift > 0 ? w : 0) * heads + head) * Wp * Wp, (unsigned)(Wp * Wp) * 4u);
    const rsrc_t rsO = make_rsrc(out + (size_t)b * T * C, active ? (unsigned)T * C * 2u : 0u);
    int tokr[KT];
    unsigned rowoff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        tokr[kt] = token_of(32 * kt + lrow);
        rowoff[kt] = (unsigned)tokr[kt] * (unsigned)(3 * C) * 2u + 16u * lhalf;   // this lane's 8 head dims of k-step s: + 32 s bytes
    }
    const unsigned hq = (unsigned)head * 64u, hk = (unsigned)(C + head * 32) * 2u, hv = (unsigned)(2 * C + head * 32) * 2u;
    // K fragments (lane = key row), both k-steps
    bf16x8 kf[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int s = 0; s < 2; s++) kf[kt][s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsQ, rowoff[kt], hk + 32u * s));
    // V^T -> LDS: item = (key pair kp, group of 8 head dims dg); the two keys' values of one head dim share a dword
    __bf16 *vt = vt_lds[wave];
#pragma unroll
    for (int it = 0; it < (Wp / 2) * 4 / 64; it++) {
        const int item = lane + 64 * it, kp = item >> 2, dg = item & 3;
        const int k0 = 2 * kp;
        const u32x4 va = buf_load_u4(rsQ, (unsigned)token_of(k0) * (unsigned)(3 * C) * 2u + 16u * dg, hv);
        const u32x4 vb = buf_load_u4(rsQ, (unsigned)token_of(k0 + 1) * (unsigned)(3 * C) * 2u + 16u * dg, hv);
        const int o = k0 & 15, pos = (k0 & ~15) + 8 * ((o >> 2) & 1) + 4 * (o >> 3) + (o & 3);
        unsigned *dst = reinterpret_cast<unsigned *>(vt + (8 * dg) * VLD + pos);
#pragma unroll
        for (int e = 0; e < 4; e++) {   // dword e of va / vb holds head dims 8 dg + 2 e, + 1
            dst[(2 * e) * (VLD / 2)] = __builtin_amdgcn_perm(vb[e], va[e], 0x05040100u);       // (va.lo, vb.lo)
            dst[(2 * e + 1) * (VLD / 2)] = __builtin_amdgcn_perm(vb[e], va[e], 0x07060302u);   // (va.hi, vb.hi)
        }
    }
    __builtin_amdgcn_wave_barrier();
    // V^T fragments: lane (d = lrow, half), k-step (kt, s): positions 32 kt + 16 s + 8 half .. + 7
    bf16x8 vf[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int s = 0; s < 2; s++) vf[kt][s] = *reinterpret_cast<const bf16x8 *>(vt + lrow * VLD + 32 * kt + 16 * s + 8 * lhalf);

    const unsigned boff = (unsigned)(4 * lhalf * Wp + lrow) * 4u;
#pragma unroll
    for (int qt = 0; qt < KT; qt++) {
        if (32 * qt >= Wt) break;
        bf16x8 qf[2];
#pragma unroll
        for (int s = 0; s < 2; s++) qf[s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsQ, rowoff[qt], hq + 32u * s));
        f32x16 sacc[KT];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                sacc[kt][r] = buf_load1(rsB, boff, (unsigned)((32 * kt + (r & 3) + 8 * (r >> 2)) * Wp + 32 * qt) * 4u);
#pragma unroll
            for (int s = 0; s < 2; s++) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s], qf[s], sacc[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            u32x4 pf[2];   // P^T as the B operand: registers 8 s .. 8 s + 7 pairwise -> k-step s
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(sacc[kt][r] - mx), e1 = __builtin_amdgcn_exp2f(sacc[kt][r + 1] - mx);   // scores carry log2(e)
                sum += e0 + e1;
                pf[r >> 3][(r & 7) >> 1] = pack_bf16(e0, e1);
            }
#pragma unroll
            for (int s = 0; s < 2; s++)
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][s], __builtin_bit_cast(bf16x8, pf[s]), oacc, 0, 0, 0);
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = fast_rcp(sum);
        // O^T tile: lane = query 32 qt + lrow, quad q = head dims 8 q + 4 half + {0..3}
        const unsigned eoff = ((unsigned)tokr[qt] * (unsigned)C + (unsigned)(head * 32 + 4 * lhalf)) * 2u;
        const bool qok = 32 * qt + lrow < Wt;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; t++) o[t] = oacc[4 * q + t] * inv;
            buf_store2(pack_bf16x4(o), rsO, qok ? eoff + 16u * q : 0x7fffffffu, 0u);
        }
    }
}

bool launch_attn_bx(const void *qkv, const float *biasT, void *out, int B, const WinGeom &g, hipStream_t s) {
    const int nW = (g.res / g.ws) * (g.res / g.ws);
    const int n_units = B * nW * g.heads;
    const dim3 grid((n_units + 3) / 4), block(256);
    if (g.C != 32 * g.heads || g.C % 8 != 0) return false;
#define AX(KT_, WS_) hipLaunchKernelGGL((attn_bx_kernel<KT_, WS_>), grid, block, 0, s, (const __bf16 *)qkv, biasT, (__bf16 *)out, B, g, n_units)
    switch (g.ws) {
        case 4: AX(1, 4); break;
        case 5: AX(1, 5); break;
        case 8: AX(2, 8); break;
        case 10: AX(4, 10); break;
        default: return false;
    }
#undef AX
    return true;
}

}  // namespace dsg
