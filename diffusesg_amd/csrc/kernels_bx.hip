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
#include <type_traits>

namespace dsg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2_c u32x2;   // pack_bf16 / pack_bf16x4: kernels_common.hip.h

__device__ __forceinline__ void buf_store2(u32x2 v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4(f32x4 v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_u4(u32x4 v, rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0); }
__device__ __forceinline__ u32x4 buf_load_u4(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }

// -------------------------------------------------------------------------------------------------
// GEMM.  At the bf16 matrix rate a 64-deep k chunk of a wave tile is ~0.4 us of MFMA work -- less than a loaded L2 / HBM round trip --
// and K is short (96..1536), so the loop is built around memory latency, not around the matrix pipe:
//   * the global loads run TWO chunks ahead of the MFMAs in two register sets (global -> VGPR -> LDS);
//   * ONE LDS stage of (BM + BN) rows x (KB + 8) bf16 (row stride 144 B / 80 B: conflict-free ds_read_b128), two barriers per chunk --
//     46 KB for the 128 x 192 tile, so two 4-wave blocks share a CU and one block's prologue, barriers and epilogue can run under the
//     other's MFMAs;
//   * blocks are persistent (grid = resident blocks, tiles dealt round-robin; a block keeps its XCD class, so the column tiles of one
//     row block meet in one L2).
// What bounds it (in-kernel clocks, tools/bx_exp.sh, COCO level-2 fc1, per 128 x 192 x 384 tile): ~10k clocks in the MFMA section
// (4.6k of matrix-pipe time per wave, two waves per SIMD), 3.5k waiting for loads, ~12k in the epilogue -- 96 GELUs per lane cost as
// many VALU clocks as the tile's MFMAs at K = 384, and the stores are HBM-bound by themselves; the proj / fc2 / C = 96 shapes sit at
// 3.5-4.3 TB/s of their minimum HBM traffic.  Tried and measured without gain (profiles/r3/bx_experiments.txt): prefetching across
// the tile boundary (spills: both register sets live through the epilogue), starting one of a CU's two blocks half a tile late.
// -------------------------------------------------------------------------------------------------
// RES: a residual is added; MOD: 0 no modulate, 1 batch-uniform (scale, shift) staged in LDS, 2 per-sample rows read in the epilogue
// (test / training-time forwards) -- compile-time so that the common epilogue has no load behind a branch.
#ifndef DSG_BX_EXP
#define DSG_BX_EXP 0   // timing experiments of tools/bx_exp.sh (wrong results): 1 no epilogue, 2 no global loads / LDS refills, 3 no MFMAs
#endif
// MT: 32-row accumulator tiles per wave (wave tile 32 MT x 96).  MT = 1 halves a wave's registers (~125): twice the waves per SIMD on the
// same block tile and LDS stage, at 1.6x the LDS fragment bytes per MFMA.
template <int WM, int WN, int KB, bool RES, int MOD, int MT = 2>
__global__ __launch_bounds__(64 * WM * WN, (MT == 2 ? 2 : 4)) void gemm_bx_kernel(BxGemm g, int tiles_m, int tiles_n, int tiles_total) {
    constexpr int NT = 64 * WM * WN, WR = 32 * MT, BM = WR * WM, BN = 96 * WN, LDP = KB + 8;
    constexpr int CPR = KB / 8;                       // 16-byte pieces per tile row
    constexpr int RPP = NT / CPR;                     // tile rows covered by one pass of the block
    constexpr int PA = BM / RPP, PW = (BN + RPP - 1) / RPP;
    constexpr int CV = (BN + NT - 1) / NT;            // column-vector entries per thread
    constexpr int TLD = 104;                          // row stride (bf16) of a wave's 32 x 96 store-transposition buffer: 208 B
    static_assert(WM * WN * 32 * TLD <= (BM + BN) * LDP, "the transposition buffers live in the tile stage");
    static_assert(NT % CPR == 0 && BM % RPP == 0, "staging layout");
    // tile stage | [3][BN] floats: the epilogue's per-column vectors (bias, 1 + scale, shift) | [BM][WN] row partials
    __shared__ __attribute__((aligned(16))) __bf16 lds[(BM + BN) * LDP + 6 * BN + 4 * BM * WN];
    float *colv = reinterpret_cast<float *>(lds + (BM + BN) * LDP);
    f32x2 *part = reinterpret_cast<f32x2 *>(colv + 3 * BN);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const __bf16 *Ap = static_cast<const __bf16 *>(g.A), *A2p = static_cast<const __bf16 *>(g.A2), *Wp = static_cast<const __bf16 *>(g.W);
    const int sc = tid % CPR, sr = tid / CPR;         // this thread's 16-byte piece / first tile row
    // per-thread byte offsets of pass 0; pass p adds p * RPP rows (a block-uniform stride)
    const unsigned voffA1 = ((unsigned)sr * g.lda + 8u * sc) * 2u, voffA2 = ((unsigned)sr * g.lda2 + 8u * sc) * 2u;
    const unsigned voffW = ((unsigned)sr * g.K + 8u * sc) * 2u;
    const unsigned strA1 = (unsigned)RPP * g.lda * 2u, strA2 = (unsigned)RPP * g.lda2 * 2u, strW = (unsigned)RPP * g.K * 2u;
    const int nk = (g.K + KB - 1) / KB;               // >= 2 (launcher)
    const int K1 = A2p ? g.K1 : g.K;

    // a tile: padded id t -> (tm, tn) with the XCD-aware order of gemm4 (ids t and t + 8 share an XCD; tn runs fastest inside one)
    struct Tile { int m0, n0, rows_m, valid; };
    auto make_tile = [&](int t) -> Tile {
        Tile c;
        for (;; t += (int)gridDim.x) {   // skip the padding ids of the last row-block group
            if (t >= tiles_total) { c.valid = 0; c.m0 = c.n0 = 0; c.rows_m = 0; break; }
            const int xcd = t & 7, seq = t >> 3;
            const int tm = (seq / tiles_n) * 8 + xcd, tn = seq % tiles_n;
            if (tm < tiles_m) { c.valid = 1; c.m0 = tm * BM; c.n0 = tn * BN; c.rows_m = min(BM, g.M - c.m0); break; }
        }
        return c;
    };
    auto next_id = [&](int t) -> int {   // the id make_tile(t) settled on is not returned; recompute the skip (cheap, scalar)
        for (;; t += (int)gridDim.x) {
            if (t >= tiles_total) return t;
            const int xcd = t & 7, seq = t >> 3;
            if ((seq / tiles_n) * 8 + xcd < tiles_m) return t;
        }
    };

    struct Stage { u32x4 a[PA], w[PW]; };
    auto issue = [&](Stage &st, const Tile &c, int kc) {
        if (DSG_BX_EXP == 2) return;
        const int k0 = kc * KB;
        // K may end inside a chunk (K = 96, KB = 64) and the rest of the row is not zero: pieces beyond K get an out-of-range offset
        // (the descriptor returns zeros) -- no branch around a load
        const unsigned kmask = (k0 + 8 * sc < g.K) ? 0u : 0x7fffffffu;
        const bool second = k0 >= K1;                  // block-uniform
        const unsigned soffA = (unsigned)(second ? k0 - K1 : k0) * 2u;
        if (second) {
            const rsrc_t rsA2 = make_rsrc(A2p + (size_t)c.m0 * g.lda2, (unsigned)c.rows_m * g.lda2 * 2u);
#pragma unroll
            for (int p = 0; p < PA; p++) st.a[p] = buf_load_u4(rsA2, (voffA2 + p * strA2) | kmask, soffA);
        } else {
            const rsrc_t rsA1 = make_rsrc(Ap + (size_t)c.m0 * g.lda, (unsigned)c.rows_m * g.lda * 2u);
#pragma unroll
            for (int p = 0; p < PA; p++) st.a[p] = buf_load_u4(rsA1, (voffA1 + p * strA1) | kmask, soffA);
        }
        const rsrc_t rsW = make_rsrc(Wp + (size_t)c.n0 * g.K, (unsigned)min(BN, g.N - c.n0) * g.K * 2u);
#pragma unroll
        for (int p = 0; p < PW; p++)
            st.w[p] = buf_load_u4(rsW, (BN % RPP == 0 || sr + RPP * p < BN) ? ((voffW + p * strW) | kmask) : 0x7fffffffu, (unsigned)k0 * 2u);
    };
    auto write = [&](const Stage &st) {
        if (DSG_BX_EXP == 2) return;
        __bf16 *As = lds, *Ws = lds + BM * LDP;
#pragma unroll
        for (int p = 0; p < PA; p++) *reinterpret_cast<u32x4 *>(As + (sr + RPP * p) * LDP + 8 * sc) = st.a[p];
#pragma unroll
        for (int p = 0; p < PW; p++)
            if (BN % RPP == 0 || sr + RPP * p < BN) *reinterpret_cast<u32x4 *>(Ws + (sr + RPP * p) * LDP + 8 * sc) = st.w[p];
    };

    f32x16 acc[MT][3];
    const __bf16 *Afr = lds + (wm * WR + lrow) * LDP + 8 * lhalf;
    const __bf16 *Wfr = lds + (BM + wn * 96 + lrow) * LDP + 8 * lhalf;
    auto compute = [&]() {
        if (DSG_BX_EXP == 3) return;
#pragma unroll
        for (int s = 0; s < KB / 16; s++) {
            bf16x8 af[MT], wf[3];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) af[mt] = *reinterpret_cast<const bf16x8 *>(Afr + 32 * mt * LDP + 16 * s);
#pragma unroll
            for (int nt = 0; nt < 3; nt++) wf[nt] = *reinterpret_cast<const bf16x8 *>(Wfr + 32 * nt * LDP + 16 * s);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 3; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    };

    // ---- epilogue of one tile: lane (m = lrow, half) owns output rows m0 + 64 wm + 32 mt + lrow; accumulator quad q of tile nt holds
    // its columns n0 + 96 wn + 32 nt + 8 q + 4 half + {0..3}.  No per-element branches (they would cut the code into basic blocks with
    // one exposed memory round trip each): columns >= N carry zeros (W rows beyond N read as zero, the column vectors are zero / one
    // there) and are dropped by out-of-range store offsets; the residual of group i + 1 is in flight while group i is processed.
    // (The reference adds the residual AFTER bias and activation; here it joins the accumulator first -- only the fp32 rounding of
    // the three-term sum changes; GELU outputs never carry a residual.)
    auto epilogue = [&](const Tile &c) {
        const int m0 = c.m0, n0 = c.n0;
        const unsigned OOB = 0x7fffffffu;
        const rsrc_t rsC = make_rsrc(g.C ? g.C + (size_t)m0 * g.ldc : nullptr, g.C ? (unsigned)c.rows_m * g.ldc * 4u : 0u);
        const rsrc_t rsR = make_rsrc(g.res ? g.res + (size_t)m0 * g.ldres : nullptr, g.res ? (unsigned)c.rows_m * g.ldres * 4u : 0u);
        __bf16 *Cbp = static_cast<__bf16 *>(g.Cb), *C2p = static_cast<__bf16 *>(g.C2b);
        const rsrc_t rsCb = make_rsrc(Cbp ? Cbp + (size_t)m0 * g.ldcb : nullptr, Cbp ? (unsigned)c.rows_m * g.ldcb * 2u : 0u);
        const rsrc_t rsC2 = make_rsrc(C2p ? C2p + (size_t)m0 * g.ldc2b : nullptr, C2p ? (unsigned)c.rows_m * g.ldc2b * 2u : 0u);
        const bool ln = g.ln_out != 0, gelu = g.act == ACT_GELU;
        const int ncol0 = wn * 96 + 4 * lhalf;          // + 32 nt + 8 q: this lane's column inside the tile
        // bf16 outputs leave through a wave-private transposition buffer in the (now idle) tile stage: in accumulator order one store
        // instruction would write 32 rows x 16 B -- the chip absorbs that footprint at 3.4 TB/s, row-contiguous 16-B pieces at 6.6
        // (tools/store_pattern.cpp, profiles/r3/store_pattern.txt).  A wave collects a 32 x 96 half tile (8 B per lane and quad),
        // then writes it out as 6 x 64 pieces of 16 B: 12 consecutive lanes cover one 192-B row segment.
        __bf16 *T = lds + wave * 32 * TLD;
        auto tput = [&](int nt, int q, const f32x4 &v) {
            *reinterpret_cast<u32x2 *>(T + lrow * TLD + 4 * lhalf + 32 * nt + 8 * q) = pack_bf16x4(v);
        };
        int lane_e = lane;                               // opaque copy: the piece addresses below are rebuilt per tile instead of
        asm volatile("" : "+v"(lane_e));                 // being hoisted out of the persistent loop into registers the K loop needs
        auto tflush = [&](const rsrc_t &rs, unsigned ld, int mt) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int i = lane_e + 64 * k, r = i / 12, p = i - 12 * r;
                const u32x4 d = *reinterpret_cast<const u32x4 *>(T + r * TLD + 8 * p);
                const int n = n0 + wn * 96 + 8 * p;
                buf_store_u4(d, rs, n < g.N ? ((unsigned)(wm * WR + 32 * mt + r) * ld + (unsigned)n) * 2u : OOB, 0u);
            }
        };
        const bool t_c2 = C2p != nullptr, t_cb = Cbp && !ln && !t_c2;   // one stream at a time goes through T inside the group loop
        // group = (mt, nt, qh): two accumulator quads = 8 values per lane at a time (register pressure: the next chunk is in flight)
        auto load_res = [&](f32x4 (&rr)[2], int grp) {
            const unsigned mrow = (unsigned)(wm * WR + 32 * (grp / 6) + lrow);
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int n = n0 + ncol0 + 32 * ((grp >> 1) % 3) + 8 * (2 * (grp & 1) + q);
                rr[q] = buf_load4(rsR, n < g.N ? (mrow * g.ldres + (unsigned)n) * 4u : OOB, 0u);
            }
        };
        f32x4 rra[2], rrb[2];
        if (RES) load_res(rra, 0);
        float ssum[MT], ssq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) { ssum[mt] = 0.f; ssq[mt] = 0.f; }
#pragma unroll
        for (int grp = 0; grp < 6 * MT; grp++) {
            const int mt = grp / 6, nt = (grp >> 1) % 3, q0 = 2 * (grp & 1);
            const unsigned mrow = (unsigned)(wm * WR + 32 * mt + lrow);
            if (RES && grp + 1 < 6 * MT) { if (grp & 1) load_res(rra, grp + 1); else load_res(rrb, grp + 1); }
            f32x4 v[2], scl[2], sft[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int nc = ncol0 + 32 * nt + 8 * (q0 + q);
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(colv + nc);
#pragma unroll
                for (int t = 0; t < 4; t++) v[q][t] = acc[mt][nt][4 * (q0 + q) + t] + b4[t];
                if (MOD == 1) {
                    scl[q] = *reinterpret_cast<const f32x4 *>(colv + BN + nc);
                    sft[q] = *reinterpret_cast<const f32x4 *>(colv + 2 * BN + nc);
                }
            }
            if (MOD == 2) {   // per-sample (scale, shift): test / training-time forwards only (the sampler's sigma is batch-uniform)
                const float *aff_row = g.mod_aff + (size_t)(min(m0 + (int)mrow, g.M - 1) / g.mod_T) * g.mod_ld + g.mod_off;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int n = min(n0 + ncol0 + 32 * nt + 8 * (q0 + q), g.N - 4);
                    scl[q] = *reinterpret_cast<const f32x4 *>(aff_row + n) + 1.0f;
                    sft[q] = *reinterpret_cast<const f32x4 *>(aff_row + g.N + n);
                }
            }
            if (gelu) {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int t = 0; t < 4; t += 2) { const f32x2_t gg = gelu_f2(v[q][t], v[q][t + 1]); v[q][t] = gg[0]; v[q][t + 1] = gg[1]; }
            }
            if (RES) {
#pragma unroll
                for (int q = 0; q < 2; q++) v[q] += (grp & 1) ? rrb[q] : rra[q];
            }
            if (t_c2) {
#pragma unroll
                for (int q = 0; q < 2; q++) tput(nt, q0 + q, v[q]);
                if (grp % 6 == 5) tflush(rsC2, (unsigned)g.ldc2b, mt);
            }
            if (MOD != 0) {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int t = 0; t < 4; t++) v[q][t] = silu_exact(fmaf(v[q][t], scl[q][t], sft[q][t]));
            }
            if (g.C) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int n = n0 + ncol0 + 32 * nt + 8 * (q0 + q);
                    buf_store4(v[q], rsC, n < g.N ? (mrow * g.ldc + (unsigned)n) * 4u : OOB, 0u);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    ssum[mt] += v[q][t]; ssq[mt] = fmaf(v[q][t], v[q][t], ssq[mt]);
                    acc[mt][nt][4 * (q0 + q) + t] = v[q][t];
                }
            if (t_cb) {
#pragma unroll
                for (int q = 0; q < 2; q++) tput(nt, q0 + q, v[q]);
                if (grp % 6 == 5) tflush(rsCb, (unsigned)g.ldcb, mt);
            } else if (Cbp && !ln) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int n = n0 + ncol0 + 32 * nt + 8 * (q0 + q);
                    buf_store2(pack_bf16x4(v[q]), rsCb, n < g.N ? (mrow * g.ldcb + (unsigned)n) * 2u : OOB, 0u);
                }
            }
        }
        if (ln) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const unsigned mrow = (unsigned)(wm * WR + 32 * mt + lrow);
                ssum[mt] += __shfl_xor(ssum[mt], 32, 64);
                ssq[mt] += __shfl_xor(ssq[mt], 32, 64);
                if (lhalf == 0) part[mrow * WN + wn] = (f32x2){ssum[mt], ssq[mt]};
            }
            __syncthreads();
            const float invn = 1.0f / (float)g.N;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const unsigned mrow = (unsigned)(wm * WR + 32 * mt + lrow);
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WN; w2++) { const f32x2 p2 = part[mrow * WN + w2]; sm += p2[0]; sq += p2[1]; }
                const float mean = sm * invn, rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * invn), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
                for (int nt = 0; nt < 3; nt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 v;
#pragma unroll
                        for (int t = 0; t < 4; t++) v[t] = fmaf(acc[mt][nt][4 * q + t], rstd, nmr);
                        tput(nt, q, v);
                    }
                tflush(rsCb, (unsigned)g.ldcb, mt);
            }
        }
    };

    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};   // DSG_BX_EXP == 4: shader clocks in MFMA section / barrier 1 / wait + LDS refill / barrier 2 / epilogue, tiles
#define BX_STAMP() (DSG_BX_EXP == 4 ? __builtin_amdgcn_s_memtime() : 0ull)
    const unsigned long long tstart = BX_STAMP();
    Stage s0, s1;
    for (int t = next_id((int)blockIdx.x); t < tiles_total; t = next_id(t + (int)gridDim.x)) {
        const Tile cur = make_tile(t);
        issue(s0, cur, 0);
        issue(s1, cur, 1);
        float cv[3][CV];   // the tile's column vectors: loaded now, written to LDS just before the epilogue
#pragma unroll
        for (int i = 0; i < CV; i++) {
            const int col = tid + NT * i, n = cur.n0 + col;
            const bool ok = col < BN && n < g.N;
            cv[0][i] = (g.bias && ok) ? g.bias[n] : 0.f;
            cv[1][i] = (MOD == 1 && ok) ? g.mod_aff[g.mod_off + n] + 1.0f : 1.0f;
            cv[2][i] = (MOD == 1 && ok) ? g.mod_aff[g.mod_off + g.N + n] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
        write(s0);
        __syncthreads();
        // invariant at the top of an iteration: LDS holds chunk kc, s1 holds chunk kc + 1 (possibly still in flight), s0 is free
        for (int kc = 0; kc < nk; kc += 2) {
            unsigned long long ta = BX_STAMP();
            if (kc + 2 < nk) issue(s0, cur, kc + 2);
            compute();
            unsigned long long tb = BX_STAMP();
            __syncthreads();                 // every wave is done reading chunk kc
            unsigned long long tc = BX_STAMP();
            ph[0] += tb - ta; ph[1] += tc - tb;
            if (kc + 1 < nk) {
                write(s1);
                if (DSG_BX_EXP == 4) __builtin_amdgcn_s_waitcnt(0xc07f);
                ta = BX_STAMP();
                __syncthreads();
                tb = BX_STAMP();
                ph[2] += ta - tc; ph[3] += tb - ta;
                if (kc + 3 < nk) issue(s1, cur, kc + 3);
                compute();
                tc = BX_STAMP();
                __syncthreads();
                ta = BX_STAMP();
                ph[0] += tc - tb; ph[1] += ta - tc;
                if (kc + 2 < nk) {
                    write(s0);
                    if (DSG_BX_EXP == 4) __builtin_amdgcn_s_waitcnt(0xc07f);
                    tb = BX_STAMP();
                    __syncthreads();
                    ph[2] += tb - ta; ph[3] += BX_STAMP() - tb;
                }
            }
        }
        const unsigned long long te = BX_STAMP();
        if (DSG_BX_EXP == 1) {           // keep the accumulators alive, store nothing of substance
            float keep = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 3; nt++) keep += acc[mt][nt][0];
            if (keep == 123.456f && g.C) g.C[0] = keep;
        } else {
#pragma unroll
            for (int i = 0; i < CV; i++) {
                const int col = tid + NT * i;
                if (col < BN) { colv[col] = cv[0][i]; colv[BN + col] = cv[1][i]; colv[2 * BN + col] = cv[2][i]; }
            }
            __syncthreads();
            epilogue(cur);
            __syncthreads();             // colv / part / the tile stage are rewritten by the next tile
        }
        if (DSG_BX_EXP == 4) { __builtin_amdgcn_s_waitcnt(0x0070); ph[4] += BX_STAMP() - te; ph[5] += 1; }
    }
    if (DSG_BX_EXP == 4 && g.dbg && tid == 0) {
        unsigned long long *d = g.dbg + 8 * (size_t)blockIdx.x;
#pragma unroll
        for (int i = 0; i < 6; i++) d[i] = ph[i];
        d[6] = BX_STAMP() - tstart;
        d[7] = __builtin_amdgcn_s_memrealtime();
    }
#undef BX_STAMP
}

static int round_up8(int x) { return (x + 7) / 8 * 8; }

static int bx_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

bool launch_gemm_bx(const BxGemm &g, hipStream_t s) {
    if (!g.A || !g.W || g.M < 1 || g.N < 1 || g.K < 8 || g.K % 8 != 0 || g.N % 4 != 0 || (g.act != ACT_NONE && g.act != ACT_GELU)) return false;
    if (g.lda % 8 != 0 || (g.A2 && (g.lda2 % 8 != 0 || g.K1 <= 0))) return false;
    if ((g.Cb || g.C2b) && (g.N % 8 != 0 || (g.Cb && g.ldcb % 8 != 0) || (g.C2b && g.ldc2b % 8 != 0))) return false;   // bf16 outputs leave in 16-byte row pieces
    if (g.act == ACT_GELU && g.res) return false;   // the epilogue adds the residual before bias / activation (fc1 has none)
    // geometry (wave grid WM x WN, block tile 64 WM x 96 WN): the tile spans the whole row when a LayerNorm output is asked for;
    // otherwise 128 x 192 where N splits into 192s, 256 x 96 for the narrow / odd widths (N = 96, 288)
    //   0: <2,2,64> 128x192 (4 waves, 2 blocks / CU)   2: <4,1,32> 256x96 (4 waves, 2 blocks / CU)   3: <2,4,64> 128x384 (8 waves)
    // (measured against 64x384 on 4 waves with 64- and 32-deep chunks and 256x192 on 8 waves: profiles/r3/bx_experiments.txt)
    int geo;
    if (g.ln_out) {
        if (!g.Cb) return false;
        if (g.N == 96) geo = 2; else if (g.N == 192) geo = 0; else if (g.N == 384) geo = 3; else return false;
    } else {
        geo = (g.N % 192 == 0) ? 0 : 2;
    }
    const int kb = geo == 2 ? 32 : 64;
    if (g.A2 && g.K1 % kb != 0) return false;
    if (geo == 2 && g.K % 32 != 0) return false;
    if ((g.K + kb - 1) / kb < 2) return false;      // the chunk loop assumes at least two chunks per tile
    const int bm = geo == 0 ? 128 : (geo == 2 ? 256 : 128), bn = geo == 0 ? 192 : (geo == 2 ? 96 : 384);
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = (g.N + bn - 1) / bn;
    const int tiles_total = round_up8(tiles_m) * tiles_n;
    // persistent grid: the blocks that are resident at once (2 per CU for the 4-wave geometries), a multiple of 8 so that a block's
    // tiles t, t + grid, ... stay in one XCD class
    const int resident = std::max(8, bx_cu_count() * (geo == 3 ? 1 : 2) / 8 * 8);
    const dim3 grid(std::min(tiles_total, resident));
    const int mod = !g.mod_aff ? 0 : (g.mod_ld == 0 ? 1 : 2);
#define BX_LAUNCH(WM_, WN_, KB_, NT_, MT_)                                                                                                 \
    do {                                                                                                                                   \
        if (g.res) {                                                                                                                       \
            if (mod == 0) DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, true, 0, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);        \
            else if (mod == 1) DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, true, 1, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);   \
            else DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, true, 2, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);                 \
        } else {                                                                                                                           \
            if (mod == 0) DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, false, 0, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);       \
            else if (mod == 1) DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, false, 1, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);  \
            else DSG_LAUNCH((gemm_bx_kernel<WM_, WN_, KB_, false, 2, MT_>), grid, dim3(NT_), 0, s, g, tiles_m, tiles_n, tiles_total);                \
        }                                                                                                                                  \
    } while (0)
    // (MT = 1 -- 32 x 96 wave tiles, ~125 registers, four waves per SIMD on the same block tiles -- measured within 3 % of these on
    //  every shape, slower on the long-K ones: profiles/r3/bx_experiments.txt; not instantiated)
    switch (geo) {
        case 0: BX_LAUNCH(2, 2, 64, 256, 2); break;
        case 2: BX_LAUNCH(4, 1, 32, 256, 2); break;
        default: BX_LAUNCH(2, 4, 64, 512, 2); break;
    }
#undef BX_LAUNCH
    return true;
}

// -------------------------------------------------------------------------------------------------
// Fused MLP half of a Swin block (diffusesg.py:275, :19-25):  x <- x + fc2(GELU(fc1(LayerNorm2(x))))  with the 4C-wide hidden tensor
// never leaving the register file -- unfused, fc1 writes and fc2 re-reads 2 x M x 4C bytes of it (more than everything else the two
// GEMMs move) and fc1's epilogue is as long as its MFMAs.
//   * one wave owns 32 tokens and keeps their normalised rows (the B operand of fc1: lane = token) in registers for the whole kernel;
//   * the hidden dimension is walked in chunks of 32: H^T = W1[chunk] . Xn^T on the matrix pipe (lane = token, register = hidden unit),
//     + b1, GELU and the bf16 rounding on the accumulator, which is then -- register pairs as they stand -- the B operand of
//     O^T += W2[:, chunk] . H^T (an accumulator tile's rows are a k-step's k index in the permuted order hidden 16 s + 8 (j >> 2) +
//     4 half + (j & 3); the W2 chunk is laid out in LDS in that order);
//   * the two weight chunks (32 x C and C x 32 bf16) stream through a double-buffered LDS stage shared by the block's 4 waves, the
//     next chunk's global loads in flight during the current chunk's MFMAs;
//   * a lane ends up with whole rows (its token's C outputs, split between the half-waves), so bias, residual, the next block's
//     modulate+SiLU and the LayerNorm of the stored row are lane-local: the same epilogue options as gemm_bx at any width.
// Registers: C/4 for Xn, C/2 for O^T (+ staging): C = 384 runs one wave per SIMD (512 registers), C = 192 two, C = 96 three.
// -------------------------------------------------------------------------------------------------
// PROJ: the attention half's tail rides in front -- x1 = x + proj(att) + bp is formed on the matrix pipe FIRST (att [M, C] bf16 as the B
// operand, the proj weight streamed through the same LDS stage in CT chunks of 32 output channels) and lands in the O^T accumulators,
// i.e. it is the initial value of the fc2 accumulation: x1 never goes to HBM (unfused: proj writes it, this kernel re-reads it), neither
// does LayerNorm-2 -- its statistics are lane-local sums of the accumulators and the normalised row, packed pairwise to bf16 AS THE
// REGISTERS STAND, is fc1's B operand (k order 16 s + 8 (j >> 2) + 4 half + (j & 3); the W1 chunk is laid out in LDS in that order).
#ifndef DSG_MLPB_EXP
#define DSG_MLPB_EXP 0   // timing experiments of mlp_bx_kernel (wrong results): 1 no GELU, 2 no MFMAs in the chunk loop, 3 no weight staging in the chunk loop, 4 no fp32 row stores, 5 no fp32 row loads
#endif
// (Tried in round 4, not kept: C = 192 as ONE eight-wave block of 256 tokens per CU instead of two four-wave blocks of 128 -- the same eight
//  waves, half the weight staging per token: 272 us against 238 (copy), 301 against 242 (modulate + LayerNorm).  Two independent blocks
//  overlap their barriers and HBM phases; one block of eight waves marches in step.)
template <int C, int MOD, bool PROJ = false>
__global__ __launch_bounds__(256, (C <= 96 ? 3 : (C <= 192 ? 2 : 1))) void mlp_bx_kernel(BxMlp g) {
    constexpr int H = 4 * C, NCH = H / 32, KS = C / 16, CT = C / 32;
    constexpr int LD1 = C + 8, LD2 = 40;                       // LDS row strides (bf16) of the W1 chunk [32][C] and the W2 chunk [C][32]
    constexpr int STAGE = 32 * LD1 + C * LD2 + 64;             // + 32 floats of b1
    constexpr int NP = (4 * C + 255) / 256;                    // 16-byte pieces per thread and chunk, for each of the two weights
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int rows = min(128, g.M - m0);
    const unsigned mrow = (unsigned)(32 * wave + lrow);        // this lane's token inside the block
    const __bf16 *W1 = static_cast<const __bf16 *>(g.W1), *W2 = static_cast<const __bf16 *>(g.W2);
    const rsrc_t rsW1 = make_rsrc(W1, (unsigned)H * C * 2u), rsW2 = make_rsrc(W2, (unsigned)C * H * 2u);
    // staging: piece q = tid + 256 p;  W1 chunk: row q / (C/8), 16-byte column q % (C/8);  W2 chunk: row q / 4, column q % 4
    u32x4 s1[NP], s2[NP];
    float s_b1 = 0.f;
    auto issue = [&](int hc) {
        if (tid < 32) s_b1 = g.b1[32 * hc + tid];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int q = tid + 256 * p;
            const bool ok = q < 4 * C;
            const unsigned o1 = ((unsigned)(32 * hc + q / (C / 8)) * C + 8u * (q % (C / 8))) * 2u;
            const unsigned o2 = ((unsigned)(q / 4) * H + 32u * hc + 8u * (q % 4)) * 2u;
            s1[p] = buf_load_u4(rsW1, ok ? o1 : 0x7fffffffu, 0u);
            s2[p] = buf_load_u4(rsW2, ok ? o2 : 0x7fffffffu, 0u);
        }
    };
    auto write = [&](int buf) {
        __bf16 *w1s = lds + buf * STAGE, *w2s = w1s + 32 * LD1;
        float *b1s = reinterpret_cast<float *>(w2s + C * LD2);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int q = tid + 256 * p;
            if (q < 4 * C) {
                if (PROJ) {   // channels 8 c8 .. + 7 of a row go to positions 16 (c8 >> 1) + 8 (e >> 2) + 4 (c8 & 1) + (e & 3)
                    const int c8 = q % (C / 8);
                    __bf16 *d1 = w1s + (q / (C / 8)) * LD1 + 16 * (c8 >> 1) + 4 * (c8 & 1);
                    *reinterpret_cast<u32x2 *>(d1) = (u32x2){s1[p][0], s1[p][1]};
                    *reinterpret_cast<u32x2 *>(d1 + 8) = (u32x2){s1[p][2], s1[p][3]};
                } else {
                    *reinterpret_cast<u32x4 *>(w1s + (q / (C / 8)) * LD1 + 8 * (q % (C / 8))) = s1[p];
                }
                // the 8 hidden units 8 c .. 8 c + 7 of the chunk go to positions 16 s + 8 (e >> 2) + 4 (c & 1) + (e & 3), s = c >> 1
                const int c = q % 4;
                __bf16 *dst = w2s + (q / 4) * LD2 + 16 * (c >> 1) + 4 * (c & 1);
                *reinterpret_cast<u32x2 *>(dst) = (u32x2){s2[p][0], s2[p][1]};
                *reinterpret_cast<u32x2 *>(dst + 8) = (u32x2){s2[p][2], s2[p][3]};
            }
        }
        if (tid < 32) b1s[tid] = s_b1;
    };
    // (PROJ) chunk ct of the proj weight: rows 32 ct .. + 31 of Wp [C, C], natural column order, through s1 and the W1 half of a stage buffer
    const rsrc_t rsWp = make_rsrc(PROJ ? static_cast<const __bf16 *>(g.Wp) : nullptr, PROJ ? (unsigned)C * C * 2u : 0u);
    auto issue_p = [&](int ct) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int q = tid + 256 * p;
            s1[p] = buf_load_u4(rsWp, q < 4 * C ? ((unsigned)(32 * ct + q / (C / 8)) * C + 8u * (q % (C / 8))) * 2u : 0x7fffffffu, 0u);
        }
    };
    auto write_p = [&](int buf) {
        __bf16 *w1s = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int q = tid + 256 * p;
            if (q < 4 * C) *reinterpret_cast<u32x4 *>(w1s + (q / (C / 8)) * LD1 + 8 * (q % (C / 8))) = s1[p];
        }
    };
    if (PROJ) issue_p(0); else issue(0);
    // the wave's B operand rows: lane (token, half) holds channels 16 s + 8 half .. + 7 of k-step s -- of the normalised rows, or (PROJ)
    // of the attention output
    const rsrc_t rsXn = make_rsrc(static_cast<const __bf16 *>(PROJ ? g.att : g.xn) + (size_t)m0 * C, (unsigned)rows * C * 2u);
    bf16x8 xf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) xf[s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsXn, (mrow * C + 16u * s + 8u * lhalf) * 2u, 0u));
    f32x16 oacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;
    // Staging schedule (proj chunks, then hidden chunks: items 0, 1, ... alternate between the two stage buffers): ONE register set, the
    // item after next requested as soon as the next one has been written to LDS, and that write placed AFTER the barrier -- a wave that
    // reaches the barrier never waits for global loads in front of it, and a request has a whole item's products + the barrier to land.
    //   barrier | write item i + 1 (registers -> the buffer item i - 1 was read from) | request item i + 2 | products of item i
    if (PROJ) {
        write_p(0);
        if (CT > 1) issue_p(1); else issue(0);
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            __syncthreads();
            if (ct + 1 < CT) write_p((ct + 1) & 1); else write(CT & 1);
            if (ct + 2 < CT) issue_p(ct + 2); else if (ct + 2 - CT < NCH) issue(ct + 2 - CT);
            const __bf16 *wps = lds + (ct & 1) * STAGE;
#pragma unroll
            for (int s = 0; s < KS; s++)
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(wps + lrow * LD1 + 16 * s + 8 * lhalf), xf[s], oacc[ct], 0, 0, 0);
        }
        // x1 = x + proj + bp (the shortcut and fc2's initial value), LayerNorm-2 of it -> fc1's B operand
        const rsrc_t rsXi = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            f32x4 rr[4];
#pragma unroll
            for (int q = 0; q < 4; q++) rr[q] = DSG_MLPB_EXP == 5 ? (f32x4){0.1f, 0.2f, 0.3f, 0.4f} : buf_load4(rsXi, (mrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.bp + 32 * ct + 8 * q + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = oacc[ct][4 * q + e] + b4[e] + rr[q][e];
                    oacc[ct][4 * q + e] = v; sm += v; sq = fmaf(v, v, sq);
                }
            }
        }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float mean = sm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; j++)
                pk[j] = pack_bf16(fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j], rstd, nmr), fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j + 1], rstd, nmr));
            xf[s] = __builtin_bit_cast(bf16x8, pk);
        }
    } else {
        write(0);
        issue(1);
    }
    for (int hc = 0; hc < NCH; hc++) {
        const int cur = (hc + (PROJ ? CT : 0)) & 1;
        __syncthreads();
        if (hc + 1 < NCH && DSG_MLPB_EXP != 3) {
            write(1 - cur);                     // chunk hc + 1 (requested one item ago; PROJ: chunk 1 by the last proj item)
            if (hc + 2 < NCH) issue(hc + 2);
        }
        const __bf16 *w1s = lds + cur * STAGE, *w2s = w1s + 32 * LD1;
        const float *b1s = reinterpret_cast<const float *>(w2s + C * LD2);
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; r++) hacc[r] = b1s[(r & 3) + 8 * (r >> 2) + 4 * lhalf];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (DSG_MLPB_EXP == 2) { hacc[s & 15] += (float)xf[s][0]; continue; }
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w1s + lrow * LD1 + 16 * s + 8 * lhalf), xf[s], hacc, 0, 0, 0);
        }
        u32x4 hf[2];
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2_t gg = DSG_MLPB_EXP == 1 ? (f32x2_t){hacc[r], hacc[r + 1]} : gelu_f2(hacc[r], hacc[r + 1]); hf[r >> 3][(r & 7) >> 1] = pack_bf16(gg[0], gg[1]); }
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int s2i = 0; s2i < 2; s2i++)
                if (DSG_MLPB_EXP == 2) oacc[ct][s2i] += __uint_as_float(hf[s2i][0]); else
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w2s + (32 * ct + lrow) * LD2 + 16 * s2i + 8 * lhalf),
                                                                   __builtin_bit_cast(bf16x8, hf[s2i]), oacc[ct], 0, 0, 0);
    }
    __syncthreads();   // the stages are free for the output transposition
    // ---- epilogue: lane (token, half) holds channels 32 ct + 8 q + 4 half + {0..3} in oacc[ct][4 q ..]
    const rsrc_t rsX = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
    __bf16 *xo = static_cast<__bf16 *>(g.xn_out);
    const rsrc_t rsO = make_rsrc(xo ? xo + (size_t)m0 * C : nullptr, xo ? (unsigned)rows * C * 2u : 0u);
    const float *aff_row = nullptr;
    if (MOD != 0) aff_row = g.mod_aff + (size_t)(MOD == 2 ? min(m0 + (int)mrow, g.M - 1) / g.mod_T : 0) * g.mod_ld + g.mod_off;
    // bf16 outputs leave through a wave-private 32 x 96 transposition in the (idle) weight stages, as in gemm_bx_kernel: row-contiguous
    // 16-byte pieces instead of 32 rows x 16 B per store instruction
    constexpr int TLD = 104;
    static_assert(4 * 32 * TLD <= 2 * STAGE, "the output transposition reuses the weight stages");
    __bf16 *T = lds + wave * 32 * TLD;
    auto tput = [&](int ct, int q, const f32x4 &v) { *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v); };
    auto tflush = [&](int third) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k, r = i / 12, pc = i - 12 * r;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(T + r * TLD + 8 * pc);
            buf_store_u4(d, rsO, ((unsigned)(32 * wave + r) * C + (unsigned)(96 * third + 8 * pc)) * 2u, 0u);
        }
    };
    // The fp32 rows leave through a wave-private 32 x 32 tile as well (round 4: tools/mlpb_exp.sh -- the direct form, where a store
    // instruction touches 32 rows x 32 bytes, cost 36 of 343 us at C = 96): a store instruction then writes 8 rows x one whole 128-byte
    // line.  The tile aliases T (used after this loop: the bf16 output is formed from the values kept in oacc).
    constexpr int XLD = 36;
    float *xt = reinterpret_cast<float *>(T);
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        f32x4 rr[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            rr[q] = PROJ ? (f32x4){0.f, 0.f, 0.f, 0.f} : (DSG_MLPB_EXP == 5 ? (f32x4){0.1f, 0.2f, 0.3f, 0.4f} : buf_load4(rsX, (mrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u));   // (PROJ: x1 is in oacc)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 32 * ct + 8 * q + 4 * lhalf;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.b2 + c);
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = oacc[ct][4 * q + t] + b4[t] + rr[q][t];
            if (MOD != 0) {
                const f32x4 scl = *reinterpret_cast<const f32x4 *>(aff_row + c), sft = *reinterpret_cast<const f32x4 *>(aff_row + C + c);
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = silu_exact(fmaf(v[t], scl[t] + 1.0f, sft[t]));
            }
            *reinterpret_cast<f32x4 *>(xt + lrow * XLD + 8 * q + 4 * lhalf) = v;
#pragma unroll
            for (int t = 0; t < 4; t++) { ssum += v[t]; ssq = fmaf(v[t], v[t], ssq); oacc[ct][4 * q + t] = v[t]; }
        }
        if (DSG_MLPB_EXP != 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = 8 * k + (lane >> 3), pc = lane & 7;
                const f32x4 d = *reinterpret_cast<const f32x4 *>(xt + r * XLD + 4 * pc);
                buf_store4(d, rsX, ((unsigned)(32 * wave + r) * C + (unsigned)(32 * ct + 4 * pc)) * 4u, 0u);
            }
        }
    }
    if (g.out_mode == 2) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = oacc[ct][4 * q + t];
                tput(ct, q, v);
                if (ct % 3 == 2 && q == 3) tflush(ct / 3);
            }
    }
    if (g.out_mode == 1) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        const float mean = ssum * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, ssq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = fmaf(oacc[ct][4 * q + t], rstd, nmr);
                tput(ct, q, v);
                if (ct % 3 == 2 && q == 3) tflush(ct / 3);
            }
    }
}

// -------------------------------------------------------------------------------------------------
// The same fused MLP at C = 384 on EIGHT waves (two per SIMD).  One wave with a whole 384-wide row needs 96 + 192 registers for Xn and
// O^T alone, runs one wave per SIMD and loses to the pair of GEMMs (mlp_bx_kernel<384>: 306 us against 110 + 149 at COCO level 2); here a
// PAIR of waves shares 32 tokens:
//   * fc1 is split along K: wave kh of the pair keeps channels 192 kh .. + 191 of the normalised rows (48 registers) and forms the
//     partial products of BOTH 32-wide hidden chunks of a chunk pair; the partial sum of the partner's chunk goes to the partner through
//     LDS (fp32), the own chunk is completed, + b1, GELU, bf16 -- every hidden unit passes through GELU once;
//   * the two bf16 hidden tiles are exchanged through LDS, and fc2 is split along N: wave kh accumulates output channels
//     192 kh .. + 191 (96 registers) over both chunks;
//   * the weights stream through ONE LDS stage per matrix (W1 pair 64 x 384, W2 pair 384 x 64), refilled from registers in the phase in
//     which the other matrix is being read (W2 during fc1, W1 during GELU / fc2); three barriers per chunk pair.
// The epilogue is mlp_bx_kernel's with the LayerNorm statistics added across the pair.
// -------------------------------------------------------------------------------------------------
// PROJ (as mlp_bx_kernel): x1 = x + att Wp^T + bp first, split along N like fc2 -- wave kh forms output channels 192 kh .. + 191 of its 32
// tokens from the whole attention row (96 registers, dead afterwards) -- which are exactly the channels of its K half of fc1 and of its
// fc2 accumulators: the LayerNorm statistics cross the pair through LDS, nothing else moves.
#ifndef DSG_MLP_EXP
#define DSG_MLP_EXP 0   // timing experiments (wrong results): 1 no GELU, 2 no MFMAs in the chunk-pair loop, 3 no weight staging (loads / LDS refills)
#endif
template <int MOD, bool PROJ = false>
__global__ __launch_bounds__(512, 1) void mlp384_bx_kernel(BxMlp g) {
    constexpr int C = 384, H = 4 * C, NPAIR = H / 64, CT = 6;
    constexpr int LD1 = C + 8, LD2 = 64 + 8, TLD = 104;
    constexpr int W1S = 64 * LD1, W2S = C * LD2, B1S = 2 * 64 * 2, X1S = 8 * 64 * 16 * 2, X2S = 8 * 64 * 16;   // bf16 elements
    static_assert(8 * 32 * TLD <= W1S + W2S, "the output transposition reuses the weight stages");
    __shared__ __attribute__((aligned(16))) __bf16 lds[W1S + W2S + B1S + X1S + X2S];
    __bf16 *w1s = lds, *w2s = lds + W1S;
    float *b1s = reinterpret_cast<float *>(lds + W1S + W2S);                    // [2][64]
    f32x4 *xch1 = reinterpret_cast<f32x4 *>(lds + W1S + W2S + B1S);             // [4 quads][8 waves][64 lanes]
    u32x4 *xch2 = reinterpret_cast<u32x4 *>(lds + W1S + W2S + B1S + X1S);       // [2 k-steps][8 waves][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int ts = wave >> 1, kh = wave & 1;
    const int m0 = blockIdx.x * 128;
    const int rows = min(128, g.M - m0);
    const unsigned mrow = (unsigned)(32 * ts + lrow);
    const __bf16 *W1 = static_cast<const __bf16 *>(g.W1), *W2 = static_cast<const __bf16 *>(g.W2);
    const rsrc_t rsW1 = make_rsrc(W1, (unsigned)H * C * 2u), rsW2 = make_rsrc(W2, (unsigned)C * H * 2u);
    // staging pieces of this thread: q = tid + 512 p, p < 6.  W1 pair: row q / 48, 16-byte column q % 48; W2 pair: row q / 8, piece q % 8
    u32x4 st[6];
    float s_b1 = 0.f;
    auto issue_w1 = [&](int cp) {
        if (DSG_MLP_EXP == 3 && cp > 0) return;
        if (tid < 64) s_b1 = g.b1[64 * cp + tid];
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p;
            st[p] = buf_load_u4(rsW1, ((unsigned)(64 * cp + q / 48) * C + 8u * (q % 48)) * 2u, 0u);
        }
    };
    auto write_w1 = [&](int cp) {
        if (DSG_MLP_EXP == 3 && cp > 0) return;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p;
            if (PROJ) {   // fc1's B operand is the accumulator order: channels 8 c8 .. + 7 -> positions 16 (c8 >> 1) + 8 (e >> 2) + 4 (c8 & 1) + (e & 3)
                const int c8 = q % 48;
                __bf16 *d1 = w1s + (q / 48) * LD1 + 16 * (c8 >> 1) + 4 * (c8 & 1);
                *reinterpret_cast<u32x2 *>(d1) = (u32x2){st[p][0], st[p][1]};
                *reinterpret_cast<u32x2 *>(d1 + 8) = (u32x2){st[p][2], st[p][3]};
            } else {
                *reinterpret_cast<u32x4 *>(w1s + (q / 48) * LD1 + 8 * (q % 48)) = st[p];
            }
        }
        if (tid < 64) b1s[64 * (cp & 1) + tid] = s_b1;
    };
    // (PROJ) stage ct of the proj weight: rows 32 ct .. + 31 (the kh = 0 waves' output channels) and 192 + 32 ct .. + 31 (kh = 1), natural order
    const rsrc_t rsWp = make_rsrc(PROJ ? static_cast<const __bf16 *>(g.Wp) : nullptr, PROJ ? (unsigned)C * C * 2u : 0u);
    auto issue_p = [&](int ct) {
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p, r = q / 48;
            st[p] = buf_load_u4(rsWp, ((unsigned)((r < 32 ? 32 * ct + r : 160 + 32 * ct + r)) * C + 8u * (q % 48)) * 2u, 0u);
        }
    };
    auto write_p = [&]() {
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p;
            *reinterpret_cast<u32x4 *>(w1s + (q / 48) * LD1 + 8 * (q % 48)) = st[p];
        }
    };
    auto issue_w2 = [&](int cp) {
        if (DSG_MLP_EXP == 3 && cp > 0) return;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p;
            st[p] = buf_load_u4(rsW2, ((unsigned)(q / 8) * H + 64u * cp + 8u * (q % 8)) * 2u, 0u);
        }
    };
    auto write_w2 = [&]() {
        if (DSG_MLP_EXP == 3) return;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int q = tid + 512 * p, c = q & 7, c4 = c & 3;
            // hidden units 8 c4 .. + 7 of chunk c >> 2 go to positions 16 (c4 >> 1) + 8 (e >> 2) + 4 (c4 & 1) + (e & 3) of the chunk
            __bf16 *dst = w2s + (q >> 3) * LD2 + 32 * (c >> 2) + 16 * (c4 >> 1) + 4 * (c4 & 1);
            *reinterpret_cast<u32x2 *>(dst) = (u32x2){st[p][0], st[p][1]};
            *reinterpret_cast<u32x2 *>(dst + 8) = (u32x2){st[p][2], st[p][3]};
        }
    };
    bf16x8 xf[12];
    f32x16 oacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;
    if (PROJ) {
        issue_p(0);
        const rsrc_t rsAt = make_rsrc(static_cast<const __bf16 *>(g.att) + (size_t)m0 * C, (unsigned)rows * C * 2u);
        bf16x8 af[24];   // the whole attention row of the lane's token: channels 16 s + 8 half .. + 7 of k-step s
#pragma unroll
        for (int s = 0; s < 24; s++) af[s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsAt, (mrow * C + 16u * s + 8u * lhalf) * 2u, 0u));
        const __bf16 *wpf = w1s + (32 * kh + lrow) * LD1 + 8 * lhalf;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            __syncthreads();                   // the previous stage has been read
            write_p();
            __syncthreads();
            if (ct + 1 < CT) issue_p(ct + 1); else issue_w1(0);
#pragma unroll
            for (int s = 0; s < 24; s++)
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(wpf + 16 * s), af[s], oacc[ct], 0, 0, 0);
        }
        // x1 = x + proj + bp on this wave's 192 channels; LayerNorm-2 statistics across the pair; the normalised values are fc1's B operand
        const rsrc_t rsXi = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            f32x4 rr[4];
#pragma unroll
            for (int q = 0; q < 4; q++) rr[q] = buf_load4(rsXi, (mrow * C + (unsigned)(192 * kh + 32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.bp + 192 * kh + 32 * ct + 8 * q + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = oacc[ct][4 * q + e] + b4[e] + rr[q][e];
                    oacc[ct][4 * q + e] = v; sm += v; sq = fmaf(v, v, sq);
                }
            }
        }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        f32x2 *part0 = reinterpret_cast<f32x2 *>(xch1);   // [128 rows][2]
        if (lhalf == 0) part0[mrow * 2 + kh] = (f32x2){sm, sq};
        __syncthreads();                       // statistics visible; the last proj stage has been read
        write_w1(0);
        issue_w2(0);
        const f32x2 q0 = part0[mrow * 2], q1 = part0[mrow * 2 + 1];
        const float tsm = q0[0] + q1[0], tsq = q0[1] + q1[1];
        const float mean = tsm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, tsq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int s = 0; s < 12; s++) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; j++)
                pk[j] = pack_bf16(fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j], rstd, nmr), fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j + 1], rstd, nmr));
            xf[s] = __builtin_bit_cast(bf16x8, pk);
        }
    } else {
        issue_w1(0);
        // the wave's half of the normalised rows: lane (token, half) holds channels 192 kh + 16 s + 8 half .. + 7 of k-step s
        const rsrc_t rsXn = make_rsrc(static_cast<const __bf16 *>(g.xn) + (size_t)m0 * C, (unsigned)rows * C * 2u);
#pragma unroll
        for (int s = 0; s < 12; s++) xf[s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsXn, (mrow * C + 192u * kh + 16u * s + 8u * lhalf) * 2u, 0u));
        write_w1(0);
        issue_w2(0);
    }
    const __bf16 *w1own = w1s + (32 * kh + lrow) * LD1 + 192 * kh + 8 * lhalf, *w1oth = w1s + (32 * (1 - kh) + lrow) * LD1 + 192 * kh + 8 * lhalf;
    const __bf16 *w2own = w2s + (192 * kh + lrow) * LD2 + 32 * kh + 8 * lhalf, *w2oth = w2s + (192 * kh + lrow) * LD2 + 32 * (1 - kh) + 8 * lhalf;
    for (int cp = 0; cp < NPAIR; cp++) {
        __syncthreads();                       // B1: fc2 of the previous pair is done (W2 stage free), W1 stage / b1 of this pair visible
        write_w2();
        if (cp + 1 < NPAIR) issue_w1(cp + 1);
        f32x16 ho, hx;                         // partial products of the own chunk (32 kh ..) and of the partner's chunk
#pragma unroll
        for (int r = 0; r < 16; r++) { ho[r] = 0.f; hx[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 12; s++) {
            if (DSG_MLP_EXP == 2) { ho[s] += (float)xf[s][0]; continue; }
            ho = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w1own + 16 * s), xf[s], ho, 0, 0, 0);
            hx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w1oth + 16 * s), xf[s], hx, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) xch1[(q * 8 + wave) * 64 + lane] = (f32x4){hx[4 * q], hx[4 * q + 1], hx[4 * q + 2], hx[4 * q + 3]};
        __syncthreads();                       // Bx: partial sums visible; nobody reads the W1 stage any more
        if (cp + 1 < NPAIR) { write_w1(cp + 1); issue_w2(cp + 1); }
        u32x4 hf[2];
        {
            const float *bb = b1s + 64 * (cp & 1) + 32 * kh + 4 * lhalf;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 pp = xch1[(q * 8 + (wave ^ 1)) * 64 + lane], b4 = *reinterpret_cast<const f32x4 *>(bb + 8 * q);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float t0 = ho[4 * q + e] + pp[e] + b4[e], t1 = ho[4 * q + e + 1] + pp[e + 1] + b4[e + 1];
                    const f32x2_t gg = DSG_MLP_EXP == 1 ? (f32x2_t){t0, t1} : gelu_f2(t0, t1);
                    v[e] = gg[0]; v[e + 1] = gg[1];
                }
                hf[q >> 1][2 * (q & 1)] = pack_bf16(v[0], v[1]);
                hf[q >> 1][2 * (q & 1) + 1] = pack_bf16(v[2], v[3]);
            }
        }
        xch2[(0 * 8 + wave) * 64 + lane] = hf[0];
        xch2[(1 * 8 + wave) * 64 + lane] = hf[1];
        __syncthreads();                       // B2: hidden tiles visible, W2 stage of this pair visible
        u32x4 hp[2];
        hp[0] = xch2[(0 * 8 + (wave ^ 1)) * 64 + lane];
        hp[1] = xch2[(1 * 8 + (wave ^ 1)) * 64 + lane];
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                if (DSG_MLP_EXP == 2) { oacc[ct][s2] += __builtin_bit_cast(float, hf[s2][0] ^ hp[s2][1]); continue; }
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w2own + 32 * ct * LD2 + 16 * s2),
                                                                   __builtin_bit_cast(bf16x8, hf[s2]), oacc[ct], 0, 0, 0);
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(w2oth + 32 * ct * LD2 + 16 * s2),
                                                                   __builtin_bit_cast(bf16x8, hp[s2]), oacc[ct], 0, 0, 0);
            }
    }
    __syncthreads();                           // the stages are free: output transposition, row statistics
    // ---- epilogue: lane (token, half) holds channels 192 kh + 32 ct + 8 q + 4 half + {0..3} in oacc[ct][4 q ..]
    const rsrc_t rsX = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
    __bf16 *xo = static_cast<__bf16 *>(g.xn_out);
    const rsrc_t rsO = make_rsrc(xo ? xo + (size_t)m0 * C : nullptr, xo ? (unsigned)rows * C * 2u : 0u);
    const float *aff_row = nullptr;
    if (MOD != 0) aff_row = g.mod_aff + (size_t)(MOD == 2 ? min(m0 + (int)mrow, g.M - 1) / g.mod_T : 0) * g.mod_ld + g.mod_off;
    __bf16 *T = lds + wave * 32 * TLD;
    f32x2 *part = reinterpret_cast<f32x2 *>(xch1);   // [128 rows][2]
    auto tflush = [&](int half96) {               // the wave's 32 x 96 tile -> row-contiguous 16-byte stores (as gemm_bx_kernel)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k, r = i / 12, pc = i - 12 * r;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(T + r * TLD + 8 * pc);
            buf_store_u4(d, rsO, ((unsigned)(32 * ts + r) * C + (unsigned)(192 * kh + 96 * half96 + 8 * pc)) * 2u, 0u);
        }
    };
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        f32x4 rr[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            rr[q] = PROJ ? (f32x4){0.f, 0.f, 0.f, 0.f} : buf_load4(rsX, (mrow * C + (unsigned)(192 * kh + 32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);   // (PROJ: x1 is in oacc)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 192 * kh + 32 * ct + 8 * q + 4 * lhalf;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.b2 + c);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = oacc[ct][4 * q + e] + b4[e] + rr[q][e];
            if (MOD != 0) {
                const f32x4 scl = *reinterpret_cast<const f32x4 *>(aff_row + c), sft = *reinterpret_cast<const f32x4 *>(aff_row + C + c);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = silu_exact(fmaf(v[e], scl[e] + 1.0f, sft[e]));
            }
            buf_store4(v, rsX, (mrow * C + (unsigned)c) * 4u, 0u);
#pragma unroll
            for (int e = 0; e < 4; e++) { ssum += v[e]; ssq = fmaf(v[e], v[e], ssq); oacc[ct][4 * q + e] = v[e]; }
            if (g.out_mode == 2) *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
        }
        if (g.out_mode == 2 && ct % 3 == 2) tflush(ct / 3);
    }
    if (g.out_mode == 1) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lhalf == 0) part[mrow * 2 + kh] = (f32x2){ssum, ssq};
        __syncthreads();
        const f32x2 p0 = part[mrow * 2], p1 = part[mrow * 2 + 1];
        const float sm = p0[0] + p1[0], sq = p0[1] + p1[1];
        const float mean = sm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(oacc[ct][4 * q + e], rstd, nmr);
                *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
            }
            if (ct % 3 == 2) tflush(ct / 3);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Round 4: the fused (proj +) MLP at C = 384 rebuilt around how the weights reach the CU.  What the eight-wave kernel above left on the
// table (profiles/r3/bx_experiments.txt: 138 of its 208 us remain with every MFMA deleted): the weights of a chunk pair -- 96 KB per 128
// tokens, 2.65 MB per block -- went global -> VGPR -> LDS in the middle of the compute phases, behind three all-wave barriers per pair
// that also held the eight waves in the same phase, so fc1 / GELU / fc2 ran one after the other on every SIMD.  Here:
//   * the weights are PRE-ARRANGED in HBM in the exact order the LDS reads want them (launch_mlp384_images: per stage of 48 KB, 16-byte
//     pieces in [k-step][k-half][row] order for fc1 / proj and [row tile][chunk][k-step][k-half][row] for fc2, K and hidden units already
//     in the accumulator order of the operand they meet), so a stage is a LINEAR copy;
//   * that copy is LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR in between, no ds_write), six instructions
//     per wave and stage, into a ring of THREE 48-KB slots that W1 and W2 stages share alternately: every stage is requested two to three
//     slots of the schedule before its first reader and retired by a counted s_waitcnt vmcnt(6) in front of a raw s_barrier;
//   * every LDS read of a weight fragment is base + lane * 16 + immediate: conflict-free ds_read_b128, one address register per matrix;
//   * fc1 is split along the HIDDEN units inside a wave pair (wave kh forms chunk 2 p + kh of pair p over the whole K = 384 from the
//     full normalised row, 96 registers), so no fp32 partial sum crosses the pair; only the bf16 hidden tiles do (as before), and fc2
//     is split along N as before;
//   * the two halves of the block (waves 0-3 / 4-7: the two waves of every SIMD) run the SAME program ONE barrier interval apart
//     (MI355X_MICROARCH.md, "Two waves per SIMD" item 9): while one half is in its GELU interval the other is in a matrix interval.
// LDS: 3 x 48 KB ring + 16 KB exchange = 160 KB, one block per CU.  PROJ / MOD / out_mode: as mlp384_bx_kernel.
// -------------------------------------------------------------------------------------------------
constexpr int M384_STAGE = 49152;                                   // bytes of a ring slot = 3072 pieces of 16 B
constexpr size_t M384_W1IMG = 0, M384_W2IMG = (size_t)1536 * 384 * 2, M384_WPIMG = (size_t)2 * 1536 * 384 * 2;   // byte offsets in the image
size_t mlp384_image_bytes() { return (size_t)(2 * 1536 * 384 + 384 * 384) * 2; }

// one thread per 16-byte piece of the image (see above); W1 [1536, 384], W2 [384, 1536], Wp [384, 384] row-major bf16 (Wp may be null)
__global__ __launch_bounds__(256) void mlp384_img_kernel(const unsigned short *__restrict__ W1, const unsigned short *__restrict__ W2,
                                                         const unsigned short *__restrict__ Wp, unsigned short *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n1 = 24 * 3072, n2 = 24 * 3072, np = 6 * 3072;
    if (idx >= n1 + n2 + (Wp ? np : 0)) return;
    unsigned short v[8];
    if (idx < n1) {
        const int p = idx / 3072, P = idx % 3072, r = P % 64, sg = P / 64, s = sg >> 1, g = sg & 1;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = W1[(size_t)(64 * p + r) * 384 + 16 * s + 4 * g + (j & 3) + 8 * (j >> 2)];
    } else if (idx < n1 + n2) {
        const int i2 = idx - n1, p = i2 / 3072, P = i2 % 3072, row = P % 32, q = P / 32;
        const int g = q & 1, s2 = (q >> 1) & 1, c = (q >> 2) & 1, ct = (q >> 3) % 6, kh = (q >> 3) / 6;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = W2[(size_t)(192 * kh + 32 * ct + row) * 1536 + 64 * p + 32 * c + 16 * s2 + 4 * g + (j & 3) + 8 * (j >> 2)];
    } else {
        const int i3 = idx - n1 - n2, ct = i3 / 3072, P = i3 % 3072, r = P % 64, sg = P / 64, s = sg >> 1, g = sg & 1;
        const int row = r < 32 ? 32 * ct + r : 192 + 32 * ct + (r - 32);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = Wp[(size_t)row * 384 + 16 * s + 8 * g + j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) img[(size_t)idx * 8 + j] = v[j];
}
void launch_mlp384_images(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s) {
    const int n = 24 * 3072 * 2 + (Wpb ? 6 * 3072 : 0);
    DSG_LAUNCH(mlp384_img_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned short *)W1b, (const unsigned short *)W2b,
               (const unsigned short *)Wpb, (unsigned short *)img);
}

typedef __attribute__((address_space(3))) void *lds_vptr;
#define M384_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))   // vmcnt(N) only (lgkmcnt / expcnt: no wait)
#define M384_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)                                      // lgkmcnt(0) only
// raw s_barrier (no vmcnt(0) in front of it: the LDS-DMA requests stay in flight across it) between two compiler-level memory barriers
// (the intrinsic alone does not keep the compiler from moving LDS accesses across it)
#define M384_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

#ifndef DSG_M384_PRIO
#define DSG_M384_PRIO 0   // 1: s_setprio 1 in the matrix intervals -- measured SLOWER (chunk-pair loop 168k clk against 145k: profiles/r4/m384_experiments.txt)
#endif
#ifndef DSG_M384_EXP
#define DSG_M384_EXP 0   // timing experiments of tools/m384_exp.sh (wrong results): 1 no LDS-DMA inside the chunk-pair loop, 2 no MFMAs there, 3 no GELU, 5 MFMAs without their LDS fragment reads
#endif
template <int MOD, bool PROJ = false>
__global__ __launch_bounds__(512, 1) void mlp384d_bx_kernel(BxMlp g) {
    constexpr int C = 384, NP = 24, CT = 6, TLD = 104;
    __shared__ __attribute__((aligned(16))) char lds[3 * M384_STAGE + 16384];
    char *xchb = lds + 3 * M384_STAGE;
    u32x4 *xch2 = reinterpret_cast<u32x4 *>(xchb);                  // [2 k-steps][8 waves][64 lanes]: the pair's bf16 hidden tiles
    const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 31, lhalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ts = wave >> 1, kh = wave & 1, team = wave >> 2;
    const int m0 = blockIdx.x * 128;
    const int rows = min(128, g.M - m0);
    const unsigned mrow = (unsigned)(32 * ts + lrow);
    const char *img = static_cast<const char *>(g.img);
    // measurement builds of the debug entry only (g.dbg != null): per-wave s_memtime stamps at the phase boundaries
    unsigned long long *dbg = g.dbg ? g.dbg + ((size_t)blockIdx.x * 8 + wave) * 16 : nullptr;
#define M384_STAMP(i) do { if (dbg && lane == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    // First-round blocks start STAGGERED (g.skew clocks x an eighth-phase of the CU index inside its XCD): with one block per CU and every
    // block starting at once, all CUs are in their HBM phases (attention / residual rows in, residual / LayerNorm rows out) at the same
    // time -- HBM saturated, ~7 B/clk per CU -- and in their compute phases at the same time, HBM idle.  Spread out, a CU's HBM phase
    // meets the others' compute phases.  The one-tile CUs of the last round absorb the delay.
    if (g.skew > 0 && blockIdx.x < 256) {
        const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)(((blockIdx.x >> 3) & 7) * g.skew);
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }
    M384_STAMP(0);
    // this wave's six 1-KiB pieces of a 48-KB stage: image bytes [src, src + 49152) -> ring slot r, linear
    const unsigned dma_voff = (unsigned)lane * 16u;
    auto dma_stage = [&](const char *src, int r) {      // src, r: wave-uniform -> scalar base + one 32-bit lane offset
        const char *sbase = src + wave * 6144;
        char *dbase = lds + r * M384_STAGE + wave * 6144;
        // (M0 is not on the clobber list -- the compiler calls it reserved and warns; it only ever writes M0 itself immediately in front of an
        // instruction that reads it, none of which these kernels contain besides the requests themselves.)
        // (inline assembly, not __builtin_amdgcn_global_load_lds: the compiler's wait-count pass books the builtin as a FLAT access that may
        // touch LDS and memory, and while one is pending every s_waitcnt it inserts -- the fragment reads' lgkmcnt, ordinary loads' vmcnt --
        // becomes a full drain.  M0 = the LDS address, one wait state between its write and the request.)
        const unsigned dlds = (unsigned)(size_t)dbase;
#pragma unroll
        for (int i = 0; i < 6; i++)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dlds + i * 1024), "v"(dma_voff), "s"(sbase + i * 1024) : "memory");
    };
    // the normalised row of the lane's token as fc1's B operand, k-step s = channels 16 s + 4 half + {0..3, 8..11}: xo = the k-steps
    // 12 kh .. + 11 (the channels this wave's proj / fc2 half owns), xp = the partner's 12 (fc1 walks own, then partner's: the order
    // of a sum's terms is free, and the register arrays keep compile-time indices)
    bf16x8 xfo[12], xfp[12];
    f32x16 oacc[CT];
    if (PROJ) {
        // x1 = x + att Wp^T + bp: the residual rows and the bias are the proj accumulators' INITIAL value -- their loads are in flight
        // together with the attention rows and the first weight stages instead of forming a second exposed HBM round trip after the proj
        const rsrc_t rsXi = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 rr = buf_load4(rsXi, (mrow * C + (unsigned)(192 * kh + 32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.bp + 192 * kh + 32 * ct + 8 * q + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; e++) oacc[ct][4 * q + e] = rr[e] + b4[e];
            }
        const char *wp = img + M384_WPIMG;
        dma_stage(wp, 0);
        // The block's 128 attention rows are ONE contiguous 96-KB piece of att: it comes in by LDS-DMA (whole cache lines; direct
        // "lane = row" loads take 16 bytes of each of 32 rows per instruction and are bound by the texture path) into ring slots 1 + 2,
        // 16-byte piece c of row r at piece position 48 r + (c ^ (r & 15)) -- the swizzle is in the SOURCE address, the LDS image of a
        // DMA instruction is lane-linear -- so that the 16 rows a ds_read_b128 lane group reads hit 16 different bank groups.
        {
            const char *asrc = reinterpret_cast<const char *>(static_cast<const __bf16 *>(g.att) + (size_t)m0 * C);
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const int P = (wave * 12 + k) * 64 + lane, r = P / 48, cp = P - 48 * r, c = (cp & 48) | ((cp ^ r) & 15);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"((unsigned)(size_t)lds + M384_STAGE + (wave * 12 + k) * 1024),
                             "v"(asrc + ((size_t)min(r, rows - 1) * 48 + c) * 16) : "memory");
            }
        }
        M384_WAIT_VM(0);
        M384_BARRIER();
        bf16x8 af[24];   // the whole attention row of the lane's token: channels 16 s + 8 half .. + 7 of k-step s
        {
            const char *arow = lds + M384_STAGE + mrow * 768;
#pragma unroll
            for (int s = 0; s < 24; s++) {
                const int c = 2 * s + lhalf;
                af[s] = *reinterpret_cast<const bf16x8 *>(arow + ((c & 48) | ((c ^ (int)mrow) & 15)) * 16);
            }
        }
        M384_WAIT_LGKM0();
        M384_BARRIER();               // slots 1 and 2 are free: the next two proj stages
        dma_stage(wp + M384_STAGE, 1); dma_stage(wp + 2 * M384_STAGE, 2);
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            if (ct < CT - 1) M384_WAIT_VM(12); else M384_WAIT_VM(6);   // stage ct has landed (at most the two younger requests are still out)
            M384_BARRIER();
            const char *a = lds + (ct % 3) * M384_STAGE + lhalf * 1024 + (32 * kh + lrow) * 16;
#pragma unroll
            for (int s = 0; s < 24; s++)
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(a + s * 2048), af[s], oacc[ct], 0, 0, 0);
            M384_WAIT_LGKM0();
            M384_BARRIER();          // every wave has read slot ct % 3
            if (ct + 3 < CT) dma_stage(wp + (ct + 3) * M384_STAGE, ct % 3);
            else if (ct == 3) dma_stage(img + M384_W1IMG, 0);          // W1 stage of pair 0 -> slot 0 (its ring position)
        }
        M384_STAMP(1);
        // LayerNorm-2 statistics of x1 (this wave's 192 channels) across the pair
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) { const float v = oacc[ct][r]; sm += v; sq = fmaf(v, v, sq); }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        f32x2 *part0 = reinterpret_cast<f32x2 *>(xchb);   // [128 rows][2]
        if (lhalf == 0) part0[mrow * 2 + kh] = (f32x2){sm, sq};
        __syncthreads();                       // statistics visible (this also drains the W1 stage request: it is needed next anyway)
        const f32x2 q0 = part0[mrow * 2], q1 = part0[mrow * 2 + 1];
        const float tsm = q0[0] + q1[0], tsq = q0[1] + q1[1];
        const float mean = tsm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, tsq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
        // the normalised values of the own 192 channels are k-steps 12 kh .. + 11 of fc1's B operand as the registers stand; the partner's
        // half crosses through ring slots 1 and 2 (free until the first W2 / second W1 stage is requested below)
        u32x4 *xx = reinterpret_cast<u32x4 *>(lds + M384_STAGE);   // [8 waves][12][64 lanes]
#pragma unroll
        for (int s = 0; s < 12; s++) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; j++)
                pk[j] = pack_bf16(fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j], rstd, nmr), fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j + 1], rstd, nmr));
            xx[(wave * 12 + s) * 64 + lane] = pk;
            xfo[s] = __builtin_bit_cast(bf16x8, pk);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 12; s++) {
            xfp[s] = __builtin_bit_cast(bf16x8, xx[((wave ^ 1) * 12 + s) * 64 + lane]);
        }
        __syncthreads();                       // slots 1 and 2 are free again
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;
        dma_stage(img + M384_W1IMG, 0);
        // the normalised rows from HBM, in the K order of the W1 image: k-step s = channels 16 s + 4 half + {0..3} and + 8 + {0..3}
        const rsrc_t rsXn = make_rsrc(static_cast<const __bf16 *>(g.xn) + (size_t)m0 * C, (unsigned)rows * C * 2u);
#pragma unroll
        for (int s = 0; s < 24; s++) {
            const unsigned ch = (unsigned)(s < 12 ? 192 * kh : 192 * (1 - kh)) + 16u * (s % 12) + 4u * lhalf;
            const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rsXn, (mrow * C + ch) * 2u, 0u, 0);
            const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rsXn, (mrow * C + ch + 8u) * 2u, 0u, 0);
            const bf16x8 v = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
            if (s < 12) xfo[s] = v; else xfp[s - 12] = v;
        }
    }
    dma_stage(img + M384_W2IMG, 1);                     // ring item 1: W2 of pair 0
    dma_stage(img + M384_W1IMG + M384_STAGE, 2);        // ring item 2: W1 of pair 1
    // ---- the chunk-pair loop.  Ring item n lives in slot n % 3: item 2 p = W1 stage of pair p, item 2 p + 1 = its W2 stage.
    // Global schedule in barrier intervals t: first half (waves 0-3) runs fc1(p) | GELU(p) | fc2(p) in t = 3 p, 3 p + 1, 3 p + 2; the second
    // half runs the same program one interval later.  Requests (by all eight waves, six pieces each): W1 of pair t/3 + 1 at the start of
    // every interval t = 1 (mod 3), W2 of pair t/3 + 1 at t = 2 (mod 3) -- the slot's previous reader finished in the interval before --
    // and vmcnt(6) in front of every barrier retires everything but the newest request.
    const int w1own = lhalf * 1024 + (32 * kh + lrow) * 16 + 12 * kh * 2048;        // k-steps 12 kh + s': + 2048 s'
    const int w1oth = lhalf * 1024 + (32 * kh + lrow) * 16 + 12 * (1 - kh) * 2048;  // the partner's channels
    const int w2own = lane * 16 + 26 * kh * 1024;                            // piece block (kh, ct, c = kh, s2): + (4 ct + s2) KiB
    const int w2oth = lane * 16 + (24 * kh + 2 * (1 - kh)) * 1024;           // c = 1 - kh
    // b1 is the fc1 accumulator's initial value: lane (token, half) holds hidden units 8 q + 4 half + {0..3} of its chunk in h[4 q ..];
    // the four loads of pair p + 1 are issued in interval C of pair p (h is dead there) and have a whole interval to land
    const float *b1w = g.b1 + 32 * kh + 4 * lhalf;
    f32x16 h;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b1w + 8 * q);
        h[4 * q] = b4[0]; h[4 * q + 1] = b4[1]; h[4 * q + 2] = b4[2]; h[4 * q + 3] = b4[3];
    }
    M384_WAIT_VM(12);                                                        // the W1 stage of pair 0 has landed (two younger requests are out)
    M384_BARRIER();
    M384_STAMP(2);
    if (team == 1) { M384_WAIT_VM(6); M384_BARRIER(); }                      // the second half starts one interval late
    int slot1 = 0, slot2 = 1;                                                // ring slots of this pair's W1 / W2 stage
    for (int p = 0; p < NP; p++) {
        // ---- interval A: fc1 of the own chunk over the whole K (h starts at b1).  The matrix intervals run at priority 1: the other half
        // of the block is in its GELU interval on the same SIMDs, and at equal priority its VALU stream takes the issue slots the
        // fragment reads and MFMAs of this one need (measured: fc1 1560 clk beside a GELU partner, 1070 beside an MFMA partner)
        if (p == 8) M384_STAMP(5);
        if (DSG_M384_PRIO) __builtin_amdgcn_s_setprio(1);
        {
            const char *a = lds + slot1 * M384_STAGE + w1own, *ax = lds + slot1 * M384_STAGE + w1oth;
            // weight fragments four k-steps ahead of their MFMA (a wave issues in order: a read issued right in front of its MFMA
            // exposes the whole LDS latency); sched_group_barrier pins read / MFMA alternation
            auto LD = [&](int s) { return DSG_M384_EXP == 5 ? xfp[s % 12] : *reinterpret_cast<const bf16x8 *>((s < 12 ? a : ax) + (s % 12) * 2048); };
            bf16x8 fr[4];
#pragma unroll
            for (int s = 0; s < 4; s++) fr[s] = LD(s);
#pragma unroll
            for (int s = 0; s < 24; s++) {
                if (DSG_M384_EXP == 2) { h[s & 15] += (float)(s < 12 ? xfo[s] : xfp[s - 12])[0] + (float)fr[s & 3][1]; }
                else h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s & 3], s < 12 ? xfo[s] : xfp[s - 12], h, 0, 0, 0);
                if (s + 4 < 24) fr[s & 3] = LD(s + 4);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int s = 0; s < 20; s++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (DSG_M384_PRIO) __builtin_amdgcn_s_setprio(0);
        if (p == 8) M384_STAMP(6);
        // (the second half's request of this interval goes LAST: nothing the compiler waits for -- the b1 loads -- has it behind itself)
        if (DSG_M384_EXP != 1 && team == 1 && p >= 1 && p + 1 < NP) dma_stage(img + M384_W1IMG + (size_t)(p + 1) * M384_STAGE, (2 * p + 2) % 3);
        if (p + 2 < NP) M384_WAIT_VM(6); else M384_WAIT_VM(0);   // (the last stages have no younger request behind them)
        M384_WAIT_LGKM0();
        M384_BARRIER();
        if (p == 8) M384_STAMP(7);
        // ---- interval B: + b1, GELU, bf16; the hidden tile goes to the partner
        if (DSG_M384_EXP != 1 && team == 0 && p >= 1 && p + 1 < NP) dma_stage(img + M384_W1IMG + (size_t)(p + 1) * M384_STAGE, (2 * p + 2) % 3);
        if (DSG_M384_EXP != 1 && team == 1 && p + 1 < NP) dma_stage(img + M384_W2IMG + (size_t)(p + 1) * M384_STAGE, (2 * p + 3) % 3);
        u32x4 hf[2];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (DSG_M384_EXP == 3) {   // no GELU
                hf[q >> 1][2 * (q & 1)] = pack_bf16(h[4 * q], h[4 * q + 1]);
                hf[q >> 1][2 * (q & 1) + 1] = pack_bf16(h[4 * q + 2], h[4 * q + 3]);
                continue;
            }
            const f32x2_t g01 = gelu_f2(h[4 * q], h[4 * q + 1]), g23 = gelu_f2(h[4 * q + 2], h[4 * q + 3]);
            hf[q >> 1][2 * (q & 1)] = pack_bf16(g01[0], g01[1]);
            hf[q >> 1][2 * (q & 1) + 1] = pack_bf16(g23[0], g23[1]);
        }
        xch2[(0 * 8 + wave) * 64 + lane] = hf[0];
        xch2[(1 * 8 + wave) * 64 + lane] = hf[1];
        if (p == 8) M384_STAMP(8);
        if (p + 2 < NP) M384_WAIT_VM(6); else M384_WAIT_VM(0);   // (the last stages have no younger request behind them)
        M384_WAIT_LGKM0();
        M384_BARRIER();
        if (p == 8) M384_STAMP(9);
        // ---- interval C: fc2 on the own 192 output channels over both chunks of the pair
        u32x4 hp[2];
        hp[0] = xch2[(0 * 8 + (wave ^ 1)) * 64 + lane];
        hp[1] = xch2[(1 * 8 + (wave ^ 1)) * 64 + lane];
        if (p + 1 < NP) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b1w + 64 * (p + 1) + 8 * q);
                h[4 * q] = b4[0]; h[4 * q + 1] = b4[1]; h[4 * q + 2] = b4[2]; h[4 * q + 3] = b4[3];
            }
        }
        if (DSG_M384_PRIO) __builtin_amdgcn_s_setprio(1);
        {
            const char *ao = lds + slot2 * M384_STAGE + w2own, *ax = lds + slot2 * M384_STAGE + w2oth;
            // MFMA i = 4 ct + 2 s2 + c' (c' = 0: own chunk, 1: the partner's), fragments four ahead as in fc1
            auto LD = [&](int i) { return DSG_M384_EXP == 5 ? __builtin_bit_cast(bf16x8, hp[i & 1]) : *reinterpret_cast<const bf16x8 *>(((i & 1) ? ax : ao) + (4 * (i >> 2) + ((i >> 1) & 1)) * 1024); };
            bf16x8 fr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) fr[i] = LD(i);
#pragma unroll
            for (int i = 0; i < 24; i++) {
                const int ct = i >> 2, s2 = (i >> 1) & 1;
                if (DSG_M384_EXP == 2) oacc[ct][i & 3] += __builtin_bit_cast(float, hf[s2][0] ^ hp[s2][1]) + (float)fr[i & 3][0];
                else oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i & 3], __builtin_bit_cast(bf16x8, (i & 1) ? hp[s2] : hf[s2]), oacc[ct], 0, 0, 0);
                if (i + 4 < 24) fr[i & 3] = LD(i + 4);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 20; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (DSG_M384_PRIO) __builtin_amdgcn_s_setprio(0);
        // (requests of a matrix interval go out BEHIND its MFMAs: six pieces cost the issuing wave ~600 clk, which it has to spare while the
        // other half finishes its GELU interval)
        if (DSG_M384_EXP != 1 && team == 0 && p + 1 < NP) dma_stage(img + M384_W2IMG + (size_t)(p + 1) * M384_STAGE, (2 * p + 3) % 3);
        if (p == 8) M384_STAMP(10);
        if (p + 2 < NP) M384_WAIT_VM(6); else M384_WAIT_VM(0);   // (the last stages have no younger request behind them)
        M384_WAIT_LGKM0();
        M384_BARRIER();
        if (p == 8) M384_STAMP(11);
        slot1 = slot1 == 0 ? 2 : slot1 - 1;     // (2 p) % 3: 0, 2, 1, 0, ...
        slot2 = slot2 == 0 ? 2 : slot2 - 1;     // (2 p + 1) % 3: 1, 0, 2, 1, ...
    }
    if (team == 0) { M384_WAIT_VM(0); M384_BARRIER(); }        // the first half waits out the second half's last interval
    __syncthreads();                           // the ring is free: output transposition, row statistics
    M384_STAMP(3);
    // ---- epilogue (as mlp384_bx_kernel): lane (token, half) holds channels 192 kh + 32 ct + 8 q + 4 half + {0..3} in oacc[ct][4 q ..]
    const rsrc_t rsX = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
    __bf16 *xo = static_cast<__bf16 *>(g.xn_out);
    const rsrc_t rsO = make_rsrc(xo ? xo + (size_t)m0 * C : nullptr, xo ? (unsigned)rows * C * 2u : 0u);
    const float *aff_row = nullptr;
    if (MOD != 0) aff_row = g.mod_aff + (size_t)(MOD == 2 ? min(m0 + (int)mrow, g.M - 1) / g.mod_T : 0) * g.mod_ld + g.mod_off;
    // per wave 19456 B of the (now free) ring: a 32 x 100-float tile through which the fp32 rows leave as whole-line stores (the direct
    // "lane = row" form stores 16 bytes into each of 32 rows per instruction) and behind it the bf16 transposition tile T
    float *xt = reinterpret_cast<float *>(lds + wave * 19456);
    __bf16 *T = reinterpret_cast<__bf16 *>(lds + wave * 19456 + 12800);
    f32x2 *part = reinterpret_cast<f32x2 *>(lds + 3 * M384_STAGE + 16384 - 2048);   // [128 rows][2], behind every wave's tiles
    auto xflush = [&](int half96) {               // the wave's 32 x 96 fp32 tile -> x rows, 24 lanes per 384-byte row segment
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int i = lane + 64 * k, r = i / 24, pc = i - 24 * r;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(xt + r * 100 + 4 * pc);
            buf_store4(d, rsX, ((unsigned)(32 * ts + r) * C + (unsigned)(192 * kh + 96 * half96 + 4 * pc)) * 4u, 0u);
        }
    };
    auto tflush = [&](int half96) {               // the wave's 32 x 96 tile -> row-contiguous 16-byte stores (as gemm_bx_kernel)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k, r = i / 12, pc = i - 12 * r;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(T + r * TLD + 8 * pc);
            buf_store_u4(d, rsO, ((unsigned)(32 * ts + r) * C + (unsigned)(192 * kh + 96 * half96 + 8 * pc)) * 2u, 0u);
        }
    };
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        f32x4 rr[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            rr[q] = PROJ ? (f32x4){0.f, 0.f, 0.f, 0.f} : buf_load4(rsX, (mrow * C + (unsigned)(192 * kh + 32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);   // (PROJ: x1 is in oacc)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 192 * kh + 32 * ct + 8 * q + 4 * lhalf;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.b2 + c);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = oacc[ct][4 * q + e] + b4[e] + rr[q][e];
            if (MOD != 0) {
                const f32x4 scl = *reinterpret_cast<const f32x4 *>(aff_row + c), sft = *reinterpret_cast<const f32x4 *>(aff_row + C + c);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = silu_exact(fmaf(v[e], scl[e] + 1.0f, sft[e]));
            }
            *reinterpret_cast<f32x4 *>(xt + lrow * 100 + 32 * (ct % 3) + 8 * q + 4 * lhalf) = v;
#pragma unroll
            for (int e = 0; e < 4; e++) { ssum += v[e]; ssq = fmaf(v[e], v[e], ssq); oacc[ct][4 * q + e] = v[e]; }
            if (g.out_mode == 2) *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
        }
        if (ct % 3 == 2) xflush(ct / 3);
        if (g.out_mode == 2 && ct % 3 == 2) tflush(ct / 3);
    }
    if (g.out_mode == 1) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lhalf == 0) part[mrow * 2 + kh] = (f32x2){ssum, ssq};
        __syncthreads();
        const f32x2 p0 = part[mrow * 2], p1 = part[mrow * 2 + 1];
        const float sm = p0[0] + p1[0], sq = p0[1] + p1[1];
        const float mean = sm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(oacc[ct][4 * q + e], rstd, nmr);
                *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
            }
            if (ct % 3 == 2) tflush(ct / 3);
        }
    }
    M384_WAIT_VM(0);
    M384_STAMP(4);
#undef M384_STAMP
}

// -------------------------------------------------------------------------------------------------
// mlp384s_bx_kernel -- the same fused (proj +) MLP on FOUR waves, one per SIMD, each with the whole 512-entry register file.
// What mlp384d's clocks showed (profiles/r4/m384_experiments.txt): at two waves per SIMD a matrix interval loses a third to a half of
// its time to the partner's VALU stream, and the pair structure costs an LDS exchange and three all-wave barriers per 64 hidden units.
// Here a wave owns 32 tokens COMPLETELY -- all 384 output channels (192 accumulator registers), the whole normalised row (96) -- so
// nothing crosses between waves but the weight ring, and the overlap of GELU with matrix work is a software pipeline INSIDE the wave:
// while the 24 MFMAs of fc1(c + 1) run, the VALU instructions of GELU(c) issue in their shadows (an MFMA holds the SIMD's issue for 8 of
// its 32 cycles), then the 24 MFMAs of fc2(c).  A ring stage is ONE chunk of 32 hidden units, [W1 rows | W2 columns] = 48 KB, in the
// piece order the fragment reads want (launch_mlp384s_images); twelve LDS-DMA pieces per wave and chunk go out between the MFMAs, one
// barrier per chunk.  b1 sits in LDS and is the fc1 accumulator's initial value.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp384s_img_kernel(const unsigned short *__restrict__ W1, const unsigned short *__restrict__ W2,
                                                          const unsigned short *__restrict__ Wp, unsigned short *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n1 = 48 * 3072, np = 6 * 3072;
    if (idx >= n1 + (Wp ? np : 0)) return;
    unsigned short v[8];
    if (idx < n1) {
        const int c = idx / 3072, P = idx % 3072;
        if (P < 1536) {     // W1 rows 32 c .. + 31: piece (s, g, r)
            const int r = P % 32, sg = P / 32, s = sg >> 1, g = sg & 1;
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = W1[(size_t)(32 * c + r) * 384 + 16 * s + 4 * g + (j & 3) + 8 * (j >> 2)];
        } else {            // W2 columns 32 c .. + 31: piece (ct, s2, g, row)
            const int Q = P - 1536, row = Q % 32, q = Q / 32, g = q & 1, s2 = (q >> 1) & 1, ct = q >> 2;
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = W2[(size_t)(32 * ct + row) * 1536 + 32 * c + 16 * s2 + 4 * g + (j & 3) + 8 * (j >> 2)];
        }
    } else {                // proj stage st: rows 64 st .. + 63, natural K order
        const int i3 = idx - n1, st = i3 / 3072, P = i3 % 3072, r = P % 64, sg = P / 64, s = sg >> 1, g = sg & 1;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = Wp[(size_t)(64 * st + r) * 384 + 16 * s + 8 * g + j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) img[(size_t)idx * 8 + j] = v[j];
}
void launch_mlp384s_images(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s) {
    const int n = 48 * 3072 + (Wpb ? 6 * 3072 : 0);
    DSG_LAUNCH(mlp384s_img_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned short *)W1b, (const unsigned short *)W2b,
               (const unsigned short *)Wpb, (unsigned short *)img);
}

template <int MOD, bool PROJ = false>
__global__ __launch_bounds__(256, 1) void mlp384s_bx_kernel(BxMlp g) {
    constexpr int C = 384, NCH = 48, CT = 12, TLD = 104;
    __shared__ __attribute__((aligned(16))) char lds[3 * M384_STAGE + 16384];
    float *b1s = reinterpret_cast<float *>(lds + 3 * M384_STAGE);   // [1536]
    const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 31, lhalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 128;
    const int rows = min(128, g.M - m0);
    const unsigned mrow = (unsigned)(32 * wave + lrow);
    const char *img = static_cast<const char *>(g.img2);
    const char *chunks = img, *wp = img + (size_t)NCH * M384_STAGE;
    unsigned long long *dbg = g.dbg ? g.dbg + ((size_t)blockIdx.x * 8 + wave) * 16 : nullptr;
#define M384_STAMP(i) do { if (dbg && lane == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    M384_STAMP(0);
    const unsigned dma_voff = (unsigned)lane * 16u;
    // The LDS-DMA requests are written as inline assembly: the compiler's wait-count pass treats the builtin as a FLAT access that may touch
    // both LDS and memory, and while one is pending it turns every lgkmcnt / vmcnt wait into a full drain (the fragment reads four ahead
    // then stall every fourth MFMA).  M0 carries the LDS address (wave-uniform), one wait state between its write and the request.
    const unsigned lds0 = (unsigned)(size_t)lds;
    auto glds_s = [&](const char *sbase, unsigned voff, unsigned ldsaddr) {        // uniform base + per-lane 32-bit offset
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsaddr), "v"(voff), "s"(sbase) : "memory");
    };
    auto glds_v = [&](const char *vptr, unsigned ldsaddr) {                        // per-lane address
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ldsaddr), "v"(vptr) : "memory");
    };
    auto dma_piece = [&](const char *src, int slot, int i) {   // piece i (0 .. 11) of this wave's share of a stage
        glds_s(src + wave * 12288 + i * 1024, dma_voff, lds0 + slot * M384_STAGE + wave * 12288 + i * 1024);
    };
    auto dma_stage = [&](const char *src, int slot) {
#pragma unroll
        for (int i = 0; i < 12; i++) dma_piece(src, slot, i);
    };
    for (int i = tid; i < 1536; i += 256) b1s[i] = g.b1[i];
    bf16x8 xf[24];
    f32x16 oacc[CT];
    if (PROJ) {
        const rsrc_t rsXi = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 rr = buf_load4(rsXi, (mrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.bp + 32 * ct + 8 * q + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; e++) oacc[ct][4 * q + e] = rr[e] + b4[e];
            }
        dma_stage(wp, 0);
        {   // the block's 128 attention rows by swizzled LDS-DMA into slots 1 + 2 (see mlp384d_bx_kernel)
            const char *asrc = reinterpret_cast<const char *>(static_cast<const __bf16 *>(g.att) + (size_t)m0 * C);
#pragma unroll
            for (int k = 0; k < 24; k++) {
                const int P = (wave * 24 + k) * 64 + lane, r = P / 48, cp = P - 48 * r, c = (cp & 48) | ((cp ^ r) & 15);
                glds_v(asrc + ((size_t)min(r, rows - 1) * 48 + c) * 16, lds0 + M384_STAGE + (wave * 24 + k) * 1024);
            }
        }
        M384_WAIT_VM(0);
        M384_BARRIER();
        bf16x8 af[24];
        {
            const char *arow = lds + M384_STAGE + mrow * 768;
#pragma unroll
            for (int s = 0; s < 24; s++) {
                const int c = 2 * s + lhalf;
                af[s] = *reinterpret_cast<const bf16x8 *>(arow + ((c & 48) | ((c ^ (int)mrow) & 15)) * 16);
            }
        }
        M384_WAIT_LGKM0();
        M384_BARRIER();
        dma_stage(wp + M384_STAGE, 1); dma_stage(wp + 2 * M384_STAGE, 2);
#pragma unroll
        for (int st = 0; st < 6; st++) {
            if (st < 5) M384_WAIT_VM(24); else M384_WAIT_VM(12);
            M384_BARRIER();
            const char *a = lds + (st % 3) * M384_STAGE + lhalf * 1024 + lrow * 16;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int s = 0; s < 24; s++)
                    oacc[2 * st + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(a + t * 512 + s * 2048), af[s], oacc[2 * st + t], 0, 0, 0);
            M384_WAIT_LGKM0();
            M384_BARRIER();
            if (st + 3 < 6) dma_stage(wp + (st + 3) * M384_STAGE, st % 3);
            else if (st < 5) dma_stage(chunks + (size_t)(st - 3) * M384_STAGE, st - 3);     // chunk stages 0, 1 -> slots 0, 1 (chunk 2 goes out in iteration 0)
        }
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) { const float v = oacc[ct][r]; sm += v; sq = fmaf(v, v, sq); }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float mean = sm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int s = 0; s < 24; s++) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; j++)
                pk[j] = pack_bf16(fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j], rstd, nmr), fmaf(oacc[s >> 1][8 * (s & 1) + 2 * j + 1], rstd, nmr));
            xf[s] = __builtin_bit_cast(bf16x8, pk);
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;
        dma_stage(chunks, 0); dma_stage(chunks + M384_STAGE, 1);
        const rsrc_t rsXn = make_rsrc(static_cast<const __bf16 *>(g.xn) + (size_t)m0 * C, (unsigned)rows * C * 2u);
#pragma unroll
        for (int s = 0; s < 24; s++) {
            const unsigned ch = 16u * s + 4u * lhalf;
            const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rsXn, (mrow * C + ch) * 2u, 0u, 0);
            const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rsXn, (mrow * C + ch + 8u) * 2u, 0u, 0);
            xf[s] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
        }
    }
    M384_STAMP(1);
    M384_WAIT_VM(0);
    M384_WAIT_LGKM0();
    M384_BARRIER();                       // chunk stages 0, 1 and b1 are in LDS
    M384_STAMP(2);
    const int w1off = lhalf * 512 + lrow * 16;            // W1 piece (s, g, r): + 1024 s
    const int w2off = 24576 + lane * 16;                  // W2 piece block (ct, s2): + 1024 (2 ct + s2)
    const float *b1l = b1s + 4 * lhalf;                   // hidden unit of h[4 q + e]: 8 q + 4 half + e
    auto b1_init = [&](int c) {
        f32x16 h;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b1l + 32 * c + 8 * q);
            h[4 * q] = b4[0]; h[4 * q + 1] = b4[1]; h[4 * q + 2] = b4[2]; h[4 * q + 3] = b4[3];
        }
        return h;
    };
    f32x16 h = b1_init(0);
    {
        const char *a = lds + w1off;
#pragma unroll
        for (int s = 0; s < 24; s++) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(a + s * 1024), xf[s], h, 0, 0, 0);
    }
    int slot = 0;                          // ring slot of chunk c
    for (int c = 0; c < NCH; c++) {
        const int slot_n = slot == 2 ? 0 : slot + 1, slot_r = slot == 0 ? 2 : slot - 1;   // slots of chunk c + 1 and of chunk c + 2 (= c - 1's)
        if (c == 8) M384_STAMP(5);
        if (c >= 1) {                      // every wave has finished chunk c - 1: its slot takes chunk c + 2; chunk c + 1 has landed
            M384_WAIT_VM(0);
            M384_WAIT_LGKM0();
            M384_BARRIER();
        }
        if (c == 8) M384_STAMP(6);
        // (unconditional, so that the half stays one basic block for the scheduler: the last iterations re-fetch chunk 47 into a finished slot)
        const char *rsrc_stage = chunks + (size_t)min(c + 2, NCH - 1) * M384_STAGE;
        // ---- the chunk's instruction stream is laid out BY HAND: one segment per MFMA -- the MFMA, the fragment read four ahead, a share
        // of the GELU arithmetic (which issues in the MFMA's shadow: 32 matrix cycles leave room for seven VALU instructions), every
        // other time one LDS-DMA piece -- and a sched_barrier(0) after each so that the compiler keeps the segments in this order
        // (sched_group_barrier pipelines with VALU groups were dropped by the solver on this block).
        // GELU(c): values 0 .. 11 behind fc1(c + 1)'s MFMAs (one PAIR per four), values 12 .. 15 behind the first twelve MFMAs of fc2(c),
        // which run s2 = 0 (hidden units 0 .. 15 = hf[0]) first.
        u32x4 hf[2];
        f32x2_t ga[8], gp[8];
        auto gelu_a = [&](int v) {             // pair v (values 2 v, 2 v + 1): the exponent arguments (packed fp32: see gelu_f2)
            if (DSG_M384_EXP == 3) { gp[v] = (f32x2_t){h[2 * v], h[2 * v + 1]}; return; }
            const f32x2_t a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(h[2 * v]), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(h[2 * v + 1]), 0.0f, 6.0f)};
            f32x2_t pp = __builtin_elementwise_fma(a, (f32x2_t){3.3159643839e-05f, 3.3159643839e-05f}, (f32x2_t){-7.6972447974e-04f, -7.6972447974e-04f});
            pp = __builtin_elementwise_fma(a, pp, (f32x2_t){8.0821445939e-03f, 8.0821445939e-03f});
            pp = __builtin_elementwise_fma(a, pp, (f32x2_t){-5.3413999628e-02f, -5.3413999628e-02f});
            pp = __builtin_elementwise_fma(a, pp, (f32x2_t){-4.5876976689e-01f, -4.5876976689e-01f});
            pp = __builtin_elementwise_fma(a, pp, (f32x2_t){-1.1512020345f, -1.1512020345f});
            gp[v] = __builtin_elementwise_fma(a, pp, (f32x2_t){-9.9999303260e-01f, -9.9999303260e-01f});
            ga[v] = a;
        };
        auto gelu_b = [&](int v) {             // pair v done and packed
            if (DSG_M384_EXP == 3) { hf[v >> 2][v & 3] = pack_bf16(gp[v][0], gp[v][1]); return; }
            const f32x2_t e = {__builtin_amdgcn_exp2f(gp[v][0]), __builtin_amdgcn_exp2f(gp[v][1])};
            const f32x2_t r = {__builtin_amdgcn_fmed3f(h[2 * v], 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(h[2 * v + 1], 0.0f, 3.0e38f)};
            const f32x2_t o = __builtin_elementwise_fma(-ga[v], e, r);
            hf[v >> 2][v & 3] = pack_bf16(o[0], o[1]);
        };
        f32x16 hn = h;
        if (c + 1 < NCH) {
            hn = b1_init(c + 1);
            const char *a = lds + slot_n * M384_STAGE + w1off;
            bf16x8 fr[4];
#pragma unroll
            for (int s = 0; s < 4; s++) fr[s] = *reinterpret_cast<const bf16x8 *>(a + s * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 24; s++) {
                if (DSG_M384_EXP != 2) hn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s & 3], xf[s], hn, 0, 0, 0);
                if (s + 4 < 24 && DSG_M384_EXP != 5) fr[s & 3] = *reinterpret_cast<const bf16x8 *>(a + (s + 4) * 1024);
                if (s % 4 == 0) gelu_a(s >> 2);
                if (s % 4 == 2) gelu_b(s >> 2);
                if ((s & 1) && DSG_M384_EXP != 1) dma_piece(rsrc_stage, slot_r, s >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int v = 0; v < 6; v++) { gelu_a(v); gelu_b(v); }
        }
        if (c == 8) M384_STAMP(7);
        // ---- fc2 of chunk c on all 384 output channels, s2-major
        {
            const char *a = lds + slot * M384_STAGE + w2off;
            bf16x8 fr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) fr[i] = *reinterpret_cast<const bf16x8 *>(a + 2 * i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 24; i++) {
                const int ct = i % 12, s2 = i / 12;
                if (DSG_M384_EXP != 2) oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i & 3], __builtin_bit_cast(bf16x8, hf[s2]), oacc[ct], 0, 0, 0);
                if (i + 4 < 24 && DSG_M384_EXP != 5) fr[i & 3] = *reinterpret_cast<const bf16x8 *>(a + (2 * ((i + 4) % 12) + (i + 4) / 12) * 1024);
                if (i < 12) { if (i % 6 == 0) gelu_a(6 + i / 6); else if (i % 6 == 3) gelu_b(6 + i / 6); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (c == 8) M384_STAMP(8);
        h = hn;
        slot = slot_n;
    }
    M384_STAMP(3);
    M384_WAIT_VM(0);
    __syncthreads();                           // the ring is free: row tiles for the stores
    // ---- epilogue: lane (token, half) holds channels 32 ct + 8 q + 4 half + {0..3} in oacc[ct][4 q ..]; the row's statistics are lane-local
    const rsrc_t rsX = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
    __bf16 *xo = static_cast<__bf16 *>(g.xn_out);
    const rsrc_t rsO = make_rsrc(xo ? xo + (size_t)m0 * C : nullptr, xo ? (unsigned)rows * C * 2u : 0u);
    const float *aff_row = nullptr;
    if (MOD != 0) aff_row = g.mod_aff + (size_t)(MOD == 2 ? min(m0 + (int)mrow, g.M - 1) / g.mod_T : 0) * g.mod_ld + g.mod_off;
    float *xt = reinterpret_cast<float *>(lds + wave * 19456);
    __bf16 *T = reinterpret_cast<__bf16 *>(lds + wave * 19456 + 12800);
    auto xflush = [&](int q96) {                  // the wave's 32 x 96 fp32 tile -> x rows
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int i = lane + 64 * k, r = i / 24, pc = i - 24 * r;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(xt + r * 100 + 4 * pc);
            buf_store4(d, rsX, ((unsigned)(32 * wave + r) * C + (unsigned)(96 * q96 + 4 * pc)) * 4u, 0u);
        }
    };
    auto tflush = [&](int q96) {                  // the wave's 32 x 96 bf16 tile -> xn_out rows
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k, r = i / 12, pc = i - 12 * r;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(T + r * TLD + 8 * pc);
            buf_store_u4(d, rsO, ((unsigned)(32 * wave + r) * C + (unsigned)(96 * q96 + 8 * pc)) * 2u, 0u);
        }
    };
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        f32x4 rr[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            rr[q] = PROJ ? (f32x4){0.f, 0.f, 0.f, 0.f} : buf_load4(rsX, (mrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cch = 32 * ct + 8 * q + 4 * lhalf;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.b2 + cch);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = oacc[ct][4 * q + e] + b4[e] + rr[q][e];
            if (MOD != 0) {
                const f32x4 scl = *reinterpret_cast<const f32x4 *>(aff_row + cch), sft = *reinterpret_cast<const f32x4 *>(aff_row + C + cch);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = silu_exact(fmaf(v[e], scl[e] + 1.0f, sft[e]));
            }
            *reinterpret_cast<f32x4 *>(xt + lrow * 100 + 32 * (ct % 3) + 8 * q + 4 * lhalf) = v;
#pragma unroll
            for (int e = 0; e < 4; e++) { ssum += v[e]; ssq = fmaf(v[e], v[e], ssq); oacc[ct][4 * q + e] = v[e]; }
            if (g.out_mode == 2) *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
        }
        if (ct % 3 == 2) xflush(ct / 3);
        if (g.out_mode == 2 && ct % 3 == 2) tflush(ct / 3);
    }
    if (g.out_mode == 1) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        const float mean = ssum * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, ssq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(oacc[ct][4 * q + e], rstd, nmr);
                *reinterpret_cast<u32x2 *>(T + lrow * TLD + 32 * (ct % 3) + 8 * q + 4 * lhalf) = pack_bf16x4(v);
            }
            if (ct % 3 == 2) tflush(ct / 3);
        }
    }
    M384_STAMP(4);
#undef M384_STAMP
}

// -------------------------------------------------------------------------------------------------
// mlp96r_bx_kernel -- the level-0 (C = 96) proj + MLP half of a Swin block with the weights RESIDENT in LDS.
// tools/mlpb_exp.sh on mlp_bx_kernel<96> (343 us at M = 819200): without the per-block weight staging 247 us -- every 128-token block
// re-stages all 147 KB of W1 | W2 through registers into a double-buffered LDS stage, twelve barriers per block.  At C = 96 the two MLP
// weights are 2 x 72 KB as bf16: they fit the CU's 160 KB of LDS whole.  So: ONE persistent block per CU (eight waves, two per SIMD)
// copies a plan-time fragment-order image of W1 | W2 into LDS once (LDS-DMA, 144 requests), b1 | bp | b2 into the LDS left over, and then
// every wave walks 32-token tiles on its own: no barrier, no staging, no other wave's data -- a fragment read is base + lane 16 + immediate
// (the proj weight's 18 fragments do not fit beside them: one contiguous-KiB load each per tile, L2-resident).  (C = 192 would need 590 KB.)
// -------------------------------------------------------------------------------------------------
constexpr int R96_W2 = 73728, R96_WP = 147456, R96_IMG = 165888, R96_B = 147456;   // image: W1 frags | W2 frags | Wp frags;  LDS: ... | b1 b p b2 floats
__global__ __launch_bounds__(256) void mlp96r_img_kernel(const unsigned short *__restrict__ W1, const unsigned short *__restrict__ W2,
                                                         const unsigned short *__restrict__ Wp, unsigned short *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;            // one 16-byte piece
    if (idx >= R96_IMG / 16) return;
    const int lane = idx & 63, lrow = lane & 31, lhalf = lane >> 5;
    unsigned short v[8];
    if (idx < 4608) {                                          // W1 fragment block (hc, s): hidden units 32 hc + lrow; channels in the ACCUMULATOR order of the
        const int blk = idx >> 6, hc = blk / 6, sk = blk % 6;  // normalised row that fc1 meets (it is packed straight from the proj accumulators)
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = W1[(size_t)(32 * hc + lrow) * 96 + 16 * sk + 8 * (j >> 2) + 4 * lhalf + (j & 3)];
    } else if (idx < 9216) {                                   // W2 fragment block (hc, ct, s2): channels 32 ct + lrow, hidden units in accumulator order
        const int blk = (idx - 4608) >> 6, hc = blk / 6, ct = (blk % 6) >> 1, s2 = blk & 1;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = W2[(size_t)(32 * ct + lrow) * 384 + 32 * hc + 16 * s2 + 8 * (j >> 2) + 4 * lhalf + (j & 3)];
    } else {                                                   // Wp fragment block (ct, s)
        const int blk = (idx - 9216) >> 6, ct = blk / 6, sk = blk % 6;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = Wp[(size_t)(32 * ct + lrow) * 96 + 16 * sk + 8 * lhalf + j];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) img[(size_t)idx * 8 + j] = v[j];
}
size_t mlp96r_image_bytes() { return R96_IMG; }
void launch_mlp96r_image(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s) {
    DSG_LAUNCH(mlp96r_img_kernel, dim3((R96_IMG / 16 + 255) / 256), dim3(256), 0, s, (const unsigned short *)W1b, (const unsigned short *)W2b,
               (const unsigned short *)Wpb, (unsigned short *)img);
}

__global__ __launch_bounds__(512, 1) void mlp96r_bx_kernel(BxMlp g, int ntiles) {
    constexpr int C = 96, KS = 6, CT = 3, NCH = 12;
    __shared__ __attribute__((aligned(16))) char lds[R96_B + (384 + 96 + 96) * 4];
    float *b1s = reinterpret_cast<float *>(lds + R96_B), *bps = b1s + 384, *b2s = bps + 96;
    const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 31, lhalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char *img = static_cast<const char *>(g.img96);
    {   // W1 | W2 image -> LDS, 144 KiB: eighteen 1-KiB requests per wave (inline assembly: see mlp384d_bx_kernel)
        const unsigned lds0 = (unsigned)(size_t)lds, voff = (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < 18; i++)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + (wave * 18 + i) * 1024), "v"(voff), "s"(img + (wave * 18 + i) * 1024) : "memory");
    }
    for (int i = tid; i < 384 + 96 + 96; i += 512) b1s[i] = i < 384 ? g.b1[i] : (i < 480 ? g.bp[i - 384] : g.b2[i - 480]);
    M384_WAIT_VM(0);
    __syncthreads();
    const char *w1l = lds + lane * 16, *w2l = lds + R96_W2 + lane * 16;
    __bf16 *xo = static_cast<__bf16 *>(g.xn_out);
    for (int wt = blockIdx.x * 8 + wave; wt < ntiles; wt += (int)gridDim.x * 8) {
        const int m0 = wt * 32, rows = min(32, g.M - m0);
        const rsrc_t rsA = make_rsrc(static_cast<const __bf16 *>(g.att) + (size_t)m0 * C, (unsigned)rows * C * 2u);
        const rsrc_t rsX = make_rsrc(g.x + (size_t)m0 * C, (unsigned)rows * C * 4u);
        const rsrc_t rsO = make_rsrc(xo ? xo + (size_t)m0 * C : nullptr, xo ? (unsigned)rows * C * 2u : 0u);
        bf16x8 xf[KS];
#pragma unroll
        for (int sk = 0; sk < KS; sk++) xf[sk] = __builtin_bit_cast(bf16x8, buf_load_u4(rsA, ((unsigned)lrow * C + 16u * sk + 8u * lhalf) * 2u, 0u));
        bf16x8 wpf[CT][KS];                    // the proj weight's 18 fragments (L2-resident, one contiguous KiB per load; they do not fit the LDS
#pragma unroll                                 // beside W1 | W2, and held for the block's lifetime they cost 72 registers through the chunk loop)
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int sk = 0; sk < KS; sk++) wpf[ct][sk] = *reinterpret_cast<const bf16x8 *>(img + R96_WP + (ct * 6 + sk) * 1024 + lane * 16);
        f32x4 rr[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) rr[ct][q] = buf_load4(rsX, ((unsigned)lrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 4u, 0u);
        // x1 = x + att Wp^T + bp: the shortcut and fc2's initial value; LayerNorm-2 of it -> fc1's B operand
        f32x16 oacc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;
#pragma unroll
            for (int sk = 0; sk < KS; sk++) oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpf[ct][sk], xf[sk], oacc[ct], 0, 0, 0);
        }
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bps + 32 * ct + 8 * q + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = oacc[ct][4 * q + e] + b4[e] + rr[ct][q][e];
                    oacc[ct][4 * q + e] = v; sm += v; sq = fmaf(v, v, sq);
                }
            }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        {
            const float mean = sm * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
            for (int sk = 0; sk < KS; sk++) {
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    pk[j] = pack_bf16(fmaf(oacc[sk >> 1][8 * (sk & 1) + 2 * j], rstd, nmr), fmaf(oacc[sk >> 1][8 * (sk & 1) + 2 * j + 1], rstd, nmr));
                xf[sk] = __builtin_bit_cast(bf16x8, pk);
            }
        }
        // (xf's k order: k-step s, half g, element e <-> channel 16 s + 8 (e >> 2) + 4 g + (e & 3) -- the proj accumulators' register order;
        //  W1's image has the same order)
        // The chunk loop is software-pipelined by hand and pinned with sched_barrier(0): left to itself the compiler reads every fragment
        // immediately in front of its MFMA (one register quad for all of them, a full LDS latency per MFMA).  fc2's six fragments are
        // requested before fc1's MFMAs, the next chunk's fc1 fragments before the GELU.
        bf16x8 f1[KS], f2[2 * CT];
#pragma unroll
        for (int sk = 0; sk < KS; sk++) f1[sk] = *reinterpret_cast<const bf16x8 *>(w1l + sk * 1024);
        for (int hc = 0; hc < NCH; hc++) {
            f32x16 hacc;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b1s + 32 * hc + 8 * q + 4 * lhalf);
                hacc[4 * q] = b4[0]; hacc[4 * q + 1] = b4[1]; hacc[4 * q + 2] = b4[2]; hacc[4 * q + 3] = b4[3];
            }
#pragma unroll
            for (int i = 0; i < 2 * CT; i++) f2[i] = *reinterpret_cast<const bf16x8 *>(w2l + (hc * 6 + i) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sk = 0; sk < KS; sk++) hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[sk], xf[sk], hacc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (hc + 1 < NCH) {
#pragma unroll
                for (int sk = 0; sk < KS; sk++) f1[sk] = *reinterpret_cast<const bf16x8 *>(w1l + ((hc + 1) * 6 + sk) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32x4 hf[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2_t gg = gelu_f2(hacc[r], hacc[r + 1]); hf[r >> 3][(r & 7) >> 1] = pack_bf16(gg[0], gg[1]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
                    oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[2 * ct + s2], __builtin_bit_cast(bf16x8, hf[s2]), oacc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: lane (token, half) holds channels 32 ct + 8 q + 4 half + {0..3} in oacc[ct][4 q ..]
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = 32 * ct + 8 * q + 4 * lhalf;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b2s + c);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) { v[e] = oacc[ct][4 * q + e] + b4[e]; ssum += v[e]; ssq = fmaf(v[e], v[e], ssq); oacc[ct][4 * q + e] = v[e]; }
                buf_store4(v, rsX, ((unsigned)lrow * C + (unsigned)c) * 4u, 0u);
                if (g.out_mode == 2) __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(v), rsO, ((unsigned)lrow * C + (unsigned)c) * 2u, 0u, 0);
            }
        if (g.out_mode == 1) {
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            const float mean = ssum * (1.0f / C), rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, ssq * (1.0f / C)), 0.f) + LN_EPS), nmr = -mean * rstd;
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaf(oacc[ct][4 * q + e], rstd, nmr);
                    __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(v), rsO, ((unsigned)lrow * C + (unsigned)(32 * ct + 8 * q + 4 * lhalf)) * 2u, 0u, 0);
                }
        }
    }
}

bool launch_mlp_bx(const BxMlp &g_in, hipStream_t s) {
    const BxMlp &g = g_in;
    if ((!g.xn && !g.att) || !g.x || !g.W1 || !g.b1 || !g.W2 || !g.b2 || g.M < 1 || (g.out_mode && !g.xn_out)) return false;
    const dim3 grid((g.M + 127) / 128), block(256);
    const int mod = !g.mod_aff ? 0 : (g.mod_ld == 0 ? 1 : 2);
    const bool proj = g.att != nullptr;
    if (proj && (!g.Wp || !g.bp)) return false;
#define MLP_LAUNCH(C_)                                                                                   \
    do {                                                                                                 \
        if (proj) {                                                                                      \
            if (mod == 0) DSG_LAUNCH((mlp_bx_kernel<C_, 0, true>), grid, block, 0, s, g);        \
            else if (mod == 1) DSG_LAUNCH((mlp_bx_kernel<C_, 1, true>), grid, block, 0, s, g);   \
            else DSG_LAUNCH((mlp_bx_kernel<C_, 2, true>), grid, block, 0, s, g);                 \
        } else {                                                                                         \
            if (mod == 0) DSG_LAUNCH((mlp_bx_kernel<C_, 0>), grid, block, 0, s, g);              \
            else if (mod == 1) DSG_LAUNCH((mlp_bx_kernel<C_, 1>), grid, block, 0, s, g);         \
            else DSG_LAUNCH((mlp_bx_kernel<C_, 2>), grid, block, 0, s, g);                       \
        }                                                                                                \
    } while (0)
    if (g.C == 96 && proj && mod == 0 && g.img96) {   // level 0: weights resident in LDS, persistent blocks (mlp96r_bx_kernel)
        const int ntiles = (g.M + 31) / 32, nb = std::max(1, std::min(bx_cu_count(), (ntiles + 7) / 8));
        DSG_LAUNCH(mlp96r_bx_kernel, dim3(nb), dim3(512), 0, s, g, ntiles);
        return true;
    }
    switch (g.C) {
        case 96: MLP_LAUNCH(96); break;
        case 192: MLP_LAUNCH(192); break;
        case 384:
            if (g.wide8 == 3 && g.img2) {  // four waves, one per SIMD, chunk-major weight images (mlp384s_bx_kernel)
                const dim3 block4(256);
                if (proj) {
                    if (mod == 0) DSG_LAUNCH((mlp384s_bx_kernel<0, true>), grid, block4, 0, s, g_in);
                    else if (mod == 1) DSG_LAUNCH((mlp384s_bx_kernel<1, true>), grid, block4, 0, s, g_in);
                    else DSG_LAUNCH((mlp384s_bx_kernel<2, true>), grid, block4, 0, s, g_in);
                } else {
                    if (mod == 0) DSG_LAUNCH((mlp384s_bx_kernel<0>), grid, block4, 0, s, g_in);
                    else if (mod == 1) DSG_LAUNCH((mlp384s_bx_kernel<1>), grid, block4, 0, s, g_in);
                    else DSG_LAUNCH((mlp384s_bx_kernel<2>), grid, block4, 0, s, g_in);
                }
            } else if (g.wide8 == 1 && g.img) {   // eight waves, pre-arranged weight images streamed by LDS-DMA (mlp384d_bx_kernel)
                const dim3 block8(512);
                static const int skew_env = getenv("DSG_M384_SKEW") ? atoi(getenv("DSG_M384_SKEW")) : -1;   // dev knob: stagger step in clocks
                BxMlp g = g_in;
                if (skew_env >= 0) g.skew = skew_env;
                if (proj) {
                    if (mod == 0) DSG_LAUNCH((mlp384d_bx_kernel<0, true>), grid, block8, 0, s, g);
                    else if (mod == 1) DSG_LAUNCH((mlp384d_bx_kernel<1, true>), grid, block8, 0, s, g);
                    else DSG_LAUNCH((mlp384d_bx_kernel<2, true>), grid, block8, 0, s, g);
                } else {
                    if (mod == 0) DSG_LAUNCH((mlp384d_bx_kernel<0>), grid, block8, 0, s, g);
                    else if (mod == 1) DSG_LAUNCH((mlp384d_bx_kernel<1>), grid, block8, 0, s, g);
                    else DSG_LAUNCH((mlp384d_bx_kernel<2>), grid, block8, 0, s, g);
                }
            } else if (g.wide8) {   // round 3's eight-wave kernel (register-staged weights): wide8 = 2, or no image
                const dim3 block8(512);
                if (proj) {
                    if (mod == 0) DSG_LAUNCH((mlp384_bx_kernel<0, true>), grid, block8, 0, s, g);
                    else if (mod == 1) DSG_LAUNCH((mlp384_bx_kernel<1, true>), grid, block8, 0, s, g);
                    else DSG_LAUNCH((mlp384_bx_kernel<2, true>), grid, block8, 0, s, g);
                } else {
                    if (mod == 0) DSG_LAUNCH((mlp384_bx_kernel<0>), grid, block8, 0, s, g);
                    else if (mod == 1) DSG_LAUNCH((mlp384_bx_kernel<1>), grid, block8, 0, s, g);
                    else DSG_LAUNCH((mlp384_bx_kernel<2>), grid, block8, 0, s, g);
                }
            } else {
                MLP_LAUNCH(384);
            }
            break;
        default: return false;
    }
#undef MLP_LAUNCH
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
    DSG_LAUNCH(ln_bx_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, aff, aff_ld, aff_off, (__bf16 *)xn, T, C, M, ln ? 1 : 0);
}

// -------------------------------------------------------------------------------------------------
// Window attention on bf16 q, k, v (diffusesg.py:108-139; window partition / cyclic shift / reverse folded into the token index as in
// window_attn_kernel).  One wave per (sample, window, head) unit; Wp = 32 KT >= WS^2 positions.
//   S^T[key][query] = K Q^T + bias   A operand = K rows, B operand = Q rows (q pre-scaled by d^-1/2 log2 e in the QKV weights): a lane
//                                    owns one query column, so the softmax is lane-local (+ one exchange between the half-waves);
//   O^T[d][query]  = V^T P^T         B operand = the S^T accumulators themselves, converted pairwise to bf16 (an accumulator tile's
//                                    rows are exactly a k-step's k index, in the permuted order below); A operand = V^T, staged through
//                                    LDS once per wave: vt[d][pos(key)] with pos(16 s + o) = 16 s + 8 ((o >> 2) & 1) + 4 (o >> 3) + (o & 3),
//                                    so that one ds_read_b128 of lane (d, half) returns keys 16 s + 8 (j >> 2) + 4 half + (j & 3), j = 0..7
//                                    -- the k order in which registers 8 s' .. 8 s' + 7 of a 32x32 accumulator enumerate its rows.
// The result is a lane's 4 consecutive head dims per accumulator quad: 8-byte bf16 stores.
// -------------------------------------------------------------------------------------------------
// A block works on ONE (window position, head) pair and walks through samples: the pair's bias tile -- relative-position bias +
// shift mask, 64 KB of fp32 for 100-token windows, which a wave-per-unit kernel re-read from L2 for every sample (2.5x the kernel's
// HBM traffic) -- is converted once into LDS as fp16 pairs (keys 2j, 2j + 1 of one query share a dword: 32 KB, conflict-free
// ds_read_b32 along the query lanes; fp16 keeps 11 bits of the O(1) bias values, below what rounding P to bf16 costs) and every
// sample's scores start from there.  Each of the 4 waves takes every 4th sample of the block's share.
template <int KT, int WS>
__global__ __launch_bounds__(256, 2) void attn_bx_kernel(const __bf16 *__restrict__ qkv, const float *__restrict__ biasT, __bf16 *__restrict__ out,
                                                         int B, WinGeom g, int split) {
    constexpr int Wp = 32 * KT, Wt = WS * WS, VLD = Wp + 8;   // vt row stride in bf16 (16-B aligned, conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) __bf16 vt_lds[4][32 * VLD];
    __shared__ __attribute__((aligned(16))) unsigned bias_lds[(Wp / 2) * Wp];   // [key pair][query] fp16x2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int heads = g.heads, C = g.C, res = g.res;
    const int nwr = res / WS, T = res * res;
    const int pair = blockIdx.x / split, part = blockIdx.x % split;   // pair = (window position, head)
    const int head = pair % heads, w = pair / heads;
    const int wi = w / nwr, wj = w % nwr;
    {   // bias tile -> LDS (fp16 pairs along the key dimension); -1e30 of the padded key slots saturates to a finite fp16
        const float *src = biasT + ((size_t)(g.shift > 0 ? w : 0) * heads + head) * Wp * Wp;
        for (int idx = tid; idx < (Wp / 2) * Wp; idx += 256) {
            const int kp = idx / Wp, q = idx % Wp;
            const float lo = fmaxf(src[(2 * kp) * Wp + q], -60000.f), hi = fmaxf(src[(2 * kp + 1) * Wp + q], -60000.f);
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 pk = {(_Float16)lo, (_Float16)hi};
            bias_lds[idx] = __builtin_bit_cast(unsigned, pk);
        }
    }
    __syncthreads();
    auto token_of = [&](int p) -> int {   // window position -> token of the sample (padded positions: token 0)
        if (p >= Wt) return 0;
        int ti = wi * WS + p / WS + g.shift, tj = wj * WS + p % WS + g.shift;
        if (ti >= res) ti -= res;
        if (tj >= res) tj -= res;
        return ti * res + tj;
    };
    int tokr[KT];
    unsigned rowoff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        tokr[kt] = token_of(32 * kt + lrow);
        rowoff[kt] = (unsigned)tokr[kt] * (unsigned)(3 * C) * 2u + 16u * lhalf;   // this lane's 8 head dims of k-step s: + 32 s bytes
    }
    // V^T staging items of this lane: (key pair kp, group of 8 head dims dg); token offsets are sample-independent
    unsigned vtoff_a[KT], vtoff_b[KT];
    int vtdst[KT];
#pragma unroll
    for (int it = 0; it < KT; it++) {
        const int item = lane + 64 * it, kp = item >> 2, dg = item & 3;
        const int k0 = 2 * kp;
        vtoff_a[it] = (unsigned)token_of(k0) * (unsigned)(3 * C) * 2u + 16u * dg;
        vtoff_b[it] = (unsigned)token_of(k0 + 1) * (unsigned)(3 * C) * 2u + 16u * dg;
        const int o = k0 & 15, pos = (k0 & ~15) + 8 * ((o >> 2) & 1) + 4 * (o >> 3) + (o & 3);
        vtdst[it] = (8 * dg) * VLD + pos;
    }
    const unsigned hq = (unsigned)head * 64u, hk = (unsigned)(C + head * 32) * 2u, hv = (unsigned)(2 * C + head * 32) * 2u;
    __bf16 *vt = vt_lds[wave];
    // the block's share of the batch: samples [b_lo, b_hi), this wave takes b_lo + wave, + 4, ...
    const int per = (B + split - 1) / split, b_lo = part * per, b_hi = min(B, b_lo + per);
    for (int b = b_lo + wave; b < b_hi; b += 4) {
        const rsrc_t rsQ = make_rsrc(qkv + (size_t)b * T * 3 * C, (unsigned)T * 3u * C * 2u);
        const rsrc_t rsO = make_rsrc(out + (size_t)b * T * C, (unsigned)T * C * 2u);
        // K fragments (lane = key row), both k-steps
        bf16x8 kf[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int s = 0; s < 2; s++) kf[kt][s] = __builtin_bit_cast(bf16x8, buf_load_u4(rsQ, rowoff[kt], hk + 32u * s));
        // V^T -> LDS: the two keys' values of one head dim share a dword
        u32x4 va[KT], vb[KT];
#pragma unroll
        for (int it = 0; it < KT; it++) { va[it] = buf_load_u4(rsQ, vtoff_a[it], hv); vb[it] = buf_load_u4(rsQ, vtoff_b[it], hv); }
        __builtin_amdgcn_wave_barrier();   // (the previous sample's fragment reads of vt are done: same wave, in order)
#pragma unroll
        for (int it = 0; it < KT; it++) {
            unsigned *dst = reinterpret_cast<unsigned *>(vt + vtdst[it]);
#pragma unroll
            for (int e = 0; e < 4; e++) {   // dword e of va / vb holds head dims 8 dg + 2 e, + 1
                dst[(2 * e) * (VLD / 2)] = __builtin_amdgcn_perm(vb[it][e], va[it][e], 0x05040100u);       // (va.lo, vb.lo)
                dst[(2 * e + 1) * (VLD / 2)] = __builtin_amdgcn_perm(vb[it][e], va[it][e], 0x07060302u);   // (va.hi, vb.hi)
            }
        }
        __builtin_amdgcn_wave_barrier();
        // V^T fragments: lane (d = lrow, half), k-step (kt, s): positions 32 kt + 16 s + 8 half .. + 7
        bf16x8 vf[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int s = 0; s < 2; s++) vf[kt][s] = *reinterpret_cast<const bf16x8 *>(vt + lrow * VLD + 32 * kt + 16 * s + 8 * lhalf);
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
                // registers r, r + 1 (r even) are keys 32 kt + 8 (r >> 2) + 4 half + (r & 3), + 1: one fp16 pair
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int kp = (32 * kt + 8 * (r >> 2) + (r & 3)) / 2 + 2 * lhalf;
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 pk = __builtin_bit_cast(h2, bias_lds[kp * Wp + 32 * qt + lrow]);
                    sacc[kt][r] = (float)pk[0]; sacc[kt][r + 1] = (float)pk[1];
                }
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
}

// -------------------------------------------------------------------------------------------------
// QKV projection + window attention in one kernel (diffusesg.py:108-139 with the window partition / cyclic shift / reverse of :246-270
// folded into the token index): q, k, v never reach HBM.  A block owns 128 window positions -- one 100-token window padded to 128, two
// 64-token windows, four 16/25-token windows padded to 32 -- of ONE head:
//   1. [128 positions x C] (rows gathered from xn through the partition) x the head's [96 x C] slice of the QKV weight (32 q, 32 k,
//      32 v rows; q pre-scaled by d^-1/2 log2 e, LayerNorm-1's gamma / beta folded in), the K loop of gemm_bx_kernel with 32 x 96 wave
//      tiles (transposed product: a lane owns one window position, its accumulators are 16 features of q, of k and of v);
//   2. q stays in registers as the B operand of S^T = K Q^T (register r <-> feature (r & 3) + 8 (r >> 2) + 4 half: the same feature
//      order on both operands, so the contraction does not care); k goes to LDS in exactly that order (the writer's 16 bytes are the
//      reader's A fragment); v goes to LDS transposed, vt[d][pos(key)], as attn_bx_kernel stages it;
//   3. softmax and O^T = V^T P^T as in attn_bx_kernel, the bias tile read as fp16 in accumulator order (bias_permute_bx_kernel: 64
//      bytes per lane and key tile, straight into the score accumulators' initial values).
// Blocks of one window group run back to back on one XCD (head fastest), so the group's xn rows are read from HBM once.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_permute_bx_kernel(const float *__restrict__ biasT, _Float16 *__restrict__ out, int n_tiles, int Wp) {
    // out[tile][query][half][kt][r] = biasT[tile][key = 32 kt + (r & 3) + 8 (r >> 2) + 4 half][query]  (clamped into fp16 range)
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n_tiles * Wp * Wp) return;
    const int KT = Wp / 32;
    const int r = idx % 16, kt = (idx / 16) % KT, half = (idx / (16 * KT)) % 2, q = (idx / (32 * KT)) % Wp;
    const size_t tile = idx / ((size_t)Wp * Wp);
    const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
    out[idx] = (_Float16)fmaxf(biasT[(tile * Wp + key) * Wp + q], -60000.f);
}
// The tiles in fp32 for qkv_attn_wx_kernel, which loads them straight into the score accumulators (no conversion, no moves), LANE-INTERLEAVED:
// out[tile][query tile][key tile][quad q][lane][4] -- one load instruction of a wave is one contiguous KiB (whole cache lines; with a lane's
// 256 bytes contiguous instead, every instruction touched 64 lines and the loads, not the arithmetic, bounded the attention phase).
__global__ __launch_bounds__(256) void bias_permute_f32_kernel(const float *__restrict__ biasT, float *__restrict__ out, int n_tiles, int Wp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n_tiles * Wp * Wp) return;
    const int KT = Wp / 32;
    const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, kt = (idx >> 10) % KT, qt = (idx / ((size_t)1024 * KT)) % KT;
    const size_t tile = idx / ((size_t)Wp * Wp);
    const int r = 4 * q + e, half = lane >> 5, query = 32 * qt + (lane & 31);
    const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
    out[idx] = biasT[(tile * Wp + key) * Wp + query];
}
void launch_bias_permute_f32(const float *biasT, float *out, int n_tiles, int Wp, hipStream_t s) {
    const size_t n = (size_t)n_tiles * Wp * Wp;
    DSG_LAUNCH(bias_permute_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, biasT, out, n_tiles, Wp);
}
void launch_bias_permute_bx(const float *biasT, void *out, int n_tiles, int Wp, hipStream_t s) {
    const size_t n = (size_t)n_tiles * Wp * Wp;
    DSG_LAUNCH(bias_permute_bx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, biasT, (_Float16 *)out, n_tiles, Wp);
}

#ifndef DSG_QA_EXP
#define DSG_QA_EXP 0   // timing experiments (wrong results): 1 no attention arithmetic, 2 no MFMAs in the K loop, 3 no global loads in the K loop, 4 no LDS staging writes
#endif
#ifndef DSG_QA_OCC
#define DSG_QA_OCC 4
#endif
template <int KT, int WS>
__global__ __launch_bounds__(256, DSG_QA_OCC) void qkv_attn_bx_kernel(BxQkvAttn a, int nblk, int U) {
    constexpr int KB = 64, LDP = KB + 8, Wp = 32 * KT, Wt = WS * WS, UPB = 4 / KT, KLD = 40, VLD = Wp + 8;
    constexpr int STAGE = (128 + 96) * LDP, OLD = 40;
    static_assert(128 * KLD + UPB * 32 * VLD <= STAGE, "k / v^T live in the tile stage after the K loop");
    __shared__ __attribute__((aligned(16))) __bf16 lds[STAGE + 2 * 96];
    float *colv = reinterpret_cast<float *>(lds + STAGE);      // the head's 96 bias values (q | k | v)
    __bf16 *kl = lds, *vtl = lds + 128 * KLD;
    const unsigned OOB = 0x7fffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int heads = a.g.heads, C = a.g.C, res = a.g.res, shift = a.g.shift;
    const int nwr = res / WS, nW = nwr * nwr, T = res * res;
    const int t = blockIdx.x, xcd = t & 7, seq = t >> 3;
    const int ub = (seq / heads) * 8 + xcd, head = seq % heads;
    if (ub >= nblk) return;
    // A unit's geometry (sample, window row / column -> first token row / column after the cyclic shift) is WAVE-UNIFORM: the two runtime
    // divisions are done once per unit on scalar values, not once per row and lane (round 4: the kernel ran 16 VALU instructions per MFMA,
    // a third of them this index arithmetic; tools/mfma_rate.cpp: every VALU instruction beyond two per MFMA costs matrix time).
    struct Unit { int tb, i0, j0; };
    auto unit_of = [&](int ul) -> Unit {               // ul: uniform
        const int u = __builtin_amdgcn_readfirstlane(ub * UPB + ul);
        if (u >= U) return {-1, 0, 0};
        const int b = u / nW, w = u - b * nW, wi = w / nwr, wj = w - wi * nwr;
        return {b * T, wi * WS + shift, wj * WS + shift};
    };
    // token of window position pos of a unit: element offset in xn / out, or -1 (padding position / no such unit)
    auto pos_token = [&](const Unit &un, int pos) -> int {
        const int pi = pos / WS, pj = pos - pi * WS;   // WS is a template value: multiply + shift
        int ti = un.i0 + pi, tj = un.j0 + pj;
        if (ti >= res) ti -= res;
        if (tj >= res) tj -= res;
        return (pos >= Wt || un.tb < 0) ? -1 : un.tb + ti * res + tj;
    };
    const __bf16 *xn = static_cast<const __bf16 *>(a.xn), *Wq = static_cast<const __bf16 *>(a.W);
    const rsrc_t rsA = make_rsrc(xn, (unsigned)((size_t)a.B * T * C * 2u));
    const rsrc_t rsW = make_rsrc(Wq, (unsigned)(3u * C * C * 2u));
    const int sc = tid & 7, sr = tid >> 3;             // this thread's 16-byte piece / first stage row
    unsigned offA[4], offW[3];
#pragma unroll
    for (int p = 0; p < 4; p++) {                      // stage row sr + 32 p: unit (32 p) / Wp, window position sr + (32 p) % Wp
        const Unit un = unit_of((32 * p) / Wp);
        const int tk = pos_token(un, sr + (32 * p) % Wp);
        offA[p] = tk < 0 ? OOB : ((unsigned)tk * C + 8u * sc) * 2u;
    }
#pragma unroll
    for (int p = 0; p < 3; p++) offW[p] = ((unsigned)(p * C + head * 32 + sr) * C + 8u * sc) * 2u;
    const int nk = (C + KB - 1) / KB;                  // >= 2 (launcher)
    struct Stage { u32x4 a[4], w[3]; };
    auto issue = [&](Stage &st, int kc) {
        const int k0 = kc * KB;
        if (k0 + KB > C) {                               // C = 96: the second chunk is half valid (a uniform branch: no mask arithmetic on full chunks)
            const unsigned kmask = (k0 + 8 * sc < C) ? 0u : OOB;
#pragma unroll
            for (int p = 0; p < 4; p++) st.a[p] = buf_load_u4(rsA, offA[p] | kmask, (unsigned)k0 * 2u);
#pragma unroll
            for (int p = 0; p < 3; p++) st.w[p] = buf_load_u4(rsW, offW[p] | kmask, (unsigned)k0 * 2u);
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) st.a[p] = buf_load_u4(rsA, offA[p], (unsigned)k0 * 2u);
#pragma unroll
            for (int p = 0; p < 3; p++) st.w[p] = buf_load_u4(rsW, offW[p], (unsigned)k0 * 2u);
        }
    };
    auto write = [&](const Stage &st) {
        if (DSG_QA_EXP == 4 && st.a[0][0] != 0x12345u) return;
#pragma unroll
        for (int p = 0; p < 4; p++) *reinterpret_cast<u32x4 *>(lds + (sr + 32 * p) * LDP + 8 * sc) = st.a[p];
#pragma unroll
        for (int p = 0; p < 3; p++) *reinterpret_cast<u32x4 *>(lds + (128 + sr + 32 * p) * LDP + 8 * sc) = st.w[p];
    };
    f32x16 acc[3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;
    const __bf16 *Afr = lds + (32 * wave + lrow) * LDP + 8 * lhalf;
    const __bf16 *Wfr = lds + (128 + lrow) * LDP + 8 * lhalf;
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < KB / 16; s++) {
            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(Afr + 16 * s);
#pragma unroll
            for (int nt = 0; nt < 3; nt++) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(Wfr + 32 * nt * LDP + 16 * s);
                if (DSG_QA_EXP != 2) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[nt], 0, 0, 0);
                else acc[nt][0] += (float)wf[0] + (float)af[0];
            }
        }
    };
    Stage s0, s1;
    issue(s0, 0);
    issue(s1, 1);
    float cvb = 0.f;
    if (tid < 96) cvb = a.bias[(tid >> 5) * C + head * 32 + (tid & 31)];
    write(s0);
    __syncthreads();
    for (int kc = 0; kc < nk; kc += 2) {   // LDS holds chunk kc, s1 holds chunk kc + 1, s0 is free
        if (kc + 2 < nk && DSG_QA_EXP != 3) issue(s0, kc + 2);
        compute();
        __syncthreads();
        if (kc + 1 < nk) {
            write(s1);
            __syncthreads();
            if (kc + 3 < nk && DSG_QA_EXP != 3) issue(s1, kc + 3);
            compute();
            __syncthreads();
            if (kc + 2 < nk) { write(s0); __syncthreads(); }
        }
    }
    // ---- this wave's attention unit: window positions 32 wave .. + 31 are query tile qt of unit ul ----
    const int ul = wave / KT, qt = wave % KT;
    const int u = __builtin_amdgcn_readfirstlane(ub * UPB + ul), w_of_u = (u < U ? u : 0) % nW;
    // bias tile of this lane's query in accumulator order (fp16): issued now, consumed after the LDS phase
    const u32x4 *bp = reinterpret_cast<const u32x4 *>(a.biasP) +
                      ((((size_t)(shift > 0 ? w_of_u : 0) * heads + head) * Wp + 32 * qt + lrow) * 2 + lhalf) * (KT * 2);
    u32x4 bb[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) { bb[kt][0] = bp[2 * kt]; bb[kt][1] = bp[2 * kt + 1]; }
    if (tid < 96) colv[tid] = cvb;
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(colv + 32 * nt + 8 * q + 4 * lhalf);
#pragma unroll
            for (int e = 0; e < 4; e++) acc[nt][4 * q + e] += b4[e];
        }
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        u32x4 pq, pk;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            pq[j] = pack_bf16(acc[0][8 * s + 2 * j], acc[0][8 * s + 2 * j + 1]);
            pk[j] = pack_bf16(acc[1][8 * s + 2 * j], acc[1][8 * s + 2 * j + 1]);
        }
        qf[s] = __builtin_bit_cast(bf16x8, pq);
        *reinterpret_cast<u32x4 *>(kl + (32 * wave + lrow) * KLD + 16 * s + 8 * lhalf) = pk;
    }
    {   // v^T: this lane's key -> column pos(key) (bits 2 and 3 of the position swapped), its 16 features -> rows
        const int pos = 32 * qt + lrow, pp = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
        __bf16 *vcol = vtl + (ul * 32 + 4 * lhalf) * VLD + pp;
#pragma unroll
        for (int r = 0; r < 16; r++) vcol[((r & 3) + 8 * (r >> 2)) * VLD] = (__bf16)acc[2][r];
    }
    __syncthreads();
    bf16x8 kf[KT][2], vf[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            kf[kt][s] = *reinterpret_cast<const bf16x8 *>(kl + (ul * Wp + 32 * kt + lrow) * KLD + 16 * s + 8 * lhalf);
            vf[kt][s] = *reinterpret_cast<const bf16x8 *>(vtl + (ul * 32 + lrow) * VLD + 32 * kt + 16 * s + 8 * lhalf);
        }
    __syncthreads();   // every wave holds its k / v^T fragments: the stage is free for the output transposition at the end
    f32x16 sacc[KT];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const unsigned dw = bb[kt][r >> 3][(r & 7) >> 1];   // (a bit_cast of the vector element itself reads element 0 with this compiler)
            const h2 pk = __builtin_bit_cast(h2, dw);
            sacc[kt][r] = (float)pk[0]; sacc[kt][r + 1] = (float)pk[1];
        }
#pragma unroll
        for (int s = 0; s < 2; s++) if (DSG_QA_EXP != 1) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s], qf[s], sacc[kt], 0, 0, 0);
        // register r holds key 32 kt + (r & 3) + 8 (r >> 2) + 4 half: quads whose first key is beyond the window (the last key tile of a
        // 100-token window keeps 1 quad of 4) are padding on every lane -- no maximum, no exponential, P = 0
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (32 * kt + 8 * (r >> 2) < Wt && DSG_QA_EXP != 1) mx = fmaxf(mx, sacc[kt][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        u32x4 pf[2];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            if (32 * kt + 8 * (r >> 2) < Wt && DSG_QA_EXP != 1) {
                const float e0 = __builtin_amdgcn_exp2f(sacc[kt][r] - mx), e1 = __builtin_amdgcn_exp2f(sacc[kt][r + 1] - mx);
                sum += e0 + e1;
                pf[r >> 3][(r & 7) >> 1] = pack_bf16(e0, e1);
            } else {
                pf[r >> 3][(r & 7) >> 1] = 0u;
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (32 * kt + 16 * s < Wt && DSG_QA_EXP != 1) oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][s], __builtin_bit_cast(bf16x8, pf[s]), oacc, 0, 0, 0);
            else if (DSG_QA_EXP == 1) oacc[0] += (float)vf[kt][s][0] + (float)kf[kt][s][0] + __uint_as_float(bb[kt][s][0]);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = fast_rcp(sum);
    // O^T tile: lane = query, quad q = head dims 8 q + 4 half + {0..3}.  Through a wave-private 32 x 32 transposition in the (idle) K
    // loop stage so that a store instruction writes 16 tokens x 64 B instead of 32 tokens x 16 B (tools/store_pattern.cpp)
    __bf16 *Tq = lds + wave * 32 * OLD;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = oacc[4 * q + e] * inv;
        *reinterpret_cast<u32x2 *>(Tq + lrow * OLD + 8 * q + 4 * lhalf) = pack_bf16x4(o);
    }
    const rsrc_t rsO = make_rsrc(a.out, (unsigned)((size_t)a.B * T * C * 2u));
    const Unit uo = unit_of(ul);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = (lane >> 2) + 16 * k, pc = lane & 3;          // token row of the tile, 16-byte piece of its 64 bytes
        const int tq = pos_token(uo, 32 * qt + r);
        const u32x4 d = *reinterpret_cast<const u32x4 *>(Tq + r * OLD + 8 * pc);
        buf_store_u4(d, rsO, tq < 0 ? OOB : ((unsigned)tq * (unsigned)C + (unsigned)(head * 32 + 8 * pc)) * 2u, 0u);
    }
}

// -------------------------------------------------------------------------------------------------
// qkv_attn_wx_kernel -- the same fused QKV projection + window attention with ONE WAVE PER (window, head) unit (10 x 10 windows).
// What the block-per-head kernel above pays for (tools/qa_exp.sh: its time is the SUM of global loads, LDS staging, K-loop MFMAs and
// attention arithmetic, ~1/4 each, and 2 blocks per CU run as fast as 4): every wave reads a fresh 1-KB weight fragment per MFMA, every
// head's block stages the window's 96-KB xn tile again, and the tiles go global -> registers -> LDS.  Here
//   * a wave owns all 128 positions of its window and the head's 96 QKV columns: 12 accumulator tiles (192 registers, one wave per
//     SIMD), a weight fragment feeds FOUR MFMAs (one per row tile) and an xn fragment three: 7 KB of LDS reads per 12 MFMAs instead of 16;
//   * a block is four consecutive units, head fastest: its (at most two) windows' xn rows are staged once for all of them;
//   * staging is LDS-DMA (global_load_lds_dwordx4 by inline assembly, see mlp384d_bx_kernel): K chunks of 32 in a three-slot ring, ten
//     requests per wave and chunk with loop-invariant per-lane offsets on a scalar base, one barrier per chunk; the LDS image of a request
//     is lane-linear, so the bank swizzle (16-byte piece c of row r at slot c ^ ((r >> 2) & 3)) is applied to the SOURCE address;
//   * (tried, not kept: PERSISTENT blocks that request the next set's first chunks during the attention phase -- the per-set setup and first
//     wait (2.7k clocks) disappear, but every s_waitcnt vmcnt the compiler puts in front of a bias-tile use then also waits for those
//     ~90 KB of requests, which are younger but counted by the same in-order counter: the attention phase grew by 2.4k clocks, L2 97 -> 107 us.
//     It needs the bias loads in assembly with hand-counted waits as well.)
//   * the attention is wave-private: k and v^T go through the wave's own LDS region (no block barrier), their fragments are read once
//     for the unit's four query tiles.
// -------------------------------------------------------------------------------------------------
#ifndef DSG_WX_EXP
#define DSG_WX_EXP 0
#endif
// The QKV weight once more in the order qkv_attn_wx_kernel streams and reads it: per (head, 32-deep K chunk) the 96 rows' 6 KB as six
// fragment blocks (q | k | v) x (k-step 0 | 1), a block = 64 lanes x 16 bytes exactly as an MFMA operand wants them -- a DMA request is one
// contiguous KiB (whole cache lines: a row-gathered request of 16 x 64-byte row segments runs at half the rate) and a fragment read is
// lane-linear (no bank conflicts by construction).
__global__ __launch_bounds__(256) void qkv_img_kernel(const unsigned short *__restrict__ W, unsigned short *__restrict__ img, int C, int heads) {
    const int nk = C / 32;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;           // one 16-byte piece
    if (idx >= (size_t)heads * nk * 6 * 64) return;
    const int lane = idx & 63, blkf = (idx >> 6) % 6, kc = (idx / (6 * 64)) % nk, head = idx / ((size_t)6 * 64 * nk);
    const int nt = blkf >> 1, s2 = blkf & 1, lrow = lane & 31, lhalf = lane >> 5;
    const unsigned short *src = W + (size_t)(nt * C + head * 32 + lrow) * C + kc * 32 + 16 * s2 + 8 * lhalf;
#pragma unroll
    for (int j = 0; j < 8; j++) img[idx * 8 + j] = src[j];
}
void launch_qkv_image(const void *Wb, void *img, int C, int heads, hipStream_t s) {
    const size_t n = (size_t)heads * (C / 32) * 6 * 64;
    DSG_LAUNCH(qkv_img_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned short *)Wb, (unsigned short *)img, C, heads);
}
// TWO: a block's four units may lie in two windows (heads not a multiple of 4): two xn slabs per ring slot, three slots; otherwise one slab
// and FOUR slots -- three chunks (2300 matrix clocks) in flight, which is what the ~2000-clock request latency under load asks for.
template <bool TWO>
__global__ __launch_bounds__(256, 1) void qkv_attn_wx_kernel(BxQkvAttn a, int nblk, int U, int G) {
    constexpr int KT = 4, WS = 10, Wp = 128, Wt = 100, OLD = 40;
    constexpr int ASZ = 128 * 64, WSZ = 96 * 64, NA = TWO ? 2 : 1, SLOT = NA * ASZ + 4 * WSZ;   // bytes: the windows' rows, four units' weight rows
    constexpr int NS = TWO ? 3 : 4;                                                           // ring slots
    __shared__ __attribute__((aligned(16))) char lds[NS * SLOT + 4 * 32 * OLD * 2];           // the ring | a 32 x 32 output tile per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int heads = a.g.heads, C = a.g.C, res = a.g.res, shift = a.g.shift;
    const int nwr = res / WS, nW = nwr * nwr, T = res * res;
    // blocks of one group of G (= the blocks that share a window) run back to back on one XCD
    const int t = blockIdx.x, xcd = t & 7, seq = t >> 3;
    const int blk = ((seq / G) * 8 + xcd) * G + seq % G;
    if (blk >= nblk) return;
    const int n_units = U * heads;
    unsigned long long *dbg = a.dbg ? a.dbg + ((size_t)blk * 4 + wave) * 8 : nullptr;   // measurement launches of the debug entry only
#define WX_STAMP(i) do { if (dbg && lane == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    WX_STAMP(0);
    const int uid = min(blk * 4 + wave, n_units - 1);                     // (a wave beyond the last unit repeats it and stores nothing)
    const bool live = blk * 4 + wave < n_units;
    const int u = uid / heads, head = uid - u * heads;
    const int win0 = (blk * 4) / heads, win1 = min(blk * 4 + 3, n_units - 1) / heads;
    const int aw = u - win0;                                              // the A slot of this wave's window
    struct Unit { int tb, i0, j0; };
    auto unit_of = [&](int uu) -> Unit {
        const int b = uu / nW, w = uu - b * nW, wi = w / nwr, wj = w - wi * nwr;
        return {b * T, wi * WS + shift, wj * WS + shift};
    };
    auto pos_token = [&](const Unit &un, int pos) -> int {               // pos < Wt
        const int pi = pos / WS, pj = pos - pi * WS;
        int ti = un.i0 + pi, tj = un.j0 + pj;
        if (ti >= res) ti -= res;
        if (tj >= res) tj -= res;
        return un.tb + ti * res + tj;
    };
    const char *xn = static_cast<const char *>(a.xn), *Wq = static_cast<const char *>(a.Wimg) + (size_t)head * (C / 32) * 6144;
    const unsigned lds0 = (unsigned)(size_t)lds;
    auto glds = [&](const char *sbase, unsigned voff, unsigned ldsaddr) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsaddr), "v"(voff), "s"(sbase) : "memory");
    };
    // ---- this wave's DMA requests per chunk: six for its unit's weight rows, two per staged window for the xn rows (wave w: rows 32 w .. + 31) ----
    unsigned voffA[2][2];
    const unsigned voffW = (unsigned)lane * 16u;       // the weight image: chunk kc of this head is one contiguous 6 KB, request i its i-th KiB
#pragma unroll
    for (int w2 = 0; w2 < 2; w2++) {
        const Unit un = unit_of(w2 ? win1 : win0);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int P = (wave * 2 + i) * 64 + lane, row = P >> 2, cp = (P & 3) ^ ((row >> 2) & 3);
            const int tk = pos_token(un, row < Wt ? row : 0);            // padding positions repeat position 0: finite values, never used
            voffA[w2][i] = ((unsigned)tk * (unsigned)C) * 2u + (unsigned)cp * 16u;
        }
    }
    const int nk = C / 32;
    const bool two = TWO && win1 != win0;
    const int npc = two ? 10 : 8;              // requests per wave and chunk
    auto wait_vm = [&](int n) {                // s_waitcnt vmcnt(n) for the counts that occur (the immediate is an instruction field)
        switch (n) { case 0: M384_WAIT_VM(0); break; case 8: M384_WAIT_VM(8); break; case 10: M384_WAIT_VM(10); break;
                     case 16: M384_WAIT_VM(16); break; default: M384_WAIT_VM(20); break; }
    };
    // request i (0 .. 9) of chunk kc: 0-1 first window's rows, 2-3 second window's (when the block has one), 4-9 weight rows
    auto dma_req = [&](int kc, int i) {
        const unsigned base = lds0 + (kc % NS) * SLOT;
        if (i < 2) glds(xn + kc * 64, voffA[0][i], base + (wave * 2 + i) * 1024);
        else if (i < 4) { if (two) glds(xn + kc * 64, voffA[1][i - 2], base + ASZ + (wave * 2 + i - 2) * 1024); }   // (uniform: the block's second window, if it has one)
        else glds(Wq + kc * 6144 + (i - 4) * 1024, voffW, base + NA * ASZ + wave * WSZ + (i - 4) * 1024);
    };
    // accumulators: q and k as D[feature][position] (lane = position: the B operand of S^T = K Q^T and, for the keys, its A operand --
    // every key tile of the window belongs to this wave, so k never leaves the registers); v with the operands SWAPPED, D[position][feature]
    // (lane = feature, registers = positions in exactly the order the probabilities' B operand has its keys): the A operand of O^T = V^T P^T.
    f32x16 aq[4], ak[4], av[4];                // (written by the first chunk's MFMAs, whose C operand is the constant 0: no initialisation pass)
#pragma unroll
    for (int c = 0; c < NS - 1; c++)           // (C >= 96: at least three chunks)
#pragma unroll
        for (int i = 0; i < 10; i++) dma_req(c, i);
    WX_STAMP(1);
    // fragment addresses: row (32 tile + lrow), k-step s of the chunk -> slot ((2 s + lhalf) ^ ((lrow >> 2) & 3))
    const int sw = (lrow >> 2) & 3;
    const int fo0 = lrow * 64 + ((lhalf ^ sw) << 4), fo1 = lrow * 64 + (((2 + lhalf) ^ sw) << 4);
    struct Frag { bf16x8 w[3], x[4]; };
    auto fread = [&](Frag &f, int kc, int fo, int s2) {
        const char *ab = lds + (kc % NS) * SLOT + aw * ASZ + fo, *wb = lds + (kc % NS) * SLOT + NA * ASZ + wave * WSZ + s2 * 1024 + lane * 16;
#pragma unroll
        for (int nt = 0; nt < 3; nt++) f.w[nt] = *reinterpret_cast<const bf16x8 *>(wb + nt * 2048);
#pragma unroll
        for (int rt = 0; rt < 4; rt++) f.x[rt] = *reinterpret_cast<const bf16x8 *>(ab + rt * 2048);
    };
    wait_vm((NS - 2) * npc);                   // chunk 0 has landed (the later chunks' requests may be in flight)
    M384_BARRIER();
    Frag f0, f1;
    fread(f0, 0, fo0, 0);
    // One chunk: k-step 0 from f0 (read behind the previous barrier) while f1 is being read, then k-step 1 from f1 while the next chunk's f0
    // is.  FIRST: the accumulators start here (C = 0).  MORE: chunk kc + NS - 1's requests go out between the MFMAs (its slot was chunk
    // kc - 1's: every wave passed the last barrier behind its reads of it).  Separate instantiations, so that each is one basic block per
    // half -- a run-time `if` around every request cut the stream into ten pieces and cost a full lgkmcnt(0) at the top.
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](int kc, auto first_c, auto more_c) {
        constexpr bool FIRST = decltype(first_c)::value, MORE = decltype(more_c)::value && DSG_WX_EXP != 1;
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
            if (DSG_WX_EXP != 2) aq[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.w[0], f0.x[rt], FIRST ? zero16 : aq[rt], 0, 0, 0);
            if (rt == 0) {                     // f1's reads are issued BEHIND the first MFMA: the wait in front of it covers f0 only
                __builtin_amdgcn_sched_barrier(0);
                if (DSG_WX_EXP != 3 || kc == 0) fread(f1, kc, fo1, 1);
            }
            if (MORE) dma_req(kc + NS - 1, 3 * rt);
            if (DSG_WX_EXP != 2) ak[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.w[1], f0.x[rt], FIRST ? zero16 : ak[rt], 0, 0, 0);
            if (MORE && rt < 3) dma_req(kc + NS - 1, 3 * rt + 1);
            if (DSG_WX_EXP != 2) av[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0.x[rt], f0.w[2], FIRST ? zero16 : av[rt], 0, 0, 0);
            if (MORE && rt < 3) dma_req(kc + NS - 1, 3 * rt + 2);
        }
        if (kc + 1 < nk) {
            // this wave's share of chunk kc + 1 has landed: at most the requests of the NS - 2 chunks behind it are in flight
            const int later = min(nk - 1, kc + NS - 1) - (kc + 1);
            wait_vm(later * npc);
            M384_WAIT_LGKM0();                 // ... and its reads of chunk kc are complete
            if (DSG_WX_EXP != 4) M384_BARRIER();
            if (DSG_WX_EXP != 3) fread(f0, kc + 1, fo0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
            if (DSG_WX_EXP != 2) aq[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.w[0], f1.x[rt], aq[rt], 0, 0, 0);
            if (DSG_WX_EXP != 2) ak[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.w[1], f1.x[rt], ak[rt], 0, 0, 0);
            if (DSG_WX_EXP != 2) av[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.x[rt], f1.w[2], av[rt], 0, 0, 0);
        }
    };
    chunk(0, std::true_type{}, std::true_type{});                              // (nk >= NS: chunk 0 always has a request to send)
    for (int kc = 1; kc + NS - 1 < nk; kc++) chunk(kc, std::false_type{}, std::true_type{});
    for (int kc = max(1, nk - NS + 1); kc < nk; kc++) chunk(kc, std::false_type{}, std::false_type{});
    WX_STAMP(2);
    // ---- bias, then everything the attention needs as bf16 operand fragments, in registers ----
    bf16x8 qf[4][2], kf[KT][2], vf[KT][2];
    {   // q | k bias per feature = register pair (packed adds); v's bias per feature = lane
        f32x4 bq[4], bk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bq[q] = *reinterpret_cast<const f32x4 *>(a.bias + head * 32 + 8 * q + 4 * lhalf);
            bk[q] = *reinterpret_cast<const f32x4 *>(a.bias + C + head * 32 + 8 * q + 4 * lhalf);
        }
        const float bv = a.bias[2 * C + head * 32 + lrow];
        const f32x2_t bv2 = {bv, bv};
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                u32x4 pq, pk, pv;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int r = 8 * s + 2 * jj;
                    const f32x2_t q2 = (f32x2_t){aq[rt][r], aq[rt][r + 1]} + (f32x2_t){bq[r >> 2][r & 3], bq[r >> 2][(r & 3) + 1]};
                    const f32x2_t k2 = (f32x2_t){ak[rt][r], ak[rt][r + 1]} + (f32x2_t){bk[r >> 2][r & 3], bk[r >> 2][(r & 3) + 1]};
                    const f32x2_t v2 = (f32x2_t){av[rt][r], av[rt][r + 1]} + bv2;
                    pq[jj] = pack_bf16(q2[0], q2[1]);
                    pk[jj] = pack_bf16(k2[0], k2[1]);
                    pv[jj] = pack_bf16(v2[0], v2[1]);
                }
                qf[rt][s] = __builtin_bit_cast(bf16x8, pq);
                kf[rt][s] = __builtin_bit_cast(bf16x8, pk);
                vf[rt][s] = __builtin_bit_cast(bf16x8, pv);
            }
    }
    WX_STAMP(3);
    const int w_of_u = u % nW;
    const Unit uo = unit_of(u);
    const rsrc_t rsO = make_rsrc(a.out, (unsigned)((size_t)a.B * T * C * 2u));
    const unsigned OOB = 0x7fffffffu;
    __bf16 *Tq = reinterpret_cast<__bf16 *>(lds + NS * SLOT) + wave * 32 * OLD;
    // the bias tile (fp32, accumulator order, lane-interleaved: bias_permute_f32_kernel) is loaded straight into the score accumulators, one query tile ahead
    const f32x4 *bp0 = reinterpret_cast<const f32x4 *>(a.biasF) + ((size_t)(shift > 0 ? w_of_u : 0) * heads + head) * (Wp * Wp / 4) + lane;
    f32x16 sbuf[2][KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v4 = 32 * kt + 8 * q < Wt ? bp0[(4 * kt + q) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};   // (quads of padding keys: never looked at)
            sbuf[0][kt][4 * q] = v4[0]; sbuf[0][kt][4 * q + 1] = v4[1]; sbuf[0][kt][4 * q + 2] = v4[2]; sbuf[0][kt][4 * q + 3] = v4[3];
        }
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
        if (qt + 1 < 4) {
            const f32x4 *bp = bp0 + (size_t)(qt + 1) * (KT * 4 * 64);
#pragma unroll
            for (int kt = 0; kt < KT; kt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 v4 = 32 * kt + 8 * q < Wt ? bp[(4 * kt + q) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    sbuf[(qt + 1) & 1][kt][4 * q] = v4[0]; sbuf[(qt + 1) & 1][kt][4 * q + 1] = v4[1]; sbuf[(qt + 1) & 1][kt][4 * q + 2] = v4[2]; sbuf[(qt + 1) & 1][kt][4 * q + 3] = v4[3];
                }
        }
        f32x16 sacc[KT];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            sacc[kt] = sbuf[qt & 1][kt];
#pragma unroll
            for (int s = 0; s < 2; s++) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s], qf[qt][s], sacc[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (32 * kt + 8 * (r >> 2) < Wt) mx = fmaxf(mx, sacc[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        f32x2_t sum2 = {0.f, 0.f};
        const f32x2_t mx2 = {mx, mx};
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
            u32x4 pf[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (32 * kt + 8 * (r >> 2) < Wt) {
                    const f32x2_t d = (f32x2_t){sacc[kt][r], sacc[kt][r + 1]} - mx2;          // v_pk_add_f32
                    const f32x2_t e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                    sum2 += e;
                    pf[r >> 3][(r & 7) >> 1] = pack_bf16(e[0], e[1]);
                } else {
                    pf[r >> 3][(r & 7) >> 1] = 0u;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (32 * kt + 16 * s < Wt) oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][s], __builtin_bit_cast(bf16x8, pf[s]), oacc, 0, 0, 0);
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 32, 64);
        const float inv = fast_rcp(sum);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = oacc[4 * q + e] * inv;
            *reinterpret_cast<u32x2 *>(Tq + lrow * OLD + 8 * q + 4 * lhalf) = pack_bf16x4(o);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int r = (lane >> 2) + 16 * k, pc = lane & 3, pos = 32 * qt + r;
            const int tq = (live && pos < Wt) ? pos_token(uo, pos) : -1;
            const u32x4 d = *reinterpret_cast<const u32x4 *>(Tq + r * OLD + 8 * pc);
            buf_store_u4(d, rsO, tq < 0 ? OOB : ((unsigned)tq * (unsigned)C + (unsigned)(head * 32 + 8 * pc)) * 2u, 0u);
        }
    }
    WX_STAMP(4);
#undef WX_STAMP
}

bool launch_qkv_attn_bx(const BxQkvAttn &a, hipStream_t s) {
    const WinGeom &g = a.g;
    if (!a.xn || !a.W || !a.bias || !a.biasP || !a.out || a.B < 1) return false;
    if (g.C != 32 * g.heads || g.C % 8 != 0 || g.C < 96 || g.res % g.ws != 0) return false;
    if ((size_t)a.B * g.res * g.res * g.C * 2u >= 0x7fffffffull) return false;   // 32-bit buffer offsets
    const int nW = (g.res / g.ws) * (g.res / g.ws), U = a.B * nW;
    if (g.ws == 10 && a.variant == 0 && a.Wimg && a.biasF) {   // one wave per (window, head): qkv_attn_wx_kernel
        if ((size_t)3 * g.C * g.C * 2u >= 0x7fffffffull) return false;
        const int n_units = U * g.heads, nblk = (n_units + 3) / 4, G = (g.heads + 3) / 4;
        const int groups = (nblk + G - 1) / G;
        const dim3 grid((unsigned)(((groups + 7) / 8) * 8 * G)), block(256);
        if (g.heads % 4 == 0) DSG_LAUNCH(qkv_attn_wx_kernel<false>, grid, block, 0, s, a, nblk, U, G);
        else DSG_LAUNCH(qkv_attn_wx_kernel<true>, grid, block, 0, s, a, nblk, U, G);
        return true;
    }
    int kt;
    switch (g.ws) { case 4: case 5: kt = 1; break; case 8: kt = 2; break; case 10: kt = 4; break; default: return false; }
    const int upb = 4 / kt, nblk = (U + upb - 1) / upb;
    const dim3 grid((unsigned)(((nblk + 7) / 8) * 8 * g.heads)), block(256);
#define QA(KT_, WS_) DSG_LAUNCH((qkv_attn_bx_kernel<KT_, WS_>), grid, block, 0, s, a, nblk, U)
    switch (g.ws) {
        case 4: QA(1, 4); break;
        case 5: QA(1, 5); break;
        case 8: QA(2, 8); break;
        default: QA(4, 10); break;
    }
#undef QA
    return true;
}

bool launch_attn_bx(const void *qkv, const float *biasT, void *out, int B, const WinGeom &g, hipStream_t s) {
    const int nW = (g.res / g.ws) * (g.res / g.ws);
    if (g.C != 32 * g.heads || g.C % 8 != 0 || g.res % g.ws != 0) return false;
    // one block per (window position, head, share of the batch): enough shares to fill the chip twice over, at least 4 samples each
    const int pairs = nW * g.heads;
    int split = std::max(1, std::min((B + 3) / 4, (2 * bx_cu_count() + pairs - 1) / pairs));
    const dim3 grid(pairs * split), block(256);
#define AX(KT_, WS_) DSG_LAUNCH((attn_bx_kernel<KT_, WS_>), grid, block, 0, s, (const __bf16 *)qkv, biasT, (__bf16 *)out, B, g, split)
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
