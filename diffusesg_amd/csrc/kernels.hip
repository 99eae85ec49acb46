// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DiffuseSG sampling hot path.
//
// All arithmetic is fp32.  The dense contractions run on the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157 TFLOP/s chip peak), everything else is HBM/LDS bound
// elementwise work fused around them.  Reference semantics are cited per kernel
// (R/ = DiffuseSG/ of the reference tree).
#include "kernels_common.hip.h"

namespace dsg {

// =================================================================================================
// GEMM: C[M,N] = act(pro(A)[M,K] . W[N,K]^T + bias) (+res)
//
// Block = 256 threads = 4 waves, block tile 128 (M) x 96 (N), K step 32.  Every channel count of
// the network is a multiple of 96 (= 3 MFMA tiles of 32), so a 96-wide N tile wastes nothing where
// a power-of-two tile would.  Wave w owns rows [32w, 32w+32) x all 96 columns = 3 accumulators of
// v_mfma_f32_32x32x2_f32.  The MFMA takes ONE f32 per lane for A and for B (lane l: row/col l&31,
// k = l>>5); the k index inside a step is a free permutation as long as A and B agree, so each lane
// fetches 4 consecutive k with one ds_read_b128 (k = 8s + 4*(l>>5) + t) and feeds element t to the
// t-th MFMA: 4 LDS reads feed 12 MFMAs (768 matrix-pipe cycles) -- LDS is never the limiter.
// A tiles are register-staged (global -> VGPR -> LDS) so that LayerNorm ((a-mean)*rstd*g+b) and the
// two-source concat are applied on the way in; the next tile's global loads are issued before the
// MFMAs of the current one.  LDS rows are padded to 36 floats: conflict-free for ds_read_b128.
// =================================================================================================

// Scheduling: a wave issues in order -- once it has issued an MFMA it sits on the next one for 64 cycles, so every
// non-MFMA instruction clumped between two MFMA blocks is exposed time (a first version that staged a tile, then ran
// 48 MFMAs, then wrote LDS kept the matrix pipe only 68 % busy).  Here every k-chunk is one basic block in which the
// fragment reads, the LDS writes of the NEXT chunk (from registers loaded one chunk earlier) and the global loads of
// the chunk after that are spread across the 48 MFMAs (sched_group_barrier pins the interleave).  Two register staging
// sets => the loop is unrolled by two.  Two blocks per CU (two waves per SIMD) cover each other's barrier and epilogue.
// -------------------------------------------------------------------------------------------------
#ifdef DSG_CLOCK_DIAG
__device__ unsigned long long *g_diag_buf = nullptr;
#endif
// -------------------------------------------------------------------------------------------------
// VALU budget.  The K loop carries (almost) no VALU
// instructions.  Measured on gfx950: v_mfma_f32_32x32x2_f32 and ordinary VALU instructions of the same
// SIMD do NOT overlap (the f32 MFMA runs at the vector rate; two dependent VALU ops per MFMA cost +10 cycles per MFMA
// with one wave per SIMD, tools/mfma_rate.cpp) -- so every address add, select or LayerNorm multiply in the loop is
// paid in matrix throughput.  Therefore:
//   * operands are fetched with buffer loads: the per-thread byte offset is loop-invariant, the k offset is a scalar
//     (SALU) -- no per-load address arithmetic; rows beyond M / N read as zero through the descriptor's range check,
//     so there is no masking code either;
//   * LayerNorm in the A path is one FMA per element (x*rstd - mean*rstd); gamma is folded into the weight columns and
//     beta into the bias when the weights are packed (dsg_finalize_weights);
//   * the epilogue uses buffer stores / loads with scalar row offsets.
// -------------------------------------------------------------------------------------------------
// EPI (epilogue extension, see GemmArgs): 0 none; 1 row-statistics partials; 2 batch-uniform modulate+SiLU + partials;
// 3 per-sample modulate+SiLU + partials; 4 fused window attention: the tile is (two 64-token windows | one 100-token window
// padded to 128 rows) x (q|k|v of one head), tile rows are gathered through the window partition / cyclic shift
// (diffusesg.py:28-57, :246-256), q, k, v go to LDS instead of HBM and softmax(q k^T + bias) v runs from there (same operand
// scheme as window_attn_kernel below; padded key slots carry -1e30 in the bias table, padded rows are never stored).
// AMODE 1: PatchMerging gather in the A path (GemmArgs::a4_res).
// R rows' partial (sum, sumsq) pairs [nparts][2] each, added in tile order.  All R rows' loads are issued before the first
// result is used and the switch on nparts sits outside the row loop: a row-by-row version cost the prologue one memory
// latency per row (branches between the rows pin an s_waitcnt vmcnt(0) behind each row's loads).
template <int R>
__device__ __forceinline__ void row_partials_n(const float *const (&pp)[R], int nparts, float (&sm)[R], float (&sq)[R]) {
    if (nparts == 1) {
        float2 a[R];
#pragma unroll
        for (int i = 0; i < R; i++) a[i] = *reinterpret_cast<const float2 *>(pp[i]);
#pragma unroll
        for (int i = 0; i < R; i++) { sm[i] = a[i].x; sq[i] = a[i].y; }
    } else if (nparts == 2) {
        f32x4 a[R];
#pragma unroll
        for (int i = 0; i < R; i++) a[i] = *reinterpret_cast<const f32x4 *>(pp[i]);
#pragma unroll
        for (int i = 0; i < R; i++) { sm[i] = a[i][0] + a[i][2]; sq[i] = a[i][1] + a[i][3]; }
    } else if (nparts == 4) {
        f32x4 a[R], b[R];
#pragma unroll
        for (int i = 0; i < R; i++) { a[i] = *reinterpret_cast<const f32x4 *>(pp[i]); b[i] = *reinterpret_cast<const f32x4 *>(pp[i] + 4); }
#pragma unroll
        for (int i = 0; i < R; i++) {
            sm[i] = (a[i][0] + a[i][2]) + (b[i][0] + b[i][2]);
            sq[i] = (a[i][1] + a[i][3]) + (b[i][1] + b[i][3]);
        }
    } else if (nparts == 8) {
        f32x4 a[R], b[R], c[R], d[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            a[i] = *reinterpret_cast<const f32x4 *>(pp[i]); b[i] = *reinterpret_cast<const f32x4 *>(pp[i] + 4);
            c[i] = *reinterpret_cast<const f32x4 *>(pp[i] + 8); d[i] = *reinterpret_cast<const f32x4 *>(pp[i] + 12);
        }
#pragma unroll
        for (int i = 0; i < R; i++) {
            sm[i] = ((a[i][0] + a[i][2]) + (b[i][0] + b[i][2])) + ((c[i][0] + c[i][2]) + (d[i][0] + d[i][2]));
            sq[i] = ((a[i][1] + a[i][3]) + (b[i][1] + b[i][3])) + ((c[i][1] + c[i][3]) + (d[i][1] + d[i][3]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < R; i++) {
            sm[i] = 0.f; sq[i] = 0.f;
            for (int t = 0; t < nparts; t++) { sm[i] += pp[i][2 * t]; sq[i] += pp[i][2 * t + 1]; }
        }
    }
}

template <bool LN, int ACT, bool RES, int EPI, int WS = 8, int AMODE = 0>   // WS: window side of the fused attention (EPI == 4): 8 or 10
__global__ __launch_bounds__(256, 2) void gemm4_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (GBM + GBN) * GLD];
    constexpr int BUF = (GBM + GBN) * GLD;

    int bid = blockIdx.x;
    // AMODE 2 (training, weight gradients): `batch` independent products of the same shape in one launch -- the K slices of a
    // split-K product -- operands / outputs batch_stride{A,W,C} floats apart; everything else as AMODE 0
    const float *gA = g.A, *gW = g.W;
    float *gC = g.C;
    if (AMODE == 2) {
        const int per = ((tiles_m + 7) / 8) * 8 * tiles_n, z = bid / per;
        bid -= z * per;
        gA += (size_t)z * g.batch_strideA; gW += (size_t)z * g.batch_strideW; gC += (size_t)z * g.batch_strideC;
    }
    const int xcd = bid & 7, seq = bid >> 3;
    const int tm = (seq / tiles_n) * 8 + xcd, tn = seq % tiles_n;
    if (tm >= tiles_m) return;
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int rows_m = min(GBM, g.M - m0), rows_n = min(GBN, g.N - n0);
    // measurement mode (dsg_profile_forward): first-block-start / last-block-end on the 100 MHz constant clock
    unsigned long long prof_t0 = 0, prof_c0 = 0;   // 100 MHz constant clock / shader clock (block 0 only: the held clock)
    if (g.prof && tid == 0) { prof_t0 = __builtin_amdgcn_s_memrealtime(); if (bid == 0) prof_c0 = __builtin_amdgcn_s_memtime(); }

    // EPI == 4: tile row r -> token row of the activation: window WPT*tm + r/Wp, position r%Wp inside the (shifted) window
    constexpr int A_WT = WS * WS, A_WP = (A_WT + 31) / 32 * 32, A_KT = A_WP / 32, A_WPT = GBM / A_WP;   // 64/64/2/2 or 100/128/4/1
    const int a_res = g.wg.res, a_nwr = (EPI == 4) ? a_res / WS : 1, a_nW = a_nwr * a_nwr, a_T = a_res * a_res;
    const int a_nwin = g.attn_batch * a_nW;
    auto win_row = [&](int r) -> int {   // global row of tile row r, or -1 if the window / the position does not exist
        const int gw = A_WPT * tm + r / A_WP, pos = r % A_WP;
        if (gw >= a_nwin || pos >= A_WT) return -1;
        const int b = gw / a_nW, w = gw - b * a_nW, wi = w / a_nwr, wj = w - wi * a_nwr;
        int ti = wi * WS + pos / WS + g.wg.shift, tj = wj * WS + pos % WS + g.wg.shift;
        if (ti >= a_res) ti -= a_res;
        if (tj >= a_res) tj -= a_res;
        return b * a_T + ti * a_res + tj;
    };
    // block-uniform descriptors: base = first row of the tile, range = the valid rows (out-of-range reads give 0)
    const int a4_C = g.K >> 2;   // AMODE 1: channels of one source row
    const rsrc_t rsA1 = (EPI == 4) ? make_rsrc(g.A, (unsigned)g.M * g.lda * 4u)
                        : (AMODE == 1) ? make_rsrc(g.A, (unsigned)g.M * (unsigned)g.K * 4u)   // 4*M fine rows of K/4 floats
                                       : make_rsrc(gA + (size_t)m0 * g.lda, (unsigned)rows_m * g.lda * 4u);
    const rsrc_t rsA2 = make_rsrc(g.A2 ? g.A2 + (size_t)m0 * g.lda2 : g.A, g.A2 ? (unsigned)rows_m * g.lda2 * 4u : 0u);
    const rsrc_t rsW = (EPI == 4) ? make_rsrc(g.W, (unsigned)(3 * g.wg.C) * g.K * 4u) : make_rsrc(gW + (size_t)n0 * g.K, (unsigned)rows_n * g.K * 4u);
    unsigned voffA1[4], voffA2[4], voffW[3];
    unsigned voffA4[4][4];   // AMODE 1: [part][staging row]
    float a_rstd[4], a_nmr[4];
    int my_tok = -1;   // EPI == 4: token row of tile row `tid` (threads 0..127), for the output scatter
    if (EPI == 4 && tid < GBM) my_tok = win_row(tid);
    int a_grow[4];   // EPI == 4: token row of staging row p; AMODE 1: its first fine row
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int r = r0 + 32 * p;
        voffA1[p] = ((unsigned)r * g.lda + 4u * c4) * 4u;
        voffA2[p] = ((unsigned)r * g.lda2 + 4u * c4) * 4u;
        a_grow[p] = 0;
        if (EPI == 4) {
            a_grow[p] = win_row(r);
            voffA1[p] = a_grow[p] >= 0 ? ((unsigned)a_grow[p] * g.lda + 4u * c4) * 4u : 0x7fffffffu;
        }
        if (AMODE == 1) {
            // coarse row -> its four fine rows (order x00, x10, x01, x11: part q has di = q&1, dj = q>>1)
            const int m = min(m0 + r, g.M - 1), r2 = g.a4_res >> 1, T2 = r2 * r2;
            const int b = m / T2, t = m - b * T2, i = t / r2, j = t - i * r2;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int srow = b * g.a4_res * g.a4_res + (2 * i + (q & 1)) * g.a4_res + 2 * j + (q >> 1);
                voffA4[q][p] = ((unsigned)srow * (unsigned)a4_C + 4u * c4) * 4u;
            }
            a_grow[p] = b * g.a4_res * g.a4_res + 2 * i * g.a4_res + 2 * j;   // fine row of part 0
        }
    }
    // LayerNorm statistics of the staging rows.  (Issuing the first two chunks' tile loads before these loads -- one memory
    // latency in the prologue instead of two -- measured 0.15 % slower in the in-box A/B, tools/ab_libs.sh.)
    auto load_row_stats = [&]() {
        const float invk = 1.0f / (float)g.K;
        auto finish = [&](int p, float sm, float sq) {
            const float mean = sm * invk, rstd = fast_rsqrt(fmaxf(fmaf(-mean, mean, sq * invk), 0.f) + LN_EPS);
            a_rstd[p] = rstd;
            a_nmr[p] = -mean * rstd;
        };
        if (AMODE == 1) {
            // LayerNorm(4C) statistics = the four fine rows' partial (sum, sumsq) pairs added in a fixed order
            const float *pp[16];
            float psm[16], psq[16];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    pp[4 * p + q] = g.ln_part + (size_t)(a_grow[p] + (q & 1) * g.a4_res + (q >> 1)) * g.ln_nparts * 2;
            row_partials_n<16>(pp, g.ln_nparts, psm, psq);
#pragma unroll
            for (int p = 0; p < 4; p++)
                finish(p, (psm[4 * p] + psm[4 * p + 1]) + (psm[4 * p + 2] + psm[4 * p + 3]),
                       (psq[4 * p] + psq[4 * p + 1]) + (psq[4 * p + 2] + psq[4 * p + 3]));
        } else if (LN) {
            int m[4];
#pragma unroll
            for (int p = 0; p < 4; p++) m[p] = (EPI == 4) ? max(a_grow[p], 0) : min(m0 + r0 + 32 * p, g.M - 1);
            if (g.ln_part) {   // partial (sum, sumsq) per 96-column tile of the producer: [M][nparts][2], added in tile order
                const float *pp[4];
                float sm[4], sq[4];
#pragma unroll
                for (int p = 0; p < 4; p++) pp[p] = g.ln_part + (size_t)m[p] * g.ln_nparts * 2;
                row_partials_n<4>(pp, g.ln_nparts, sm, sq);
#pragma unroll
                for (int p = 0; p < 4; p++) finish(p, sm[p], sq[p]);
            } else {
                float2 st[4];
#pragma unroll
                for (int p = 0; p < 4; p++) st[p] = *reinterpret_cast<const float2 *>(g.ln_stats + 2 * m[p]);
#pragma unroll
                for (int p = 0; p < 4; p++) { a_rstd[p] = st[p].y; a_nmr[p] = -st[p].x * st[p].y; }
            }
        }
    };
#pragma unroll
    for (int p = 0; p < 3; p++) {
        voffW[p] = ((unsigned)(r0 + 32 * p) * g.K + 4u * c4) * 4u;
        if (EPI == 4) voffW[p] = ((unsigned)(p * g.wg.C + tn * 32 + r0) * g.K + 4u * c4) * 4u;   // rows [q_h | k_h | v_h] of head tn
    }
    const int nk = g.K / GBK;
    const int nk1 = g.A2 ? g.K1 / GBK : nk;  // chunks served by the first source

    struct Stage { f32x4 a[4], w[3]; };
    auto issue = [&](Stage &st, int kc) {
        kc = kc < nk ? kc : nk - 1;  // the tail iterations re-load a valid chunk they never use
#if defined(DSG_EXP) && DSG_EXP >= 2
        if (kc > 1) return;          // power experiment (tools/gemm_bench, wrong results): no global loads in the K loop
#elif defined(DSG_EXP) && DSG_EXP == 1
        kc = kc & 1;                 // power experiment: every chunk re-reads chunk 0/1 (cache hits, nothing streams)
#endif
        const bool second = kc >= nk1;
        const unsigned soffA = (unsigned)(second ? kc - nk1 : kc) * (GBK * 4u), soffW = (unsigned)kc * (GBK * 4u);
        if (AMODE == 1) {
            const int cpp = a4_C / GBK, q = kc / cpp;   // chunks per part; part of this chunk (block-uniform)
            const unsigned soff4 = (unsigned)(kc - q * cpp) * (GBK * 4u);
            if (q == 0) {
#pragma unroll
                for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA1, voffA4[0][p], soff4);
            } else if (q == 1) {
#pragma unroll
                for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA1, voffA4[1][p], soff4);
            } else if (q == 2) {
#pragma unroll
                for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA1, voffA4[2][p], soff4);
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA1, voffA4[3][p], soff4);
            }
        } else if (second) {
#pragma unroll
            for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA2, voffA2[p], soffA);
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) st.a[p] = buf_load4(rsA1, voffA1[p], soffA);
        }
#pragma unroll
        for (int p = 0; p < 3; p++) st.w[p] = buf_load4(rsW, voffW[p], soffW);
    };
    bool exp_skip_write = false;
    (void)exp_skip_write;
    auto write = [&](const Stage &st, int buf) {
        float *As = lds + buf * BUF, *Ws = As + GBM * GLD;
#if defined(DSG_EXP) && DSG_EXP >= 3
        if (exp_skip_write) return;  // power experiment: no tile staging stores in the K loop
#endif
#pragma unroll
        for (int p = 0; p < 4; p++) {
            f32x4 v = st.a[p];
            if (LN) {
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = fmaf(v[t], a_rstd[p], a_nmr[p]);
            }
            *reinterpret_cast<f32x4 *>(As + (r0 + 32 * p) * GLD + 4 * c4) = v;
        }
#pragma unroll
        for (int p = 0; p < 3; p++) *reinterpret_cast<f32x4 *>(Ws + (r0 + 32 * p) * GLD + 4 * c4) = st.w[p];
    };
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int a_off = (wave * 32 + lrow) * GLD + 4 * lhalf;
    const int w_off = GBM * GLD + lrow * GLD + 4 * lhalf;
    struct Frag { f32x4 a, b[3]; };
    auto fread = [&](Frag &f, int buf, int s) {
        const float *base = lds + buf * BUF;
        f.a = *reinterpret_cast<const f32x4 *>(base + a_off + 8 * s);
#pragma unroll
        for (int j = 0; j < 3; j++) f.b[j] = *reinterpret_cast<const f32x4 *>(base + w_off + 32 * j * GLD + 8 * s);
    };

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    auto mfma12 = [&](const Frag &f) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t], f.b[j][t], acc[j], 0, 0, 0);
    };

#ifdef DSG_CLOCK_DIAG
    unsigned long long diag_t0 = 0, diag_r0 = 0;
    if (tid == 0) { diag_t0 = __builtin_amdgcn_s_memtime(); diag_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    Stage s0, s1;
    Frag f0, f1;
#ifdef DSG_PHASE_DIAG
    unsigned long long ph1 = 0, ph2 = 0;
#endif
#define GEMM4_CHUNK(CUR, ST_W, ST_L, KC)                                                               \
    do {                                                                                               \
        fread(f1, CUR, 1);                                                                             \
        mfma12(f0);                                                                                    \
        write(ST_W, 1 - (CUR));                                                                        \
        fread(f0, CUR, 2);                                                                             \
        mfma12(f1);                                                                                    \
        issue(ST_L, (KC) + 2);                                                                         \
        fread(f1, CUR, 3);                                                                             \
        mfma12(f0);                                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 36; q_++) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x3f6, 2, 0);                                         \
        }                                                                                              \
        __syncthreads();                                                                               \
        fread(f0, 1 - (CUR), 0);                                                                       \
        mfma12(f1);                                                                                    \
    } while (0)

    load_row_stats();
    issue(s0, 0);
    write(s0, 0);
    issue(s1, 1);
    __syncthreads();
#ifdef DSG_PHASE_DIAG
    if (tid == 0) ph1 = __builtin_amdgcn_s_memrealtime();
#endif
    fread(f0, 0, 0);
    exp_skip_write = true;
    int kc = 0;
    for (; kc + 1 < nk; kc += 2) {
        GEMM4_CHUNK(0, s1, s0, kc);
        GEMM4_CHUNK(1, s0, s1, kc + 1);
    }
    if (kc < nk) GEMM4_CHUNK(0, s1, s0, kc);
#undef GEMM4_CHUNK
#ifdef DSG_PHASE_DIAG
    if (tid == 0) ph2 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef DSG_CLOCK_DIAG
    if (tid == 0 && g_diag_buf) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        g_diag_buf[2 * bid] = t1 - diag_t0;
        g_diag_buf[2 * bid + 1] = r1 - diag_r0;
    }
#endif

    if (EPI == 4) {
        // ---- fused window attention: q, k, v of (two windows, one head) -> LDS -> S^T = K Q^T + bias -> softmax -> P V ----
        constexpr int QKV = GBM * GLD;   // one [128][36] slab each for q, k, v; the token table sits behind them
        float *Qs = lds, *Ks = lds + QKV, *Vs = lds + 2 * QKV;
        int *toks = reinterpret_cast<int *>(lds + 3 * QKV);
        __syncthreads();                 // every wave is done with the K loop's tiles
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float bias = g.bias ? g.bias[j * g.wg.C + tn * 32 + lrow] : 0.f;
            float *dst = lds + j * QKV + (wave * 32 + 4 * lhalf) * GLD + lrow;
#pragma unroll
            for (int r = 0; r < 16; r++) dst[((r & 3) + 8 * (r >> 2)) * GLD] = acc[j][r] + bias;
        }
        if (tid < GBM) toks[tid] = my_tok;
        __syncthreads();
        const int wl = wave / A_KT, qh = wave % A_KT;   // window of the tile / 32-query block inside the window
        const int gw = A_WPT * tm + wl;
        if (gw < a_nwin && 32 * qh < A_WT) {   // wave-uniform: the last tile's second window may not exist
            const int wbase = A_WP * wl;
            f32x4 qf[4];
#pragma unroll
            for (int sx = 0; sx < 4; sx++) qf[sx] = *reinterpret_cast<const f32x4 *>(Qs + (32 * wave + lrow) * GLD + 8 * sx + 4 * lhalf);
            const int wtype = g.wg.shift > 0 ? gw % a_nW : 0;
            const rsrc_t rsB = make_rsrc(g.attn_bias + ((size_t)wtype * g.wg.heads + tn) * (A_WP * A_WP), (unsigned)(A_WP * A_WP) * 4u);
            const unsigned boff = (unsigned)(4 * lhalf * A_WP + 32 * qh + lrow) * 4u;
            f32x16 sacc[A_KT];
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < A_KT; kt++) {
                f32x4 kf[4];
#pragma unroll
                for (int sx = 0; sx < 4; sx++)
                    kf[sx] = *reinterpret_cast<const f32x4 *>(Ks + (wbase + 32 * kt + lrow) * GLD + 8 * sx + 4 * lhalf);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    sacc[kt][r] = buf_load1(rsB, boff, (unsigned)((32 * kt + (r & 3) + 8 * (r >> 2)) * A_WP) * 4u);
#pragma unroll
                for (int sx = 0; sx < 4; sx++)
#pragma unroll
                    for (int t = 0; t < 4; t++) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[sx][t], qf[sx][t], sacc[kt], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[kt][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
            f32x2_t sum2 = {0.f, 0.f};
            const f32x2_t mx2 = {mx, mx};
#pragma unroll
            for (int kt = 0; kt < A_KT; kt++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {   // (subtract and sum on packed fp32; the sum's association changes: pairs of lanes' partials)
                    const f32x2_t d2 = (f32x2_t){sacc[kt][r], sacc[kt][r + 1]} - mx2;
                    const f32x2_t e2 = {__builtin_amdgcn_exp2f(d2[0]), __builtin_amdgcn_exp2f(d2[1])};   // scores carry the log2(e) factor
                    sacc[kt][r] = e2[0]; sacc[kt][r + 1] = e2[1];
                    sum2 += e2;
                }
            sum = sum2[0] + sum2[1];
            sum += __shfl_xor(sum, 32, 64);
            const float inv = fast_rcp(sum);
            f32x16 oacc;
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < A_KT; kt++) {
                float vf[16];
#pragma unroll
                for (int r = 0; r < 16; r++) vf[r] = Vs[(wbase + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * GLD + lrow];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2_t p2 = (f32x2_t){sacc[kt][r], sacc[kt][r + 1]} * (f32x2_t){inv, inv};
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(p2[0], vf[r], oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(p2[1], vf[r + 1], oacc, 0, 0, 0);
                }
            }
            // O tile: column d = lrow of head tn, row = query (r&3)+8(r>>2)+4*half of this wave's 32 tokens
            const rsrc_t rsO = make_rsrc(g.C, (unsigned)g.M * g.ldc * 4u);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = toks[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lhalf];
                buf_store1(oacc[r], rsO, tok >= 0 ? ((unsigned)tok * g.ldc + (unsigned)(tn * 32 + lrow)) * 4u : 0x7fffffffu, 0u);
            }
        }
        if (g.prof && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            atomicMin(g.prof, prof_t0);
            atomicMax(g.prof + 1, t1);
            if (bid == 0) { g.prof[2] = __builtin_amdgcn_s_memtime() - prof_c0; g.prof[3] = t1 - prof_t0; }
        }
        return;
    }
    // epilogue: buffer stores; lane-dependent part of the address in voffset, (register, tile) part in a scalar offset.
    // rows >= M fall outside the descriptor and are dropped; columns >= N get an out-of-range voffset.
    const rsrc_t rsC = make_rsrc(gC + (size_t)m0 * g.ldc, (unsigned)rows_m * g.ldc * 4u);
    const rsrc_t rsC2 = make_rsrc(g.C2 ? g.C2 + (size_t)m0 * g.ldc2 : g.C, g.C2 ? (unsigned)rows_m * g.ldc2 * 4u : 0u);
    const rsrc_t rsR = make_rsrc(RES ? g.res + (size_t)m0 * g.ldres : g.C, RES ? (unsigned)rows_m * g.ldres * 4u : 0u);
    const unsigned rowl = (unsigned)(wave * 32 + 4 * lhalf);
    const unsigned OOB = 0x7fffffffu;
    float st_s[16], st_q[16];   // EPI: this lane's share of (sum, sumsq) of its 16 rows
    int brow[16];               // EPI == 3: sample index of each of the lane's rows
    if (EPI >= 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) { st_s[r] = 0.f; st_q[r] = 0.f; }
    }
    if (EPI == 3) {   // per-sample (scale,shift): row -> sample through a 128-entry LDS table (one division per thread)
        __syncthreads();   // every wave is done with the K loop's tiles
        int *bt = reinterpret_cast<int *>(lds) + 4 * 2304;   // behind the four per-wave reduction slabs used below
        if (tid < GBM) bt[tid] = min(m0 + tid, g.M - 1) / g.mod_T;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) brow[r] = bt[rowl + (r & 3) + 8 * (r >> 2)];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int n = n0 + 32 * j + lrow;
        const bool nok = n < g.N;
        const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
        const unsigned vC = nok ? (rowl * g.ldc + (unsigned)n) * 4u : OOB;
        const unsigned vC2 = nok ? (rowl * g.ldc2 + (unsigned)n) * 4u : OOB;
        const unsigned vR = nok ? (rowl * g.ldres + (unsigned)n) * 4u : OOB;
        float msc = 0.f, msh = 0.f;
        if (EPI == 2 && nok) { msc = g.mod_aff[g.mod_off + n] + 1.0f; msh = g.mod_aff[g.mod_off + g.N + n]; }
        float rres[16];
        if (RES) {
#pragma unroll
            for (int r = 0; r < 16; r++) rres[r] = buf_load1(rsR, vR, (unsigned)((r & 3) + 8 * (r >> 2)) * g.ldres * 4u);
        }
        // phase by phase over the 16 rows of this 32-column slab, not row by row: the block-uniform `if (g.C2)` would otherwise
        // cut the loop into 16 basic blocks, each one serial dependency chain (bias -> activation -> store) that the
        // scheduler cannot interleave with its neighbours
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2_t t2 = (f32x2_t){acc[j][r], acc[j][r + 1]} + (f32x2_t){bias, bias}; v[r] = t2[0]; v[r + 1] = t2[1]; }
        if (ACT == ACT_GELU) {   // pairs on the packed-fp32 instructions (gelu_f2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2_t gg = gelu_f2(v[r], v[r + 1]); v[r] = gg[0]; v[r + 1] = gg[1]; }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (ACT == ACT_SILU) v[r] = silu_exact(v[r]);
        }
        if (ACT != ACT_DGELU && RES) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2_t t2 = (f32x2_t){v[r], v[r + 1]} + (f32x2_t){rres[r], rres[r + 1]}; v[r] = t2[0]; v[r + 1] = t2[1]; }
        }
        if (ACT == ACT_DGELU) {   // (training form) d_pre = d_hid * GELU'(pre), pre read through the residual path (dgelu_f2: pairs, packed fp32)
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2_t dg = dgelu_f2(rres[r], rres[r + 1]); v[r] *= dg[0]; v[r + 1] *= dg[1]; }
        }
        if (g.C2) {
#pragma unroll
            for (int r = 0; r < 16; r++) buf_store1(v[r], rsC2, vC2, (unsigned)((r & 3) + 8 * (r >> 2)) * g.ldc2 * 4u);
        }
        if (ACT == ACT_GELU_KEEP) {   // (training form) C2 just received the pre-activation
#pragma unroll
            for (int r = 0; r < 16; r += 2) { const f32x2_t gg = gelu_f2(v[r], v[r + 1]); v[r] = gg[0]; v[r + 1] = gg[1]; }
        }
        if (EPI >= 2) {   // silu(shift + x (1 + scale)) = a / (1 + exp(-a)): the FMA, the add and the product on packed fp32 (same arithmetic per element)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t sc2 = {msc, msc}, sh2 = {msh, msh};
                if (EPI == 3 && nok) {
                    const float *ar0 = g.mod_aff + (size_t)brow[r] * g.mod_ld + g.mod_off + n, *ar1 = g.mod_aff + (size_t)brow[r + 1] * g.mod_ld + g.mod_off + n;
                    sc2 = (f32x2_t){ar0[0], ar1[0]} + (f32x2_t){1.0f, 1.0f}; sh2 = (f32x2_t){ar0[g.N], ar1[g.N]};
                }
                const f32x2_t a2 = __builtin_elementwise_fma((f32x2_t){v[r], v[r + 1]}, sc2, sh2);
                const f32x2_t d2 = (f32x2_t){1.0f, 1.0f} + (f32x2_t){__expf(-a2[0]), __expf(-a2[1])};
                const f32x2_t o2 = a2 * (f32x2_t){fast_rcp(d2[0]), fast_rcp(d2[1])};
                v[r] = o2[0]; v[r + 1] = o2[1];
            }
        }
        if (EPI >= 1 && nok) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_t v2 = {v[r], v[r + 1]};
                const f32x2_t s2 = (f32x2_t){st_s[r], st_s[r + 1]} + v2, q2 = __builtin_elementwise_fma(v2, v2, (f32x2_t){st_q[r], st_q[r + 1]});
                st_s[r] = s2[0]; st_s[r + 1] = s2[1]; st_q[r] = q2[0]; st_q[r + 1] = q2[1];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) buf_store1(v[r], rsC, vC, (unsigned)((r & 3) + 8 * (r >> 2)) * g.ldc * 4u);
    }
    if (EPI >= 1) {
        // Row statistics of the stored tile.  A row's 96 values sit in the 32 lanes of a half-wave: transpose through a
        // per-wave LDS slab (the K loop's tiles are dead) so that lane (row, q) adds the 32 lane-partials of quantity q
        // of one row in a fixed order, then store the (sum, sumsq) pair of this column tile.
        if (EPI != 3) __syncthreads();
        float *red = lds + wave * 2304;   // [2 quantities][32 rows][36]
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            red[row * GLD + lrow] = st_s[r];
            red[(32 + row) * GLD + lrow] = st_q[r];
        }
        __builtin_amdgcn_wave_barrier();
        const float *src = red + (lhalf * 32 + lrow) * GLD;
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f32x4 q4 = *reinterpret_cast<const f32x4 *>(src + 4 * i);
            tot += (q4[0] + q4[1]) + (q4[2] + q4[3]);
        }
        const rsrc_t rsP = make_rsrc(g.stats_out + (size_t)m0 * tiles_n * 2, (unsigned)(rows_m * tiles_n) * 8u);   // [M][tiles_n][2]
        buf_store1(tot, rsP, (unsigned)(((wave * 32 + lrow) * tiles_n + tn) * 2 + lhalf) * 4u, 0u);
    }
#ifdef DSG_PHASE_DIAG
    if (tid == 0 && g.prof) {   // g.prof doubles as a [4 x blocks] stamp buffer in this build
        const unsigned long long ph3 = __builtin_amdgcn_s_memrealtime();   // all stores issued
        __builtin_amdgcn_s_waitcnt(0);
        g.prof[4 * bid] = ph3 - ph2; g.prof[4 * bid + 1] = ph1 - prof_t0; g.prof[4 * bid + 2] = ph2 - ph1;
        g.prof[4 * bid + 3] = __builtin_amdgcn_s_memrealtime() - ph3;
    }
    return;
#endif
    if (g.prof && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been issued and acknowledged
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(g.prof, prof_t0);
        atomicMax(g.prof + 1, t1);
        // block 0's lifetime on both clocks: shader cycles / 100 MHz ticks = the clock the chip holds under this load
        if (bid == 0) { g.prof[2] = __builtin_amdgcn_s_memtime() - prof_c0; g.prof[3] = t1 - prof_t0; }
    }
}
static float *g_gelu_tab_dev = nullptr;
const float *gelu_table() {
    if (!g_gelu_tab_dev) {
        static float host_tab[GELU_TAB_FLOATS];
        for (int i = 0; i < GELU_NODES; i++) {
            const double x = -6.0 + i / 64.0;
            const double phi = std::exp(-0.5 * x * x) / std::sqrt(2.0 * M_PI);
            host_tab[4 * i + 0] = (float)(0.5 * std::erfc(-x / std::sqrt(2.0)));
            host_tab[4 * i + 1] = (float)phi;
            host_tab[4 * i + 2] = (float)(-0.5 * x * phi);
            host_tab[4 * i + 3] = 0.f;
        }
        if (hipMalloc((void **)&g_gelu_tab_dev, sizeof(host_tab)) != hipSuccess) return nullptr;
        if (hipMemcpy(g_gelu_tab_dev, host_tab, sizeof(host_tab), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    return g_gelu_tab_dev;
}

thread_local bool g_dry_run = false;

// false = this argument combination is not built; nothing has been launched and the caller reports DSG_ERR_INVALID (the host
// validates a whole forward's launches in a dry run at plan time, so a forward never meets this half-way)
bool launch_gemm(const GemmArgs &g, hipStream_t s) {
    if (g.M < 1 || g.N < 1 || g.K < GBK || g.K % GBK != 0 || !g.A || !g.W || !g.C) return false;
    if ((g.Ws3 || g.Wb) && launch_gemm_lp(g, s)) return true;   // opt-in bf16-MFMA modes (kernels_lp.hip)
    if (g.a_bf16 || g.c_bf16) return false;   // bf16 tensors need the bf16 GEMM kernel
    const int tiles_m = (g.M + GBM - 1) / GBM, tiles_n = (g.N + GBN - 1) / GBN;
    const dim3 grid(((tiles_m + 7) / 8) * 8 * tiles_n), block(256);
    const bool ln = g.ln_stats != nullptr || g.ln_part != nullptr, res = g.res != nullptr;
    if (g.batch > 1) {   // split-K slices of a training weight-gradient product: plain epilogue only
        if (ln || res || g.act != ACT_NONE || g.stats_out || g.a4_res > 0 || g.A2 || g.C2 || g.bias) return false;
        const dim3 gridb(grid.x * (unsigned)g.batch);
        DSG_LAUNCH((gemm4_f32_kernel<false, ACT_NONE, false, 0, 8, 2>), gridb, block, 0, s, g, tiles_m, tiles_n);
        return true;
    }
#define GEMM_CASE(L, A, R) DSG_LAUNCH((gemm4_f32_kernel<L, A, R, 0>), grid, block, 0, s, g, tiles_m, tiles_n)
#define GEMM_EPI(R, E) DSG_LAUNCH((gemm4_f32_kernel<false, ACT_NONE, R, E>), grid, block, 0, s, g, tiles_m, tiles_n)
    if (g.a4_res > 0) {   // PatchMerging gather + LayerNorm(4C) from partials; epilogue: plain, or premod + stats (dual store allowed)
        if (!g.ln_part || g.A2 || g.act != ACT_NONE || res || (g.K >> 2) % GBK != 0 || (g.stats_out && !g.mod_aff)) return false;
#define GEMM_MERGE(E) DSG_LAUNCH((gemm4_f32_kernel<true, ACT_NONE, false, E, 8, 1>), grid, block, 0, s, g, tiles_m, tiles_n)
        if (!g.stats_out) GEMM_MERGE(0); else if (g.mod_ld == 0) GEMM_MERGE(2); else GEMM_MERGE(3);
#undef GEMM_MERGE
        return true;
    }
    if (g.stats_out) {   // epilogue extensions: only the shapes the forward uses (plain A path, no activation)
        if (ln || g.act != ACT_NONE) return false;   // stats_out with LayerNorm / an activation is not built
        const int epi = !g.mod_aff ? 1 : (g.mod_ld == 0 ? 2 : 3);
        if (res) { if (epi == 1) GEMM_EPI(true, 1); else if (epi == 2) GEMM_EPI(true, 2); else GEMM_EPI(true, 3); }
        else { if (epi == 1) GEMM_EPI(false, 1); else if (epi == 2) GEMM_EPI(false, 2); else GEMM_EPI(false, 3); }
        return true;
    }
    if (g.act == ACT_GELU_KEEP || g.act == ACT_DGELU) {
        if (ln || (g.act == ACT_GELU_KEEP && (res || !g.C2)) || (g.act == ACT_DGELU && (!res || g.C2))) return false;
        if (g.act == ACT_GELU_KEEP) GEMM_CASE(false, ACT_GELU_KEEP, false); else GEMM_CASE(false, ACT_DGELU, true);
    }
    else if (ln && g.act == ACT_NONE && !res) GEMM_CASE(true, ACT_NONE, false);
    else if (ln && g.act == ACT_GELU && !res) GEMM_CASE(true, ACT_GELU, false);
    else if (!ln && g.act == ACT_NONE && res) GEMM_CASE(false, ACT_NONE, true);
    else if (!ln && g.act == ACT_NONE && !res) GEMM_CASE(false, ACT_NONE, false);
    else if (!ln && g.act == ACT_GELU && !res) GEMM_CASE(false, ACT_GELU, false);
    else if (!ln && g.act == ACT_SILU && !res) GEMM_CASE(false, ACT_SILU, false);
    else if (ln && g.act == ACT_NONE && res) GEMM_CASE(true, ACT_NONE, true);
    else if (ln && g.act == ACT_GELU && res) GEMM_CASE(true, ACT_GELU, true);
    else if (ln && g.act == ACT_SILU && !res) GEMM_CASE(true, ACT_SILU, false);
    else if (ln && g.act == ACT_SILU && res) GEMM_CASE(true, ACT_SILU, true);
    else if (!ln && g.act == ACT_GELU && res) GEMM_CASE(false, ACT_GELU, true);
    else GEMM_CASE(false, ACT_SILU, true);
#undef GEMM_EPI
#undef GEMM_CASE
    return true;
}

bool launch_gemm_qkv_attn(const GemmArgs &g, hipStream_t s) {
    const WinGeom &wg = g.wg;
    if ((wg.ws != 8 && wg.ws != 10) || wg.res % wg.ws != 0 || wg.C != 32 * wg.heads || g.N != 3 * wg.C || g.K % GBK != 0 ||
        g.act != ACT_NONE || g.res || g.A2 || !g.attn_bias || !(g.ln_stats || g.ln_part) || g.Wb || g.Ws3)
        return false;
    const int n_windows = g.attn_batch * (wg.res / wg.ws) * (wg.res / wg.ws);
    const int wpt = wg.ws == 8 ? 2 : 1;   // windows per 128-row tile: two of 64 tokens, or one of 100 padded to 128
    const int tiles_m = (n_windows + wpt - 1) / wpt, tiles_n = wg.heads;
    const dim3 grid(((tiles_m + 7) / 8) * 8 * tiles_n), block(256);
    if (wg.ws == 8) DSG_LAUNCH((gemm4_f32_kernel<true, ACT_NONE, false, 4, 8>), grid, block, 0, s, g, tiles_m, tiles_n);
    else DSG_LAUNCH((gemm4_f32_kernel<true, ACT_NONE, false, 4, 10>), grid, block, 0, s, g, tiles_m, tiles_n);
    return true;
}

// =================================================================================================
// Fused MLP half of a Swin block (diffusesg.py:275, :19-25):  x <- x + fc2(GELU(fc1(LayerNorm2(x))))
// for the narrow levels (C = 96, 192), where the unfused version is bound by the 4C-wide hidden tensor's
// round trip through HBM.  Everything stays in registers, there is no LDS and no barrier:
//   * one wave owns 32 tokens.  Products are formed TRANSPOSED, H^T = W1 . Xn^T, so a lane (m, half) always owns
//     token m: the fc1 accumulator (lane = token, register r = hidden unit (r&3)+8(r>>2)+4*half) is, register by
//     register, exactly the B operand of k-step r of the second product O^T = W2 . H^T -- the hidden activations
//     never leave the accumulator file (the f32 MFMA takes one VGPR per operand, so no repacking is needed);
//   * LayerNorm statistics come from the same fragments (a row is split over the two half-waves);
//   * weights are pre-packed fragment-major at load time ([tile][k-step][lane][4]) so every operand fetch is one
//     fully coalesced 1-KiB wave load served by L2; the next tile's fragments are in flight during the current MFMAs.
// =================================================================================================
template <int C>
__global__ __launch_bounds__(256, (C <= 96 ? 2 : 1)) void fused_mlp_kernel(float *__restrict__ x, const float *__restrict__ gam,
                                                           const float *__restrict__ bet, const float *__restrict__ W1p,
                                                           const float *__restrict__ b1, const float *__restrict__ W2p,
                                                           const float *__restrict__ b2, int M, float *__restrict__ stats_out) {
    constexpr int S = C / 8, CT = C / 32, NT = 4 * C / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int m = (blockIdx.x * 4 + wave) * 32 + lrow;
    const bool ok = m < M;
    float *xr = x + (size_t)(ok ? m : M - 1) * C + 4 * lhalf;

    f32x4 xn[S];
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < S; s++) {
        xn[s] = *reinterpret_cast<const f32x4 *>(xr + 8 * s);
        sum += (xn[s][0] + xn[s][1]) + (xn[s][2] + xn[s][3]);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) { const float d = xn[s][t] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = fast_rsqrt(var * (1.0f / C) + LN_EPS);
#pragma unroll
    for (int s = 0; s < S; s++) {
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(gam + 8 * s + 4 * lhalf);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(bet + 8 * s + 4 * lhalf);
        xn[s] = (xn[s] - mean) * rstd * gg + bb;
    }

    f32x16 oacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[ct][r] = 0.f;

    // packed weights via buffer loads: lane offset in the VGPR, tile / k-step offset scalar -> no address VALU
    const rsrc_t rs1 = make_rsrc(W1p, (unsigned)(4 * C * C) * 4u);  // [NT][S][64] float4
    const rsrc_t rs2 = make_rsrc(W2p, (unsigned)(4 * C * C) * 4u);  // [NT][CT][4][64] float4
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4 w1f[S], w2f[CT * 4];
#pragma unroll
    for (int s = 0; s < S; s++) w1f[s] = buf_load4(rs1, lane16, (unsigned)s * 1024u);
    for (int nt = 0; nt < NT; nt++) {
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(b1 + 32 * nt + 8 * g + 4 * lhalf);
#pragma unroll
            for (int t = 0; t < 4; t++) hacc[4 * g + t] = bv[t];
        }
#pragma unroll
        for (int q = 0; q < CT * 4; q++) w2f[q] = buf_load4(rs2, lane16, (unsigned)(nt * CT * 4 + q) * 1024u);
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int t = 0; t < 4; t++) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[s][t], xn[s][t], hacc, 0, 0, 0);
        if (nt + 1 < NT) {
#pragma unroll
            for (int s = 0; s < S; s++) w1f[s] = buf_load4(rs1, lane16, (unsigned)((nt + 1) * S + s) * 1024u);
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2_t gg = gelu_f2(hacc[r], hacc[r + 1]); hacc[r] = gg[0]; hacc[r + 1] = gg[1]; }   // closed form: no LDS, no barrier in this kernel (table version: -0.2 % in the A/B)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[ct * 4 + g][t], hacc[4 * g + t], oacc[ct], 0, 0, 0);
    }
    // epilogue: channel of oacc[ct][4g+t] is 32ct + 8g + 4*half + t -- the same pattern as the input fragments
    float rs = 0.f, rq = 0.f;   // (sum, sumsq) of the row written back: a row is split over the two half-waves
    if (ok) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int c = 32 * ct + 8 * g;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + c);
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(b2 + c + 4 * lhalf);
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; t++) { o[t] = oacc[ct][4 * g + t] + bv[t] + xv[t]; rs += o[t]; rq = fmaf(o[t], o[t], rq); }
                *reinterpret_cast<f32x4 *>(xr + c) = o;
            }
    }
    if (stats_out) {   // same format as the GEMM epilogue's partials with one column tile... C/96 tiles folded into one pair
        rs += __shfl_xor(rs, 32, 64);
        rq += __shfl_xor(rq, 32, 64);
        if (ok && lhalf == 0) { stats_out[2 * (size_t)m] = rs; stats_out[2 * (size_t)m + 1] = rq; }
    }
}

void launch_fused_mlp(float *x, const float *gam, const float *bet, const float *W1p, const float *b1, const float *W2p,
                      const float *b2, int M, int C, float *stats_out, hipStream_t s) {
    const dim3 grid((M + 127) / 128), block(256);
    if (C == 96) DSG_LAUNCH(fused_mlp_kernel<96>, grid, block, 0, s, x, gam, bet, W1p, b1, W2p, b2, M, stats_out);
    else if (C == 192) DSG_LAUNCH(fused_mlp_kernel<192>, grid, block, 0, s, x, gam, bet, W1p, b1, W2p, b2, M, stats_out);
}

// =================================================================================================
// Fused attention half of a Swin block for C = 96 (3 heads of 32), diffusesg.py:238-272:
//   x <- silu(shift + x*(1+scale));  x <- x + proj(window_attention(LayerNorm1(x)))
// One wave owns one window (MB blocks of 32 tokens) and chains every product through the accumulator file:
//   Q^T, K^T = W . Xn^T      (lane = token, register = head dim d)        -- "swapped" products
//   V        = Xn . Wv^T     (lane = head dim d, register = key)           -- unswapped product, same Xn fragments
//   S^T     += K^T[r] x Q^T[r]   over the 16 registers r (both operands already have lane = key / query, slot = d)
//   softmax over keys: lane-local over registers + one half-wave exchange
//   O^T      = V[r] x P[r]   (lane = query, register = d)  ->  Y^T += Wproj[:, head] x O^T[r]
// No LDS, no barrier, no intermediate tensor in HBM; weights are read as pre-packed fragment-major 1-KiB wave loads.
// =================================================================================================
template <int MB, bool PREMOD = false>   // PREMOD: x arrives already modulated (the producer applied silu(shift + x(1+scale)))
__global__ __launch_bounds__(256, 1) void fused_attn96_kernel(float *__restrict__ x, const float *__restrict__ aff, int aff_ld,
                                                             int aff_off, const float *__restrict__ gam,
                                                             const float *__restrict__ bet, const float *__restrict__ Wqp,
                                                             const float *__restrict__ bqkv, const float *__restrict__ biasT,
                                                             const float *__restrict__ Wpp, const float *__restrict__ bproj,
                                                             WinGeom g, int n_windows) {
    constexpr int C = 96, S = 12, CT = 3, HEADS = 3, Wp = 32 * MB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int win = blockIdx.x * 4 + wave;
    if (win >= n_windows) return;  // no block-level synchronisation below
    const int res = g.res, ws = g.ws, nwr = res / ws, nW = nwr * nwr, Wt = ws * ws, T = res * res;
    const int w = win % nW, b = win / nW;
    const int wi = w / nwr, wj = w % nwr;
    const float *bias_w = biasT + (size_t)(g.shift > 0 ? w : 0) * HEADS * Wp * Wp;
    const float *scale = aff + (size_t)b * aff_ld + aff_off + 4 * lhalf, *shift = scale + C;

    // this lane's token in each 32-token block of the window (cyclic shift folded in)
    float *xrow[MB];
    bool valid[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const int p = 32 * mb + lrow;
        valid[mb] = p < Wt;
        int t = 0;
        if (valid[mb]) {
            const int si = wi * ws + p / ws, sj = wj * ws + p % ws;
            t = ((si + g.shift) % res) * res + ((sj + g.shift) % res);
        }
        xrow[mb] = x + ((size_t)b * T + t) * C + 4 * lhalf;
    }

    // modulate + SiLU, LayerNorm-1 (statistics of the modulated row; a row is split over the two half-waves)
    f32x4 xn[MB][S];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow[mb] + 8 * s);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 8 * s);
            const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 8 * s);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                xn[mb][s][t] = PREMOD ? v[t] : silu_exact(sh[t] + v[t] * (sc[t] + 1.0f));
                sum += xn[mb][s][t];
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float var = 0.f;
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int t = 0; t < 4; t++) { const float d = xn[mb][s][t] - mean; var = fmaf(d, d, var); }
        var += __shfl_xor(var, 32, 64);
        const float rstd = fast_rsqrt(var * (1.0f / C) + LN_EPS);
#pragma unroll
        for (int s = 0; s < S; s++) {
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(gam + 8 * s + 4 * lhalf);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bet + 8 * s + 4 * lhalf);
            xn[mb][s] = (xn[mb][s] - mean) * rstd * gg + bb;
        }
    }

    f32x16 yacc[MB][CT];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) yacc[mb][ct][r] = 0.f;

    const rsrc_t rsq = make_rsrc(Wqp, (unsigned)(3 * C * C) * 4u);  // [9 tiles][S][64] float4
    const rsrc_t rsp = make_rsrc(Wpp, (unsigned)(C * C) * 4u);      // [3 heads][CT][4][64] float4
    const unsigned lane16 = (unsigned)lane * 16u;

    for (int hd = 0; hd < HEADS; hd++) {
        f32x16 qa[MB], ka[MB], va[MB];
        // Q^T and K^T tiles: bias per register (= per head dim)
#pragma unroll
        for (int which = 0; which < 2; which++) {
            const int nt = which * 3 + hd;
            f32x4 wf[S];
#pragma unroll
            for (int s = 0; s < S; s++) wf[s] = buf_load4(rsq, lane16, (unsigned)(nt * S + s) * 1024u);
            f32x16 init;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bqkv + 32 * nt + 8 * gq + 4 * lhalf);
#pragma unroll
                for (int t = 0; t < 4; t++) init[4 * gq + t] = bv[t];
            }
#pragma unroll
            for (int mb = 0; mb < MB; mb++) {
                f32x16 a = init;
#pragma unroll
                for (int s = 0; s < S; s++)
#pragma unroll
                    for (int t = 0; t < 4; t++) a = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s][t], xn[mb][s][t], a, 0, 0, 0);
                if (which == 0) qa[mb] = a;  // q weights/bias are pre-scaled by 32^-0.5 * log2(e) at pack time
                else ka[mb] = a;
            }
        }
        // V tile, unswapped: lane = head dim, register = key; bias per lane
        {
            const int nt = 6 + hd;
            f32x4 wf[S];
#pragma unroll
            for (int s = 0; s < S; s++) wf[s] = buf_load4(rsq, lane16, (unsigned)(nt * S + s) * 1024u);
            const float bv = bqkv[32 * nt + lrow];
#pragma unroll
            for (int mb = 0; mb < MB; mb++) {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; r++) a[r] = bv;
#pragma unroll
                for (int s = 0; s < S; s++)
#pragma unroll
                    for (int t = 0; t < 4; t++) a = __builtin_amdgcn_mfma_f32_32x32x2f32(xn[mb][s][t], wf[s][t], a, 0, 0, 0);
                va[mb] = a;
            }
        }
        f32x4 pf[CT * 4];
#pragma unroll
        for (int q = 0; q < CT * 4; q++) pf[q] = buf_load4(rsp, lane16, (unsigned)(hd * CT * 4 + q) * 1024u);
        const float *bias_h = bias_w + (size_t)hd * Wp * Wp;
#pragma unroll
        for (int qb = 0; qb < MB; qb++) {
            if (32 * qb >= Wt) break;
            f32x16 sa[MB];
            float mx = -3.0e38f;
#pragma unroll
            for (int kb = 0; kb < MB; kb++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    sa[kb][r] = bias_h[(size_t)key * Wp + 32 * qb + lrow];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) sa[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[kb][r], qa[qb][r], sa[kb], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) mx = fmaxf(mx, sa[kb][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < MB; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float e = __builtin_amdgcn_exp2f(sa[kb][r] - mx);  // scores carry the log2(e) factor
                    sa[kb][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = fast_rcp(sum);
            f32x16 ot;
#pragma unroll
            for (int r = 0; r < 16; r++) ot[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < MB; kb++)
#pragma unroll
                for (int r = 0; r < 16; r++) ot = __builtin_amdgcn_mfma_f32_32x32x2f32(va[kb][r], sa[kb][r] * inv, ot, 0, 0, 0);
            // proj partial of this head: Y^T[c][m] += Wp[c][32hd + d] * O[m][d]
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        yacc[qb][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[ct * 4 + gq][t], ot[4 * gq + t], yacc[qb][ct], 0, 0, 0);
        }
    }

    // epilogue: x_new = silu(shift + x*(1+scale)) (recomputed: the shortcut) + Y + proj bias
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        if (!valid[mb]) continue;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int c = 32 * ct + 8 * gq;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow[mb] + c);
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + c);
                const f32x4 bp = *reinterpret_cast<const f32x4 *>(bproj + c + 4 * lhalf);
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; t++) o[t] = (PREMOD ? v[t] : silu_exact(sh[t] + v[t] * (sc[t] + 1.0f))) + yacc[mb][ct][4 * gq + t] + bp[t];
                *reinterpret_cast<f32x4 *>(xrow[mb] + c) = o;
            }
    }
}

void launch_fused_attn96(float *x, const float *aff, int aff_ld, int aff_off, const float *gam, const float *bet, const float *Wqp,
                         const float *bqkv, const float *biasT, const float *Wpp, const float *bproj, int B, const WinGeom &g,
                         bool premod, hipStream_t s) {
    const int nW = (g.res / g.ws) * (g.res / g.ws), n_windows = B * nW;
    const int MB = (g.ws * g.ws + 31) / 32;
    const dim3 grid((n_windows + 3) / 4), block(256);
#define FA(MB_, PM_) DSG_LAUNCH((fused_attn96_kernel<MB_, PM_>), grid, block, 0, s, x, aff, aff_ld, aff_off, gam, bet, Wqp, bqkv, biasT, Wpp, bproj, g, n_windows)
    if (MB == 1) { if (premod) FA(1, true); else FA(1, false); }
    else { if (premod) FA(2, true); else FA(2, false); }
#undef FA
}

// =================================================================================================
// Fused read-out for E = 96 (diffusesg.py:758-761, 806-813, 825): final LayerNorm, the three 1x1 read_out convs and
// readout_adj_mlp.fc1 are one affine map before the GELU (folded on the host at weight load: Fa = F1.W2.W1.W0^T),
// so per 32 tokens:  H^T = Fa . LN(x)^T -> GELU -> out^T = F2 . H^T  -> masked store in [B,C_adj,N,N] layout.
// The same pass accumulates the node head's padding-aware pooling of LN(x) (the pooled shared_rep is recovered
// from it by linearity in the node chain): pool[b,i,:] += (f_i f_j / N) * LN(x)[b,i,j,:].
// =================================================================================================
// BF (the opt-in bf16 mode's block pipeline only): the two products on v_mfma_f32_32x32x16_bf16 -- 18 + 6 instructions per wave instead
// of 144 + 48 f32 ones; LN(x) pairs as they stand in the registers are the first product's B operand (k order 16 s + 8 (j >> 2) + 4 half +
// (j & 3), the packed weights Wfp / W2p follow it: dsg_api.cpp), the GELU'd accumulators the second's.  The pooling stays fp32.
// -------------------------------------------------------------------------------------------------
// Row tiles through LDS.  The register-chained C = 96 kernels give a lane ONE token row, so a direct load / store instruction touches
// 32 rows x 32 bytes: thirty-two cache lines a quarter used, and the texture path (not HBM) bounds the kernel's I/O (round 4:
// profiles/r4/m384_experiments.txt -- the HBM phases of a block do not speed up when the other CUs are kept off the memory).  A wave's
// 32 consecutive token rows are ONE contiguous 12-KB piece of the [M, 96] tensor: it is copied linearly (1 KiB per wave-instruction, whole
// cache lines) and only crosses between "lane = row" and "lane = 16 consecutive bytes" inside the wave's own LDS tile (rows padded to 100
// floats: conflict-free both ways).  No barrier: only the owning wave touches its tile, and a wave's LDS operations complete in order.
// -------------------------------------------------------------------------------------------------
constexpr int T96_LD = 100;                       // floats per padded row
constexpr int T96_FLOATS = 32 * T96_LD;           // one wave's tile
// fragment values of the lane's row -> the wave's tile; v(e4) = the f32x4 of channels 4 e4 .. + 3 owned by this lane (e4 = 2 q + half)
__device__ __forceinline__ void t96_put(float *tile, int lrow, int e4, f32x4 v) { *reinterpret_cast<f32x4 *>(tile + lrow * T96_LD + 4 * e4) = v; }
__device__ __forceinline__ f32x4 t96_get(const float *tile, int lrow, int e4) { return *reinterpret_cast<const f32x4 *>(tile + lrow * T96_LD + 4 * e4); }
// the tile's rows [0, rows) -> 32 consecutive rows of a [*, 96] fp32 tensor starting at g (row-contiguous 16-byte pieces, whole lines)
__device__ __forceinline__ void t96_store_rows(const float *tile, float *g, int rows, int lane) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int p = 64 * k + lane, r = p / 24, c = p - 24 * r;
        if (r < rows) *reinterpret_cast<f32x4 *>(g + (size_t)r * 96 + 4 * c) = *reinterpret_cast<const f32x4 *>(tile + r * T96_LD + 4 * c);
    }
}
__device__ __forceinline__ void t96_load_rows(float *tile, const float *g, int rows, int lane) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int p = 64 * k + lane, r = p / 24, c = p - 24 * r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) v = *reinterpret_cast<const f32x4 *>(g + (size_t)r * 96 + 4 * c);
        *reinterpret_cast<f32x4 *>(tile + r * T96_LD + 4 * c) = v;
    }
}
// the same for a bf16 [*, 96] tensor: the tile holds packed pairs in its first 48 floats per row (12 pieces of 16 bytes)
__device__ __forceinline__ void t96_put_bf16(float *tile, int lrow, int e4, u32x2_c v) { *reinterpret_cast<u32x2_c *>(tile + lrow * T96_LD + 2 * e4) = v; }
__device__ __forceinline__ void t96_store_rows_bf16(const float *tile, unsigned short *g, int rows, int lane) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int p = 64 * k + lane, r = p / 12, c = p - 12 * r;
        if (r < rows) *reinterpret_cast<f32x4 *>(g + (size_t)r * 96 + 8 * c) = *reinterpret_cast<const f32x4 *>(tile + r * T96_LD + 4 * c);
    }
}

template <bool BF>
__global__ __launch_bounds__(256, 2) void fused_readout96_kernel(const float *__restrict__ x, const float *__restrict__ gam,
                                                                 const float *__restrict__ bet, const float *__restrict__ Wfp,
                                                                 const float *__restrict__ fa, const float *__restrict__ W2p,
                                                                 const float *__restrict__ f2, const uint8_t *__restrict__ flags,
                                                                 float *__restrict__ out_adj, float *__restrict__ pool_part,
                                                                 int nseg, int B, int N, int Ca) {
    constexpr int C = 96, S = 12;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int T = N * N, M = B * T;
    const int m = (blockIdx.x * 4 + wave) * 32 + lrow;
    const bool ok = m < M;
    const int mc = ok ? m : M - 1;
    const int b = mc / T, i = (mc / N) % N, j = mc % N;
    const float *xr = x + (size_t)mc * C + 4 * lhalf;
    f32x4 xn[S];
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < S; s++) {
        xn[s] = *reinterpret_cast<const f32x4 *>(xr + 8 * s);   // (through the wave's LDS row tile instead: measured slower, 1.40 vs 1.38 ms for the six C = 96 launches)
        sum += (xn[s][0] + xn[s][1]) + (xn[s][2] + xn[s][3]);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) { const float d = xn[s][t] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = fast_rsqrt(var * (1.0f / C) + LN_EPS);
#pragma unroll
    for (int s = 0; s < S; s++) {
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(gam + 8 * s + 4 * lhalf);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(bet + 8 * s + 4 * lhalf);
        xn[s] = (xn[s] - mean) * rstd * gg + bb;
    }
    const bool fi = flags[(size_t)b * N + i] != 0, fj = flags[(size_t)b * N + j] != 0;
    const bool valid = ok && fi && fj;

    // pooled LN(x): segmented sum over the tokens of this wave that share a row (b,i).  No atomics: the wave stores its
    // partial for row r into slot (tile - first tile of r) of pool_part[r]; pool_finish_kernel adds a row's slots in slot
    // order, so the node head is bitwise reproducible.
    {
        const int tile = blockIdx.x * 4 + wave;
        const int rowid = ok ? b * N + i : -1;
        const int r_first = __shfl(rowid, 0, 64);
        int r_last = __shfl(rowid, 31, 64);
        if (r_last < 0) r_last = (M - 1) / N;  // partially filled last tile
        const float wgt = valid ? 1.0f / (float)N : 0.f;
        if (r_first < 0) r_last = -2;          // a tile entirely beyond M (padding wave of the last block) owns no row
        for (int r = r_first; r <= r_last; r++) {
            const float sel = (rowid == r) ? wgt : 0.f;
            float *dst = pool_part + ((size_t)r * nseg + (tile - (r * N) / 32)) * C;
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float v = sel * xn[s][t];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    if (lrow == 0) dst[8 * s + 4 * lhalf + t] = v;
                }
        }
    }

    const f32x4 *w1 = reinterpret_cast<const f32x4 *>(Wfp) + lane;
    const f32x4 *w2 = reinterpret_cast<const f32x4 *>(W2p) + lane;
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t *w1b = reinterpret_cast<const u32x4_t *>(Wfp) + lane, *w2b = reinterpret_cast<const u32x4_t *>(W2p) + lane;   // BF: bf16 fragments
    u32x4_t xb[S / 2];
    if (BF) {
#pragma unroll
        for (int s2 = 0; s2 < S / 2; s2++) {
            xb[s2][0] = pack_bf16(xn[2 * s2][0], xn[2 * s2][1]); xb[s2][1] = pack_bf16(xn[2 * s2][2], xn[2 * s2][3]);
            xb[s2][2] = pack_bf16(xn[2 * s2 + 1][0], xn[2 * s2 + 1][1]); xb[s2][3] = pack_bf16(xn[2 * s2 + 1][2], xn[2 * s2 + 1][3]);
        }
    }
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(fa + 32 * nt + 8 * g + 4 * lhalf);
#pragma unroll
            for (int t = 0; t < 4; t++) hacc[4 * g + t] = bv[t];
        }
        if (BF) {
#pragma unroll
            for (int s2 = 0; s2 < S / 2; s2++)
                hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1b[((size_t)nt * (S / 2) + s2) * 64]),
                                                               __builtin_bit_cast(bf16x8_t, xb[s2]), hacc, 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const f32x4 a = w1[((size_t)nt * S + s) * 64];
#pragma unroll
                for (int t = 0; t < 4; t++) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], xn[s][t], hacc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2_t gg = gelu_f2(hacc[r], hacc[r + 1]); hacc[r] = gg[0]; hacc[r + 1] = gg[1]; }
        if (BF) {
#pragma unroll
            for (int g2 = 0; g2 < 2; g2++) {
                u32x4_t hb;
#pragma unroll
                for (int j = 0; j < 4; j++) hb[j] = pack_bf16(hacc[8 * g2 + 2 * j], hacc[8 * g2 + 2 * j + 1]);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w2b[((size_t)nt * 2 + g2) * 64]),
                                                               __builtin_bit_cast(bf16x8_t, hb), oacc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 a = w2[((size_t)nt * 4 + g) * 64];
#pragma unroll
                for (int t = 0; t < 4; t++) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], hacc[4 * g + t], oacc, 0, 0, 0);
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            if (c < Ca) out_adj[(((size_t)b * Ca + c) * N + i) * N + j] = valid ? oacc[r] + f2[c] : 0.f;
        }
    }
}

// pool_ext [B*N,128]: columns 0..95 = the row's partial sums added in slot order (row r spans the 32-token tiles
// (r*N)/32 .. (r*N+N-1)/32), column 96 = f_i * (#valid nodes) / N (the factor of the folded constant term), rest zero
__global__ void pool_finish_kernel(const uint8_t *flags, const float *pool_part, int nseg, float *pool_ext, int B, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N * 128) return;
    const int col = idx & 127, row = idx >> 7, b = row / N;
    float v = 0.f;
    if (col < 96) {
        const int segs = (row * N + N - 1) / 32 - (row * N) / 32 + 1;
        for (int sg = 0; sg < segs; sg++) v += pool_part[((size_t)row * nseg + sg) * 96 + col];
    } else if (col == 96 && flags[row]) {
        int cnt = 0;
        for (int jj = 0; jj < N; jj++) cnt += flags[(size_t)b * N + jj] ? 1 : 0;
        v = (float)cnt / (float)N;
    }
    pool_ext[idx] = v;
}

int readout_pool_segments(int N) { return (N + 30) / 32 + 1; }  // most 32-token tiles a row of N tokens can touch

void launch_fused_readout96(const float *x, const float *gam, const float *bet, const float *Wfp, const float *fa, const float *W2p,
                            const float *f2, const uint8_t *flags, float *out_adj, float *pool_part, float *pool_ext, int B, int N,
                            int Ca, hipStream_t s, bool bf16_frags) {
    const int M = B * N * N, nseg = readout_pool_segments(N);
    if (bf16_frags)
        DSG_LAUNCH(fused_readout96_kernel<true>, dim3((M + 127) / 128), dim3(256), 0, s, x, gam, bet, Wfp, fa, W2p, f2, flags, out_adj,
                           pool_part, nseg, B, N, Ca);
    else
        DSG_LAUNCH(fused_readout96_kernel<false>, dim3((M + 127) / 128), dim3(256), 0, s, x, gam, bet, Wfp, fa, W2p, f2, flags, out_adj,
                           pool_part, nseg, B, N, Ca);
    const int n = B * N * 128;
    DSG_LAUNCH(pool_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, flags, pool_part, nseg, pool_ext, B, N);
}

// =================================================================================================
// Fused PatchEmbed for E = 96 (diffusesg.py:784-802 input assembly, :562-577 1x1 conv + LayerNorm + modulate+SiLU).
// Per 32 tokens: the (i,j) input vector [sc_adj, adj, sc_node(i), node(i), sc_node(j), node(j)] is gathered straight
// into MFMA B-fragments (lane = token; consecutive lanes = consecutive j, so the channel-major adjacency reads are
// coalesced), Y^T = Wpe . in^T on the matrix cores, LayerNorm over the 96 outputs (split over registers and the two
// half-waves), modulate, SiLU, 16-B stores.  KP = in_chans rounded up to 32.
// =================================================================================================
template <int KP, int CA = 0, int CN = 0>   // CA, CN > 0: compile-time channel counts (the index divisions become multiplies)
__global__ __launch_bounds__(256, 2) void fused_patch_embed96_kernel(const float *__restrict__ adj, const float *__restrict__ node,
                                                                    const float *__restrict__ sc_adj, const float *__restrict__ sc_node,
                                                                    const int *__restrict__ has_sc, const uint8_t *__restrict__ flags,
                                                                    const float *__restrict__ Wp, const float *__restrict__ bias,
                                                                    const float *__restrict__ gam, const float *__restrict__ bet,
                                                                    const float *__restrict__ aff, int aff_ld, int aff_off, int aff_off2,
                                                                    float *__restrict__ x, int B, int N, int Ca_rt, int Cn_rt, int self_cond,
                                                                    void *__restrict__ xn) {
    constexpr int C = 96, S = KP / 8;
    const int Ca = CA > 0 ? CA : Ca_rt, Cn = CN > 0 ? CN : Cn_rt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int T = N * N, M = B * T;
    const int m = (blockIdx.x * 4 + wave) * 32 + lrow;
    const bool ok = m < M;
    const int mc = ok ? m : M - 1;
    const int b = mc / T, i = (mc / N) % N, j = mc % N;
    const bool sc_on = self_cond && sc_adj != nullptr && (has_sc == nullptr || *has_sc != 0);
    const int nsc = self_cond ? 2 : 1;
    const bool mask = flags[(size_t)b * N + i] && flags[(size_t)b * N + j];
    f32x4 in[S];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int c = 8 * s + 4 * lhalf + t;
            float v = 0.f;
            if (c < nsc * Ca) {
                const int which = self_cond ? c / Ca : 1, a = c % Ca;
                const float *src = (which == 0) ? (sc_on ? sc_adj : nullptr) : adj;
                if (src) v = src[(((size_t)b * Ca + a) * N + i) * N + j];
            } else if (c < nsc * (Ca + 2 * Cn)) {
                const int cc = c - nsc * Ca;
                const int colpart = cc / (nsc * Cn), c2 = cc % (nsc * Cn);
                const int which = self_cond ? c2 / Cn : 1, a = c2 % Cn;
                const float *src = (which == 0) ? (sc_on ? sc_node : nullptr) : node;
                if (src && mask) v = src[((size_t)b * N + (colpart ? j : i)) * Cn + a];
            }
            in[s][t] = v;
        }
    const f32x4 *w = reinterpret_cast<const f32x4 *>(Wp) + lane;  // [3][S][64] float4
    f32x16 acc[3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + 32 * nt + 8 * g + 4 * lhalf);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[nt][4 * g + t] = bv[t];
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            const f32x4 a = w[((size_t)nt * S + s) * 64];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], in[s][t], acc[nt], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) sum += acc[nt][r];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);
    float var = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) { const float d = acc[nt][r] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = fast_rsqrt(var * (1.0f / C) + LN_EPS);
    __shared__ __attribute__((aligned(16))) float tiles[4 * T96_FLOATS];
    float *tile = tiles + wave * T96_FLOATS;
    const int m_w = (blockIdx.x * 4 + wave) * 32, rows_w = min(32, M - m_w);   // this wave's rows (<= 0: a padding wave)
    const float *scale = aff + (size_t)b * aff_ld + aff_off + 4 * lhalf, *shift = scale + C;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int e = 32 * nt + 8 * g;
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(gam + e + 4 * lhalf);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bet + e + 4 * lhalf);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + e);
            const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + e);
            f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float n = (acc[nt][4 * g + t] - mean) * rstd * gg[t] + bb[t];
                o[t] = silu_exact(sh[t] + n * (sc[t] + 1.0f));
            }
            if (aff_off2 >= 0) {   // the first Swin block's modulate+SiLU on top (it then takes its input pre-modulated)
                const float *scale2 = aff + (size_t)b * aff_ld + aff_off2 + 4 * lhalf;
                const f32x4 sc2 = *reinterpret_cast<const f32x4 *>(scale2 + e), sh2 = *reinterpret_cast<const f32x4 *>(scale2 + C + e);
#pragma unroll
                for (int t = 0; t < 4; t++) o[t] = silu_exact(sh2[t] + o[t] * (sc2[t] + 1.0f));
            }
            t96_put(tile, lrow, (e >> 2) + lhalf, o);
            if (xn) {
#pragma unroll
                for (int t = 0; t < 4; t++) { acc[nt][4 * g + t] = o[t]; s1 += o[t]; s2 = fmaf(o[t], o[t], s2); }
            }
        }
    if (rows_w > 0) t96_store_rows(tile, x + (size_t)m_w * C, rows_w, lane);
    if (xn) {   // bf16 block pipeline: LayerNorm-1 (no affine) of the stored row as the bf16 tensor the first QKV GEMM reads
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float mu = s1 * (1.0f / C), rs = fast_rsqrt(fmaxf(fmaf(-mu, mu, s2 * (1.0f / C)), 0.f) + LN_EPS), nmr = -mu * rs;
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = fmaf(acc[nt][4 * g + t], rs, nmr);
                t96_put_bf16(tile, lrow, (32 * nt + 8 * g) / 4 + lhalf, pack_bf16x4(v));
            }
        if (rows_w > 0) t96_store_rows_bf16(tile, reinterpret_cast<unsigned short *>(xn) + (size_t)m_w * C, rows_w, lane);
    }
}

bool launch_fused_patch_embed96(const float *adj, const float *node, const float *sc_adj, const float *sc_node, const int *has_sc,
                                const uint8_t *flags, const float *Wp, const float *bias, const float *gam, const float *bet,
                                const float *aff, int aff_ld, int aff_off, int aff_off2, float *x, int B, int N, int Ca, int Cn,
                                int self_cond, int Kp, hipStream_t s, void *xn) {
    const int M = B * N * N;
    const dim3 grid((M + 127) / 128), block(256);
#define PE_ARGS adj, node, sc_adj, sc_node, has_sc, flags, Wp, bias, gam, bet, aff, aff_ld, aff_off, aff_off2, x, B, N, Ca, Cn, self_cond, xn
    if (Kp == 64 && Ca == 6 && Cn == 12) DSG_LAUNCH((fused_patch_embed96_kernel<64, 6, 12>), grid, block, 0, s, PE_ARGS);       // VG bits
    else if (Kp == 64 && Ca == 3 && Cn == 12) DSG_LAUNCH((fused_patch_embed96_kernel<64, 3, 12>), grid, block, 0, s, PE_ARGS);  // COCO bits
    else if (Kp == 32) DSG_LAUNCH((fused_patch_embed96_kernel<32>), grid, block, 0, s, PE_ARGS);
    else if (Kp == 64) DSG_LAUNCH((fused_patch_embed96_kernel<64>), grid, block, 0, s, PE_ARGS);
    else return false;
#undef PE_ARGS
    return true;
}

// =================================================================================================
// Window attention core (R/model/diffusesg/diffusesg.py:108-139 with the window partition, cyclic
// shift and their inverses of :28-57, :246-271 folded into the token index).
//
// One wave per (sample, window, head).  q/k/v of a head are 32 floats = one 128-B segment of a
// token's qkv row, so operands are loaded straight from global memory into MFMA fragments:
//   S^T[key][query] = K . (scale*Q)^T      (A = K rows, B = Q rows; "swapped" product so that a
//                                           lane owns one query column and softmax is lane-local)
//   O[query][d]     = P . V                (A = P taken from the S^T accumulators in place -- a lane
//                                           already holds P[query][key] for exactly the keys the A
//                                           operand of k-step r needs; B = V row of that key)
// Relative-position bias and the shifted-window mask (-100) come pre-combined and transposed
// (key-major) from a dense table built at weight-load time; padded key slots hold -1e30.
// =================================================================================================
template <int KT, int WS, bool OBF = false, bool IBF = false>  // KT 32-token tiles per window (Wp = 32*KT >= WS*WS); WS compile-time:
                                             // cheap index math; OBF: the output is stored as bf16 (bf16 mode: the proj GEMM rounds it
                                             // to bf16 anyway); IBF: qkv is a bf16 tensor (widened to fp32 on load, math unchanged)
__global__ __launch_bounds__(256) void window_attn_kernel(const float *__restrict__ qkv, const float *__restrict__ biasT,
                                                          float *__restrict__ out, int B, WinGeom g, int n_units) {
    constexpr int Wp = 32 * KT, Wt = WS * WS, VLD = 36;
    __shared__ __attribute__((aligned(16))) float v_lds[4][Wp * VLD];  // V rows of the window, one slab per wave
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

    // this lane's token per 32-position block (byte offset of its qkv row); padded positions read token 0
    unsigned rowoff[KT];
    int tokr[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const int p = 32 * kt + lrow;
        int t = 0;
        if (p < Wt) {
            const int si = wi * WS + p / WS, sj = wj * WS + p % WS;
            int ti = si + g.shift, tj = sj + g.shift;
            if (ti >= res) ti -= res;
            if (tj >= res) tj -= res;
            t = ti * res + tj;
        }
        tokr[kt] = t;
        rowoff[kt] = IBF ? (unsigned)t * (unsigned)(3 * C) * 2u + 8u * lhalf : (unsigned)t * (unsigned)(3 * C) * 4u + 16u * lhalf;
    }
    // descriptors: the sample's qkv rows / the (window-type, head) bias tile / the sample's output rows
    const rsrc_t rsQ = IBF ? make_rsrc(reinterpret_cast<const unsigned short *>(qkv) + (size_t)b * T * 3 * C, (unsigned)T * 3u * C * 2u)
                           : make_rsrc(qkv + (size_t)b * T * 3 * C, (unsigned)T * 3u * C * 4u);
    // q | k | v fragment of this lane: 4 consecutive head dims at 8s + 4*half of head `head`, part 0 / 1 / 2
    auto qkv_frag = [&](unsigned roff, int part, int s4) -> f32x4 {
        return IBF ? buf_load4_bf16(rsQ, roff, (unsigned)(part * C) * 2u + (unsigned)head * 64u + 16u * s4)
                   : buf_load4(rsQ, roff, (unsigned)(part * C) * 4u + (unsigned)head * 128u + 32u * s4);
    };
    const rsrc_t rsB = make_rsrc(biasT + ((size_t)(g.shift > 0 ? w : 0) * heads + head) * Wp * Wp, (unsigned)(Wp * Wp) * 4u);
    const rsrc_t rsO = OBF ? make_rsrc(reinterpret_cast<unsigned short *>(out) + (size_t)b * T * C, active ? (unsigned)T * C * 2u : 0u)
                           : make_rsrc(out + (size_t)b * T * C, active ? (unsigned)T * C * 4u : 0u);
    const unsigned hoff = (unsigned)head * 128u;

    // K fragments (lane = key), V rows -> LDS (for the transposed read lane = head dim)
    f32x4 kf[KT][4];
    float *vl = v_lds[wave];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
#pragma unroll
        for (int s = 0; s < 4; s++) kf[kt][s] = qkv_frag(rowoff[kt], 1, s);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const f32x4 v = qkv_frag(rowoff[kt], 2, s);
            *reinterpret_cast<f32x4 *>(vl + (32 * kt + lrow) * VLD + 8 * s + 4 * lhalf) = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // V fragments: lane (d, half) holds V[key = 32kt + (r&3) + 8(r>>2) + 4*half][d]
    float vf[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) vf[kt][r] = vl[(32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * VLD + lrow];

    const unsigned boff = (unsigned)(4 * lhalf * Wp + lrow) * 4u;
#pragma unroll
    for (int qt = 0; qt < KT; qt++) {
        if (32 * qt >= Wt) break;
        f32x4 qf[4];
#pragma unroll
        for (int s = 0; s < 4; s++) qf[s] = qkv_frag(rowoff[qt], 0, s);  // pre-scaled by 32^-0.5 * log2(e)
        f32x16 sacc[KT];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                sacc[kt][r] = buf_load1(rsB, boff, (unsigned)((32 * kt + (r & 3) + 8 * (r >> 2)) * Wp + 32 * qt) * 4u);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt][s][t], qf[s][t], sacc[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float e = __builtin_amdgcn_exp2f(sacc[kt][r] - mx);  // scores carry the log2(e) factor
                sacc[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = fast_rcp(sum);
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sacc[kt][r] * inv, vf[kt][r], oacc, 0, 0, 0);
        // O tile: col = d = lrow, row = query 32*qt + (r&3) + 8(r>>2) + 4*half; the row's token comes from the lane that owns it
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int q0 = (r & 3) + 8 * (r >> 2);
            const int ta = __builtin_amdgcn_readlane(tokr[qt], q0), tb = __builtin_amdgcn_readlane(tokr[qt], q0 + 4);
            const int q = 32 * qt + q0 + 4 * lhalf;
            const unsigned eoff = (unsigned)(lhalf ? tb : ta) * (unsigned)C + (unsigned)lrow;
            if (OBF) buf_store_bf16(oacc[r], rsO, (q < Wt) ? eoff * 2u : 0x7fffffffu, hoff >> 1);
            else buf_store1(oacc[r], rsO, (q < Wt) ? eoff * 4u : 0x7fffffffu, hoff);
        }
    }
}

bool launch_window_attn(const float *qkv, const float *biasT, float *out, int B, const WinGeom &g, hipStream_t s, bool out_bf16, bool in_bf16) {
    if (g.ws < 1 || g.res % g.ws != 0 || g.C != 32 * g.heads || B < 1) return false;
    const int nW = (g.res / g.ws) * (g.res / g.ws);
    const int n_units = B * nW * g.heads;
    const dim3 grid((n_units + 3) / 4), block(256);
    if (in_bf16 && !out_bf16) return false;   // bf16 qkv is only built with a bf16 output
#define WA(KT_, WS_)                                                                                                                \
    do {                                                                                                                            \
        if (in_bf16) DSG_LAUNCH((window_attn_kernel<KT_, WS_, true, true>), grid, block, 0, s, qkv, biasT, out, B, g, n_units);  \
        else if (out_bf16) DSG_LAUNCH((window_attn_kernel<KT_, WS_, true>), grid, block, 0, s, qkv, biasT, out, B, g, n_units);  \
        else DSG_LAUNCH((window_attn_kernel<KT_, WS_, false>), grid, block, 0, s, qkv, biasT, out, B, g, n_units);               \
    } while (0)
    switch (g.ws) {
        case 2: WA(1, 2); break;
        case 4: WA(1, 4); break;
        case 5: WA(1, 5); break;
        case 8: WA(2, 8); break;
        case 10: WA(4, 10); break;
        default: return false;  // (dsg_create only admits these window sizes)
    }
#undef WA
    return true;
}

// =================================================================================================
// Row kernels: one wave per token row, values held in registers between the passes.
// =================================================================================================
constexpr int ROW_MAXV = 24;  // rows up to 64*24 = 1536 channels (PatchBreakup at the deepest level)
constexpr int ROW_MAXV4 = 6;  // the same in float4 units per lane

// x <- silu(shift + x*(1+scale)); stats = LayerNorm statistics of the NEW row (norm1 runs on the
// modulated tensor, which is also the residual shortcut: diffusesg.py:238-243).  16 B per lane per access.
__global__ __launch_bounds__(256) void mod_stats_kernel(float *x, const float *aff, int aff_ld, int aff_off, float *stats,
                                                        int T, int C, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = m / T, C4 = C >> 2;
    const f32x4 *scale = reinterpret_cast<const f32x4 *>(aff + (size_t)b * aff_ld + aff_off), *shift = scale + C4;
    f32x4 *xr = reinterpret_cast<f32x4 *>(x + (size_t)m * C);
    f32x4 v[ROW_MAXV4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV4; i++) {
        const int c = lane + 64 * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C4) {
            const f32x4 xv = xr[c], sc = scale[c], sh = shift[c];
#pragma unroll
            for (int t = 0; t < 4; t++) { v[i][t] = silu_exact(sh[t] + xv[t] * (sc[t] + 1.0f)); sum += v[i][t]; }
            xr[c] = v[i];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV4; i++)
        if (lane + 64 * i < C4)
#pragma unroll
            for (int t = 0; t < 4; t++) { const float d = v[i][t] - mean; var += d * d; }
    var = wave_sum(var) / (float)C;
    if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = fast_rsqrt(var + LN_EPS); }
}
void launch_mod_stats(float *x, const float *aff, int aff_ld, int aff_off, float *stats, int B, int T, int C, hipStream_t s) {
    const int M = B * T;
    DSG_LAUNCH(mod_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, aff, aff_ld, aff_off, stats, T, C, M);
}

__global__ __launch_bounds__(256) void ln_stats_kernel(const float *x, float *stats, int C, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int C4 = C >> 2;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (size_t)m * C);
    f32x4 v[ROW_MAXV4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV4; i++) {
        const int c = lane + 64 * i;
        v[i] = (c < C4) ? xr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV4; i++)
        if (lane + 64 * i < C4)
#pragma unroll
            for (int t = 0; t < 4; t++) { const float d = v[i][t] - mean; var += d * d; }
    var = wave_sum(var) / (float)C;
    if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = fast_rsqrt(var + LN_EPS); }
}
void launch_ln_stats(const float *x, float *stats, int M, int C, hipStream_t s) {
    DSG_LAUNCH(ln_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, stats, C, M);
}

// PatchEmbed tail (diffusesg.py:570-576): y = silu(shift + LN(x)*(1+scale))
__global__ __launch_bounds__(256) void ln_mod_kernel(const float *x, const float *g, const float *bta, const float *aff,
                                                     int aff_ld, int aff_off, float *y, int T, int C, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = m / T;
    const float *scale = aff + (size_t)b * aff_ld + aff_off, *shift = scale + C;
    const float *xr = x + (size_t)m * C;
    float v[ROW_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV; i++) {
        const int c = lane + 64 * i;
        v[i] = (c < C) ? xr[c] : 0.f;
        sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < C) { const float d = v[i] - mean; var += d * d; }
    }
    const float rstd = fast_rsqrt(wave_sum(var) / (float)C + LN_EPS);
#pragma unroll
    for (int i = 0; i < ROW_MAXV; i++) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float n = (v[i] - mean) * rstd * g[c] + bta[c];
            y[(size_t)m * C + c] = silu_exact(shift[c] + n * (scale[c] + 1.0f));
        }
    }
}
void launch_ln_mod(const float *x, const float *g, const float *b, const float *aff, int aff_ld, int aff_off, float *y,
                   int B, int T, int C, hipStream_t s) {
    const int M = B * T;
    DSG_LAUNCH(ln_mod_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, aff, aff_ld, aff_off, y, T, C, M);
}

// PatchMerging (diffusesg.py:323-332): out row (i,j) = LN_4C(cat[x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)])
template <bool OBF>   // OBF: y is a bf16 tensor (the bf16 block pipeline's reduction GEMM reads bf16)
__global__ __launch_bounds__(256) void merge_ln_kernel(const float *x, const float *g, const float *bta, float *y, int res,
                                                       int C, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int r2 = res / 2, T2 = r2 * r2;
    const int b = m / T2, t = m % T2, i = t / r2, j = t % r2;
    const int C4 = C >> 2, D4 = C;  // float4 per source row / per output row (4C/4)
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x + (size_t)b * res * res * C);
    f32x4 v[ROW_MAXV4];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        v[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < D4) {
            const int part = c / C4, cc = c % C4;
            const int di = part & 1, dj = part >> 1;
            v[q] = xb[(size_t)((2 * i + di) * res + (2 * j + dj)) * C4 + cc];
            sum += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)(4 * C);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++)
        if (lane + 64 * q < D4)
#pragma unroll
            for (int tt = 0; tt < 4; tt++) { const float d = v[q][tt] - mean; var += d * d; }
    const float rstd = fast_rsqrt(wave_sum(var) / (float)(4 * C) + LN_EPS);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g), *b4 = reinterpret_cast<const f32x4 *>(bta);
    f32x4 *yr = reinterpret_cast<f32x4 *>(y + (size_t)m * 4 * C);
    u32x2_c *yb = reinterpret_cast<u32x2_c *>(reinterpret_cast<unsigned short *>(y) + (size_t)m * 4 * C);
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        if (c < D4) {
            const f32x4 o = (v[q] - mean) * rstd * g4[c] + b4[c];
            if (OBF) yb[c] = pack_bf16x4(o); else yr[c] = o;
        }
    }
}
void launch_merge_ln(const float *x, const float *g, const float *b, float *y, int B, int res, int C, hipStream_t s, bool out_bf16) {
    const int M = B * (res / 2) * (res / 2);
    if (out_bf16) DSG_LAUNCH(merge_ln_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, y, res, C, M);
    else DSG_LAUNCH(merge_ln_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, y, res, C, M);
}

// PatchBreakup middle (diffusesg.py:386-400): LN_D(row) -> chunk q -> token (2i+(q&1), 2j+(q>>1)) -> LN_{D/4}
template <bool OBF>   // OBF: z is a bf16 tensor (the bf16 block pipeline's post_linear GEMM reads bf16)
__global__ __launch_bounds__(256) void breakup_ln_kernel(const float *y, const float *g, const float *bta, const float *pg,
                                                         const float *pb, float *z, int res, int D, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int T = res * res, b = m / T, t = m % T, i = t / res, j = t % res;
    const int Co = D / 4, R = 2 * res, D4 = D >> 2, Co4 = Co >> 2;
    const f32x4 *yr = reinterpret_cast<const f32x4 *>(y + (size_t)m * D);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g), *b4 = reinterpret_cast<const f32x4 *>(bta);
    f32x4 v[ROW_MAXV4];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        v[q] = (c < D4) ? yr[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        sum += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
    const float mean = wave_sum(sum) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++)
        if (lane + 64 * q < D4)
#pragma unroll
            for (int tt = 0; tt < 4; tt++) { const float d = v[q][tt] - mean; var += d * d; }
    const float rstd = fast_rsqrt(wave_sum(var) / (float)D + LN_EPS);
    // per-chunk statistics of the normalised row (a float4 never straddles a chunk: Co % 4 == 0)
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        if (c < D4) {
            v[q] = (v[q] - mean) * rstd * g4[c] + b4[c];
            const int part = c / Co4;
            const float s4 = (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
#pragma unroll
            for (int p = 0; p < 4; p++) csum[p] += (part == p) ? s4 : 0.f;
        }
    }
    float cmean[4], cvar[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; p++) cmean[p] = wave_sum(csum[p]) / (float)Co;
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        if (c < D4) {
            const int part = c / Co4;
            float mu = cmean[0];
#pragma unroll
            for (int p = 1; p < 4; p++) if (part == p) mu = cmean[p];
            float d2 = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; tt++) { const float d = v[q][tt] - mu; d2 += d * d; }
#pragma unroll
            for (int p = 0; p < 4; p++) cvar[p] += (part == p) ? d2 : 0.f;
        }
    }
    float crstd[4];
#pragma unroll
    for (int p = 0; p < 4; p++) crstd[p] = fast_rsqrt(wave_sum(cvar[p]) / (float)Co + LN_EPS);
    const f32x4 *pg4 = reinterpret_cast<const f32x4 *>(pg), *pb4 = reinterpret_cast<const f32x4 *>(pb);
#pragma unroll
    for (int q = 0; q < ROW_MAXV4; q++) {
        const int c = lane + 64 * q;
        if (c < D4) {
            const int part = c / Co4, cc = c % Co4;
            const int di = part & 1, dj = part >> 1;
            float mu = cmean[0], rs = crstd[0];
#pragma unroll
            for (int p = 1; p < 4; p++) if (part == p) { mu = cmean[p]; rs = crstd[p]; }
            const size_t orow = (size_t)b * 4 * T + (size_t)(2 * i + di) * R + (2 * j + dj);
            const f32x4 o = (v[q] - mu) * rs * pg4[cc] + pb4[cc];
            if (OBF) reinterpret_cast<u32x2_c *>(reinterpret_cast<unsigned short *>(z) + orow * Co)[cc] = pack_bf16x4(o);
            else reinterpret_cast<f32x4 *>(z + orow * Co)[cc] = o;
        }
    }
}
void launch_breakup_ln(const float *y, const float *g, const float *b, const float *pg, const float *pb, float *z, int B,
                       int res, int D, hipStream_t s, bool out_bf16) {
    const int M = B * res * res;
    if (out_bf16) DSG_LAUNCH(breakup_ln_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, y, g, b, pg, pb, z, res, D, M);
    else DSG_LAUNCH(breakup_ln_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, y, g, b, pg, pb, z, res, D, M);
}

// PositionalEmbedding (diffusesg.py:507-513): freqs = (1/10000)^(k/(E/2)); [cos(x f), sin(x f)]
__global__ void noise_pe_kernel(const float *c_noise, float *pe, int B, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = E / 2;
    if (idx >= B * half) return;
    const int b = idx / half, k = idx % half;
    const float f = powf(1.0f / 10000.0f, (float)k / (float)half);
    const float v = c_noise[b] * f;
    pe[(size_t)b * E + k] = cosf(v);
    pe[(size_t)b * E + half + k] = sinf(v);
}
void launch_noise_pe(const float *c_noise, float *pe, int B, int E, hipStream_t s) {
    const int n = B * (E / 2);
    DSG_LAUNCH(noise_pe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, c_noise, pe, B, E);
}

// Input assembly (diffusesg.py:784-802), token-major with the K dim zero-padded to Kp.
// channel order: [sc_adj, adj, sc_node(i), node(i), sc_node(j), node(j)]; node part masked by flags[i]&flags[j]
__global__ void assemble_kernel(const float *adj, const float *node, const float *sc_adj, const float *sc_node,
                                const int *has_sc, const uint8_t *flags, float *out, int B, int N, int Ca, int Cn,
                                int self_cond, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * N * N * Kp;
    if (idx >= total) return;
    const int c = idx % Kp;
    const size_t tokg = idx / Kp;
    const int j = tokg % N, i = (tokg / N) % N, b = tokg / ((size_t)N * N);
    const bool sc_on = self_cond && (has_sc == nullptr || *has_sc != 0) && sc_adj != nullptr;
    const int nsc = self_cond ? 2 : 1;
    float v = 0.f;
    if (c < nsc * Ca) {
        const int which = self_cond ? c / Ca : 1, a = c % Ca;  // 0 = self-cond copy, 1 = current
        const float *src = (which == 0) ? (sc_on ? sc_adj : nullptr) : adj;
        if (src) v = src[(((size_t)b * Ca + a) * N + i) * N + j];
    } else if (c < nsc * (Ca + 2 * Cn)) {
        const int cc = c - nsc * Ca;
        const int colpart = cc / (nsc * Cn);  // 0: node_mat (row index i), 1: node_mat_t (column index j)
        const int c2 = cc % (nsc * Cn);
        const int which = self_cond ? c2 / Cn : 1, a = c2 % Cn;
        const int nodei = colpart ? j : i;
        const float *src = (which == 0) ? (sc_on ? sc_node : nullptr) : node;
        const bool m = flags[(size_t)b * N + i] && flags[(size_t)b * N + j];
        if (src && m) v = src[((size_t)b * N + nodei) * Cn + a];
    }
    out[idx] = v;
}
void launch_assemble(const float *adj, const float *node, const float *sc_adj, const float *sc_node, const int *has_sc,
                     const uint8_t *flags, float *out, int B, int N, int Ca, int Cn, int self_cond, int Kp, hipStream_t s) {
    const size_t total = (size_t)B * N * N * Kp;
    DSG_LAUNCH(assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, adj, node, sc_adj, sc_node,
                       has_sc, flags, out, B, N, Ca, Cn, self_cond, Kp);
}

// Adjacency head tail (diffusesg.py:806-809, :825): out[b][c][i][j] = mask * (h[tok] . W[c] + bias[c])
// one wave per token: lanes split the E-long dot product
__global__ __launch_bounds__(256) void head_adj_kernel(const float *h, const float *W, const float *bias,
                                                       const uint8_t *flags, float *out, int N, int E, int Ca, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int T = N * N, b = m / T, t = m % T, i = t / N, j = t % N;
    const bool ok = flags[(size_t)b * N + i] && flags[(size_t)b * N + j];
    const float *hr = h + (size_t)m * E;
    for (int c = 0; c < Ca; c++) {
        float acc = 0.f;
        if (ok)
            for (int e = lane; e < E; e += 64) acc += hr[e] * W[(size_t)c * E + e];
        acc = wave_sum(acc);
        if (lane == 0) out[(((size_t)b * Ca + c) * N + i) * N + j] = ok ? acc + bias[c] : 0.f;
    }
}
void launch_head_adj(const float *h, const float *W, const float *bias, const uint8_t *flags, float *out, int B, int N,
                     int E, int Ca, hipStream_t s) {
    const int M = B * N * N;
    DSG_LAUNCH(head_adj_kernel, dim3((M + 3) / 4), dim3(256), 0, s, h, W, bias, flags, out, N, E, Ca, M);
}

// Padding-aware pooling (diffusesg.py:812-813): mean over j of the row/col-masked rep, divided by N_pad
__global__ void pool_kernel(const float *rep, const uint8_t *flags, float *pool, int B, int N, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N * E) return;
    const int e = idx % E, i = (idx / E) % N, b = idx / (E * N);
    float acc = 0.f;
    if (flags[(size_t)b * N + i])
        for (int j = 0; j < N; j++)
            if (flags[(size_t)b * N + j]) acc += rep[(((size_t)b * N + i) * N + j) * E + e];
    pool[idx] = acc / (float)N;
}
void launch_pool(const float *rep, const uint8_t *flags, float *pool, int B, int N, int E, hipStream_t s) {
    const int n = B * N * E;
    DSG_LAUNCH(pool_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rep, flags, pool, B, N, E);
}

__global__ __launch_bounds__(256) void head_node_kernel(const float *h, const float *W, const float *bias,
                                                        const uint8_t *flags, float *out, int N, int E, int Cn, int M) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const bool ok = flags[m];
    const float *hr = h + (size_t)m * E;
    for (int c = 0; c < Cn; c++) {
        float acc = 0.f;
        if (ok)
            for (int e = lane; e < E; e += 64) acc += hr[e] * W[(size_t)c * E + e];
        acc = wave_sum(acc);
        if (lane == 0) out[(size_t)m * Cn + c] = ok ? acc + bias[c] : 0.f;
    }
}
void launch_head_node(const float *h, const float *W, const float *bias, const uint8_t *flags, float *out, int B, int N,
                      int E, int Cn, hipStream_t s) {
    const int M = B * N;
    DSG_LAUNCH(head_node_kernel, dim3((M + 3) / 4), dim3(256), 0, s, h, W, bias, flags, out, N, E, Cn, M);
}

// =================================================================================================
// Preconditioning and sampler elementwise kernels.  One launch covers the adjacency part
// [B,Ca,N,N] and the node part [B,N,Cn]; a flat index below n_adj addresses the former.
// The arithmetic mirrors torch's op-by-op rounding of the reference (no fused multiply-add where the
// reference has two rounded ops), see R/model/precond/precond.py:100-105, R/runner/mcmc_sampler/edm.py:355-427.
// =================================================================================================
struct ElemIdx { bool is_adj; size_t off; int b; bool valid; };

__device__ __forceinline__ ElemIdx elem_index(size_t idx, const Dims &d, const uint8_t *flags) {
    ElemIdx e;
    const size_t n_adj = (size_t)d.B * d.Ca * d.N * d.N;
    if (idx < n_adj) {
        e.is_adj = true; e.off = idx;
        const int j = idx % d.N, i = (idx / d.N) % d.N;
        e.b = idx / ((size_t)d.Ca * d.N * d.N);
        e.valid = flags ? (flags[(size_t)e.b * d.N + i] && flags[(size_t)e.b * d.N + j]) : true;
    } else {
        e.is_adj = false; e.off = idx - n_adj;
        const int i = (e.off / d.Cn) % d.N;
        e.b = e.off / ((size_t)d.N * d.Cn);
        e.valid = flags ? (bool)flags[(size_t)e.b * d.N + i] : true;
    }
    return e;
}
__host__ __device__ inline size_t total_elems(const Dims &d) { return (size_t)d.B * ((size_t)d.Ca * d.N * d.N + (size_t)d.N * d.Cn); }

#define FMUL(a, b) __fmul_rn((a), (b))
#define FADD(a, b) __fadd_rn((a), (b))
#define FSUB(a, b) __fsub_rn((a), (b))

__global__ void precond_in_kernel(CStatePtrs x, const float *sigmas, StatePtrs in, float *c_noise, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)d.B) c_noise[idx] = __fdiv_rn(logf(sigmas[idx]), 4.0f);  // objectives/edm.py:126
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, nullptr);
    const float s = sigmas[e.b];
    const float c_in = __fdiv_rn(1.0f, __fsqrt_rn(FADD(0.25f, FMUL(s, s))));       // objectives/edm.py:125
    if (e.is_adj) in.adj[e.off] = FMUL(c_in, x.adj[e.off]);
    else in.node[e.off] = FMUL(c_in, x.node[e.off]);
}
__global__ void cnoise_kernel(const float *sigmas, float *c_noise, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c_noise[i] = __fdiv_rn(logf(sigmas[i]), 4.0f);  // objectives/edm.py:126
}
void launch_cnoise(const float *sigmas, float *c_noise, int n, hipStream_t s) {
    DSG_LAUNCH(cnoise_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sigmas, c_noise, n);
}
void launch_precond_in(CStatePtrs x, const float *sigmas, StatePtrs in, float *c_noise, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(precond_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, sigmas, in, c_noise, d);
}

__global__ void precond_out_kernel(CStatePtrs x, CStatePtrs F, const float *sigmas, const uint8_t *flags, StatePtrs D,
                                   StatePtrs D2, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, flags);
    const float s = sigmas[e.b];
    const float s2 = FADD(FMUL(s, s), 0.25f);
    const float c_skip = __fdiv_rn(0.25f, s2);                                       // objectives/edm.py:123
    const float c_out = __fdiv_rn(FMUL(s, 0.5f), __fsqrt_rn(s2));                    // objectives/edm.py:124
    const float xv = e.is_adj ? x.adj[e.off] : x.node[e.off];
    const float fv = e.is_adj ? F.adj[e.off] : F.node[e.off];
    const float v = e.valid ? FADD(FMUL(c_skip, xv), FMUL(c_out, fv)) : 0.f;         // precond.py:102-105
    if (e.is_adj) { D.adj[e.off] = v; if (D2.adj) D2.adj[e.off] = v; }
    else { D.node[e.off] = v; if (D2.node) D2.node[e.off] = v; }
}
void launch_precond_out(CStatePtrs x, CStatePtrs F, const float *sigmas, const uint8_t *flags, StatePtrs D, StatePtrs D2,
                        Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(precond_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, F, sigmas, flags, D, D2, d);
}

// Philox4x32-10 counter-based generator -> one N(0,1) per (seed, stream, element)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream, 0x5eedu};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; r++) philox_round(c, k);
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ void init_kernel(CStatePtrs init, float scale, uint64_t seed, uint32_t stream, const uint8_t *flags, StatePtrs x, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, flags);
    float v;
    if (init.adj) v = e.is_adj ? init.adj[e.off] : init.node[e.off];
    else v = philox_normal(seed, stream, idx);                   // stream 0: gen_init_sample, edm.py:279,285
    v = e.valid ? FMUL(v, scale) : 0.f;                          // edm.py:346-347
    if (e.is_adj) x.adj[e.off] = v; else x.node[e.off] = v;
}
void launch_init(CStatePtrs init, float scale, uint64_t seed, uint32_t stream, const uint8_t *flags, StatePtrs x, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, init, scale, seed, stream, flags, x, d);
}

// ---- reverse-loop kernels: scalars from StepRow[ctl->step] (one captured step body serves every step; edm.py:355-427) ----
__global__ void churn_tab_kernel(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs xhat, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const int step = ctl->step;
    const float coef = tab[step].noise_coef;
    const ElemIdx e = elem_index(idx, d, flags);
    const float xv = e.is_adj ? x.adj[e.off] : x.node[e.off];
    float eps;
    if (ctl->noise_adj) {
        const size_t sa = (size_t)d.B * d.Ca * d.N * d.N, sn = (size_t)d.B * d.N * d.Cn;
        eps = e.is_adj ? ctl->noise_adj[(size_t)step * sa + e.off] : ctl->noise_node[(size_t)step * sn + e.off];
    } else eps = (coef != 0.f && e.valid) ? philox_normal(ctl->seed, (uint32_t)step + 1u, idx) : 0.f;
    const float v = e.valid ? FADD(xv, FMUL(coef, eps)) : 0.f;  // edm.py:361-366
    if (e.is_adj) xhat.adj[e.off] = v; else xhat.node[e.off] = v;
}
void launch_churn_tab(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs xhat, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(churn_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, tab, ctl, flags, xhat, d);
}
__global__ void precond_in_tab_kernel(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, StatePtrs in, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, nullptr);
    const float s = tab[ctl->step].sigma;
    const float c_in = __fdiv_rn(1.0f, __fsqrt_rn(FADD(0.25f, FMUL(s, s))));       // objectives/edm.py:125
    if (e.is_adj) in.adj[e.off] = FMUL(c_in, x.adj[e.off]);
    else in.node[e.off] = FMUL(c_in, x.node[e.off]);
}
void launch_precond_in_tab(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, StatePtrs in, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(precond_in_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, tab, ctl, in, d);
}
__global__ void precond_out_tab_kernel(CStatePtrs x, CStatePtrs F, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags,
                                       StatePtrs D, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, flags);
    const float s = tab[ctl->step].sigma;
    const float s2 = FADD(FMUL(s, s), 0.25f);
    const float c_skip = __fdiv_rn(0.25f, s2);                                       // objectives/edm.py:123
    const float c_out = __fdiv_rn(FMUL(s, 0.5f), __fsqrt_rn(s2));                    // objectives/edm.py:124
    const float xv = e.is_adj ? x.adj[e.off] : x.node[e.off];
    const float fv = e.is_adj ? F.adj[e.off] : F.node[e.off];
    const float v = e.valid ? FADD(FMUL(c_skip, xv), FMUL(c_out, fv)) : 0.f;         // precond.py:102-105
    if (e.is_adj) D.adj[e.off] = v; else D.node[e.off] = v;
}
void launch_precond_out_tab(CStatePtrs x, CStatePtrs F, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs D, Dims d,
                            hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(precond_out_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, F, tab, ctl, flags, D, d);
}
__global__ void euler_tab_kernel(CStatePtrs xhat, CStatePtrs D, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs x,
                                 Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const StepRow r = tab[ctl->step];
    const ElemIdx e = elem_index(idx, d, flags);
    const float xh = e.is_adj ? xhat.adj[e.off] : xhat.node[e.off];
    const float dn = e.is_adj ? D.adj[e.off] : D.node[e.off];
    const float dc = FSUB(FMUL(r.inv_t, xh), FMUL(r.inv_t, dn));     // edm.py:384-385
    const float v = e.valid ? FADD(xh, FMUL(r.h, dc)) : 0.f;         // edm.py:395-396, :421-422
    if (e.is_adj) x.adj[e.off] = v; else x.node[e.off] = v;
}
void launch_euler_tab(CStatePtrs xhat, CStatePtrs D, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs x, Dims d,
                      hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(euler_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xhat, D, tab, ctl, flags, x, d);
}
__global__ void heun_tab_kernel(CStatePtrs xhat, CStatePtrs D1, CStatePtrs D2, const StepRow *tab, const RunCtl *ctl,
                                const uint8_t *flags, StatePtrs x, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_elems(d)) return;
    const StepRow r = tab[ctl->step];
    const ElemIdx e = elem_index(idx, d, flags);
    const float xh = e.is_adj ? xhat.adj[e.off] : xhat.node[e.off];
    const float d1 = e.is_adj ? D1.adj[e.off] : D1.node[e.off];
    const float d2 = e.is_adj ? D2.adj[e.off] : D2.node[e.off];
    const float dc = FSUB(FMUL(r.inv_t, xh), FMUL(r.inv_t, d1));        // edm.py:384-385
    const float xp = FADD(xh, FMUL(r.h, dc));                           // edm.py:389-390 (alpha = 1)
    const float dp = FSUB(FMUL(r.inv_tp, xp), FMUL(r.inv_tp, d2));      // edm.py:414-417
    const float avg = FADD(FMUL(0.5f, dc), FMUL(0.5f, dp));
    const float v = e.valid ? FADD(xh, FMUL(r.h, avg)) : 0.f;           // edm.py:418-422
    if (e.is_adj) x.adj[e.off] = v; else x.node[e.off] = v;
}
void launch_heun_tab(CStatePtrs xhat, CStatePtrs D1, CStatePtrs D2, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags,
                     StatePtrs x, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(heun_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xhat, D1, D2, tab, ctl, flags, x, d);
}
__global__ void step_row_kernel(const float *table, int n, const RunCtl *ctl, float *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = table[(size_t)ctl->step * n + i];
}
void launch_step_row(const float *table, int n, const RunCtl *ctl, float *dst, hipStream_t s) {
    DSG_LAUNCH(step_row_kernel, dim3((n + 255) / 256), dim3(256), 0, s, table, n, ctl, dst);
}
__global__ void step_advance_kernel(RunCtl *ctl) { ctl->step += 1; }
void launch_step_advance(RunCtl *ctl, hipStream_t s) { DSG_LAUNCH(step_advance_kernel, dim3(1), dim3(1), 0, s, ctl); }

// Training-time objective (forward): R/runner/objectives/edm.py:160-180 (sigma ~ exp(N(P_mean, P_std)), loss weight) and
// :239-281 / graph_utils.add_sym_normal_noise with non_symmetric=True (noisy inputs).  Op-by-op fp32 rounding like torch.
__global__ void train_inputs_kernel(CStatePtrs clean, const float *rnd, CStatePtrs eps, uint64_t seed, const uint8_t *flags,
                                    float *sigmas, float *weights, StatePtrs noisy, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto sigma_of = [&](int b) {
        const float r = rnd ? rnd[b] : philox_normal(seed, 0x7000u, (uint64_t)b);
        return expf(FADD(FMUL(r, 1.2f), -1.2f));                                        // edm.py:176-177
    };
    if (idx < (size_t)d.B) {
        const float s = sigma_of((int)idx), sd = FMUL(s, 0.5f);
        sigmas[idx] = s;
        weights[idx] = __fdiv_rn(FADD(FMUL(s, s), 0.25f), FMUL(sd, sd));                // edm.py:178
    }
    if (idx >= total_elems(d)) return;
    const ElemIdx e = elem_index(idx, d, flags);
    const float s = sigma_of(e.b);
    const float ev = eps.adj ? (e.is_adj ? eps.adj[e.off] : eps.node[e.off]) : philox_normal(seed, 0x7001u, idx);
    const float nz = FMUL(ev, s);
    if (e.is_adj) noisy.adj[e.off] = e.valid ? FADD(clean.adj[e.off], nz) : 0.f;        // graph_utils.py:139-144
    else noisy.node[e.off] = FADD(clean.node[e.off], e.valid ? nz : 0.f);               // edm.py:249-256
}
void launch_train_inputs(CStatePtrs clean, const float *rnd, CStatePtrs eps, uint64_t seed, const uint8_t *flags, float *sigmas,
                         float *weights, StatePtrs noisy, Dims d, hipStream_t s) {
    const size_t n = total_elems(d);
    DSG_LAUNCH(train_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, clean, rnd, eps, seed, flags, sigmas,
                       weights, noisy, d);
}

// The bounding-box term of one node (R/runner/trainer/trainer_node_adj.py:130-153): both boxes go (x+1)/2 -> cxcywh -> xyxy -> clamp[0,1],
// then by iou_loss_type
//   0 'iou'          -(box_iou)^2                         torchvision.ops.box_iou: inter / (area_a + area_b - inter), overlap clamp(min=0)
//   1 'giou'         generalized_box_iou_loss             1 - iou + (area_c - union) / (area_c + eps)
//   2 'giou_squared' (that)^2
//   3 'diou'         distance_box_iou_loss                1 - iou + centre distance^2 / (enclosing diagonal^2 + eps)
//   4 'ciou'         complete_box_iou_loss                diou + alpha v, v = 4/pi^2 (atan(w/h) - atan(wg/hg))^2, alpha = v/(1 - iou + v + eps) (no grad)
// (types 1-4: torchvision.ops' *_box_iou_loss with reduction='none', eps = 1e-7, iou = inter / (union + eps), intersection only where
// both overlaps are open; restated from torchvision's published source -- torchvision itself is not installed here.)
// Forward value in fp32, operation by operation; the gradient with respect to the four raw (cx, cy, w, h)-space channels follows
// autograd: clamp passes it inside [0,1] inclusive, max/min to the selected argument (a tie splits it evenly), in double.
struct BoxTerm { float loss; double g[4]; };
__device__ BoxTerm box_iou_term(const float *pred4, const float *tgt4, int type, bool want_grad) {
    float bx[2][4], raw[4];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const float *src = q ? tgt4 : pred4;
        const float cx = __fdiv_rn(FADD(src[0], 1.0f), 2.0f), cy = __fdiv_rn(FADD(src[1], 1.0f), 2.0f);
        const float bw = __fdiv_rn(FADD(src[2], 1.0f), 2.0f), bh = __fdiv_rn(FADD(src[3], 1.0f), 2.0f);
        const float v[4] = {FSUB(cx, FMUL(0.5f, bw)), FSUB(cy, FMUL(0.5f, bh)), FADD(cx, FMUL(0.5f, bw)), FADD(cy, FMUL(0.5f, bh))};
#pragma unroll
        for (int t = 0; t < 4; t++) { bx[q][t] = fminf(fmaxf(v[t], 0.0f), 1.0f); if (!q) raw[t] = v[t]; }
    }
    const float *p = bx[0], *g = bx[1];
    BoxTerm out;
    out.g[0] = out.g[1] = out.g[2] = out.g[3] = 0.0;
    double gc[4] = {0.0, 0.0, 0.0, 0.0};   // d loss / d clamped corner of the prediction
    const float pw = FSUB(p[2], p[0]), ph = FSUB(p[3], p[1]), gw = FSUB(g[2], g[0]), gh = FSUB(g[3], g[1]);
    const float a0 = FMUL(pw, ph), a1 = FMUL(gw, gh);
    const float dwf = FSUB(fminf(p[2], g[2]), fmaxf(p[0], g[0])), dhf = FSUB(fminf(p[3], g[3]), fmaxf(p[1], g[1]));
    // selection weights of max / min (1 selected, 0.5 tie, 0 not): lo = the prediction's corner is the larger lower corner, ...
    auto sel_gt = [](float a, float b) { return a > b ? 1.0 : (a == b ? 0.5 : 0.0); };
    if (type == 0) {
        const float iw = fmaxf(dwf, 0.0f), ih = fmaxf(dhf, 0.0f);
        const float inter = FMUL(iw, ih), iou = __fdiv_rn(inter, FSUB(FADD(a0, a1), inter));
        out.loss = -FMUL(iou, iou);
        if (want_grad) {
            const double dw = dwf, dh = dhf, I = (double)iw * ih, U = (double)a0 + a1 - I, io = I / U;
            const double g_iou = -2.0 * io, g_I = g_iou * (U + I) / (U * U), g_A = g_iou * (-I) / (U * U);
            gc[0] = g_I * ((dw >= 0 && p[0] > g[0]) ? -(double)ih : 0.0) + g_A * (-(double)ph);
            gc[1] = g_I * ((dh >= 0 && p[1] > g[1]) ? -(double)iw : 0.0) + g_A * (-(double)pw);
            gc[2] = g_I * ((dw >= 0 && p[2] < g[2]) ? (double)ih : 0.0) + g_A * (double)ph;
            gc[3] = g_I * ((dh >= 0 && p[3] < g[3]) ? (double)iw : 0.0) + g_A * (double)pw;
        }
    } else {
        const float eps = 1e-7f;
        const bool open = dhf > 0.0f && dwf > 0.0f;
        const float inter = open ? FMUL(dwf, dhf) : 0.0f;
        const float uni = FSUB(FADD(a0, a1), inter), iou = __fdiv_rn(inter, FADD(uni, eps));
        const float cw = FSUB(fmaxf(p[2], g[2]), fminf(p[0], g[0])), ch = FSUB(fmaxf(p[3], g[3]), fminf(p[1], g[1]));
        // d inter, d area_p, d union, d iou per clamped corner
        double dI[4] = {0, 0, 0, 0};
        if (open) {
            dI[0] = -sel_gt(p[0], g[0]) * (double)dhf; dI[1] = -sel_gt(p[1], g[1]) * (double)dwf;
            dI[2] = sel_gt(g[2], p[2]) * (double)dhf;  dI[3] = sel_gt(g[3], p[3]) * (double)dwf;
        }
        const double dA[4] = {-(double)ph, -(double)pw, (double)ph, (double)pw};
        const double Ue = (double)uni + (double)eps;
        double dU[4], dIoU[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { dU[t] = dA[t] - dI[t]; dIoU[t] = (dI[t] * Ue - (double)inter * dU[t]) / (Ue * Ue); }
        // enclosing box: d cw / d corner, d ch / d corner
        const double dcw[4] = {-sel_gt(g[0], p[0]), 0.0, sel_gt(p[2], g[2]), 0.0};
        const double dch[4] = {0.0, -sel_gt(g[1], p[1]), 0.0, sel_gt(p[3], g[3])};
        if (type == 1 || type == 2) {
            const float area_c = FMUL(cw, ch);
            const float miou = FSUB(iou, __fdiv_rn(FSUB(area_c, uni), FADD(area_c, eps)));
            const float l = FSUB(1.0f, miou);
            out.loss = type == 2 ? FMUL(l, l) : l;
            if (want_grad) {
                const double Ace = (double)area_c + (double)eps, scale = type == 2 ? 2.0 * (double)l : 1.0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double dAc = dcw[t] * (double)ch + dch[t] * (double)cw;
                    const double dfrac = ((dAc - dU[t]) * Ace - ((double)area_c - (double)uni) * dAc) / (Ace * Ace);
                    gc[t] = scale * (-dIoU[t] + dfrac);
                }
            }
        } else {
            const float diag2 = FADD(FADD(FMUL(cw, cw), FMUL(ch, ch)), eps);
            const float xp = __fdiv_rn(FADD(p[2], p[0]), 2.0f), yp = __fdiv_rn(FADD(p[3], p[1]), 2.0f);
            const float xg = __fdiv_rn(FADD(g[0], g[2]), 2.0f), yg = __fdiv_rn(FADD(g[1], g[3]), 2.0f);
            const float dx = FSUB(xp, xg), dy = FSUB(yp, yg);
            const float cd2 = FADD(FMUL(dx, dx), FMUL(dy, dy));
            float l = FADD(FSUB(1.0f, iou), __fdiv_rn(cd2, diag2));
            float v = 0.f, alpha = 0.f, dat = 0.f;
            if (type == 4) {
                dat = FSUB(atanf(__fdiv_rn(pw, ph)), atanf(__fdiv_rn(gw, gh)));
                v = FMUL(__fdiv_rn(4.0f, FMUL(3.14159265358979323846f, 3.14159265358979323846f)), FMUL(dat, dat));
                alpha = __fdiv_rn(v, FADD(FADD(FSUB(1.0f, iou), v), eps));
                l = FADD(l, FMUL(alpha, v));
            }
            out.loss = l;
            if (want_grad) {
                const double D2 = diag2;
                const double dcd2[4] = {(double)dx, (double)dy, (double)dx, (double)dy};
                const double kv = type == 4 ? (double)alpha * (8.0 / (M_PI * M_PI)) * (double)dat / ((double)ph * ph + (double)pw * pw) : 0.0;
                const double dv[4] = {-kv * (double)ph, kv * (double)pw, kv * (double)ph, -kv * (double)pw};   // dv/dw = k h, dv/dh = -k w
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double dD2 = 2.0 * (double)cw * dcw[t] + 2.0 * (double)ch * dch[t];
                    gc[t] = -dIoU[t] + (dcd2[t] * D2 - (double)cd2 * dD2) / (D2 * D2) + dv[t];
                }
            }
        }
    }
    if (want_grad) {
#pragma unroll
        for (int t = 0; t < 4; t++) if (!(raw[t] >= 0.0f && raw[t] <= 1.0f)) gc[t] = 0.0;   // clamp(min=0, max=1)
        // corners = (cx - w/2, cy - h/2, cx + w/2, cy + h/2), (cx, cy, w, h) = (x + 1)/2
        out.g[0] = 0.5 * (gc[0] + gc[2]); out.g[1] = 0.5 * (gc[1] + gc[3]);
        out.g[2] = 0.25 * (gc[2] - gc[0]); out.g[3] = 0.25 * (gc[3] - gc[1]);
    }
    return out;
}

// NodeAdjRainbowLoss.forward(reduction='none') (R/loss/rainbow_loss.py:37-101) and the bbox term of the trainer
// (R/runner/trainer/trainer_node_adj.py:130-159, every iou_loss_type: box_iou_term above).  One 256-thread block per sample; every
// thread sums a fixed strided subset in double, the block adds the 256 partials in a fixed tree: deterministic.
__global__ __launch_bounds__(256) void rainbow_loss_kernel(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w,
                                                           float edge_w, float node_w, float iou_w, int iou_type, float *loss_adj,
                                                           float *loss_node, Dims d) {
    __shared__ double red[3][256];
    __shared__ int cnt[2];
    const int b = blockIdx.x, tid = threadIdx.x, N = d.N, Ca = d.Ca, Cn = d.Cn;
    const uint8_t *f = flags + (size_t)b * N;
    if (tid == 0) {
        int n = 0, nt = 0;
        for (int i = 0; i < N; i++) n += f[i] ? 1 : 0;
        for (int k = 0; k < d.B * N; k++) nt += flags[k] ? 1 : 0;
        cnt[0] = n; cnt[1] = nt;
    }
    const float wb = w ? w[b] : 1.0f;
    double sa = 0.0, sn = 0.0, si = 0.0;
    const size_t na = (size_t)Ca * N * N;
    for (size_t k = tid; k < na; k += 256) {
        const int j = k % N, i = (k / N) % N;
        if (f[i] && f[j]) { const float dv = pred.adj[(size_t)b * na + k] - tgt.adj[(size_t)b * na + k]; sa += (double)(dv * dv * wb); }
    }
    const size_t nn = (size_t)N * Cn;
    for (size_t k = tid; k < nn; k += 256) {
        const int i = k / Cn;
        if (f[i]) { const float dv = pred.node[(size_t)b * nn + k] - tgt.node[(size_t)b * nn + k]; sn += (double)(dv * dv * wb); }
    }
    if (iou_w != 0.f)
        for (int i = tid; i < N; i += 256)
            if (f[i]) {
                const size_t o = ((size_t)b * N + i) * Cn + (Cn - 4);
                si += (double)box_iou_term(pred.node + o, tgt.node + o, iou_type, false).loss;
            }
    red[0][tid] = sa; red[1][tid] = sn; red[2][tid] = si;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double n = (double)cnt[0];
        loss_adj[b] = (float)(red[0][0] / (n * n) / (double)Ca) * edge_w;
        loss_node[b] = (float)(red[1][0] / n / (double)Cn) * node_w + (iou_w != 0.f ? iou_w * (float)(red[2][0] / (double)cnt[1]) * wb : 0.f);
    }
}
void launch_rainbow_loss(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w, float edge_w, float node_w, float iou_w,
                         int iou_type, float *loss_adj, float *loss_node, Dims d, hipStream_t s) {
    DSG_LAUNCH(rainbow_loss_kernel, dim3(d.B), dim3(256), 0, s, pred, tgt, flags, w, edge_w, node_w, iou_w, iou_type, loss_adj,
                       loss_node, d);
}

// Backward of  loss = mean_b(loss_adj) + mean_b(loss_node)  (trainer_node_adj.py:163) with respect to the preconditioned
// outputs D, and with respect to the raw network outputs F:  D = mask(c_skip x + c_out F)  (precond.py:101-104)  =>
// dL/dF = c_out(sigma_b) dL/dD.  The bbox term follows autograd (box_iou_term).
// One block per sample; the first stage of the training backward, checked against the reference's autograd.
__global__ __launch_bounds__(256) void rainbow_loss_backward_kernel(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w,
                                                                    float edge_w, float node_w, float iou_w, int iou_type,
                                                                    const float *sigmas, StatePtrs grad, StatePtrs gradF, Dims d) {
    __shared__ int cnt[2];
    const int b = blockIdx.x, tid = threadIdx.x, N = d.N, Ca = d.Ca, Cn = d.Cn;
    const uint8_t *f = flags + (size_t)b * N;
    if (tid == 0) {
        int n = 0, nt = 0;
        for (int i = 0; i < N; i++) n += f[i] ? 1 : 0;
        for (int k = 0; k < d.B * N; k++) nt += flags[k] ? 1 : 0;
        cnt[0] = n; cnt[1] = nt;
    }
    __syncthreads();
    const double n = (double)cnt[0], wb = w ? (double)w[b] : 1.0, invB = 1.0 / (double)d.B;
    const double ka = (double)edge_w * 2.0 * wb / (n * n * (double)Ca) * invB;
    const double kn = (double)node_w * 2.0 * wb / (n * (double)Cn) * invB;
    const double ki = (double)iou_w * wb / (double)cnt[1] * invB;
    float c_out = 1.0f;
    if (sigmas) { const float sg = sigmas[b], sd = 0.5f; c_out = __fdiv_rn(FMUL(sg, sd), __fsqrt_rn(FADD(FMUL(sg, sg), FMUL(sd, sd)))); }
    const size_t na = (size_t)Ca * N * N;
    for (size_t k = tid; k < na; k += 256) {
        const int j = k % N, i = (k / N) % N;
        const float g = (f[i] && f[j]) ? (float)(ka * (double)FSUB(pred.adj[(size_t)b * na + k], tgt.adj[(size_t)b * na + k])) : 0.0f;
        grad.adj[(size_t)b * na + k] = g;
        if (gradF.adj) gradF.adj[(size_t)b * na + k] = FMUL(c_out, g);
    }
    for (int i = tid; i < N; i += 256) {
        double gb[4] = {0.0, 0.0, 0.0, 0.0};
        const size_t row = ((size_t)b * N + i) * Cn;
        if (f[i] && iou_w != 0.f) {
            const BoxTerm bt = box_iou_term(pred.node + row + (Cn - 4), tgt.node + row + (Cn - 4), iou_type, true);
#pragma unroll
            for (int t = 0; t < 4; t++) gb[t] = ki * bt.g[t];
        }
        for (int c = 0; c < Cn; c++) {
            double g = f[i] ? kn * (double)FSUB(pred.node[row + c], tgt.node[row + c]) : 0.0;
            if (f[i] && c >= Cn - 4) g += gb[c - (Cn - 4)];
            grad.node[row + c] = (float)g;
            if (gradF.node) gradF.node[row + c] = FMUL(c_out, (float)g);
        }
    }
}
void launch_rainbow_loss_backward(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w, float edge_w, float node_w,
                                  float iou_w, int iou_type, const float *sigmas, StatePtrs grad, StatePtrs gradF, Dims d, hipStream_t s) {
    DSG_LAUNCH(rainbow_loss_backward_kernel, dim3(d.B), dim3(256), 0, s, pred, tgt, flags, w, edge_w, node_w, iou_w, iou_type, sigmas,
                       grad, gradF, d);
}

// Post-decode of the samples (sampler_node_adj.py:222-285), one thread per adjacency entry / node, for the three encodings of
// `--node_encoding` / `--edge_encoding` (R/utils/attribute_code.py):
//   bits    (enc 0; bin2dec :319-328): value > 0 -> bit 1, channel 0 is the MSB; clamp to [0, n_type-1]
//   one_hot (enc 1; attribute_one_hot_to_int :212-237 after the +-1 threshold of :225 / :245): the FIRST channel whose value is > 0
//           (torch.argmax over 0/1 entries returns the first maximum), 0 when none is
//   ddpm    (enc 2; attribute_ddpm_to_int :121-177): the single channel clamped to [-1, 1] is assigned the class i whose interval
//           (lo_i, hi_i] holds it, lo/hi = center_i -+ L/2 with L = 2/(k-1), center_i = -1 + i L formed in DOUBLE by the reference's
//           Python floats and compared in fp32 (torch casts the scalar to the tensor's dtype): the same double operations are issued
//           here with the correctly rounded intrinsics; classes are tried in ascending order and the last match wins, like the
//           reference's loop; a NaN matches nothing and stays -1, the reference's fill value
// masked entries and the adjacency diagonal are 0; bbox = node[..., -4:]*0.5+0.5 masked (sampler_node_adj.py:201-209)
__device__ __forceinline__ int ddpm_class(float x, int k) {
    if (x != x) return -1;
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    const double L = __ddiv_rn(2.0, (double)(k - 1)), half = __dmul_rn(L, 0.5);
    const int guess = (int)rintf((x + 1.0f) * 0.5f * (float)(k - 1));
    int out = -1;
    for (int i = max(guess - 2, 0); i <= min(guess + 2, k - 1); i++) {
        const double center = __dadd_rn(-1.0, __dmul_rn((double)i, L));
        const float lo = i == 0 ? -INFINITY : __double2float_rn(__dsub_rn(center, half));
        const float hi = i == k - 1 ? INFINITY : __double2float_rn(__dadd_rn(center, half));
        if (x > lo && x <= hi) out = i;
    }
    return out;
}
__global__ void decode_kernel(const float *adj, const float *node, const uint8_t *flags, int enc_adj, int enc_node, int n_adj_type,
                              int n_node_type, int node_chans, int32_t *out_adj, int32_t *out_node, float *out_bbox, Dims d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_a = (size_t)d.B * d.N * d.N, n_n = (size_t)d.B * d.N;
    if (idx < n_a) {
        const int j = idx % d.N, i = (idx / d.N) % d.N, b = idx / ((size_t)d.N * d.N);
        int v = 0;
        if (flags[(size_t)b * d.N + i] && flags[(size_t)b * d.N + j] && i != j) {
            const float *a = adj + ((size_t)b * d.Ca * d.N + i) * d.N + j;
            const size_t cs = (size_t)d.N * d.N;
            if (enc_adj == 0) {
                for (int c = 0; c < d.Ca; c++) v = (v << 1) | (a[c * cs] > 0.f ? 1 : 0);
                v = min(max(v, 0), n_adj_type - 1);
            } else if (enc_adj == 1) {
                for (int c = d.Ca - 1; c >= 0; c--) if (a[c * cs] > 0.f) v = c;
            } else v = ddpm_class(a[0], n_adj_type);
        }
        out_adj[idx] = v;
    } else if (idx < n_a + n_n) {
        const size_t m = idx - n_a;
        const bool ok = flags[m];
        int v = 0;
        if (ok) {
            const float *x = node + m * d.Cn;
            if (enc_node == 0) {
                for (int c = 0; c < node_chans; c++) v = (v << 1) | (x[c] > 0.f ? 1 : 0);
                v = min(max(v, 0), n_node_type - 1);
            } else if (enc_node == 1) {
                for (int c = node_chans - 1; c >= 0; c--) if (x[c] > 0.f) v = c;
            } else v = ddpm_class(x[0], n_node_type);
        }
        out_node[m] = v;
        if (out_bbox)
            for (int c = 0; c < 4; c++) out_bbox[m * 4 + c] = ok ? node[m * d.Cn + (d.Cn - 4) + c] * 0.5f + 0.5f : 0.f;
    }
}
void launch_decode(const float *adj, const float *node, const uint8_t *flags, int enc_adj, int enc_node, int n_adj_type, int n_node_type,
                   int node_chans, int32_t *out_adj, int32_t *out_node, float *out_bbox, Dims d, hipStream_t s) {
    const size_t n = (size_t)d.B * d.N * d.N + (size_t)d.B * d.N;
    DSG_LAUNCH(decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, adj, node, flags, enc_adj, enc_node, n_adj_type,
               n_node_type, node_chans, out_adj, out_node, out_bbox, d);
}

}  // namespace dsg
