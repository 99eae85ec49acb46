// train_kernels.hip -- training-time kernels of one SwinTransformerBlock (second half, first piece): the block's
// forward in the form that keeps what the backward needs, and its backward.  R = DiffuseSG/ of the reference.
//
//   forward   R/model/diffusesg/diffusesg.py:232-277 (block), :108-139 (WindowAttention), :19-25 (Mlp)
//   backward  what torch.autograd derives for those lines; checked against the reference's own autograd
//             (tests/golden/block_backward.npz, tools/gen_golden.py::gen_block_backward)
//
// CORRECTNESS FIRST: these are straightforward fp32 kernels (an LDS-tiled FMA GEMM, one thread per element / row / window row;
// only the activation-side GEMMs are forwarded to the sampling path's MFMA kernel) -- they exist so that every gradient formula is
// pinned to the reference before the fast versions are written.  Nothing on the sampling path calls into this file.
#include "kernels_common.hip.h"

#include <stdio.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace dsg {

// ---------------------------------------------------------------------------------------------------------------------
// Scratch of the training kernels (split-K partials, transposed weights, column-sum partials): one set PER STREAM.  Work
// on a stream is serialised, so launches that share a set never overlap; two streams (or two handles driven from two
// streams) get two sets.  Buffers only grow; a failed allocation marks the stream's set (t_scratch_failed) and the launch
// that needed it is skipped -- the C entry point then returns DSG_ERR_HIP instead of producing garbage.
// ---------------------------------------------------------------------------------------------------------------------
struct TScratch {
    float *sk = nullptr;  size_t sk_cap = 0;    // split-K partial products / transposed slices
    float *wt = nullptr;  size_t wt_cap = 0;    // transposed weight of a dx product
    double *cs = nullptr; size_t cs_cap = 0;    // column-sum / modulate-gradient partials
    void *opt = nullptr;  size_t opt_cap = 0;   // optimiser step: tensor table + partial sums
    bool failed = false;
};
static std::mutex g_ts_mutex;
static std::map<hipStream_t, TScratch> g_ts;
static TScratch &t_scratch(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ts_mutex);
    return g_ts[s];
}
template <typename T>
static T *t_scratch_get(hipStream_t s, T *&buf, size_t &cap, size_t need, TScratch &ts) {
    if (need > cap) {
        // the previous buffer may still be read by work queued on this stream: drain it before freeing
        if (buf) { (void)hipStreamSynchronize(s); (void)hipFree(buf); }
        buf = nullptr; cap = 0;
        if (hipMalloc((void **)&buf, sizeof(T) * need) == hipSuccess) cap = need;
        else { buf = nullptr; ts.failed = true; }
    }
    return buf;
}
bool t_scratch_failed(hipStream_t s, bool clear) {
    TScratch &ts = t_scratch(s);
    const bool f = ts.failed;
    if (clear) ts.failed = false;
    return f;
}
void t_scratch_release() {   // dsg_destroy of the last handle / tests
    std::lock_guard<std::mutex> lk(g_ts_mutex);
    for (auto &kv : g_ts) { (void)hipFree(kv.second.sk); (void)hipFree(kv.second.wt); (void)hipFree(kv.second.cs); (void)hipFree(kv.second.opt); }
    g_ts.clear();
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] (+)= op(A) op(B) (+ bias[n]);  op(A)(m,k) = TA ? A[k*lda+m] : A[m*lda+k];  op(B)(k,n) = TB ? B[n*ldb+k] : B[k*ldb+n]
// 32x32 tile per 256-thread block, 4 outputs per thread, k in chunks of 32, fixed summation order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
// the k loop of one 32 x 32 tile: acc[i] += sum_k op(A)(m0 + ty + 8 i, k) op(B)(k, n0 + tx), k in [kbeg, kend)
template <bool TA, bool TB>
__device__ __forceinline__ void t_gemm_accum(float (&acc)[4], const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, int M, int N,
                                             int kbeg, int K, int m0, int n0, float (*As)[33], float (*Bs)[33]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
    for (int k0 = kbeg; k0 < K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = ty + 8 * i;   // tile row
            {   // As[r][tx] = op(A)(m0 + r, k0 + tx)
                const int m = m0 + r, k = k0 + tx;
                float v = 0.f;
                if (TA) { const int mm = m0 + tx, kk = k0 + r; v = (mm < M && kk < K) ? A[(size_t)kk * lda + mm] : 0.f; As[tx][r] = v; }
                else { v = (m < M && k < K) ? A[(size_t)m * lda + k] : 0.f; As[r][tx] = v; }
            }
            {   // Bs[r][tx] = op(B)(k0 + r, n0 + tx)
                float v = 0.f;
                if (TB) { const int nn = n0 + r, kk = k0 + tx; v = (nn < N && kk < K) ? B[(size_t)nn * ldb + kk] : 0.f; Bs[tx][r] = v; }
                else { const int kk = k0 + r, nn = n0 + tx; v = (kk < K && nn < N) ? B[(size_t)kk * ldb + nn] : 0.f; Bs[r][tx] = v; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; kk++) {
            const float b = Bs[kk][tx];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = fmaf(As[ty + 8 * i][kk], b, acc[i]);
        }
        __syncthreads();
    }
}
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void t_gemm_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                     const float *__restrict__ bias, float *__restrict__ C, int ldc, int M, int N, int K,
                                                     int accumulate, int kslice) {
    // kslice > 0: split-K -- block z handles k in [z kslice, (z+1) kslice) and writes its partial product to C + z M N (ldc = N)
    __shared__ float As[32][33], Bs[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int kbeg = kslice > 0 ? blockIdx.z * kslice : 0;
    if (kslice > 0) { C += (size_t)blockIdx.z * M * N; K = min(K, kbeg + kslice); }
    t_gemm_accum<TA, TB>(acc, A, lda, B, ldb, M, N, kbeg, K, m0, n0, As, Bs);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + ty + 8 * i, n = n0 + tx;
        if (m < M && n < N) {
            float v = acc[i] + (bias ? bias[n] : 0.f);
            if (accumulate) v += C[(size_t)m * ldc + n];
            C[(size_t)m * ldc + n] = v;
        }
    }
}
// Grouped small products (the 2C-wide `affine` linears of all blocks hang off ONE [B, 512] embedding: 13-19 products of a few MFLOP
// each, forward and backward -- one launch per kind instead of one per block).
//   SUM = false: problem z = blockIdx.z writes its own C_z = op(A_z) op(B_z) (+ bias_z)
//   SUM = true : ONE output C = sum_z op(A_z) op(B_z) (all problems share M, N and C; k runs through the problems in order)
template <bool TA, bool TB, bool SUM>
__global__ __launch_bounds__(256) void t_gemm_grouped_kernel(TGemmGroup g) {
    __shared__ float As[32][33], Bs[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (SUM) {
        for (int z = 0; z < g.n; z++) {
            const TGemmProb &p = g.p[z];
            t_gemm_accum<TA, TB>(acc, p.A, p.lda, p.B, p.ldb, p.M, p.N, 0, p.K, m0, n0, As, Bs);
        }
    } else {
        const TGemmProb &p = g.p[blockIdx.z];
        if (m0 >= p.M || n0 >= p.N) return;   // block-uniform
        t_gemm_accum<TA, TB>(acc, p.A, p.lda, p.B, p.ldb, p.M, p.N, 0, p.K, m0, n0, As, Bs);
    }
    const TGemmProb &p = g.p[SUM ? 0 : blockIdx.z];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + ty + 8 * i, n = n0 + tx;
        if (m < p.M && n < p.N) p.C[(size_t)m * p.ldc + n] = acc[i] + (p.bias ? p.bias[n] : 0.f);
    }
}
// out[i] = sum_z C_z[i] over the group's (contiguous, equally sized M x N) outputs, in order
__global__ __launch_bounds__(256) void t_sum_grouped_kernel(TGemmGroup g, float *out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int z = 0; z < g.n; z++) v += g.p[z].C[i];
    out[i] = v;
}
// out_z[n] = sum_m X_z[m][n] for a group of short matrices (M <= a few hundred rows: the bias gradients of the affine linears)
__global__ __launch_bounds__(256) void t_colsum_grouped_kernel(TGemmGroup g) {
    const TGemmProb &p = g.p[blockIdx.y];   // A = X [M, N] (lda), C = out [N]
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.N) return;
    double sacc = 0.0;
    for (int m = 0; m < p.M; m++) sacc += (double)p.A[(size_t)m * p.lda + n];
    p.C[n] = (float)sacc;
}
// C[i] = (accumulate ? C[i] : 0) + sum_z part[z][i]   (fixed order); the blocks behind the M N elements add the slices of the
// column sums that gemm_tn_f32_kernel<true> wrote next to its partial products: cs_out[m] = sum_z cs_part[z][m]
// ZL slice lanes per output (a block = 256 / ZL consecutive outputs x ZL lanes): lane q adds slices q, q + ZL, ... in order, then the ZL lane
// sums are added in order -- a small weight's 256 slices are no longer walked by one thread (round 3: 36 blocks for a 96 x 96 gradient,
// 256 dependent strided reads per thread, 25 us a call)
template <int ZL>
__global__ __launch_bounds__(256) void t_splitk_reduce_kernel(const float *part, float *C, int ldc, int M, int N, int S, int accumulate, const float *cs_part, float *cs_out) {
    constexpr int OUT = 256 / ZL;
    __shared__ float red[ZL][OUT + 1];
    const int ol = threadIdx.x % OUT, q = threadIdx.x / OUT;
    const size_t mn = (size_t)M * N, n_blk = (mn + OUT - 1) / OUT;
    const bool cs = blockIdx.x >= n_blk;                               // the blocks behind the M N outputs: the column sums
    const size_t i = cs ? (size_t)(blockIdx.x - n_blk) * OUT + ol : (size_t)blockIdx.x * OUT + ol;
    const size_t lim = cs ? (size_t)M : mn;
    const float *src = cs ? cs_part : part;
    float v = 0.f;
    if (i < lim) for (int z = q; z < S; z += ZL) v += src[(size_t)z * lim + i];
    red[q][ol] = v;
    __syncthreads();
    if (q == 0 && i < lim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < ZL; k++) t += red[k][ol];
        if (cs) cs_out[i] = t;
        else { const size_t m = i / N, n = i % N; C[m * ldc + n] = (accumulate ? C[m * ldc + n] : 0.f) + t; }
    }
}
static void t_splitk_reduce(const float *part, float *C, int ldc, int M, int N, int S, bool accumulate, const float *cs_part, float *cs_out, hipStream_t s) {
    const size_t mn = (size_t)M * N;
    // slice lanes so that the launch has >= ~1024 blocks' worth of threads where the output is small and the slices are many
    const int zl = (S >= 64 && mn < ((size_t)1 << 17)) ? 16 : ((S >= 16 && mn < ((size_t)1 << 19)) ? 4 : 1);
    const int out = 256 / zl;
    const unsigned rb = (unsigned)((mn + out - 1) / out) + (cs_out ? (unsigned)((M + out - 1) / out) : 0u);
    if (zl == 16) hipLaunchKernelGGL((t_splitk_reduce_kernel<16>), dim3(rb), dim3(256), 0, s, part, C, ldc, M, N, S, (int)accumulate, cs_part, cs_out);
    else if (zl == 4) hipLaunchKernelGGL((t_splitk_reduce_kernel<4>), dim3(rb), dim3(256), 0, s, part, C, ldc, M, N, S, (int)accumulate, cs_part, cs_out);
    else hipLaunchKernelGGL((t_splitk_reduce_kernel<1>), dim3(rb), dim3(256), 0, s, part, C, ldc, M, N, S, (int)accumulate, cs_part, cs_out);
}
// [R][Cc] -> [Cc][R] (weights on their way into the MFMA GEMM, which wants both operands K-contiguous)
__global__ __launch_bounds__(256) void t_transpose_kernel(const float *src, int ld, float *dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = (r0 + i < R && c0 + tx < Cc) ? src[(size_t)(r0 + i) * ld + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}
// src [R][ld] (R = K tokens, Cc channels used) -> dst [S][Cc][kslice]: slice z holds rows z*kslice .. of src, transposed, zero-padded
__global__ __launch_bounds__(256) void t_transpose_sliced_kernel(const float *src, int ld, float *dst, int R, int Cc, int kslice) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int z = blockIdx.z, c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;   // k0: offset inside the slice
    for (int i = ty; i < 32; i += 8) {
        const int r = z * kslice + k0 + i;
        tile[i][tx] = (k0 + i < kslice && r < R && c0 + tx < Cc) ? src[(size_t)r * ld + c0 + tx] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && k0 + tx < kslice) dst[((size_t)z * Cc + c0 + i) * kslice + k0 + tx] = tile[tx][i];
}
// W [N][K] (pitch ld) -> [N][Kp] with zero columns K .. Kp - 1 (t_gemm: a K that is not a multiple of the GEMM's 32-deep chunk)
__global__ void t_pad_cols_kernel(const float *__restrict__ W, int ld, float *__restrict__ out, int N, int K, int Kp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    out[i] = k < K ? W[(size_t)n * ld + k] : 0.f;
}
// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients on the f32 matrix pipe without transposes:  C[M,N] = sum_k A[k][m] B[k][n]  (dW = dy^T x: A = dy [tokens, out],
// B = x [tokens, in], both as the forward left them, K = tokens).  v_mfma_f32_32x32x2_f32 takes ONE value per lane and operand
// (lane l: A[row l&31][k l>>5], B[k l>>5][col l&31]), so with K-major tiles in LDS ([k][m] / [k][n]) a fragment is one
// conflict-free ds_read_b32 of 32 consecutive floats per k -- the operand layout the token-major activations already have.
// Block tile 128 (m) x 96 (n), wave tile 32 x 96, k chunks of 32, register-staged double buffer; split-K over blockIdx (slice z
// writes its partial product to Cp + z M N, added in slice order afterwards: deterministic).  CS: the block column tn == 0 also
// leaves the column sums of its A slice, sum_k A[k][m] (the bias gradient db = colsum(dy)), in cs_part[z][m].
// ---------------------------------------------------------------------------------------------------------------------
template <bool CS>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                             float *__restrict__ Cp, int M, int N, int K, int kslice, int tiles_m, int tiles_n,
                                                             float *__restrict__ cs_part) {
    constexpr int BM = 128, BN = 96, BK = 32;
    __shared__ __attribute__((aligned(16))) float lds[2][BK * (BM + BN)];
    const int tiles = tiles_m * tiles_n;
    const int z = blockIdx.x / tiles, t = blockIdx.x % tiles, tm = t / tiles_n, tn = t % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kb = z * kslice, ke = min(K, kb + kslice);
    const int nk = (ke - kb + BK - 1) / BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    // rows [kb, ke) of A / B; rows past the slice and columns past M / N read as zero
    const rsrc_t rsA = make_rsrc(A + (size_t)kb * lda, (unsigned)(ke - kb) * lda * 4u);
    const rsrc_t rsB = make_rsrc(B + (size_t)kb * ldb, (unsigned)(ke - kb) * ldb * 4u);
    unsigned voffA[4], voffB[3];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int idx = tid + 256 * p, kr = idx >> 5, c4 = idx & 31;
        voffA[p] = (m0 + 4 * c4 < M) ? ((unsigned)kr * lda + (unsigned)(m0 + 4 * c4)) * 4u : 0x7fffffffu;
    }
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int idx = tid + 256 * p, kr = idx / 24, c4 = idx % 24;
        voffB[p] = (n0 + 4 * c4 < N) ? ((unsigned)kr * ldb + (unsigned)(n0 + 4 * c4)) * 4u : 0x7fffffffu;
    }
    f32x4 sa[4], sb[3];
    auto issue = [&](int kc) {
#pragma unroll
        for (int p = 0; p < 4; p++) sa[p] = buf_load4(rsA, voffA[p], (unsigned)kc * BK * lda * 4u);
#pragma unroll
        for (int p = 0; p < 3; p++) sb[p] = buf_load4(rsB, voffB[p], (unsigned)kc * BK * ldb * 4u);
    };
    auto write = [&](int buf) {
        float *As = lds[buf], *Bs = As + BK * BM;
#pragma unroll
        for (int p = 0; p < 4; p++) { const int idx = tid + 256 * p; *reinterpret_cast<f32x4 *>(As + (idx >> 5) * BM + 4 * (idx & 31)) = sa[p]; }
#pragma unroll
        for (int p = 0; p < 3; p++) { const int idx = tid + 256 * p; *reinterpret_cast<f32x4 *>(Bs + (idx / 24) * BN + 4 * (idx % 24)) = sb[p]; }
    };
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float asum = 0.f;
    if (nk > 0) {
        issue(0);
        write(0);
        __syncthreads();
    }
    for (int kc = 0; kc < nk; kc++) {
        const int cur = kc & 1;
        if (kc + 1 < nk) issue(kc + 1);
        const float *As = lds[cur] + lhalf * BM + wave * 32 + lrow, *Bs = lds[cur] + BK * BM + lhalf * BN + lrow;
#pragma unroll
        for (int s2 = 0; s2 < BK / 2; s2++) {
            const float a = As[2 * s2 * BM];
            if (CS) asum += a;
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[2 * s2 * BN + 32 * j], acc[j], 0, 0, 0);
        }
        if (kc + 1 < nk) write(1 - cur);
        __syncthreads();
    }
    // partial product of this slice: rows m0 + 32 wave + (r&3) + 8 (r>>2) + 4 half, columns n0 + 32 j + lrow
    float *Cz = Cp + (size_t)z * M * N;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int n = n0 + 32 * j + lrow;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            if (m < M && n < N) Cz[(size_t)m * N + n] = acc[j][r];
        }
    }
    if (CS && tn == 0) {
        asum += __shfl_xor(asum, 32, 64);
        const int m = m0 + 32 * wave + lrow;
        if (lhalf == 0 && m < M) cs_part[(size_t)z * M + m] = asum;
    }
}

void t_gemm(bool ta, bool tb, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K,
            bool accumulate, hipStream_t s, float *a_colsum, const float *res, int act, float *c2) {
    // res (activation-side products only): C = res + product, res a tensor of C's shape and pitch -- the MFMA kernel reads it in its
    // epilogue; the fallbacks copy it into C first and accumulate
    if (act == ACT_NONE && res && res == C) { res = nullptr; accumulate = true; }
    TScratch &ts = t_scratch(s);
    // the fallbacks below form the plain product; the fused activation forms are finished by an elementwise pass afterwards
    struct ActAfter { int act; const float *res; float *C, *c2; size_t n; hipStream_t s; bool done;
        ~ActAfter() {
            if (done || act == ACT_NONE) return;
            if (act == ACT_GELU_KEEP) t_gelu(c2, nullptr, C, n, false, s); else t_gelu(res, C, C, n, true, s);
        } } act_after{act, res, C, c2, (size_t)M * N, s, false};
    float *C_plain = (act == ACT_GELU_KEEP) ? c2 : C;   // where the fallbacks put the product
    bool colsum_done = false;
    struct ColsumAfter { const float *A; int lda, M, K; float *out; hipStream_t s; bool *done; bool ta;
        ~ColsumAfter() { if (out && !*done) t_colsum(A, lda, out, K, M, s); } } colsum_after{A, lda, M, K, a_colsum, s, &colsum_done, ta};
    (void)colsum_after;   // a_colsum (dW products only: out[m] = sum_k A[k][m]) is produced by the MFMA kernel below or, on the fallbacks, by t_colsum
    // The activation-side products y = x W^T and dx = dy W have the sampling path's GEMM form (A [M,K] row-major, second operand
    // [N,K]): large ones go to gemm4_f32_kernel (dx after transposing the weight into a scratch tile).  The weight-gradient
    // products (ta: K = tokens) and everything small or oddly shaped stay on the plain kernel below.
    static const bool use_mfma = getenv("DSG_TRAIN_PLAIN_GEMM") == nullptr;
    // (K not a multiple of the 32-deep chunk -- PatchEmbed's 60 input channels: the weight goes into the scratch tile zero-padded to Kp
    // columns; the activation rows keep their pitch, the chunk's overhang reads the next row's first values (finite) against those zeros,
    // and past the last row the buffer descriptor returns zeros)
    const int Kp = (K + 31) / 32 * 32;
    if (use_mfma && !ta && M >= 512 && N % 32 == 0 && lda == K && (K % 32 == 0 || (tb && lda % 4 == 0 && !res && !accumulate)) && (size_t)N * Kp <= ((size_t)32 << 20)) {
        const float *Wop = B;
        bool ok = true;
        if (tb && K % 32 != 0) {
            float *wt = t_scratch_get(s, ts.wt, ts.wt_cap, (size_t)N * Kp, ts);
            ok = wt != nullptr;
            if (ok) {
                hipLaunchKernelGGL(t_pad_cols_kernel, dim3((unsigned)(((size_t)N * Kp + 255) / 256)), dim3(256), 0, s, B, ldb, wt, N, K, Kp);
                Wop = wt;
            }
        } else if (tb) ok = ldb == K;
        else {
            float *wt = t_scratch_get(s, ts.wt, ts.wt_cap, (size_t)N * K, ts);
            ok = wt != nullptr;
            if (ok) {
                hipLaunchKernelGGL(t_transpose_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, s, B, ldb, wt, K, N);   // B [K][N] -> [N][K]
                Wop = wt;
            }
        }
        if (ok) {
            GemmArgs g;
            g.A = A; g.lda = K; g.K1 = Kp; g.K = Kp; g.M = M; g.N = N; g.W = Wop; g.bias = bias; g.C = C; g.ldc = ldc;
            if (res) { g.res = res; g.ldres = ldc; }
            else if (accumulate) { g.res = C; g.ldres = ldc; }
            g.act = act;
            if (act == ACT_GELU_KEEP) { g.C2 = c2; g.ldc2 = ldc; }
            if (!launch_gemm(g, s)) ts.failed = true;   // reported by the entry point as DSG_ERR_HIP / INVALID (t_scratch_failed)
            act_after.done = true;
            return;
        }
    }
    if (act != ACT_NONE && (ta || ldc != N || accumulate)) { ts.failed = true; act_after.done = true; return; }   // (activation-side products only)
    if (res && act == ACT_NONE) {
        if (hipMemcpy2DAsync(C, sizeof(float) * ldc, res, sizeof(float) * ldc, sizeof(float) * N, M, hipMemcpyDeviceToDevice, s) != hipSuccess) { ts.failed = true; return; }
        accumulate = true;
    }
    C = C_plain;
    // Weight gradients dW [M = out, N = in] = dy^T x with K = tokens: gemm_tn_f32_kernel reads both token-major operands as they
    // are (no transposed copies), S slices of K in one launch, partial products added in slice order.
    // (M, N, lda, ldb need not be multiples of 4: a 16-byte buffer load wants dword alignment only, what a row's last load reads of the
    // next row lands in rows / columns >= M / N, which the epilogue drops, and past the slice the descriptor returns zeros -- the
    // adjacency head's 6-wide layer ran on the plain kernel with 64 blocks for K = 262144)
    if (use_mfma && ta && !tb && !bias && K >= 2048) {
        const int tiles_m = (M + 127) / 128, tiles_n = (N + 95) / 96;
        int S = std::max(1, std::min(256, 1024 / (tiles_m * tiles_n)));   // ~1024 blocks: two rounds of the chip's 512 slots
        S = std::max(1, std::min(S, K / 512));
        const int kslice = ((K + S - 1) / S + 31) / 32 * 32;
        S = (K + kslice - 1) / kslice;
        const size_t nC = (size_t)S * M * N, nS = a_colsum ? (size_t)S * M : 0;
        float *buf = t_scratch_get(s, ts.sk, ts.sk_cap, nC + nS, ts);
        if (buf) {
            const dim3 grid((unsigned)(tiles_m * tiles_n * S));
            if (a_colsum) hipLaunchKernelGGL((gemm_tn_f32_kernel<true>), grid, dim3(256), 0, s, A, lda, B, ldb, buf, M, N, K, kslice, tiles_m, tiles_n, buf + nC);
            else hipLaunchKernelGGL((gemm_tn_f32_kernel<false>), grid, dim3(256), 0, s, A, lda, B, ldb, buf, M, N, K, kslice, tiles_m, tiles_n, nullptr);
            t_splitk_reduce(buf, C, ldc, M, N, S, accumulate, buf + nC, a_colsum, s);
            if (a_colsum) colsum_done = true;
            return;
        }
    }
    static const bool log_plain = getenv("DSG_TGEMM_LOG") != nullptr;   // dev: which products still run on the plain kernel
    if (log_plain) fprintf(stderr, "t_gemm plain: ta %d tb %d M %d N %d K %d lda %d ldb %d ldc %d bias %d acc %d act %d\n", (int)ta, (int)tb, M, N, K, lda, ldb, ldc, bias != nullptr, (int)accumulate, act);
    dim3 grid((N + 31) / 32, (M + 31) / 32), block(256);
    // weight gradients (K = tokens, M x N = the weight): few output tiles, long K -> split K over up to 64 slices of >= 1024 and add
    // the partial products in slice order (deterministic)
    int S = 1, kslice = 0;
    float *Cout = C;
    int ldo = ldc;
    if (K >= 4096 && !bias && (size_t)grid.x * grid.y < 1024) {
        S = min(64, K / 1024);
        kslice = ((K + S - 1) / S + 31) / 32 * 32;
        S = (K + kslice - 1) / kslice;
        float *buf = t_scratch_get(s, ts.sk, ts.sk_cap, (size_t)S * M * N, ts);
        if (buf) { Cout = buf; ldo = N; grid.z = S; } else { S = 1; kslice = 0; }
    }
    const int acc1 = S > 1 ? 0 : (int)accumulate;
    if (!ta && !tb) hipLaunchKernelGGL((t_gemm_kernel<false, false>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else if (!ta && tb) hipLaunchKernelGGL((t_gemm_kernel<false, true>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else if (ta && !tb) hipLaunchKernelGGL((t_gemm_kernel<true, false>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else hipLaunchKernelGGL((t_gemm_kernel<true, true>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    if (S > 1) t_splitk_reduce(ts.sk, C, ldc, M, N, S, accumulate, nullptr, nullptr, s);
}

void t_gemm_grouped(bool ta, bool tb, bool sum, const TGemmGroup &g, hipStream_t s) {
    if (g.n < 1) return;
    int mx_m = 0, mx_n = 0;
    for (int z = 0; z < g.n; z++) { mx_m = std::max(mx_m, g.p[z].M); mx_n = std::max(mx_n, g.p[z].N); }
    const dim3 grid((mx_n + 31) / 32, (mx_m + 31) / 32, sum ? 1 : g.n), block(256);
    if (!ta && tb && !sum) hipLaunchKernelGGL((t_gemm_grouped_kernel<false, true, false>), grid, block, 0, s, g);
    else if (ta && !tb && !sum) hipLaunchKernelGGL((t_gemm_grouped_kernel<true, false, false>), grid, block, 0, s, g);
    else if (!ta && !tb && sum) hipLaunchKernelGGL((t_gemm_grouped_kernel<false, false, true>), grid, block, 0, s, g);
    else if (!ta && !tb && !sum) hipLaunchKernelGGL((t_gemm_grouped_kernel<false, false, false>), grid, block, 0, s, g);
    else t_scratch(s).failed = true;   // (no other form is used)
}
void t_sum_grouped(const TGemmGroup &g, float *out, int n, hipStream_t s) {
    if (g.n < 1) return;
    hipLaunchKernelGGL(t_sum_grouped_kernel, dim3((n + 255) / 256), dim3(256), 0, s, g, out, n);
}
void t_colsum_grouped(const TGemmGroup &g, hipStream_t s) {
    if (g.n < 1) return;
    int mx_n = 0;
    for (int z = 0; z < g.n; z++) mx_n = std::max(mx_n, g.p[z].N);
    hipLaunchKernelGGL(t_colsum_grouped_kernel, dim3((mx_n + 255) / 256, g.n), dim3(256), 0, s, g);
}

// out[n] = sum_m X[m*ld + n]: row chunks summed by separate blocks (double, fixed order inside a chunk), then the chunks in order.
// Stage 1: block = 64 columns x 4 row lanes over one chunk of rows; stage 2: block = 64 columns x 4 lanes over the chunks.
__global__ __launch_bounds__(256) void t_colsum_part_kernel(const float *X, int ld, double *part, int M, int N, int rows_per) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    double sacc = 0.0;
    if (c < N) for (int m = r0 + q; m < r1; m += 4) sacc += (double)X[(size_t)m * ld + c];
    red[q][threadIdx.x & 63] = sacc;
    __syncthreads();
    if (q == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void t_colsum_final_kernel(const double *part, float *out, int N, int R) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    double sacc = 0.0;
    if (c < N) for (int r = q; r < R; r += 4) sacc += part[(size_t)r * N + c];
    red[q][threadIdx.x & 63] = sacc;
    __syncthreads();
    if (q == 0 && c < N) out[c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
void t_colsum(const float *X, int ld, float *out, int M, int N, hipStream_t s) {
    // ~2048 blocks in stage 1 (the pass is HBM-bound: it needs the whole chip), at least 32 rows and at most 512 chunks
    const int colblocks = (N + 63) / 64;
    int R = std::max(1, std::min(512, 2048 / colblocks));
    R = std::max(1, std::min(R, (M + 31) / 32));
    const int rows_per = (M + R - 1) / R;
    R = (M + rows_per - 1) / rows_per;
    TScratch &ts = t_scratch(s);
    double *buf = t_scratch_get(s, ts.cs, ts.cs_cap, (size_t)R * N, ts);
    if (!buf) return;   // (the stream's scratch is marked failed)
    hipLaunchKernelGGL(t_colsum_part_kernel, dim3(colblocks, R), dim3(256), 0, s, X, ld, buf, M, N, rows_per);
    hipLaunchKernelGGL(t_colsum_final_kernel, dim3(colblocks), dim3(256), 0, s, buf, out, N, R);
}

__device__ __forceinline__ float t_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// y = silu(shift_b + x (1 + scale_b))   (diffusesg.py:238-240); aff [B][2C] = (scale | shift)
__global__ void t_modulate_fwd_kernel(const float *x, const float *aff, float *y, int T, int C, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % C;
    const size_t b = i / ((size_t)T * C);
    const float u = aff[b * 2 * C + C + c] + x[i] * (aff[b * 2 * C + c] + 1.0f);
    y[i] = u * t_sigmoid(u);
}
// dx = du (1 + scale), du = dy silu'(u); d_aff[b] = (sum_t du x | sum_t du).  Stage 1: block (64 channels x 4 row lanes) over a chunk of
// sample b's tokens writes dx and double partials part[b][chunk][2][C]; stage 2 adds the chunks in order.
__global__ __launch_bounds__(256) void t_modulate_bwd_kernel(const float *x, const float *aff, const float *dy, float *dx, double *part, int T, int C,
                                                              int rows_per, int chunks) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6, b = blockIdx.z, ch = blockIdx.y;
    const int t0 = ch * rows_per, t1 = min(T, t0 + rows_per);
    double ds = 0.0, dh = 0.0;
    if (c < C) {
        const float sc = aff[(size_t)b * 2 * C + c] + 1.0f, sh = aff[(size_t)b * 2 * C + C + c];
        for (int t = t0 + q; t < t1; t += 4) {
            const size_t k = ((size_t)b * T + t) * C + c;
            const float xv = x[k], u = sh + xv * sc, sg = t_sigmoid(u);
            const float du = dy[k] * (sg * (1.0f + u * (1.0f - sg)));
            dx[k] = du * sc;
            ds += (double)(du * xv);
            dh += (double)du;
        }
    }
    red[0][q][threadIdx.x & 63] = ds; red[1][q][threadIdx.x & 63] = dh;
    __syncthreads();
    if (q == 0 && c < C) {
        const int l = threadIdx.x;
        double *p = part + (((size_t)b * chunks + ch) * 2) * C;
        p[c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        p[C + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}
__global__ void t_modulate_bwd_final_kernel(const double *part, float *d_aff, int B, int C, int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * C) return;
    const int b = i / (2 * C), j = i % (2 * C);
    double sacc = 0.0;
    for (int ch = 0; ch < chunks; ch++) sacc += part[(((size_t)b * chunks + ch) * 2) * C + j];
    d_aff[i] = (float)sacc;
}

// LayerNorm with affine: one wave per row (lanes stride over the channels: coalesced), two-pass statistics.  stats[m] = (mean, rstd)
// LayerNorm forward, the row read ONCE with 16-byte accesses: LPR lanes share a row (32 for C <= 128, else 64), each with KV float4s
// (float4 index sub + LPR k), so a wave covers 64 / LPR rows and every load / store instruction moves whole 16-byte pieces along rows
// (round 3 / early round 4: one wave per row with 4-byte accesses -- a C = 96 row kept 1.5 of 64 lanes' worth busy per load).
// With aff != null the row is modulated first -- u = shift + x (1 + scale), x_mod = u sigmoid(u) (diffusesg.py:238-243), written to y_mod:
// the block's shortcut and the tensor LayerNorm-1 normalises -- so the modulate pass and its second trip over [M, C] are gone.
template <int LPR, int KV>
__global__ __launch_bounds__(256) void t_ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ aff, float *__restrict__ y_mod,
                                                        const float *__restrict__ gam, const float *__restrict__ bet, float *__restrict__ y,
                                                        float *__restrict__ stats, int M, int C, int T) {
    constexpr int RW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int m = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + lane / LPR;
    const bool ok = m < M;
    const int C4 = C >> 2;
    const f32x4 *r = reinterpret_cast<const f32x4 *>(x + (size_t)(ok ? m : M - 1) * C);
    const f32x4 *af = aff ? reinterpret_cast<const f32x4 *>(aff + (size_t)((ok ? m : M - 1) / T) * 2 * C) : nullptr;
    f32x4 v[KV];
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < KV; k++) {
        const int q = sub + LPR * k;
        v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < C4) {
            v[k] = r[q];
            if (af) {
                const f32x4 sc = af[q], sh = af[C4 + q];
#pragma unroll
                for (int e = 0; e < 4; e++) { const float u = sh[e] + v[k][e] * (sc[e] + 1.0f); v[k][e] = u * t_sigmoid(u); }
                if (ok) reinterpret_cast<f32x4 *>(y_mod + (size_t)m * C)[q] = v[k];
            }
            sacc += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    const float mean = sacc / (float)C;
    float qv = 0.f;
#pragma unroll
    for (int k = 0; k < KV; k++)
        if (sub + LPR * k < C4) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = v[k][e] - mean; qv = fmaf(d, d, qv); }
        }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) qv += __shfl_xor(qv, o, 64);
    const float rstd = 1.0f / sqrtf(qv / (float)C + LN_EPS);
    if (!ok) return;
#pragma unroll
    for (int k = 0; k < KV; k++) {
        const int q = sub + LPR * k;
        if (q < C4) {
            const f32x4 g4 = reinterpret_cast<const f32x4 *>(gam)[q], b4 = reinterpret_cast<const f32x4 *>(bet)[q];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (v[k][e] - mean) * rstd * g4[e] + b4[e];
            reinterpret_cast<f32x4 *>(y + (size_t)m * C)[q] = o;
        }
    }
    if (sub == 0) { stats[2 * (size_t)m] = mean; stats[2 * (size_t)m + 1] = rstd; }
}
// x (+ modulate with aff [B, 2C], T tokens per sample -> y_mod) -> LayerNorm -> y, stats
static void t_ln_fwd_launch(const float *x, const float *aff, float *y_mod, const float *gam, const float *bet, float *y, float *stats, int M, int C, int T,
                            hipStream_t s) {
#define T_LNF(LPR_, KV_) hipLaunchKernelGGL((t_ln_fwd_kernel<LPR_, KV_>), dim3((M + 4 * (64 / LPR_) - 1) / (4 * (64 / LPR_))), dim3(256), 0, s, x, aff, y_mod, gam, bet, y, stats, M, C, T)
    if (C % 4 != 0 || C > 1536) { t_scratch(s).failed = true; return; }   // (every width of the networks is a multiple of 96; 4 x 384 is the widest LayerNorm)
    if (C <= 128) T_LNF(32, 1); else if (C <= 256) T_LNF(64, 1); else if (C <= 512) T_LNF(64, 2); else if (C <= 768) T_LNF(64, 3); else T_LNF(64, 6);
#undef T_LNF
}
// dx_out = (dx_in ? dx_in : 0) + rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;  d_gamma = sum_m dy xhat, d_beta = sum_m dy.
// Same lane layout as the forward (LPR lanes per row, KV float4s each, 16-byte accesses).  A block of 4 waves walks a chunk of rows;
// every lane keeps the two column sums of its 4 KV channels in registers (fp32 over the rows of its slot), the block adds its
// 4 x (64 / LPR) slots in order and writes part[block][2][C] (double); t_ln_bwd_final_kernel adds the blocks in order.
template <int LPR, int KV>
__global__ __launch_bounds__(256) void t_ln_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gam, const float *__restrict__ stats,
                                                        const float *__restrict__ dy, const float *dx_in, float *dx_out, double *part, int M, int C,
                                                        int rows_per) {
    constexpr int RW = 64 / LPR, SLOTS = 4 * RW, CP = LPR * KV * 4;   // CP: padded channel count
    __shared__ float red[SLOTS][2][CP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane % LPR, slot = wave * RW + lane / LPR;
    const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
    const int C4 = C >> 2;
    f32x4 ag[KV], ab[KV], gm[KV];
#pragma unroll
    for (int k = 0; k < KV; k++) {
        ag[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab[k] = ag[k];
        gm[k] = (sub + LPR * k < C4) ? reinterpret_cast<const f32x4 *>(gam)[sub + LPR * k] : ag[k];
    }
    for (int mb = r0; mb < r1; mb += SLOTS) {     // every lane takes part in the shuffles: rows beyond r1 run with zero weight
        const int m = mb + slot;
        const bool ok = m < r1;
        const size_t mc = (size_t)(ok ? m : r1 - 1);
        const float mean = stats[2 * mc], rstd = stats[2 * mc + 1];
        const f32x4 *r = reinterpret_cast<const f32x4 *>(x + mc * C), *d = reinterpret_cast<const f32x4 *>(dy + mc * C);
        f32x4 xh[KV], gg[KV];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < KV; k++) {
            const int q = sub + LPR * k;
            xh[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; gg[k] = xh[k];
            if (q < C4) {
                const f32x4 dv = d[q], xv = r[q];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    xh[k][e] = (xv[e] - mean) * rstd; gg[k][e] = dv[e] * gm[k][e];
                    sg += gg[k][e]; sgx = fmaf(gg[k][e], xh[k][e], sgx);
                    if (ok) { ag[k][e] = fmaf(dv[e], xh[k][e], ag[k][e]); ab[k][e] += dv[e]; }
                }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
        const float mg = sg / (float)C, mgx = sgx / (float)C;
        if (ok) {
#pragma unroll
            for (int k = 0; k < KV; k++) {
                const int q = sub + LPR * k;
                if (q < C4) {
                    f32x4 o4 = dx_in ? reinterpret_cast<const f32x4 *>(dx_in + mc * C)[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; e++) o4[e] += rstd * (gg[k][e] - mg - xh[k][e] * mgx);
                    reinterpret_cast<f32x4 *>(dx_out + mc * C)[q] = o4;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KV; k++) {
        *reinterpret_cast<f32x4 *>(&red[slot][0][4 * (sub + LPR * k)]) = ag[k];
        *reinterpret_cast<f32x4 *>(&red[slot][1][4 * (sub + LPR * k)]) = ab[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double *p = part + (size_t)blockIdx.x * 2 * C;
        double a0 = 0.0, b0 = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) { a0 += (double)red[sl][0][c]; b0 += (double)red[sl][1][c]; }
        p[c] = a0; p[C + c] = b0;
    }
}
// (d_gamma | d_beta)[c] = sum over the R block partials, in a fixed order: block = 4 columns x 64 row lanes (lane q adds partials
// q, q + 64, ...; the 64 lane sums are added in order).  (Round 3 ran 16 columns x 16 row lanes: 12 blocks for C = 96, every thread
// walking 64 strided partials one after the other -- 67 us per call, 5 % of a training iteration for a 1.5 MB reduction.)
__global__ __launch_bounds__(256) void t_ln_bwd_final_kernel(const double *part, float *d_gamma, float *d_beta, int C, int R) {
    __shared__ double red[64][5];
    const int cl = threadIdx.x & 3, q = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;   // c over 2C
    double sacc = 0.0;
    if (c < 2 * C) for (int r = q; r < R; r += 64) sacc += part[(size_t)r * 2 * C + c];
    red[q][cl] = sacc;
    __syncthreads();
    if (q == 0 && c < 2 * C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 64; k++) t += red[k][cl];
        const float v = (float)t;
        if (c < C) { if (d_gamma) d_gamma[c] = v; } else if (d_beta) d_beta[c - C] = v;
    }
}

// exact-erf GELU and its derivative  Phi(x) + x phi(x)
__global__ void t_gelu_fwd_kernel(const float *x, float *y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752f));
}
__global__ void t_gelu_bwd_kernel(const float *x, const float *dy, float *dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    dx[i] = dy[i] * (0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v));
}
__global__ void t_add_kernel(float *a, const float *b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Window attention core, forward and backward, one block per (sample, window, head).  qkv [B*T][3C] token-major (as the QKV linear writes it), out / d_out [B*T][C].
//   S = scale q k^T + table[index(i,j)][h] (+ -100 where the shifted window's regions differ);  P = softmax_j S;  O = P v
// token of (window (wi,wj), position p): rolled coordinate r = (wi ws + p / ws, wj ws + p % ws), original (r + shift) mod res.
// ---------------------------------------------------------------------------------------------------------------------
struct TAttnGeom { int res, ws, shift, heads, C; };
__device__ __forceinline__ int t_token_of(const TAttnGeom &g, int w, int p, int &ri, int &rj) {
    const int nwr = g.res / g.ws, wi = w / nwr, wj = w % nwr;
    ri = wi * g.ws + p / g.ws; rj = wj * g.ws + p % g.ws;
    int ti = ri + g.shift, tj = rj + g.shift;
    if (ti >= g.res) ti -= g.res;
    if (tj >= g.res) tj -= g.res;
    return ti * g.res + tj;
}
__device__ __forceinline__ int t_region(const TAttnGeom &g, int r) { return r < g.res - g.ws ? 0 : (r < g.res - g.shift ? 1 : 2); }

// Four threads per window row: thread (i, p) owns keys j = p, p+4, ... of query row i in the row passes and head dims 8p..8p+7 (HD = 32)
// of row i in the passes that produce [Wt, HD] results; 4-lane shuffles combine a row's partial max / sums.
template <bool BWD>
__global__ void t_attn_kernel(const float *qkv, const float *table, float *out, const float *d_out, float *d_qkv, float *d_table,
                              TAttnGeom g) {
    extern __shared__ float sm[];
    const int Wt = g.ws * g.ws, HD = g.C / g.heads, LD = HD + 1, PL = Wt + 1;
    float *qs = sm, *ks = qs + Wt * LD, *vs = ks + Wt * LD, *dos = vs + Wt * LD, *Ps = dos + (BWD ? Wt * LD : 0), *dPs = Ps + Wt * PL;
    const int nW = (g.res / g.ws) * (g.res / g.ws), T = g.res * g.res;
    const int h = blockIdx.x % g.heads, w = (blockIdx.x / g.heads) % nW, b = blockIdx.x / (g.heads * nW);
    const int i = threadIdx.x >> 2, p = threadIdx.x & 3, d0 = p * (HD / 4), DG = HD / 4;
    const bool act = i < Wt;
    const float scale = 1.0f / sqrtf((float)HD);
    __shared__ int regs[128], toks[128];
    __shared__ float dtab[448];   // this block's share of the relative-position-bias gradient ((2 ws - 1)^2 <= 441 entries of head h)
    const int ntab = (2 * g.ws - 1) * (2 * g.ws - 1);
    if (BWD) for (int t = threadIdx.x; t < ntab; t += blockDim.x) dtab[t] = 0.f;
    int ri = 0, rj = 0, tok = 0, reg = 0;
    if (act) {
        tok = t_token_of(g, w, i, ri, rj);
        reg = g.shift > 0 ? 3 * t_region(g, ri) + t_region(g, rj) : 0;
        const float *row = qkv + ((size_t)b * T + tok) * 3 * g.C + h * HD;
        for (int d = d0; d < d0 + DG; d++) { qs[i * LD + d] = row[d]; ks[i * LD + d] = row[g.C + d]; vs[i * LD + d] = row[2 * g.C + d]; }
        if (BWD) { const float *dr = d_out + ((size_t)b * T + tok) * g.C + h * HD; for (int d = d0; d < d0 + DG; d++) dos[i * LD + d] = dr[d]; }
        if (p == 0) { regs[i] = reg; toks[i] = tok; }
    }
    __syncthreads();
    const int yi = i / g.ws, xi = i % g.ws;
    // ---- P = softmax(scale q k^T + bias (+ mask)) ----
    float mx = -3.0e38f;
    if (act)
        for (int j = p; j < Wt; j += 4) {
            float sc = 0.f;
            for (int d = 0; d < HD; d++) sc = fmaf(qs[i * LD + d], ks[j * LD + d], sc);
            const int idx = (yi - j / g.ws + g.ws - 1) * (2 * g.ws - 1) + (xi - j % g.ws + g.ws - 1);
            sc = sc * scale + table[(size_t)idx * g.heads + h];
            if (g.shift > 0 && regs[j] != reg) sc += -100.0f;
            Ps[i * PL + j] = sc;
            mx = fmaxf(mx, sc);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    if (act)
        for (int j = p; j < Wt; j += 4) { const float e = expf(Ps[i * PL + j] - mx); Ps[i * PL + j] = e; sum += e; }
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
    if (act) { const float inv = 1.0f / sum; for (int j = p; j < Wt; j += 4) Ps[i * PL + j] *= inv; }
    __syncthreads();
    if (!BWD) {
        if (act) {
            float *o = out + ((size_t)b * T + tok) * g.C + h * HD;
            for (int d = d0; d < d0 + DG; d++) {
                float a = 0.f;
                for (int j = 0; j < Wt; j++) a = fmaf(Ps[i * PL + j], vs[j * LD + d], a);
                o[d] = a;
            }
        }
        return;
    }
    float *dq = d_qkv + ((size_t)b * T + tok) * 3 * g.C + h * HD;
    // ---- dV[i] = sum_q P[q][i] dO[q];  dP[i][j] = dO[i] . v[j] and t_i = sum_j dP P ----
    float tsum = 0.f;
    if (act) {
        for (int d = d0; d < d0 + DG; d++) {
            float a = 0.f;
            for (int q = 0; q < Wt; q++) a = fmaf(Ps[q * PL + i], dos[q * LD + d], a);
            dq[2 * g.C + d] = a;
        }
        for (int j = p; j < Wt; j += 4) {
            float dp = 0.f;
            for (int d = 0; d < HD; d++) dp = fmaf(dos[i * LD + d], vs[j * LD + d], dp);
            dPs[i * PL + j] = dp;
            tsum = fmaf(dp, Ps[i * PL + j], tsum);
        }
    }
    tsum += __shfl_xor(tsum, 1, 64); tsum += __shfl_xor(tsum, 2, 64);
    __syncthreads();   // every dV column pass has read P before it is overwritten
    // ---- dS = P (dP - t) overwrites P; the bias gradient ----
    if (act)
        for (int j = p; j < Wt; j += 4) {
            const float ds = Ps[i * PL + j] * (dPs[i * PL + j] - tsum);
            Ps[i * PL + j] = ds;
            const int idx = (yi - j / g.ws + g.ws - 1) * (2 * g.ws - 1) + (xi - j % g.ws + g.ws - 1);
            atomicAdd(&dtab[idx], ds);   // LDS atomics; one global atomic per table entry and block below (4096 -> 225 per 8x8 window)
        }
    __syncthreads();
    for (int t = threadIdx.x; t < ntab; t += blockDim.x) atomicAdd(d_table + (size_t)t * g.heads + h, dtab[t]);
    // ---- dQ[i] = scale sum_j dS[i][j] k[j];  dK[i] = scale sum_q dS[q][i] q[q] ----
    if (act)
        for (int d = d0; d < d0 + DG; d++) {
            float a = 0.f, c = 0.f;
            for (int j = 0; j < Wt; j++) { a = fmaf(Ps[i * PL + j], ks[j * LD + d], a); c = fmaf(Ps[j * PL + i], qs[j * LD + d], c); }
            dq[d] = a * scale;
            dq[g.C + d] = c * scale;
        }
}
// ---------------------------------------------------------------------------------------------------------------------
// The same forward / backward on the f32 matrix pipe (v_mfma_f32_32x32x2_f32: exact f32 products, so the gradient fixtures hold at
// the same bars).  One block per (sample, window, head), one wave per 32 window positions; q, k, v (and dO) of the window sit in
// LDS as [position][33] (conflict-free ds_read_b32 along rows and along head dims); Wp = 32 KT padded positions.
//   phase 1, lane = QUERY i (a wave's 32 queries x all keys):
//       S^T[j][i] = k_j . q_i  ->  P^T = softmax over the keys: lane-local (+ one exchange between the half-waves); (m_i, l_i) -> LDS
//       forward:  O^T[d][i] = sum_j V^T[d][j] P^T[j][i]   -- the P^T accumulators ARE the B operand: register r of a 32x32 tile is
//                 key (r&3) + 8 (r>>2) + 4 half of the tile, and the A operand is simply fetched for that key
//       backward: dP^T[j][i] = v_j . dO_i;  t_i = sum_j dP P;  dS^T = P (dP - t);  t_i -> LDS;  bias gradient (LDS atomics);
//                 dQ^T[d][i] = scale sum_j K^T[d][j] dS^T[j][i]
//   phase 2 (backward), lane = KEY j (a wave's 32 keys x all queries), because dV and dK sum over the queries -- the LANE index of
//   phase 1 -- and turning the accumulators would need Wp x Wp floats of LDS for P and for dS (132 KB at 100 tokens):
//       S[i][j], dP[i][j] again (2 of the 7 products are recomputed), P from (m_i, l_i), dS from t_i;
//       dV^T[d][j] = sum_i dO^T[d][i] P[i][j];   dK^T[d][j] = scale sum_i Q^T[d][i] dS[i][j]
// Results leave as 4 consecutive head dims per lane (16-byte stores).
// ---------------------------------------------------------------------------------------------------------------------
template <int KT, bool BWD>
__global__ __launch_bounds__(64 * KT) void t_attn_mfma_kernel(const float *__restrict__ qkv, const float *__restrict__ table, float *__restrict__ out,
                                                              const float *__restrict__ d_out, float *__restrict__ d_qkv, float *__restrict__ d_table,
                                                              TAttnGeom g) {
    constexpr int Wp = 32 * KT, LD = 33, NT = 64 * KT;
    extern __shared__ float sm[];
    float *qs = sm, *ks = qs + Wp * LD, *vs = ks + Wp * LD, *dos = vs + Wp * LD;      // dos only with BWD
    float *stat = dos + (BWD ? Wp * LD : 0);                                           // [Wp][4]: m, 1/l, t
    float *tabh = stat + Wp * 4;                                                       // this head's bias column [(2 ws - 1)^2 <= 448]
    float *dtab = tabh + 448;                                                          // its gradient (BWD)
    int *ypos = reinterpret_cast<int *>(dtab + (BWD ? 448 : 0)), *xpos = ypos + Wp, *regs = xpos + Wp, *toks = regs + Wp;
    const int Wt = g.ws * g.ws, C = g.C;
    const int nW = (g.res / g.ws) * (g.res / g.ws), T = g.res * g.res;
    const int h = blockIdx.x % g.heads, w = (blockIdx.x / g.heads) % nW, b = blockIdx.x / (g.heads * nW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const float scale = 1.0f / sqrtf(32.0f);
    const int ntab = (2 * g.ws - 1) * (2 * g.ws - 1);
    for (int t = tid; t < 448; t += NT) { tabh[t] = t < ntab ? table[(size_t)t * g.heads + h] : 0.f; if (BWD) dtab[t] = 0.f; }
    for (int p = tid; p < Wp; p += NT) {
        int ri = 0, rj = 0, tok = 0;
        if (p < Wt) tok = t_token_of(g, w, p, ri, rj);
        ypos[p] = p / g.ws; xpos[p] = p % g.ws; toks[p] = tok;
        regs[p] = (g.shift > 0 && p < Wt) ? 3 * t_region(g, ri) + t_region(g, rj) : 0;
    }
    __syncthreads();
    // q | k | v | dO rows -> LDS (float4 loads; padded positions are zero)
    for (int idx = tid; idx < Wp * 8; idx += NT) {
        const int p = idx >> 3, c4 = idx & 7;
        f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, v4 = q4, o4 = q4;
        if (p < Wt) {
            const float *row = qkv + ((size_t)b * T + toks[p]) * 3 * C + h * 32 + 4 * c4;
            q4 = *reinterpret_cast<const f32x4 *>(row); k4 = *reinterpret_cast<const f32x4 *>(row + C); v4 = *reinterpret_cast<const f32x4 *>(row + 2 * C);
            if (BWD) o4 = *reinterpret_cast<const f32x4 *>(d_out + ((size_t)b * T + toks[p]) * C + h * 32 + 4 * c4);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            qs[p * LD + 4 * c4 + t] = q4[t]; ks[p * LD + 4 * c4 + t] = k4[t]; vs[p * LD + 4 * c4 + t] = v4[t];
            if (BWD) dos[p * LD + 4 * c4 + t] = o4[t];
        }
    }
    __syncthreads();
    // bias + mask of the pair (query position pi, key position pj); -1e30 for a padded key
    auto bias_of = [&](int pi, int pj) -> float {
        if (pj >= Wt) return -1.0e30f;
        const int idx = (ypos[pi] - ypos[pj] + g.ws - 1) * (2 * g.ws - 1) + (xpos[pi] - xpos[pj] + g.ws - 1);
        float bv = tabh[idx];
        if (g.shift > 0 && regs[pi] != regs[pj]) bv += -100.0f;
        return bv;
    };
    auto tab_idx = [&](int pi, int pj) -> int { return (ypos[pi] - ypos[pj] + g.ws - 1) * (2 * g.ws - 1) + (xpos[pi] - xpos[pj] + g.ws - 1); };
    // X^T[j][i] = sum_d Arow[j][d] Brow[i][d] for this wave's 32 columns i = 32 wave + lrow and key tile kt (rows j = 32 kt + ...)
    auto prod_T = [&](const float *Arows, const float *Brows, int kt) -> f32x16 {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const float *ap = Arows + (32 * kt + lrow) * LD + lhalf, *bp = Brows + (32 * wave + lrow) * LD + lhalf;
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], bp[2 * s2], acc, 0, 0, 0);
        return acc;
    };
    // Y^T[d][c] += sum over the 32 rows of tile rt of Mrow[row][d] * X[row][c], X = an accumulator tile (rows = its register index)
    auto prod_acc = [&](const float *Mrows, int rt, const f32x16 &X, f32x16 &Y) {
        const float *mp = Mrows + (32 * rt + 4 * lhalf) * LD + lrow;
#pragma unroll
        for (int r = 0; r < 16; r++) Y = __builtin_amdgcn_mfma_f32_32x32x2f32(mp[((r & 3) + 8 * (r >> 2)) * LD], X[r], Y, 0, 0, 0);
    };
    auto store_T = [&](const f32x16 &Y, float mul, float *dst_row) {   // Y^T[d][.]: this lane's 16 head dims -> dst_row[h*32 + d]
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; t++) o[t] = Y[4 * q + t] * mul;
            *reinterpret_cast<f32x4 *>(dst_row + 8 * q + 4 * lhalf) = o;
        }
    };
    const int pi = 32 * wave + lrow;   // phase 1: this lane's query position; phase 2: its key position
    const bool pi_ok = pi < Wt;
    // ---------------- phase 1 ----------------
    f32x16 P[KT];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        P[kt] = prod_T(ks, qs, kt);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pj = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            P[kt][r] = fmaf(P[kt][r], scale, bias_of(pi, pj));
            mx = fmaxf(mx, P[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) { P[kt][r] = __expf(P[kt][r] - mx); sum += P[kt][r]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) P[kt][r] *= inv;
    if (!BWD) {
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; r++) O[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) prod_acc(vs, kt, P[kt], O);
        if (pi_ok) store_T(O, 1.0f, out + ((size_t)b * T + toks[pi]) * C + h * 32);
        return;
    }
    f32x16 dS[KT];
    float tsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        dS[kt] = prod_T(vs, dos, kt);   // dP^T[j][i] = v_j . dO_i
#pragma unroll
        for (int r = 0; r < 16; r++) tsum = fmaf(dS[kt][r], P[kt][r], tsum);
    }
    tsum += __shfl_xor(tsum, 32, 64);
    if (lhalf == 0) { stat[pi * 4 + 0] = mx; stat[pi * 4 + 1] = inv; stat[pi * 4 + 2] = tsum; }
    {
        f32x16 dQ;
#pragma unroll
        for (int r = 0; r < 16; r++) dQ[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float ds = P[kt][r] * (dS[kt][r] - tsum);
                dS[kt][r] = ds;
                const int pj = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                if (pi_ok && pj < Wt) atomicAdd(&dtab[tab_idx(pi, pj)], ds);   // LDS atomics; one global atomic per table entry and block below
            }
            prod_acc(ks, kt, dS[kt], dQ);
        }
        if (pi_ok) store_T(dQ, scale, d_qkv + ((size_t)b * T + toks[pi]) * 3 * C + h * 32);
    }
    __syncthreads();   // (m, 1/l, t) of every query are in LDS; dtab is complete
    for (int t = tid; t < ntab; t += NT) atomicAdd(d_table + (size_t)t * g.heads + h, dtab[t]);
    // ---------------- phase 2: lane = key pi ----------------
    f32x16 dV, dK;
#pragma unroll
    for (int r = 0; r < 16; r++) { dV[r] = 0.f; dK[r] = 0.f; }
#pragma unroll
    for (int it = 0; it < KT; it++) {
        // S[i][j]: rows = queries of tile it (A = q rows), columns = this wave's keys (B = k rows)
        f32x16 Pq, dSq;
#pragma unroll
        for (int r = 0; r < 16; r++) { Pq[r] = 0.f; dSq[r] = 0.f; }
        {
            const float *ap = qs + (32 * it + lrow) * LD + lhalf, *bp = ks + (32 * wave + lrow) * LD + lhalf;
            const float *ap2 = dos + (32 * it + lrow) * LD + lhalf, *bp2 = vs + (32 * wave + lrow) * LD + lhalf;
#pragma unroll
            for (int s2 = 0; s2 < 16; s2++) {
                Pq = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], bp[2 * s2], Pq, 0, 0, 0);
                dSq = __builtin_amdgcn_mfma_f32_32x32x2f32(ap2[2 * s2], bp2[2 * s2], dSq, 0, 0, 0);   // dP[i][j] = dO_i . v_j
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int qi = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lhalf;   // query position of this register
            const float m = stat[qi * 4 + 0], il = stat[qi * 4 + 1], tq = stat[qi * 4 + 2];
            float pv = __expf(fmaf(Pq[r], scale, bias_of(qi, pi)) - m) * il;
            if (qi >= Wt) pv = 0.f;   // a padded query row (its statistics are those of an all-padding softmax): contributes nothing
            Pq[r] = pv;
            dSq[r] = pv * (dSq[r] - tq);
        }
        prod_acc(dos, it, Pq, dV);    // dV^T[d][j] += sum_i dO[i][d] P[i][j]
        prod_acc(qs, it, dSq, dK);    // dK^T[d][j] += sum_i Q[i][d] dS[i][j]
    }
    if (pi_ok) {
        float *row = d_qkv + ((size_t)b * T + toks[pi]) * 3 * C + h * 32;
        store_T(dK, scale, row + C);
        store_T(dV, 1.0f, row + 2 * C);
    }
}

template <int KT>
static bool t_attn_mfma_launch(bool bwd, const float *qkv, const float *table, float *out, const float *d_out, float *d_qkv, float *d_table,
                               int B, TAttnGeom g, hipStream_t s) {
    constexpr int Wp = 32 * KT;
    const size_t lds = sizeof(float) * ((size_t)(bwd ? 4 : 3) * Wp * 33 + (size_t)Wp * 4 + 448 + (bwd ? 448 : 0) + 4 * (size_t)Wp);
    const int nW = (g.res / g.ws) * (g.res / g.ws);
    const dim3 grid(B * nW * g.heads), block(64 * KT);
    if (bwd) {
        if (hipFuncSetAttribute((const void *)t_attn_mfma_kernel<KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        hipLaunchKernelGGL((t_attn_mfma_kernel<KT, true>), grid, block, lds, s, qkv, table, out, d_out, d_qkv, d_table, g);
    } else {
        if (hipFuncSetAttribute((const void *)t_attn_mfma_kernel<KT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        hipLaunchKernelGGL((t_attn_mfma_kernel<KT, false>), grid, block, lds, s, qkv, table, out, d_out, d_qkv, d_table, g);
    }
    return true;
}

static bool t_attn_launch(bool bwd, const float *qkv, const float *table, float *out, const float *d_out, float *d_qkv, float *d_table,
                          int B, TAttnGeom g, hipStream_t s) {
    const int Wt = g.ws * g.ws, HD = g.C / g.heads, LD = HD + 1;
    if (Wt > 128 || HD % 4 != 0) return false;
    static const bool plain = getenv("DSG_TRAIN_PLAIN_ATTN") != nullptr;   // dev knob: the scalar kernel below
    if (!plain && HD == 32 && g.C % 4 == 0) {
        if (Wt <= 32) return t_attn_mfma_launch<1>(bwd, qkv, table, out, d_out, d_qkv, d_table, B, g, s);
        if (Wt <= 64) return t_attn_mfma_launch<2>(bwd, qkv, table, out, d_out, d_qkv, d_table, B, g, s);
        return t_attn_mfma_launch<4>(bwd, qkv, table, out, d_out, d_qkv, d_table, B, g, s);
    }
    const size_t lds = sizeof(float) * ((size_t)(bwd ? 4 : 3) * Wt * LD + (size_t)(bwd ? 2 : 1) * Wt * (Wt + 1));
    const int nW = (g.res / g.ws) * (g.res / g.ws);
    const dim3 grid(B * nW * g.heads), block((4 * Wt + 63) / 64 * 64);
    if (bwd) {
        if (hipFuncSetAttribute((const void *)t_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        hipLaunchKernelGGL((t_attn_kernel<true>), grid, block, lds, s, qkv, table, out, d_out, d_qkv, d_table, g);
    } else {
        if (hipFuncSetAttribute((const void *)t_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        hipLaunchKernelGGL((t_attn_kernel<false>), grid, block, lds, s, qkv, table, out, d_out, d_qkv, d_table, g);
    }
    return true;
}

// ---- small kernels of the rest of the network (training form) ----
__global__ void t_silu_fwd_kernel(const float *x, float *y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * t_sigmoid(x[i]);
}
__global__ void t_silu_bwd_kernel(const float *x, const float *dy, float *dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sg = t_sigmoid(x[i]);
    dx[i] = dy[i] * (sg * (1.0f + x[i] * (1.0f - sg)));
}
// PatchMerging's 2x2 regrouping (diffusesg.py:322-327) and its inverse (PatchBreakup's scatter, :389-398, is the inverse with C = D/4):
// coarse[((b T2 + i r2 + j) 4 + q) C + c] <-> fine[(b T + (2i + (q&1)) res + 2j + (q>>1)) C + c];  gather: coarse <- fine
__global__ void t_regroup_kernel(const float *src, float *dst, int B, int res, int C, int gather, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = idx % C;
    const size_t tok = idx / C;
    const int T = res * res, r2 = res / 2;
    const int b = tok / T, t = tok % T, ti = t / res, tj = t % res;
    const int q = (ti & 1) + 2 * (tj & 1);
    const size_t coarse = (((size_t)b * (T / 4) + (size_t)(ti / 2) * r2 + tj / 2) * 4 + q) * C + c;
    if (gather) dst[coarse] = src[idx]; else dst[idx] = src[coarse];
}
// cat[m] = (x[m] | skip[m])  and the split of its gradient (d_skip accumulates: the skip tensor also feeds the encoder's next stage)
__global__ void t_concat_kernel(const float *x, const float *skip, float *cat, int C, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t m = i / (2 * C);
    const int c = i % (2 * C);
    cat[i] = c < C ? x[m * C + c] : skip[m * C + c - C];
}
__global__ void t_split_kernel(const float *dcat, float *dx, float *dskip_acc, int C, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t m = i / (2 * C);
    const int c = i % (2 * C);
    if (c < C) dx[m * C + c] = dcat[i]; else dskip_acc[m * C + c - C] += dcat[i];
}
// adjacency head tail: F_adj[b,a,i,j] = f_i f_j oa[(b,i,j)][a] (mask_adjs, diffusesg.py:825) and its transpose
__global__ void t_adj_out_kernel(const float *oa, const uint8_t *flags, float *F, const float *dF, float *d_oa, int B, int N, int Ca, int bwd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * N * Ca) return;
    const int a = idx % Ca;
    const size_t t = idx / Ca;
    const int j = t % N, i = (t / N) % N, b = t / ((size_t)N * N);
    const bool ok = flags[(size_t)b * N + i] && flags[(size_t)b * N + j];
    const size_t k = (((size_t)b * Ca + a) * N + i) * N + j;
    if (bwd) d_oa[idx] = ok ? dF[k] : 0.f; else F[k] = ok ? oa[idx] : 0.f;
}
// node pooling backward: d_rep[(b,i,j)][e] += f_i f_j d_pool[(b,i)][e] / N   (forward: launch_pool, diffusesg.py:812-815)
__global__ void t_pool_bwd_kernel(const float *d_pool, const uint8_t *flags, float *d_rep_acc, int B, int N, int E) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * N * N * E) return;
    const int e = idx % E;
    const size_t t = idx / E;
    const int j = t % N, i = (t / N) % N, b = t / ((size_t)N * N);
    if (flags[(size_t)b * N + i] && flags[(size_t)b * N + j]) d_rep_acc[idx] += d_pool[((size_t)b * N + i) * E + e] / (float)N;
}
// y[m][c] = f_m x[m][c]   (mask_nodes on [B*N, Cn]; its own transpose)
__global__ void t_rowmask_kernel(const float *x, const uint8_t *flags, float *y, int C, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = flags[i / C] ? x[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// One block, forward (training form) + backward.  All buffers caller-provided (see dsg_block_train in dsg_api.cpp).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int NOISE_EMB = 512;   // width of the mapped noise embedding (diffusesg.py: noise_emb_channels)
static inline unsigned t_blocks(size_t n) { return (unsigned)((n + 255) / 256); }

bool train_block(const TrainBlockArgs &a, hipStream_t s) {
    const int B = a.B, T = a.res * a.res, C = a.C, M = B * T, H = a.hidden;
    TAttnGeom g{a.res, a.ws, a.shift, a.heads, C};
    // ---- forward ----
    if (!a.aff_grouped)   // params = affine(emb); the whole-network step computes all blocks' rows in one grouped launch up front
        t_gemm(false, true, a.emb, NOISE_EMB, a.W.aff_w, NOISE_EMB, a.W.aff_b, a.aff, 2 * C, B, 2 * C, NOISE_EMB, false, s);
    t_ln_fwd_launch(a.x_in, a.aff, a.x_mod, a.W.n1_w, a.W.n1_b, a.xn1, a.stats1, M, C, T, s);   // x_mod = modulate(x_in), xn1 = LayerNorm-1(x_mod): one row pass
    t_gemm(false, true, a.xn1, C, a.W.qkv_w, C, a.W.qkv_b, a.qkv, 3 * C, M, 3 * C, C, false, s);
    if (!t_attn_launch(false, a.qkv, a.W.rpb, a.att, nullptr, nullptr, nullptr, B, g, s)) return false;
    t_gemm(false, true, a.att, C, a.W.proj_w, C, a.W.proj_b, a.x1, C, M, C, C, false, s, nullptr, a.x_mod);                  // x1 = shortcut + proj(att)
    t_ln_fwd_launch(a.x1, nullptr, nullptr, a.W.n2_w, a.W.n2_b, a.xn2, a.stats2, M, C, T, s);
    t_gemm(false, true, a.xn2, C, a.W.fc1_w, C, a.W.fc1_b, a.hid, H, M, H, C, false, s, nullptr, nullptr, ACT_GELU_KEEP, a.pre);   // pre = fc1(xn2), hid = GELU(pre)
    t_gemm(false, true, a.hid, H, a.W.fc2_w, H, a.W.fc2_b, a.x_out, C, M, C, H, false, s, nullptr, a.x1);                    // x_out = x1 + fc2(gelu(fc1(ln2)))
    if (!a.grad_out) return hipGetLastError() == hipSuccess;
    return train_block_backward(a, s);
}

// backward of the block from the tensors train_block's forward left in `a` (d_x1 accumulates in a.d_x1; scratch t_mc, t_mc2 [M,C],
// t_mh [M,H], t_m3c [M,3C]; a.grad_out is read, a.grad_in / a.grad_emb / a.G.* are written)
bool train_block_backward(const TrainBlockArgs &a, hipStream_t s) {
    const int B = a.B, T = a.res * a.res, C = a.C, M = B * T, H = a.hidden;
    TAttnGeom g{a.res, a.ws, a.shift, a.heads, C};
    const float *dY = a.grad_out;
    // MLP: x_out = x1 + hid W2^T + b2
    t_gemm(true, false, dY, C, a.hid, H, nullptr, a.G.fc2_w, H, C, H, M, false, s, a.G.fc2_b);  // dW2 [C,H] = dY^T hid, db2 = colsum(dY)
    t_gemm(false, false, dY, C, a.W.fc2_w, H, nullptr, a.t_mh, H, M, H, C, false, s, nullptr, a.pre, ACT_DGELU);   // d_pre = (dY W2) GELU'(pre)
    t_gemm(true, false, a.t_mh, H, a.xn2, C, nullptr, a.G.fc1_w, C, H, C, M, false, s, a.G.fc1_b);   // dW1 [H,C] = d_pre^T xn2, db1
    t_gemm(false, false, a.t_mh, H, a.W.fc1_w, C, nullptr, a.t_mc, C, M, C, H, false, s);       // d_xn2 = d_pre W1
    t_ln_bwd(a.x1, a.W.n2_w, a.stats2, a.t_mc, dY, a.d_x1, a.G.n2_w, a.G.n2_b, M, C, s);       // d_x1 = dY (residual branch) + LN2 backward; d_gamma2, d_beta2
    // attention half: x1 = x_mod + att Wp^T + bp
    t_gemm(true, false, a.d_x1, C, a.att, C, nullptr, a.G.proj_w, C, C, C, M, false, s, a.G.proj_b);
    t_gemm(false, false, a.d_x1, C, a.W.proj_w, C, nullptr, a.t_mc, C, M, C, C, false, s);      // d_att
    if (hipMemsetAsync(a.G.rpb, 0, sizeof(float) * (size_t)(2 * a.ws - 1) * (2 * a.ws - 1) * a.heads, s) != hipSuccess) return false;
    if (!t_attn_launch(true, a.qkv, a.W.rpb, nullptr, a.t_mc, a.t_m3c, a.G.rpb, B, g, s)) return false;   // d_qkv, d_table
    t_gemm(true, false, a.t_m3c, 3 * C, a.xn1, C, nullptr, a.G.qkv_w, C, 3 * C, C, M, false, s, a.G.qkv_b);
    t_gemm(false, false, a.t_m3c, 3 * C, a.W.qkv_w, C, nullptr, a.t_mc, C, M, C, 3 * C, false, s);   // d_xn1
    // d_xmod = d_x1 (shortcut) + LN1 backward
    t_ln_bwd(a.x_mod, a.W.n1_w, a.stats1, a.t_mc, a.d_x1, a.d_x1, a.G.n1_w, a.G.n1_b, M, C, s);
    // modulate: x_mod = silu(shift + x (1 + scale)); params = emb Wa^T + ba
    t_modulate(a.x_in, a.aff, a.d_x1, a.grad_in, a.d_aff, B, T, C, true, s);
    if (a.aff_grouped) return hipGetLastError() == hipSuccess;   // dWa, d_ba and d_emb of all blocks: grouped launches of the whole-network step
    t_gemm(true, false, a.d_aff, 2 * C, a.emb, NOISE_EMB, nullptr, a.G.aff_w, NOISE_EMB, 2 * C, NOISE_EMB, B, false, s);   // dWa = d_aff^T emb
    t_colsum(a.d_aff, 2 * C, a.G.aff_b, B, 2 * C, s);
    t_gemm(false, false, a.d_aff, 2 * C, a.W.aff_w, NOISE_EMB, nullptr, a.grad_emb, NOISE_EMB, B, NOISE_EMB, 2 * C, false, s);   // d_emb
    return hipGetLastError() == hipSuccess;
}

// ---- launch wrappers used by the whole-network training step (dsg_api.cpp) ----
void t_silu(const float *x, const float *dy, float *out, size_t n, bool bwd, hipStream_t s) {
    if (bwd) hipLaunchKernelGGL(t_silu_bwd_kernel, dim3(t_blocks(n)), dim3(256), 0, s, x, dy, out, n);
    else hipLaunchKernelGGL(t_silu_fwd_kernel, dim3(t_blocks(n)), dim3(256), 0, s, x, out, n);
}
void t_gelu(const float *x, const float *dy, float *out, size_t n, bool bwd, hipStream_t s) {
    if (bwd) hipLaunchKernelGGL(t_gelu_bwd_kernel, dim3(t_blocks(n)), dim3(256), 0, s, x, dy, out, n);
    else hipLaunchKernelGGL(t_gelu_fwd_kernel, dim3(t_blocks(n)), dim3(256), 0, s, x, out, n);
}
void t_add(float *a, const float *b, size_t n, hipStream_t s) { hipLaunchKernelGGL(t_add_kernel, dim3(t_blocks(n)), dim3(256), 0, s, a, b, n); }
void t_ln_fwd(const float *x, const float *gam, const float *bet, float *y, float *stats, int M, int C, hipStream_t s) {
    t_ln_fwd_launch(x, nullptr, nullptr, gam, bet, y, stats, M, C, 1, s);
}
void t_ln_bwd(const float *x, const float *gam, const float *stats, const float *dy, const float *dx_in, float *dx_out, float *d_gamma, float *d_beta,
              int M, int C, hipStream_t s) {
    // <= 1024 blocks, >= 4 rows each (one per wave)
    int blocks = std::max(1, std::min(1024, (M + 3) / 4));
    const int rows_per = ((M + blocks - 1) / blocks + 3) / 4 * 4;
    blocks = (M + rows_per - 1) / rows_per;
    TScratch &ts = t_scratch(s);
    double *part = t_scratch_get(s, ts.cs, ts.cs_cap, (size_t)blocks * 2 * C, ts);
    if (!part) return;   // (the stream's scratch is marked failed)
#define T_LNB(LPR_, KV_) hipLaunchKernelGGL((t_ln_bwd_kernel<LPR_, KV_>), dim3(blocks), dim3(256), 0, s, x, gam, stats, dy, dx_in, dx_out, part, M, C, rows_per)
    if (C % 4 != 0 || C > 1536) { ts.failed = true; return; }   // (4 x 384 is the widest LayerNorm of the networks)
    if (C <= 128) T_LNB(32, 1); else if (C <= 256) T_LNB(64, 1); else if (C <= 512) T_LNB(64, 2); else if (C <= 768) T_LNB(64, 3); else T_LNB(64, 6);
#undef T_LNB
    hipLaunchKernelGGL(t_ln_bwd_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, s, part, d_gamma, d_beta, C, blocks);
}
void t_modulate(const float *x, const float *aff, const float *dy, float *out, float *d_aff, int B, int T, int C, bool bwd, hipStream_t s) {
    if (bwd) {
        const int rows_per = max(64, (T + 63) / 64), chunks = (T + rows_per - 1) / rows_per;
        TScratch &ts = t_scratch(s);
        double *buf = t_scratch_get(s, ts.cs, ts.cs_cap, (size_t)B * chunks * 2 * C, ts);
        if (!buf) return;   // (the stream's scratch is marked failed)
        hipLaunchKernelGGL(t_modulate_bwd_kernel, dim3((C + 63) / 64, chunks, B), dim3(256), 0, s, x, aff, dy, out, buf, T, C, rows_per, chunks);
        hipLaunchKernelGGL(t_modulate_bwd_final_kernel, dim3((B * 2 * C + 255) / 256), dim3(256), 0, s, buf, d_aff, B, C, chunks);
    } else hipLaunchKernelGGL(t_modulate_fwd_kernel, dim3(t_blocks((size_t)B * T * C)), dim3(256), 0, s, x, aff, out, T, C, (size_t)B * T * C);
}
void t_regroup(const float *src, float *dst, int B, int res, int C, bool gather, hipStream_t s) {
    const size_t n = (size_t)B * res * res * C;
    hipLaunchKernelGGL(t_regroup_kernel, dim3(t_blocks(n)), dim3(256), 0, s, src, dst, B, res, C, (int)gather, n);
}
void t_concat(const float *x, const float *skip, float *cat, size_t M, int C, hipStream_t s) {
    hipLaunchKernelGGL(t_concat_kernel, dim3(t_blocks(M * 2 * C)), dim3(256), 0, s, x, skip, cat, C, M * 2 * C);
}
void t_split(const float *dcat, float *dx, float *dskip_acc, size_t M, int C, hipStream_t s) {
    hipLaunchKernelGGL(t_split_kernel, dim3(t_blocks(M * 2 * C)), dim3(256), 0, s, dcat, dx, dskip_acc, C, M * 2 * C);
}
void t_adj_out(const float *oa, const uint8_t *flags, float *F, const float *dF, float *d_oa, int B, int N, int Ca, bool bwd, hipStream_t s) {
    hipLaunchKernelGGL(t_adj_out_kernel, dim3(t_blocks((size_t)B * N * N * Ca)), dim3(256), 0, s, oa, flags, F, dF, d_oa, B, N, Ca, (int)bwd);
}
void t_pool_bwd(const float *d_pool, const uint8_t *flags, float *d_rep_acc, int B, int N, int E, hipStream_t s) {
    hipLaunchKernelGGL(t_pool_bwd_kernel, dim3(t_blocks((size_t)B * N * N * E)), dim3(256), 0, s, d_pool, flags, d_rep_acc, B, N, E);
}
void t_rowmask(const float *x, const uint8_t *flags, float *y, size_t M, int C, hipStream_t s) {
    hipLaunchKernelGGL(t_rowmask_kernel, dim3(t_blocks(M * C)), dim3(256), 0, s, x, flags, y, C, M * C);
}

// ---- optimiser step (torch.optim.Adam, R/utils/learning_utils.py:137-140) with nn.utils.clip_grad_norm_ in front (trainer_node_adj.py:170) ----
// Multi-tensor: ONE launch per phase over all parameter tensors (a model has 233-314 of them; one launch per tensor and phase was
// ~2800 launches per iteration).  The tensors are cut into chunks of OPT_CHUNK elements, one block per chunk; tab[t] describes
// tensor t, blk[b] = (tensor, first element) of block b.  Fixed chunking -> deterministic sums.
constexpr int OPT_CHUNK = 4096;
struct OptTensor { float *p, *g, *m, *v; unsigned long long n; unsigned first_blk, n_blk; };
struct OptBlock { unsigned tensor, start; };
// partial[b] = sum of squares of block b's chunk (double)
__global__ __launch_bounds__(256) void t_mt_sumsq_kernel(const OptTensor *tab, const OptBlock *blk, double *partial) {
    __shared__ double red[256];
    const OptBlock bk = blk[blockIdx.x];
    const OptTensor t = tab[bk.tensor];
    const unsigned long long i0 = (unsigned long long)bk.start, i1 = min(t.n, i0 + OPT_CHUNK);
    double s = 0.0;
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += 256) s += (double)t.g[i] * (double)t.g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// per tensor: fp32 norm = sqrt(sum of its chunks' partials, in chunk order) (torch: per-tensor fp32 norms, then the norm of the norms);
// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (total + 1e-6)).  One block; thread i takes tensors i, i + 256, ...;
// the per-thread sums are then added in thread order.
__global__ __launch_bounds__(256) void t_clip_coef_kernel(const OptTensor *tab, const double *partial, int n_tensors, float max_norm, float *out) {
    __shared__ double red[256];
    double tot = 0.0;
    for (int t = threadIdx.x; t < n_tensors; t += 256) {
        double s = 0.0;
        for (unsigned b = 0; b < tab[t].n_blk; b++) s += partial[tab[t].first_blk + b];
        const float nt = sqrtf((float)s);
        tot += (double)nt * (double)nt;
    }
    red[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        double all = 0.0;
        for (int i = 0; i < 256; i++) all += red[i];
        const float total = sqrtf((float)all);
        out[0] = total;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (total + 1e-6f)) : 1.0f;
    }
}
__global__ __launch_bounds__(256) void t_mt_adam_kernel(const OptTensor *tab, const OptBlock *blk, const float *clip, float lr, float b1, float b2,
                                                        float eps, float wd, float bc1, float bc2_sqrt) {
    const OptBlock bk = blk[blockIdx.x];
    const OptTensor t = tab[bk.tensor];
    const unsigned long long i0 = (unsigned long long)bk.start, i1 = min(t.n, i0 + OPT_CHUNK);
    const float cl = clip[1];
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += 256) {
        float gr = t.g[i] * cl;
        t.g[i] = gr;   // clip_grad_norm_ scales the gradients in place
        const float pi = t.p[i];
        if (wd != 0.f) gr = fmaf(wd, pi, gr);
        const float mi = t.m[i] + (gr - t.m[i]) * (1.0f - b1);             // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = t.v[i] * b2 + (1.0f - b2) * gr * gr;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        t.m[i] = mi; t.v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        t.p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
// ema <- ema + (p - ema)(1 - decay) (ma.lerp_(current, 1 - decay)); tab[t].p = ema tensor, tab[t].g = parameter tensor
__global__ __launch_bounds__(256) void t_mt_ema_kernel(const OptTensor *tab, const OptBlock *blk, float decay) {
    const OptBlock bk = blk[blockIdx.x];
    const OptTensor t = tab[bk.tensor];
    const unsigned long long i0 = (unsigned long long)bk.start, i1 = min(t.n, i0 + OPT_CHUNK);
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += 256) t.p[i] = t.p[i] + (t.g[i] - t.p[i]) * (1.0f - decay);
}
// device tables for a tensor list, in the stream's scratch: [OptTensor n | OptBlock nb | double partial nb | float coef 2]
struct OptTables { OptTensor *tab; OptBlock *blk; double *partial; float *coef; unsigned nb; };
static bool t_opt_tables(int n, float *const *p, float *const *g, float *const *m, float *const *v, const int64_t *numel, hipStream_t s, OptTables &o) {
    std::vector<OptTensor> tab((size_t)n);
    std::vector<OptBlock> blk;
    for (int t = 0; t < n; t++) {
        if (numel[t] < 0 || numel[t] > 0xffffffffll) return false;
        tab[t] = OptTensor{p[t], g ? g[t] : nullptr, m ? m[t] : nullptr, v ? v[t] : nullptr, (unsigned long long)numel[t], (unsigned)blk.size(), 0u};
        for (int64_t i = 0; i < numel[t]; i += OPT_CHUNK) blk.push_back(OptBlock{(unsigned)t, (unsigned)i});
        tab[t].n_blk = (unsigned)blk.size() - tab[t].first_blk;
    }
    const size_t b_tab = (sizeof(OptTensor) * n + 255) / 256 * 256, b_blk = (sizeof(OptBlock) * blk.size() + 255) / 256 * 256,
                 b_par = (sizeof(double) * blk.size() + 255) / 256 * 256;
    TScratch &ts = t_scratch(s);
    char *base = nullptr;
    {
        char *cur = static_cast<char *>(ts.opt);
        base = t_scratch_get(s, cur, ts.opt_cap, b_tab + b_blk + b_par + 256, ts);
        ts.opt = cur;
    }
    if (!base) return false;
    o.tab = reinterpret_cast<OptTensor *>(base); o.blk = reinterpret_cast<OptBlock *>(base + b_tab);
    o.partial = reinterpret_cast<double *>(base + b_tab + b_blk); o.coef = reinterpret_cast<float *>(base + b_tab + b_blk + b_par);
    o.nb = (unsigned)blk.size();
    // (the stream is synchronised at the end of every optimiser / EMA call, so the previous call's tables are no longer read)
    if (hipMemcpyAsync(o.tab, tab.data(), sizeof(OptTensor) * n, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (o.nb && hipMemcpyAsync(o.blk, blk.data(), sizeof(OptBlock) * blk.size(), hipMemcpyHostToDevice, s) != hipSuccess) return false;
    return hipStreamSynchronize(s) == hipSuccess;   // the host vectors go out of scope
}
bool t_adam_step(int n, float *const *params, float *const *grads, float *const *m, float *const *v, const int64_t *numel, int step, float lr,
                 float b1, float b2, float eps, float wd, float max_norm, float *host_total_norm, hipStream_t s) {
    OptTables o;
    if (!t_opt_tables(n, params, grads, m, v, numel, s, o)) return false;
    if (o.nb) hipLaunchKernelGGL(t_mt_sumsq_kernel, dim3(o.nb), dim3(256), 0, s, o.tab, o.blk, o.partial);
    hipLaunchKernelGGL(t_clip_coef_kernel, dim3(1), dim3(256), 0, s, o.tab, o.partial, n, max_norm, o.coef);
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    if (o.nb) hipLaunchKernelGGL(t_mt_adam_kernel, dim3(o.nb), dim3(256), 0, s, o.tab, o.blk, o.coef, lr, b1, b2, eps, wd, bc1, bc2s);
    bool ok = hipGetLastError() == hipSuccess;
    if (host_total_norm) ok = ok && hipMemcpyAsync(host_total_norm, o.coef, sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    return ok;
}
bool t_ema_update(int n, float *const *ema, const float *const *params, const int64_t *numel, float decay, hipStream_t s) {
    OptTables o;
    if (!t_opt_tables(n, ema, const_cast<float *const *>(params), nullptr, nullptr, numel, s, o)) return false;
    if (o.nb) hipLaunchKernelGGL(t_mt_ema_kernel, dim3(o.nb), dim3(256), 0, s, o.tab, o.blk, decay);
    return hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
}

}  // namespace dsg
