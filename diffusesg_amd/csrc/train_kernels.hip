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

namespace dsg {

// ---------------------------------------------------------------------------------------------------------------------
// C[M,N] (+)= op(A) op(B) (+ bias[n]);  op(A)(m,k) = TA ? A[k*lda+m] : A[m*lda+k];  op(B)(k,n) = TB ? B[n*ldb+k] : B[k*ldb+n]
// 32x32 tile per 256-thread block, 4 outputs per thread, k in chunks of 32, fixed summation order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void t_gemm_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                     const float *__restrict__ bias, float *__restrict__ C, int ldc, int M, int N, int K,
                                                     int accumulate, int kslice) {
    // kslice > 0: split-K -- block z handles k in [z kslice, (z+1) kslice) and writes its partial product to C + z M N (ldc = N)
    __shared__ float As[32][33], Bs[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int kbeg = kslice > 0 ? blockIdx.z * kslice : 0;
    if (kslice > 0) { C += (size_t)blockIdx.z * M * N; K = min(K, kbeg + kslice); }
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
// C[i] = (accumulate ? C[i] : 0) + sum_z part[z][i]   (fixed order)
__global__ void t_splitk_reduce_kernel(const float *part, float *C, int ldc, int M, int N, int S, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = i / N, n = i % N;
    float v = accumulate ? C[(size_t)m * ldc + n] : 0.f;
    for (int z = 0; z < S; z++) v += part[(size_t)z * M * N + i];
    C[(size_t)m * ldc + n] = v;
}
static float *g_sk_scratch = nullptr;
static size_t g_sk_cap = 0;

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
static float *g_wt_scratch = nullptr;
static size_t g_wt_cap = 0;

void t_gemm(bool ta, bool tb, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K,
            bool accumulate, hipStream_t s) {
    // The activation-side products y = x W^T and dx = dy W have the sampling path's GEMM form (A [M,K] row-major, second operand
    // [N,K]): large ones go to gemm4_f32_kernel (dx after transposing the weight into a scratch tile).  The weight-gradient
    // products (ta: K = tokens) and everything small or oddly shaped stay on the plain kernel below.
    static const bool use_mfma = getenv("DSG_TRAIN_PLAIN_GEMM") == nullptr;
    if (use_mfma && !ta && M >= 512 && K % 32 == 0 && N % 32 == 0 && lda == K && (size_t)N * K <= ((size_t)32 << 20)) {
        const float *Wop = B;
        bool ok = true;
        if (tb) ok = ldb == K;
        else {
            const size_t need = (size_t)N * K;
            if (need > g_wt_cap) {
                if (g_wt_scratch) (void)hipFree(g_wt_scratch);   // (the stream has been drained by the callers' end-of-step sync before a larger model shows up)
                g_wt_cap = 0; g_wt_scratch = nullptr;
                if (hipMalloc((void **)&g_wt_scratch, sizeof(float) * need) == hipSuccess) g_wt_cap = need;
            }
            ok = g_wt_cap >= need;
            if (ok) {
                hipLaunchKernelGGL(t_transpose_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, s, B, ldb, g_wt_scratch, K, N);   // B [K][N] -> [N][K]
                Wop = g_wt_scratch;
            }
        }
        if (ok) {
            GemmArgs g;
            g.A = A; g.lda = K; g.K1 = K; g.K = K; g.M = M; g.N = N; g.W = Wop; g.bias = bias; g.C = C; g.ldc = ldc;
            if (accumulate) { g.res = C; g.ldres = ldc; }
            launch_gemm(g, s);
            return;
        }
    }
    // Weight gradients dW [M = out, N = in] = dy^T x with K = tokens: both operands are transposed into K-contiguous slices
    // ([S][rows][kslice], zero-padded tail) and the S slice products run as ONE batched launch of the sampling path's MFMA GEMM
    // (gemm4_f32_kernel AMODE 2); the partial products are then added in slice order.
    if (use_mfma && ta && !tb && !bias && K >= 4096 && M % 32 == 0 && N % 32 == 0) {
        int S = std::min(256, std::max(1, (int)(((size_t)K * 48) / ((size_t)M * N) + 1)));   // enough tiles to fill the chip, slices >= 512 long
        S = std::min(S, K / 512);
        const int kslice = ((K + S - 1) / S + 31) / 32 * 32;
        S = (K + kslice - 1) / kslice;
        const size_t nA = (size_t)S * M * kslice, nB = (size_t)S * N * kslice, nC = (size_t)S * M * N, need = nA + nB + nC;
        if (need > g_sk_cap) {
            if (g_sk_scratch) (void)hipFree(g_sk_scratch);
            g_sk_cap = 0; g_sk_scratch = nullptr;
            if (hipMalloc((void **)&g_sk_scratch, sizeof(float) * need) == hipSuccess) g_sk_cap = need;
        }
        if (g_sk_cap >= need) {
            float *At = g_sk_scratch, *Bt = At + nA, *Cp = Bt + nB;
            hipLaunchKernelGGL(t_transpose_sliced_kernel, dim3((M + 31) / 32, (kslice + 31) / 32, S), dim3(256), 0, s, A, lda, At, K, M, kslice);
            hipLaunchKernelGGL(t_transpose_sliced_kernel, dim3((N + 31) / 32, (kslice + 31) / 32, S), dim3(256), 0, s, B, ldb, Bt, K, N, kslice);
            GemmArgs g;
            g.A = At; g.lda = kslice; g.K1 = kslice; g.K = kslice; g.M = M; g.N = N; g.W = Bt; g.C = Cp; g.ldc = N;
            g.batch = S; g.batch_strideA = (size_t)M * kslice; g.batch_strideW = (size_t)N * kslice; g.batch_strideC = (size_t)M * N;
            launch_gemm(g, s);
            hipLaunchKernelGGL(t_splitk_reduce_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, s, Cp, C, ldc, M, N, S,
                               (int)accumulate);
            return;
        }
    }
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
        const size_t need = (size_t)S * M * N;
        if (need > g_sk_cap) {
            if (g_sk_scratch) (void)hipFree(g_sk_scratch);
            g_sk_cap = 0; g_sk_scratch = nullptr;
            if (hipMalloc((void **)&g_sk_scratch, sizeof(float) * need) == hipSuccess) g_sk_cap = need;
        }
        if (g_sk_cap >= need) { Cout = g_sk_scratch; ldo = N; grid.z = S; } else { S = 1; kslice = 0; }
    }
    const int acc1 = S > 1 ? 0 : (int)accumulate;
    if (!ta && !tb) hipLaunchKernelGGL((t_gemm_kernel<false, false>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else if (!ta && tb) hipLaunchKernelGGL((t_gemm_kernel<false, true>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else if (ta && !tb) hipLaunchKernelGGL((t_gemm_kernel<true, false>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    else hipLaunchKernelGGL((t_gemm_kernel<true, true>), grid, block, 0, s, A, lda, B, ldb, bias, Cout, ldo, M, N, K, acc1, kslice);
    if (S > 1)
        hipLaunchKernelGGL(t_splitk_reduce_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, s, g_sk_scratch, C, ldc, M, N, S,
                           (int)accumulate);
}

// out[n] = sum_m X[m*ld + n]: row chunks summed by separate blocks (double, fixed order inside a chunk), then the chunks in order
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
__global__ void t_colsum_final_kernel(const double *part, float *out, int N, int R) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double sacc = 0.0;
    for (int r = 0; r < R; r++) sacc += part[(size_t)r * N + n];
    out[n] = (float)sacc;
}
static double *g_cs_scratch = nullptr;
static size_t g_cs_cap = 0;
void t_colsum(const float *X, int ld, float *out, int M, int N, hipStream_t s) {
    const int rows_per = max(64, (M + 255) / 256), R = (M + rows_per - 1) / rows_per;
    const size_t need = (size_t)R * N;
    if (need > g_cs_cap) {
        if (g_cs_scratch) (void)hipFree(g_cs_scratch);
        g_cs_cap = 0; g_cs_scratch = nullptr;
        if (hipMalloc((void **)&g_cs_scratch, sizeof(double) * need) == hipSuccess) g_cs_cap = need;
    }
    if (g_cs_cap < need) { fprintf(stderr, "dsg: t_colsum: out of memory\n"); abort(); }
    hipLaunchKernelGGL(t_colsum_part_kernel, dim3((N + 63) / 64, R), dim3(256), 0, s, X, ld, g_cs_scratch, M, N, rows_per);
    hipLaunchKernelGGL(t_colsum_final_kernel, dim3((N + 63) / 64), dim3(64), 0, s, g_cs_scratch, out, N, R);
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
__global__ __launch_bounds__(256) void t_ln_fwd_kernel(const float *x, const float *gam, const float *bet, float *y, float *stats, int M, int C) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const float *r = x + (size_t)m * C;
    float sacc = 0.f;
    for (int c = lane; c < C; c += 64) sacc += r[c];
    const float mean = wave_sum(sacc) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = r[c] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + LN_EPS);
    for (int c = lane; c < C; c += 64) y[(size_t)m * C + c] = (r[c] - mean) * rstd * gam[c] + bet[c];
    if (lane == 0) { stats[2 * (size_t)m] = mean; stats[2 * (size_t)m + 1] = rstd; }
}
// dx (ADDED to dx_acc) = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;  xhat_dy[m][c] = dy xhat (for d_gamma = colsum)
__global__ __launch_bounds__(256) void t_ln_bwd_kernel(const float *x, const float *gam, const float *stats, const float *dy, float *dx_acc,
                                                        float *xhat_dy, int M, int C) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const float mean = stats[2 * (size_t)m], rstd = stats[2 * (size_t)m + 1];
    const float *r = x + (size_t)m * C, *d = dy + (size_t)m * C;
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (r[c] - mean) * rstd, g = d[c] * gam[c];
        sg += g; sgx = fmaf(g, xh, sgx);
        xhat_dy[(size_t)m * C + c] = d[c] * xh;
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (r[c] - mean) * rstd, g = d[c] * gam[c];
        dx_acc[(size_t)m * C + c] += rstd * (g - mg - xh * mgx);
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
static bool t_attn_launch(bool bwd, const float *qkv, const float *table, float *out, const float *d_out, float *d_qkv, float *d_table,
                          int B, TAttnGeom g, hipStream_t s) {
    const int Wt = g.ws * g.ws, HD = g.C / g.heads, LD = HD + 1;
    if (Wt > 128 || HD % 4 != 0) return false;
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
    const size_t nMC = (size_t)M * C, nMH = (size_t)M * H;
    TAttnGeom g{a.res, a.ws, a.shift, a.heads, C};
    // ---- forward ----
    t_gemm(false, true, a.emb, NOISE_EMB, a.W.aff_w, NOISE_EMB, a.W.aff_b, a.aff, 2 * C, B, 2 * C, NOISE_EMB, false, s);     // params = affine(emb)
    hipLaunchKernelGGL(t_modulate_fwd_kernel, dim3(t_blocks(nMC)), dim3(256), 0, s, a.x_in, a.aff, a.x_mod, T, C, nMC);
    hipLaunchKernelGGL(t_ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.x_mod, a.W.n1_w, a.W.n1_b, a.xn1, a.stats1, M, C);
    t_gemm(false, true, a.xn1, C, a.W.qkv_w, C, a.W.qkv_b, a.qkv, 3 * C, M, 3 * C, C, false, s);
    if (!t_attn_launch(false, a.qkv, a.W.rpb, a.att, nullptr, nullptr, nullptr, B, g, s)) return false;
    if (hipMemcpyAsync(a.x1, a.x_mod, sizeof(float) * nMC, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;
    t_gemm(false, true, a.att, C, a.W.proj_w, C, a.W.proj_b, a.x1, C, M, C, C, true, s);                                      // x1 = shortcut + proj(att)
    hipLaunchKernelGGL(t_ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.x1, a.W.n2_w, a.W.n2_b, a.xn2, a.stats2, M, C);
    t_gemm(false, true, a.xn2, C, a.W.fc1_w, C, a.W.fc1_b, a.pre, H, M, H, C, false, s);
    hipLaunchKernelGGL(t_gelu_fwd_kernel, dim3(t_blocks(nMH)), dim3(256), 0, s, a.pre, a.hid, nMH);
    if (hipMemcpyAsync(a.x_out, a.x1, sizeof(float) * nMC, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;
    t_gemm(false, true, a.hid, H, a.W.fc2_w, H, a.W.fc2_b, a.x_out, C, M, C, H, true, s);                                      // x_out = x1 + fc2(gelu(fc1(ln2)))
    if (!a.grad_out) return hipGetLastError() == hipSuccess;
    return train_block_backward(a, s);
}

// backward of the block from the tensors train_block's forward left in `a` (d_x1 accumulates in a.d_x1; scratch t_mc, t_mc2 [M,C],
// t_mh [M,H], t_m3c [M,3C]; a.grad_out is read, a.grad_in / a.grad_emb / a.G.* are written)
bool train_block_backward(const TrainBlockArgs &a, hipStream_t s) {
    const int B = a.B, T = a.res * a.res, C = a.C, M = B * T, H = a.hidden;
    const size_t nMC = (size_t)M * C, nMH = (size_t)M * H;
    TAttnGeom g{a.res, a.ws, a.shift, a.heads, C};
    const float *dY = a.grad_out;
    // MLP: x_out = x1 + hid W2^T + b2
    t_gemm(true, false, dY, C, a.hid, H, nullptr, a.G.fc2_w, H, C, H, M, false, s);            // dW2 [C,H] = dY^T hid
    t_colsum(dY, C, a.G.fc2_b, M, C, s);
    t_gemm(false, false, dY, C, a.W.fc2_w, H, nullptr, a.t_mh, H, M, H, C, false, s);          // d_hid = dY W2
    hipLaunchKernelGGL(t_gelu_bwd_kernel, dim3(t_blocks(nMH)), dim3(256), 0, s, a.pre, a.t_mh, a.t_mh, nMH);   // d_pre
    t_gemm(true, false, a.t_mh, H, a.xn2, C, nullptr, a.G.fc1_w, C, H, C, M, false, s);         // dW1 [H,C] = d_pre^T xn2
    t_colsum(a.t_mh, H, a.G.fc1_b, M, H, s);
    t_gemm(false, false, a.t_mh, H, a.W.fc1_w, C, nullptr, a.t_mc, C, M, C, H, false, s);       // d_xn2 = d_pre W1
    if (hipMemcpyAsync(a.d_x1, dY, sizeof(float) * nMC, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;   // residual branch
    hipLaunchKernelGGL(t_ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.x1, a.W.n2_w, a.stats2, a.t_mc, a.d_x1, a.t_mc2, M, C);
    t_colsum(a.t_mc2, C, a.G.n2_w, M, C, s);                                                     // d_gamma2 = colsum(d_xn2 * xhat)
    t_colsum(a.t_mc, C, a.G.n2_b, M, C, s);                                                      // d_beta2 = colsum(d_xn2)
    // attention half: x1 = x_mod + att Wp^T + bp
    t_gemm(true, false, a.d_x1, C, a.att, C, nullptr, a.G.proj_w, C, C, C, M, false, s);
    t_colsum(a.d_x1, C, a.G.proj_b, M, C, s);
    t_gemm(false, false, a.d_x1, C, a.W.proj_w, C, nullptr, a.t_mc, C, M, C, C, false, s);      // d_att
    if (hipMemsetAsync(a.G.rpb, 0, sizeof(float) * (size_t)(2 * a.ws - 1) * (2 * a.ws - 1) * a.heads, s) != hipSuccess) return false;
    if (!t_attn_launch(true, a.qkv, a.W.rpb, nullptr, a.t_mc, a.t_m3c, a.G.rpb, B, g, s)) return false;   // d_qkv, d_table
    t_gemm(true, false, a.t_m3c, 3 * C, a.xn1, C, nullptr, a.G.qkv_w, C, 3 * C, C, M, false, s);
    t_colsum(a.t_m3c, 3 * C, a.G.qkv_b, M, 3 * C, s);
    t_gemm(false, false, a.t_m3c, 3 * C, a.W.qkv_w, C, nullptr, a.t_mc, C, M, C, 3 * C, false, s);   // d_xn1
    // d_xmod = d_x1 (shortcut) + LN1 backward
    hipLaunchKernelGGL(t_ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, a.x_mod, a.W.n1_w, a.stats1, a.t_mc, a.d_x1, a.t_mc2, M, C);
    t_colsum(a.t_mc2, C, a.G.n1_w, M, C, s);
    t_colsum(a.t_mc, C, a.G.n1_b, M, C, s);
    // modulate: x_mod = silu(shift + x (1 + scale)); params = emb Wa^T + ba
    t_modulate(a.x_in, a.aff, a.d_x1, a.grad_in, a.d_aff, B, T, C, true, s);
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
    hipLaunchKernelGGL(t_ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, gam, bet, y, stats, M, C);
}
void t_ln_bwd(const float *x, const float *gam, const float *stats, const float *dy, float *dx_acc, float *xhat_dy, int M, int C, hipStream_t s) {
    hipLaunchKernelGGL(t_ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, gam, stats, dy, dx_acc, xhat_dy, M, C);
}
void t_modulate(const float *x, const float *aff, const float *dy, float *out, float *d_aff, int B, int T, int C, bool bwd, hipStream_t s) {
    if (bwd) {
        const int rows_per = max(64, (T + 63) / 64), chunks = (T + rows_per - 1) / rows_per;
        const size_t need = (size_t)B * chunks * 2 * C;
        if (need > g_cs_cap) {
            if (g_cs_scratch) (void)hipFree(g_cs_scratch);
            g_cs_cap = 0; g_cs_scratch = nullptr;
            if (hipMalloc((void **)&g_cs_scratch, sizeof(double) * need) == hipSuccess) g_cs_cap = need;
        }
        if (g_cs_cap < need) { fprintf(stderr, "dsg: t_modulate: out of memory\n"); abort(); }
        hipLaunchKernelGGL(t_modulate_bwd_kernel, dim3((C + 63) / 64, chunks, B), dim3(256), 0, s, x, aff, dy, out, g_cs_scratch, T, C, rows_per, chunks);
        hipLaunchKernelGGL(t_modulate_bwd_final_kernel, dim3((B * 2 * C + 255) / 256), dim3(256), 0, s, g_cs_scratch, d_aff, B, C, chunks);
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
// partial[b] = sum of squares of this block's grid-stride share (double); fixed block count -> deterministic
__global__ __launch_bounds__(256) void t_sumsq_kernel(const float *g, size_t n, double *partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// norms[t] = sqrt(sum of tensor t's partials) as fp32 (torch: per-tensor fp32 norms, then the norm of the norms); out[0] = total norm,
// out[1] = clip coefficient min(1, max_norm / (total + 1e-6))
__global__ void t_clip_coef_kernel(const double *partial, int n_tensors, int per, float max_norm, float *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double tot = 0.0;
    for (int t = 0; t < n_tensors; t++) {
        double s = 0.0;
        for (int b = 0; b < per; b++) s += partial[(size_t)t * per + b];
        const float nt = sqrtf((float)s);
        tot += (double)nt * (double)nt;
    }
    const float total = sqrtf((float)tot);
    out[0] = total;
    out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (total + 1e-6f)) : 1.0f;
}
__global__ void t_adam_kernel(float *p, float *g, float *m, float *v, size_t n, const float *clip, float lr, float b1, float b2, float eps, float wd,
                              float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gr = g[i] * clip[1];
    g[i] = gr;   // clip_grad_norm_ scales the gradients in place
    if (wd != 0.f) gr = fmaf(wd, p[i], gr);
    const float mi = m[i] + (gr - m[i]) * (1.0f - b1);               // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gr * gr;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}
__global__ void t_ema_kernel(float *ema, const float *p, size_t n, float decay) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ema[i] = ema[i] + (p[i] - ema[i]) * (1.0f - decay);   // ma.lerp_(current, 1 - decay)
}
bool t_adam_step(int n, float *const *params, float *const *grads, float *const *m, float *const *v, const int64_t *numel, int step, float lr,
                 float b1, float b2, float eps, float wd, float max_norm, float *host_total_norm, hipStream_t s) {
    constexpr int PER = 64;
    double *partial = nullptr;
    float *coef = nullptr;
    if (hipMalloc((void **)&partial, sizeof(double) * (size_t)n * PER) != hipSuccess) return false;
    if (hipMalloc((void **)&coef, sizeof(float) * 2) != hipSuccess) { (void)hipFree(partial); return false; }
    for (int t = 0; t < n; t++) hipLaunchKernelGGL(t_sumsq_kernel, dim3(PER), dim3(256), 0, s, grads[t], (size_t)numel[t], partial + (size_t)t * PER);
    hipLaunchKernelGGL(t_clip_coef_kernel, dim3(1), dim3(1), 0, s, partial, n, PER, max_norm, coef);
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    for (int t = 0; t < n; t++)
        hipLaunchKernelGGL(t_adam_kernel, dim3(t_blocks((size_t)numel[t])), dim3(256), 0, s, params[t], grads[t], m[t], v[t], (size_t)numel[t], coef, lr,
                           b1, b2, eps, wd, bc1, bc2s);
    bool ok = hipGetLastError() == hipSuccess;
    if (host_total_norm) ok = ok && hipMemcpyAsync(host_total_norm, coef, sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(partial); (void)hipFree(coef);
    return ok;
}
bool t_ema_update(int n, float *const *ema, const float *const *params, const int64_t *numel, float decay, hipStream_t s) {
    for (int t = 0; t < n; t++)
        hipLaunchKernelGGL(t_ema_kernel, dim3(t_blocks((size_t)numel[t])), dim3(256), 0, s, ema[t], params[t], (size_t)numel[t], decay);
    return hipGetLastError() == hipSuccess;
}

}  // namespace dsg
