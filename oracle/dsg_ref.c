/*
 * dsg_ref.c -- ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C fp32 restatement of the DiffuseSG sampling hot path:
 *   denoiser forward  R/model/diffusesg/diffusesg.py:765-830 (DiffuseSG.forward)
 *   preconditioning   R/model/precond/precond.py:65-110      (NodeAdjPrecond.forward)
 *   reverse loop      R/runner/mcmc_sampler/edm.py:291-445   (NodeAdjEDMSampler.sample)
 * (R/ = /root/reference/DiffuseSG/).  Every function cites the lines it follows.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against vectors
 * produced by the reference's own Python modules (tools/gen_golden.py -> tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product
 * (diffusesg_amd/, libdsg.so) never links, loads or calls it.
 *
 * Layouts (same as the reference): adj [B,C_adj,N,N], node [B,N,C_node], flags [B,N] u8,
 * activations token-major [T,C] per sample with token t = i*res + j.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NOISE_EMB 512
#define MAX_LAYERS 8
#define LN_EPS 1e-5f
/* The preconditioning and sampler arithmetic mirrors torch's op-by-op fp32 rounding (each
 * elementwise op rounds once).  A fused multiply-add would, e.g., turn t_hat^2 - t_cur^2 into a
 * non-zero rounding residue when t_hat == t_cur, injecting churn noise the reference does not. */
#define NO_FMA __attribute__((optimize("-ffp-contract=off")))

typedef struct {
    char key[128];
    float *data;     /* as given: [out,in] for linears */
    float *tr;       /* lazily built [in,out] copy for the GEMM inner loop */
    int64_t numel;
} wentry;

typedef struct {
    char name[64];
    float *dst;
    int64_t cap;
} tap_t;

typedef struct dsgref {
    int N, c_adj, c_node, E, L, depths[MAX_LAYERS], heads[MAX_LAYERS], ws, mlp_ratio, self_cond;
    wentry *w;
    int nw, capw;
    tap_t taps[64];
    int ntaps;
    int cur_sample, cur_B;
    char err[256];
    long nfe; /* network forwards executed (all samples of a batch count once) */
} dsgref;

/* ------------------------------------------------------------------------------------------ */
/* handle + weights                                                                            */

dsgref *dsgref_create(const int32_t *c) {
    /* c = [N, c_adj, c_node, E, L, depths[8], heads[8], window, mlp_ratio, self_cond] */
    dsgref *h = (dsgref *)calloc(1, sizeof(dsgref));
    h->N = c[0]; h->c_adj = c[1]; h->c_node = c[2]; h->E = c[3]; h->L = c[4];
    for (int i = 0; i < MAX_LAYERS; i++) { h->depths[i] = c[5 + i]; h->heads[i] = c[13 + i]; }
    h->ws = c[21]; h->mlp_ratio = c[22]; h->self_cond = c[23];
    return h;
}

void dsgref_destroy(dsgref *h) {
    if (!h) return;
    for (int i = 0; i < h->nw; i++) { free(h->w[i].data); free(h->w[i].tr); }
    free(h->w);
    free(h);
}

const char *dsgref_last_error(dsgref *h) { return h->err; }
long dsgref_nfe(dsgref *h) { return h->nfe; }

int dsgref_set_weight(dsgref *h, const char *key, const float *data, int64_t numel) {
    for (int i = 0; i < h->nw; i++)
        if (!strcmp(h->w[i].key, key)) {
            free(h->w[i].data); free(h->w[i].tr); h->w[i].tr = NULL;
            h->w[i].data = (float *)malloc(sizeof(float) * numel);
            memcpy(h->w[i].data, data, sizeof(float) * numel);
            h->w[i].numel = numel;
            return 0;
        }
    if (h->nw == h->capw) {
        h->capw = h->capw ? 2 * h->capw : 256;
        h->w = (wentry *)realloc(h->w, sizeof(wentry) * h->capw);
    }
    wentry *e = &h->w[h->nw++];
    memset(e, 0, sizeof(*e));
    snprintf(e->key, sizeof(e->key), "%s", key);
    e->data = (float *)malloc(sizeof(float) * numel);
    memcpy(e->data, data, sizeof(float) * numel);
    e->numel = numel;
    return 0;
}

static wentry *W(dsgref *h, const char *fmt, const char *prefix) {
    char key[192];
    snprintf(key, sizeof(key), fmt, prefix);
    for (int i = 0; i < h->nw; i++)
        if (!strcmp(h->w[i].key, key)) return &h->w[i];
    snprintf(h->err, sizeof(h->err), "missing weight %s", key);
    fprintf(stderr, "dsg_ref: missing weight %s\n", key);
    abort();
}

int dsgref_tap(dsgref *h, const char *name, float *dst, int64_t cap) {
    if (h->ntaps >= 64) return -1;
    snprintf(h->taps[h->ntaps].name, 64, "%s", name);
    h->taps[h->ntaps].dst = dst;
    h->taps[h->ntaps].cap = cap;
    h->ntaps++;
    return 0;
}
void dsgref_clear_taps(dsgref *h) { h->ntaps = 0; }

/* copy one sample's stage output [numel] into slot cur_sample of a registered tap */
static void tap(dsgref *h, const char *name, const float *src, int64_t numel) {
    for (int i = 0; i < h->ntaps; i++)
        if (!strcmp(h->taps[i].name, name)) {
            int64_t off = (int64_t)h->cur_sample * numel;
            if (off + numel <= h->taps[i].cap) memcpy(h->taps[i].dst + off, src, sizeof(float) * numel);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* primitive ops                                                                               */

static inline float silu_f(float x) { return x / (1.0f + expf(-x)); }
static inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); } /* exact erf GELU, nn.GELU default */

/* y[M,N] = x[M,K] @ Wt[K,N] + b with Wt stored [in,out].  Every output element is the k-ordered chain
 * y = b; y += x[m][k] * Wt[k][n] (k = 0..K-1), exactly as a row-by-row axpy would form it; the 4-row x 24-column register
 * tile only changes how often the weights are streamed (this is also the CPU baseline bench.py times). */
#define LIN_RB 4
#define LIN_NB 24
static void linear_kn(const float *wt, const float *bias, const float *x, float *y, int M, int K, int N) {
    const int mblocks = (M + LIN_RB - 1) / LIN_RB;
#pragma omp parallel for schedule(static)
    for (int mb = 0; mb < mblocks; mb++) {
        const int m0 = mb * LIN_RB, rows = (M - m0 < LIN_RB) ? M - m0 : LIN_RB;
        for (int n0 = 0; n0 < N; n0 += LIN_NB) {
            const int nb = (N - n0 < LIN_NB) ? N - n0 : LIN_NB;
            if (rows == LIN_RB && nb == LIN_NB) {
                float acc[LIN_RB][LIN_NB];
                for (int r = 0; r < LIN_RB; r++)
                    for (int n = 0; n < LIN_NB; n++) acc[r][n] = bias ? bias[n0 + n] : 0.f;
                const float *x0 = x + (size_t)m0 * K, *x1 = x0 + K, *x2 = x1 + K, *x3 = x2 + K;
                for (int k = 0; k < K; k++) {
                    const float *wr = wt + (size_t)k * N + n0;
                    const float a0 = x0[k], a1 = x1[k], a2 = x2[k], a3 = x3[k];
                    for (int n = 0; n < LIN_NB; n++) {
                        const float wv = wr[n];
                        acc[0][n] += a0 * wv; acc[1][n] += a1 * wv; acc[2][n] += a2 * wv; acc[3][n] += a3 * wv;
                    }
                }
                for (int r = 0; r < LIN_RB; r++) memcpy(y + (size_t)(m0 + r) * N + n0, acc[r], sizeof(float) * LIN_NB);
            } else {
                for (int r = 0; r < rows; r++) {
                    float *yr = y + (size_t)(m0 + r) * N + n0;
                    const float *xr = x + (size_t)(m0 + r) * K;
                    for (int n = 0; n < nb; n++) yr[n] = bias ? bias[n0 + n] : 0.f;
                    for (int k = 0; k < K; k++) {
                        const float a = xr[k];
                        const float *wr = wt + (size_t)k * N + n0;
                        for (int n = 0; n < nb; n++) yr[n] += a * wr[n];
                    }
                }
            }
        }
    }
}

/* y[M,N] = x[M,K] @ W[N,K]^T + b   (torch.nn.functional.linear; diffusesg.py:14-16 etc.) */
static void linear(wentry *w, const float *bias, const float *x, float *y, int M, int K, int N) {
    if (!w->tr) {
#pragma omp critical(dsgref_tr)
        if (!w->tr) { /* samples of a batch may run in parallel: build the transposed copy once */
            float *t = (float *)malloc(sizeof(float) * (size_t)K * N);
            for (int n = 0; n < N; n++)
                for (int k = 0; k < K; k++) t[(size_t)k * N + n] = w->data[(size_t)n * K + k];
            w->tr = t;
        }
    }
    linear_kn(w->tr, bias, x, y, M, K, N);
}

/* y = x @ W (weight already stored [in,out]); used for ConvTranspose2d k=1 (diffusesg.py:706) */
static void linear_in_out(const float *w_in_out, const float *bias, const float *x, float *y, int M, int K, int N) {
    linear_kn(w_in_out, bias, x, y, M, K, N);
}

/* nn.LayerNorm over the last dim, eps=1e-5, biased variance */
static void layer_norm(const float *g, const float *b, const float *x, float *y, int M, int C) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++) {
        const float *xr = x + (size_t)m * C;
        float *yr = y + (size_t)m * C;
        float mean = 0.f;
        for (int c = 0; c < C; c++) mean += xr[c];
        mean /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; c++) { float d = xr[c] - mean; var += d * d; }
        var /= (float)C;
        const float rstd = 1.0f / sqrtf(var + LN_EPS);
        for (int c = 0; c < C; c++) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
    }
}

/* x = silu(shift + x*(scale+1)), (scale,shift) = chunk(affine(emb)) (diffusesg.py:238-240, 574-576) */
static void modulate_silu(dsgref *h, const char *prefix, const float *emb, float *x, int T, int C) {
    float *params = (float *)malloc(sizeof(float) * 2 * C);
    linear(W(h, "%s.affine.weight", prefix), W(h, "%s.affine.bias", prefix)->data, emb, params, 1, NOISE_EMB, 2 * C);
    const float *scale = params, *shift = params + C;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; t++)
        for (int c = 0; c < C; c++) {
            float *v = &x[(size_t)t * C + c];
            *v = silu_f(shift[c] + *v * (scale[c] + 1.0f));
        }
    free(params);
}

/* WindowAttention.forward + window_partition/reverse + cyclic shift
 * (diffusesg.py:108-139, 28-57, 246-271).  xin/xout token-major [res*res, C]. */
static void window_attention(dsgref *h, const char *prefix, const float *xin, float *xout,
                             int res, int C, int heads, int ws, int shift) {
    const int Wt = ws * ws, nwr = res / ws, nW = nwr * nwr, hd = C / heads, T = res * res;
    const float scale = 1.0f / sqrtf((float)hd); /* head_dim ** -0.5 */
    float *qkv = (float *)malloc(sizeof(float) * (size_t)T * 3 * C);
    /* qkv is a per-token linear, so it commutes with the window gather */
    linear(W(h, "%s.attn.qkv.weight", prefix), W(h, "%s.attn.qkv.bias", prefix)->data, xin, qkv, T, C, 3 * C);
    const float *table = W(h, "%s.attn.relative_position_bias_table", prefix)->data; /* [(2ws-1)^2, heads] */
    float *att_out = (float *)malloc(sizeof(float) * (size_t)T * C);
#pragma omp parallel for schedule(static) collapse(2)
    for (int w = 0; w < nW; w++)
        for (int hh = 0; hh < heads; hh++) {
            int tok[256];
            int region[256];
            float s[256];
            const int wi = w / nwr, wj = w % nwr;
            for (int p = 0; p < Wt; p++) {
                const int si = wi * ws + p / ws, sj = wj * ws + p % ws; /* coords in the rolled image */
                /* torch.roll(x, -shift): rolled[i] = x[(i+shift) % res] (diffusesg.py:248) */
                tok[p] = ((si + shift) % res) * res + ((sj + shift) % res);
                /* region id of the rolled coordinate (diffusesg.py:209-221) */
                const int ri = si < res - ws ? 0 : (si < res - shift ? 1 : 2);
                const int rj = sj < res - ws ? 0 : (sj < res - shift ? 1 : 2);
                region[p] = 3 * ri + rj;
            }
            for (int p = 0; p < Wt; p++) {
                const float *q = qkv + (size_t)tok[p] * 3 * C + hh * hd;
                float mx = -INFINITY;
                for (int r = 0; r < Wt; r++) {
                    const float *k = qkv + (size_t)tok[r] * 3 * C + C + hh * hd;
                    float acc = 0.f;
                    for (int d = 0; d < hd; d++) acc += (q[d] * scale) * k[d];
                    const int pi = p / ws, pj = p % ws, qi = r / ws, qj = r % ws;
                    const int idx = (pi - qi + ws - 1) * (2 * ws - 1) + (pj - qj + ws - 1);
                    acc += table[(size_t)idx * heads + hh];
                    if (shift > 0 && region[p] != region[r]) acc += -100.0f;
                    s[r] = acc;
                    mx = fmaxf(mx, acc);
                }
                float sum = 0.f;
                for (int r = 0; r < Wt; r++) { s[r] = expf(s[r] - mx); sum += s[r]; }
                const float inv = 1.0f / sum;
                float *o = att_out + (size_t)tok[p] * C + hh * hd;
                for (int d = 0; d < hd; d++) o[d] = 0.f;
                for (int r = 0; r < Wt; r++) {
                    const float *v = qkv + (size_t)tok[r] * 3 * C + 2 * C + hh * hd;
                    const float pr = s[r] * inv;
                    for (int d = 0; d < hd; d++) o[d] += pr * v[d];
                }
            }
        }
    /* proj is per token too; un-window + roll back map each token to itself (tok[] used on both sides) */
    linear(W(h, "%s.attn.proj.weight", prefix), W(h, "%s.attn.proj.bias", prefix)->data, att_out, xout, T, C, C);
    free(qkv);
    free(att_out);
}

/* SwinTransformerBlock.forward (diffusesg.py:232-277), in place on x [T,C] */
static void swin_block(dsgref *h, const char *prefix, const float *emb, float *x, int res, int C, int heads,
                       int ws_cfg, int shift_cfg) {
    const int T = res * res;
    int ws = ws_cfg, shift = shift_cfg;
    if (res <= ws_cfg) { ws = res; shift = 0; } /* diffusesg.py:189-192 */
    modulate_silu(h, prefix, emb, x, T, C);     /* the modulated tensor is also the shortcut (:242) */
    float *y = (float *)malloc(sizeof(float) * (size_t)T * C);
    float *a = (float *)malloc(sizeof(float) * (size_t)T * C);
    layer_norm(W(h, "%s.norm1.weight", prefix)->data, W(h, "%s.norm1.bias", prefix)->data, x, y, T, C);
    window_attention(h, prefix, y, a, res, C, heads, ws, shift);
    for (size_t i = 0; i < (size_t)T * C; i++) x[i] += a[i];
    /* FFN (:275) */
    layer_norm(W(h, "%s.norm2.weight", prefix)->data, W(h, "%s.norm2.bias", prefix)->data, x, y, T, C);
    const int Hd = h->mlp_ratio * C;
    float *hid = (float *)malloc(sizeof(float) * (size_t)T * Hd);
    linear(W(h, "%s.mlp.fc1.weight", prefix), W(h, "%s.mlp.fc1.bias", prefix)->data, y, hid, T, C, Hd);
    for (size_t i = 0; i < (size_t)T * Hd; i++) hid[i] = gelu_f(hid[i]);
    linear(W(h, "%s.mlp.fc2.weight", prefix), W(h, "%s.mlp.fc2.bias", prefix)->data, hid, a, T, Hd, C);
    for (size_t i = 0; i < (size_t)T * C; i++) x[i] += a[i];
    free(y); free(a); free(hid);
}

/* PatchMerging.forward (diffusesg.py:314-335): [res*res,C] -> [(res/2)^2, 2C] */
static float *patch_merging(dsgref *h, const char *prefix, const float *x, int res, int C) {
    const int r2 = res / 2, T2 = r2 * r2;
    float *cat = (float *)malloc(sizeof(float) * (size_t)T2 * 4 * C);
    for (int i = 0; i < r2; i++)
        for (int j = 0; j < r2; j++)
            for (int q = 0; q < 4; q++) {
                /* x0=(0::2,0::2) x1=(1::2,0::2) x2=(0::2,1::2) x3=(1::2,1::2) */
                const int di = q & 1, dj = q >> 1;
                memcpy(cat + ((size_t)(i * r2 + j) * 4 + q) * C,
                       x + (size_t)((2 * i + di) * res + (2 * j + dj)) * C, sizeof(float) * C);
            }
    float *nrm = (float *)malloc(sizeof(float) * (size_t)T2 * 4 * C);
    layer_norm(W(h, "%s.norm.weight", prefix)->data, W(h, "%s.norm.bias", prefix)->data, cat, nrm, T2, 4 * C);
    float *out = (float *)malloc(sizeof(float) * (size_t)T2 * 2 * C);
    linear(W(h, "%s.reduction.weight", prefix), NULL, nrm, out, T2, 4 * C, 2 * C);
    free(cat); free(nrm);
    return out;
}

/* PatchBreakup.forward, skip_connection=True (diffusesg.py:374-403): [res*res,D] -> [(2res)^2, D/4] */
static float *patch_breakup(dsgref *h, const char *prefix, const float *x, int res, int D) {
    const int T = res * res, Co = D / 4, R = 2 * res;
    float *y = (float *)malloc(sizeof(float) * (size_t)T * D);
    float *n = (float *)malloc(sizeof(float) * (size_t)T * D);
    linear(W(h, "%s.pre_linear.weight", prefix), NULL, x, y, T, D, D);
    layer_norm(W(h, "%s.norm.weight", prefix)->data, W(h, "%s.norm.bias", prefix)->data, y, n, T, D);
    float *sc = (float *)malloc(sizeof(float) * (size_t)4 * T * Co);
    for (int i = 0; i < res; i++)
        for (int j = 0; j < res; j++)
            for (int q = 0; q < 4; q++) {
                const int di = q & 1, dj = q >> 1; /* x_out[:,0::2,0::2]=x0, [1::2,0::2]=x1, [0::2,1::2]=x2, [1::2,1::2]=x3 */
                memcpy(sc + (size_t)((2 * i + di) * R + (2 * j + dj)) * Co,
                       n + ((size_t)(i * res + j) * 4 + q) * Co, sizeof(float) * Co);
            }
    float *pn = (float *)malloc(sizeof(float) * (size_t)4 * T * Co);
    layer_norm(W(h, "%s.post_norm.weight", prefix)->data, W(h, "%s.post_norm.bias", prefix)->data, sc, pn, 4 * T, Co);
    float *out = (float *)malloc(sizeof(float) * (size_t)4 * T * Co);
    linear(W(h, "%s.post_linear.weight", prefix), NULL, pn, out, 4 * T, Co, Co);
    free(y); free(n); free(sc); free(pn);
    return out;
}

/* PositionalEmbedding + map_layer0/1 (diffusesg.py:507-513, 768-771) */
static void noise_embedding_pe(dsgref *h, float c_noise, float *emb /*[512]*/, float *pe_out /*[E] or NULL*/) {
    const int E = h->E, half = E / 2;
    float *pe = (float *)malloc(sizeof(float) * E);
    for (int k = 0; k < half; k++) {
        const float f = powf(1.0f / 10000.0f, (float)k / (float)half);
        const float v = c_noise * f;
        pe[k] = cosf(v);
        pe[half + k] = sinf(v);
    }
    float *t0 = (float *)malloc(sizeof(float) * NOISE_EMB);
    linear(W(h, "%s", "map_layer0.weight"), W(h, "%s", "map_layer0.bias")->data, pe, t0, 1, E, NOISE_EMB);
    for (int i = 0; i < NOISE_EMB; i++) t0[i] = silu_f(t0[i]);
    linear(W(h, "%s", "map_layer1.weight"), W(h, "%s", "map_layer1.bias")->data, t0, emb, 1, NOISE_EMB, NOISE_EMB);
    for (int i = 0; i < NOISE_EMB; i++) emb[i] = silu_f(emb[i]);
    if (pe_out) memcpy(pe_out, pe, sizeof(float) * E);
    free(pe); free(t0);
}
static void noise_embedding(dsgref *h, float c_noise, float *emb) { noise_embedding_pe(h, c_noise, emb, NULL); }
/* stand-alone: PositionalEmbedding rows [rows, E] and the mapped embedding [rows, 512] (test hook for survey fixture G1) */
int dsgref_noise_embed(dsgref *h, int rows, const float *c_noise, float *pe, float *emb) {
    for (int r = 0; r < rows; r++) noise_embedding_pe(h, c_noise[r], emb + (size_t)r * NOISE_EMB, pe + (size_t)r * h->E);
    return 0;
}

/* DiffuseSG.forward for ONE sample (diffusesg.py:765-830 with forward_features :739-763) */
static void forward_one(dsgref *h, const float *adj, const float *node, const uint8_t *flags, float c_noise,
                        const float *sc_adj, const float *sc_node, float *out_adj, float *out_node) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node, E = h->E, L = h->L, T0 = N * N;
    float emb[NOISE_EMB];
    noise_embedding(h, c_noise, emb);

    /* input assembly (:784-802): channel order [sc_adj, adj, sc_node_row, node_row, sc_node_col, node_col] */
    const int nsc = h->self_cond ? 2 : 1;
    const int Cin = nsc * (Ca + 2 * Cn);
    float *in = (float *)calloc((size_t)T0 * Cin, sizeof(float)); /* token-major [T0, Cin] */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            float *r = in + (size_t)(i * N + j) * Cin;
            int c = 0;
            if (h->self_cond)
                for (int a = 0; a < Ca; a++) r[c++] = sc_adj ? sc_adj[((size_t)a * N + i) * N + j] : 0.f;
            for (int a = 0; a < Ca; a++) r[c++] = adj[((size_t)a * N + i) * N + j];
            const float m = (flags[i] && flags[j]) ? 1.f : 0.f; /* mask_adjs on the node part only (:800) */
            if (h->self_cond)
                for (int a = 0; a < Cn; a++) r[c++] = m * (sc_node ? sc_node[(size_t)i * Cn + a] : 0.f);
            for (int a = 0; a < Cn; a++) r[c++] = m * node[(size_t)i * Cn + a];
            if (h->self_cond)
                for (int a = 0; a < Cn; a++) r[c++] = m * (sc_node ? sc_node[(size_t)j * Cn + a] : 0.f);
            for (int a = 0; a < Cn; a++) r[c++] = m * node[(size_t)j * Cn + a];
        }

    /* PatchEmbed (:562-577): 1x1 conv + LN + modulate */
    float *x = (float *)malloc(sizeof(float) * (size_t)T0 * E);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)T0 * E);
    linear(W(h, "%s", "patch_embed.proj.weight"), W(h, "%s", "patch_embed.proj.bias")->data, in, tmp, T0, Cin, E);
    layer_norm(W(h, "%s", "patch_embed.norm.weight")->data, W(h, "%s", "patch_embed.norm.bias")->data, tmp, x, T0, E);
    modulate_silu(h, "patch_embed", emb, x, T0, E);
    tap(h, "patch_embed", x, (int64_t)T0 * E);
    free(in); free(tmp);

    /* encoder (:745-748) */
    float *skips[MAX_LAYERS];
    char prefix[128], name[64];
    int res = N, C = E;
    for (int l = 0; l < L; l++) {
        for (int j = 0; j < h->depths[l]; j++) {
            snprintf(prefix, sizeof(prefix), "down_layers.%d.blocks.%d", l, j);
            swin_block(h, prefix, emb, x, res, C, h->heads[l], h->ws, (j % 2 == 0) ? 0 : h->ws / 2);
            snprintf(name, sizeof(name), "down%d.block%d", l, j);
            tap(h, name, x, (int64_t)res * res * C);
        }
        if (l < L - 1) {
            snprintf(prefix, sizeof(prefix), "down_layers.%d.downsample", l);
            float *y = patch_merging(h, prefix, x, res, C);
            free(x);
            x = y; res /= 2; C *= 2;
        }
        snprintf(name, sizeof(name), "down%d", l);
        tap(h, name, x, (int64_t)res * res * C);
        skips[l] = (float *)malloc(sizeof(float) * (size_t)res * res * C);
        memcpy(skips[l], x, sizeof(float) * (size_t)res * res * C);
    }
    /* decoder (:751-756): first up layer discards the deepest skip; others cat([x, skip]) then PatchBreakup */
    for (int i = 0; i < L; i++) {
        const int lvl = L - 1 - i;
        if (i > 0) {
            const int T = res * res;
            float *cat = (float *)malloc(sizeof(float) * (size_t)T * 2 * C);
            const float *sk = skips[lvl]; /* skips.pop(): after i pops the top is skips[L-1-i] */
            for (int t = 0; t < T; t++) {
                memcpy(cat + (size_t)t * 2 * C, x + (size_t)t * C, sizeof(float) * C);
                memcpy(cat + (size_t)t * 2 * C + C, sk + (size_t)t * C, sizeof(float) * C);
            }
            snprintf(prefix, sizeof(prefix), "up_layers.%d.upsample", i);
            float *y = patch_breakup(h, prefix, cat, res, 2 * C);
            free(cat); free(x);
            x = y; res *= 2; C /= 2;
            snprintf(name, sizeof(name), "up%d.upsample", i);
            tap(h, name, x, (int64_t)res * res * C);
        }
        for (int j = 0; j < h->depths[lvl]; j++) {
            snprintf(prefix, sizeof(prefix), "up_layers.%d.blocks.%d", i, j);
            swin_block(h, prefix, emb, x, res, C, h->heads[lvl], h->ws, (j % 2 == 0) ? 0 : h->ws / 2);
            snprintf(name, sizeof(name), "up%d.block%d", i, j);
            tap(h, name, x, (int64_t)res * res * C);
        }
    }
    for (int l = 0; l < L; l++) free(skips[l]);

    /* final norm + read_out (:758-761): ConvTranspose2d(k=1) weight is [in,out]; Conv2d weight is [out,in] */
    float *y = (float *)malloc(sizeof(float) * (size_t)T0 * E);
    float *z = (float *)malloc(sizeof(float) * (size_t)T0 * E);
    layer_norm(W(h, "%s", "norm.weight")->data, W(h, "%s", "norm.bias")->data, x, y, T0, E);
    linear_in_out(W(h, "%s", "read_out.0.weight")->data, W(h, "%s", "read_out.0.bias")->data, y, z, T0, E, E);
    linear(W(h, "%s", "read_out.1.weight"), W(h, "%s", "read_out.1.bias")->data, z, y, T0, E, E);
    linear(W(h, "%s", "read_out.2.weight"), W(h, "%s", "read_out.2.bias")->data, y, z, T0, E, E);
    float *rep = z; /* shared_rep, token-major [T0,E] */
    tap(h, "read_out", rep, (int64_t)T0 * E);

    /* adjacency head (:806-809) */
    float *ha = (float *)malloc(sizeof(float) * (size_t)T0 * E);
    float *oa = (float *)malloc(sizeof(float) * (size_t)T0 * Ca);
    linear(W(h, "%s", "readout_adj_mlp.fc1.weight"), W(h, "%s", "readout_adj_mlp.fc1.bias")->data, rep, ha, T0, E, E);
    for (size_t i = 0; i < (size_t)T0 * E; i++) ha[i] = gelu_f(ha[i]);
    linear(W(h, "%s", "readout_adj_mlp.fc2.weight"), W(h, "%s", "readout_adj_mlp.fc2.bias")->data, ha, oa, T0, E, Ca);
    for (int a = 0; a < Ca; a++)
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) /* mask_adjs (:825) */
                out_adj[((size_t)a * N + i) * N + j] = (flags[i] && flags[j]) ? oa[(size_t)(i * N + j) * Ca + a] : 0.f;

    /* node head (:812-818): masked mean over j, divided by N (padded size) */
    float *pool = (float *)calloc((size_t)N * E, sizeof(float));
    for (int i = 0; i < N; i++) {
        if (!flags[i]) continue;
        for (int j = 0; j < N; j++) {
            if (!flags[j]) continue;
            const float *r = rep + (size_t)(i * N + j) * E;
            for (int e = 0; e < E; e++) pool[(size_t)i * E + e] += r[e];
        }
    }
    for (size_t i = 0; i < (size_t)N * E; i++) pool[i] /= (float)N;
    float *hn = (float *)malloc(sizeof(float) * (size_t)N * E);
    linear(W(h, "%s", "readout_node_mlp.fc1.weight"), W(h, "%s", "readout_node_mlp.fc1.bias")->data, pool, hn, N, E, E);
    for (size_t i = 0; i < (size_t)N * E; i++) hn[i] = gelu_f(hn[i]);
    linear(W(h, "%s", "readout_node_mlp.fc2.weight"), W(h, "%s", "readout_node_mlp.fc2.bias")->data, hn, out_node, N, E, Cn);
    for (int i = 0; i < N; i++) /* mask_nodes (:822) */
        if (!flags[i]) for (int c = 0; c < Cn; c++) out_node[(size_t)i * Cn + c] = 0.f;

    free(x); free(y); free(z); free(ha); free(oa); free(pool); free(hn);
}

/* DiffuseSG.forward for a batch.  sc_* may be NULL (zeros, diffusesg.py:791-793). */
int dsgref_forward(dsgref *h, int B, const float *adj, const float *node, const uint8_t *flags, const float *c_noise,
                   const float *sc_adj, const float *sc_node, float *out_adj, float *out_node) {
    const size_t sa = (size_t)h->c_adj * h->N * h->N, sn = (size_t)h->N * h->c_node;
    h->cur_B = B;
    /* samples are independent: with a batch of >= 4 (and no debug taps) run one sample per thread; the
     * parallel loops inside the ops then run serially (nested parallelism is off by default) */
#pragma omp parallel for schedule(dynamic, 1) if (B >= 4 && h->ntaps == 0)
    for (int b = 0; b < B; b++) {
        if (h->ntaps) h->cur_sample = b;
        forward_one(h, adj + b * sa, node + b * sn, flags + (size_t)b * h->N, c_noise[b],
                    sc_adj ? sc_adj + b * sa : NULL, sc_node ? sc_node + b * sn : NULL,
                    out_adj + b * sa, out_node + b * sn);
    }
    h->nfe++;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* preconditioning: NodeAdjPrecond.forward (precond.py:65-110) with get_preconditioning_params 'edm'
 * (objectives/edm.py:122-126).  `coin` = the outcome of `np.random.rand() < 0.5` (precond.py:90). */

static void mask_adj_inplace(const dsgref *h, float *a, const uint8_t *flags) {
    const int N = h->N;
    for (int c = 0; c < h->c_adj; c++)
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++)
                if (!(flags[i] && flags[j])) a[((size_t)c * N + i) * N + j] = 0.f;
}
static void mask_node_inplace(const dsgref *h, float *n, const uint8_t *flags) {
    for (int i = 0; i < h->N; i++)
        if (!flags[i]) for (int c = 0; c < h->c_node; c++) n[(size_t)i * h->c_node + c] = 0.f;
}

NO_FMA int dsgref_precond(dsgref *h, int B, const float *adj, const float *node, const uint8_t *flags, const float *sigmas,
                   const float *sc_adj, const float *sc_node, int coin, float *out_adj, float *out_node) {
    const int N = h->N;
    const size_t sa = (size_t)h->c_adj * N * N, sn = (size_t)N * h->c_node;
    const float sd = 0.5f; /* sigma_data */
    float *xin = (float *)malloc(sizeof(float) * B * sa), *nin = (float *)malloc(sizeof(float) * B * sn);
    float *cn = (float *)malloc(sizeof(float) * B);
    float *fa = (float *)malloc(sizeof(float) * B * sa), *fn = (float *)malloc(sizeof(float) * B * sn);
    float *sca = NULL, *scn = NULL;
    for (int b = 0; b < B; b++) {
        const float s = sigmas[b];
        const float c_in = 1.0f / sqrtf(sd * sd + s * s);
        cn[b] = logf(s) / 4.0f;
        for (size_t i = 0; i < sa; i++) xin[b * sa + i] = c_in * adj[b * sa + i];
        for (size_t i = 0; i < sn; i++) nin[b * sn + i] = c_in * node[b * sn + i];
    }
    const float *use_sca = sc_adj, *use_scn = sc_node;
    if (h->self_cond && coin) { /* precond.py:90-98: extra forward whose D replaces the self-cond inputs */
        sca = (float *)malloc(sizeof(float) * B * sa);
        scn = (float *)malloc(sizeof(float) * B * sn);
        dsgref_forward(h, B, xin, nin, flags, cn, sc_adj, sc_node, fa, fn);
        for (int b = 0; b < B; b++) {
            const float s = sigmas[b];
            const float c_skip = (sd * sd) / (s * s + sd * sd);
            const float c_out = s * sd / sqrtf(s * s + sd * sd);
            for (size_t i = 0; i < sa; i++) sca[b * sa + i] = c_skip * adj[b * sa + i] + c_out * fa[b * sa + i];
            for (size_t i = 0; i < sn; i++) scn[b * sn + i] = c_skip * node[b * sn + i] + c_out * fn[b * sn + i];
            mask_adj_inplace(h, sca + b * sa, flags + (size_t)b * N);
            mask_node_inplace(h, scn + b * sn, flags + (size_t)b * N);
        }
        use_sca = sca; use_scn = scn;
    }
    dsgref_forward(h, B, xin, nin, flags, cn, use_sca, use_scn, fa, fn);
    for (int b = 0; b < B; b++) {
        const float s = sigmas[b];
        const float c_skip = (sd * sd) / (s * s + sd * sd);
        const float c_out = s * sd / sqrtf(s * s + sd * sd);
        for (size_t i = 0; i < sa; i++) out_adj[b * sa + i] = c_skip * adj[b * sa + i] + c_out * fa[b * sa + i];
        for (size_t i = 0; i < sn; i++) out_node[b * sn + i] = c_skip * node[b * sn + i] + c_out * fn[b * sn + i];
        mask_adj_inplace(h, out_adj + b * sa, flags + (size_t)b * N);
        mask_node_inplace(h, out_node + b * sn, flags + (size_t)b * N);
    }
    free(xin); free(nin); free(cn); free(fa); free(fn); free(sca); free(scn);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* sampler: NodeAdjEDMSampler.sample (edm.py:291-445), schedule='linear', scaling='none' */

typedef struct {
    int32_t num_steps;
    int32_t heun;        /* 1 = 'heun', 0 = 'euler' */
    float S_churn, S_min, S_max, S_noise;
    double sigma_min, sigma_max, rho;
    int32_t max_steps;   /* >0: stop after this many steps (CPU-baseline timing of a bounded sample) */
} dsgref_sampler_cfg;

/* sigma_steps in fp64 (edm.py:84-88) */
void dsgref_sigma_steps(const dsgref_sampler_cfg *c, double *out) {
    const double a = pow(c->sigma_max, 1.0 / c->rho), b = pow(c->sigma_min, 1.0 / c->rho);
    for (int i = 0; i < c->num_steps; i++) {
        const double frac = (c->num_steps > 1) ? (double)i / (double)(c->num_steps - 1) : 0.0;
        out[i] = pow(a + frac * (b - a), c->rho);
    }
}

/*
 * noise_adj [T,B,C_adj,N,N] / noise_node [T,B,N,C_node]: the N(0,1) draws of step i (edm.py:361-364);
 * coins [n_precond_calls] u8: outcome of the coin of each preconditioned call, in call order;
 * gt_* non-NULL: sanity-check mode (edm.py:372-377) -- the denoiser output is replaced by gt.
 */
NO_FMA int dsgref_sample(dsgref *h, const dsgref_sampler_cfg *c, int B, const uint8_t *flags,
                  const float *init_adj, const float *init_node, const float *noise_adj, const float *noise_node,
                  const uint8_t *coins, const float *gt_adj, const float *gt_node, float *out_adj, float *out_node) {
    const int N = h->N, T = c->num_steps;
    const size_t sa = (size_t)B * h->c_adj * N * N, sn = (size_t)B * N * h->c_node;
    const size_t sa1 = sa / B, sn1 = sn / B;
    double *sig = (double *)malloc(sizeof(double) * T);
    dsgref_sigma_steps(c, sig);
    float *t_steps = (float *)malloc(sizeof(float) * (T + 1));
    for (int i = 0; i < T; i++) t_steps[i] = (float)sig[i]; /* round_sigma = identity; cast f64->f32 (:320-323) */
    t_steps[T] = 0.f;
    float *xa = (float *)malloc(sizeof(float) * sa), *xn = (float *)malloc(sizeof(float) * sn);
    float *ha = (float *)malloc(sizeof(float) * sa), *hn = (float *)malloc(sizeof(float) * sn);
    float *da = (float *)malloc(sizeof(float) * sa), *dn = (float *)malloc(sizeof(float) * sn);
    float *d2a = (float *)malloc(sizeof(float) * sa), *d2n = (float *)malloc(sizeof(float) * sn);
    float *dca = (float *)malloc(sizeof(float) * sa), *dcn = (float *)malloc(sizeof(float) * sn);
    float *sca = (float *)malloc(sizeof(float) * sa), *scn = (float *)malloc(sizeof(float) * sn);
    float *sig_b = (float *)malloc(sizeof(float) * B);
    int have_sc = 0, call = 0;
    for (size_t i = 0; i < sa; i++) xa[i] = init_adj[i] * t_steps[0]; /* :346-347 */
    for (size_t i = 0; i < sn; i++) xn[i] = init_node[i] * t_steps[0];
    /* python: min(S_churn / num_steps, np.sqrt(2) - 1) in double; becomes fp32 when multiplied with t_cur */
    const float gamma_on = (float)fmin((double)c->S_churn / (double)T, sqrt(2.0) - 1.0);
    const int nsteps = (c->max_steps > 0 && c->max_steps < T) ? c->max_steps : T;
    for (int i = 0; i < nsteps; i++) {
        const float t_cur = t_steps[i], t_next = t_steps[i + 1];
        const float gamma = (c->S_min <= t_cur && t_cur <= c->S_max) ? gamma_on : 0.f; /* :355 */
        const float t_hat = t_cur + gamma * t_cur;                                      /* :356 */
        const float nz = sqrtf(fmaxf(t_hat * t_hat - t_cur * t_cur, 0.f)) * c->S_noise; /* :361-364 */
        for (size_t k = 0; k < sa; k++) ha[k] = xa[k] + nz * (noise_adj ? noise_adj[(size_t)i * sa + k] : 0.f);
        for (size_t k = 0; k < sn; k++) hn[k] = xn[k] + nz * (noise_node ? noise_node[(size_t)i * sn + k] : 0.f);
        for (int b = 0; b < B; b++) {
            mask_adj_inplace(h, ha + b * sa1, flags + (size_t)b * N);
            mask_node_inplace(h, hn + b * sn1, flags + (size_t)b * N);
            sig_b[b] = t_hat;
        }
        const float hstep = t_next - t_hat; /* :369 */
        if (gt_adj) { memcpy(da, gt_adj, sizeof(float) * sa); memcpy(dn, gt_node, sizeof(float) * sn); }
        else dsgref_precond(h, B, ha, hn, flags, sig_b, have_sc ? sca : NULL, have_sc ? scn : NULL,
                            coins ? coins[call++] : 0, da, dn);
        for (int b = 0; b < B; b++) {
            mask_adj_inplace(h, da + b * sa1, flags + (size_t)b * N);
            mask_node_inplace(h, dn + b * sn1, flags + (size_t)b * N);
        }
        const float inv = 1.0f / t_hat; /* sigma_deriv/sigma with sigma(t)=t (:384-385) */
        for (size_t k = 0; k < sa; k++) dca[k] = inv * ha[k] - inv * da[k];
        for (size_t k = 0; k < sn; k++) dcn[k] = inv * hn[k] - inv * dn[k];
        for (int b = 0; b < B; b++) {
            mask_adj_inplace(h, dca + b * sa1, flags + (size_t)b * N);
            mask_node_inplace(h, dcn + b * sn1, flags + (size_t)b * N);
        }
        const float t_prime = t_hat + hstep; /* alpha = 1 (:391) */
        if (!c->heun || i == T - 1) {        /* :394-396 */
            for (size_t k = 0; k < sa; k++) xa[k] = ha[k] + hstep * dca[k];
            for (size_t k = 0; k < sn; k++) xn[k] = hn[k] + hstep * dcn[k];
        } else {
            /* stage 2 re-evaluates at (x_hat, sigma(t_hat)); only the self-cond inputs change (:400-405) */
            if (!gt_adj) {
                if (h->self_cond) { memcpy(sca, da, sizeof(float) * sa); memcpy(scn, dn, sizeof(float) * sn); have_sc = 1; }
                dsgref_precond(h, B, ha, hn, flags, sig_b, have_sc ? sca : NULL, have_sc ? scn : NULL,
                               coins ? coins[call++] : 0, d2a, d2n);
            } else { memcpy(d2a, gt_adj, sizeof(float) * sa); memcpy(d2n, gt_node, sizeof(float) * sn); }
            for (int b = 0; b < B; b++) {
                mask_adj_inplace(h, d2a + b * sa1, flags + (size_t)b * N);
                mask_node_inplace(h, d2n + b * sn1, flags + (size_t)b * N);
            }
            const float invp = 1.0f / t_prime; /* d_prime uses x_prime and t_prime (:414-417) */
            for (size_t k = 0; k < sa; k++) {
                const float xp = ha[k] + hstep * dca[k];
                const float dp = invp * xp - invp * d2a[k];
                xa[k] = ha[k] + hstep * (0.5f * dca[k] + 0.5f * dp);
            }
            for (size_t k = 0; k < sn; k++) {
                const float xp = hn[k] + hstep * dcn[k];
                const float dp = invp * xp - invp * d2n[k];
                xn[k] = hn[k] + hstep * (0.5f * dcn[k] + 0.5f * dp);
            }
            memcpy(da, d2a, sizeof(float) * sa); memcpy(dn, d2n, sizeof(float) * sn);
        }
        for (int b = 0; b < B; b++) { /* :421-422 */
            mask_adj_inplace(h, xa + b * sa1, flags + (size_t)b * N);
            mask_node_inplace(h, xn + b * sn1, flags + (size_t)b * N);
        }
        if (h->self_cond) { memcpy(sca, da, sizeof(float) * sa); memcpy(scn, dn, sizeof(float) * sn); have_sc = 1; } /* :423-424 */
    }
    memcpy(out_adj, xa, sizeof(float) * sa);
    memcpy(out_node, xn, sizeof(float) * sn);
    free(sig); free(t_steps); free(xa); free(xn); free(ha); free(hn); free(da); free(dn); free(d2a); free(d2n);
    free(dca); free(dcn); free(sca); free(scn); free(sig_b);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* post-decode of 'bits' samples                                                */
/* ------------------------------------------------------------------------------------------ */
/* R/runner/sampler/sampler_node_adj.py:222-285 ('bits' branches of _decode_node / _decode_adj) with
 * bin2dec of R/utils/attribute_code.py:319-328 (channel 0 is the most significant bit) and the bbox
 * handling of :201-209.  Step by step as the reference: clamp(-1,1); > 0 -> bit; mask; sum(bit * 2^k);
 * mask; clamp to [0, n_type-1]; zero the adjacency diagonal.  node_bits = C_node - 4 when the last four
 * node channels are the bounding box (out_bbox != NULL), else C_node.
 * Pinned by tests/golden/decode.npz (tests/test_oracle_golden.py::test_decode_bits_matches_reference). */
int dsgref_decode_bits(dsgref *h, int B, const float *adj, const float *node, const uint8_t *flags, int n_adj_type,
                       int n_node_type, int node_bits, int32_t *out_adj, int32_t *out_node, float *out_bbox) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node;
    for (int b = 0; b < B; b++) {
        const uint8_t *f = flags + (size_t)b * N;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                long v = 0;
                for (int c = 0; c < Ca; c++) {
                    float x = adj[(((size_t)b * Ca + c) * N + i) * N + j];
                    x = fminf(fmaxf(x, -1.0f), 1.0f);                       /* :243 clamp */
                    const int bit = (x > 0.0f) && f[i] && f[j];             /* :245-247, :268-269 */
                    v += (long)bit << (Ca - 1 - c);                         /* bin2dec: mask = 2^(num_bits-1 .. 0) */
                }
                if (!(f[i] && f[j])) v = 0;                                 /* :275 mask_adjs */
                if (v < 0) v = 0;
                if (v > n_adj_type - 1) v = n_adj_type - 1;                 /* :275 clamp */
                if (i == j) v = 0;                                          /* :281 remove self loops */
                out_adj[((size_t)b * N + i) * N + j] = (int32_t)v;
            }
        for (int i = 0; i < N; i++) {
            long v = 0;
            for (int c = 0; c < node_bits; c++) {
                float x = node[((size_t)b * N + i) * Cn + c];
                x = fminf(fmaxf(x, -1.0f), 1.0f);                           /* :223 */
                const int bit = (x > 0.0f) && f[i];                         /* :225-230 */
                v += (long)bit << (node_bits - 1 - c);
            }
            if (!f[i]) v = 0;
            if (v < 0) v = 0;
            if (v > n_node_type - 1) v = n_node_type - 1;                   /* :232 */
            out_node[(size_t)b * N + i] = (int32_t)v;
            if (out_bbox)
                for (int c = 0; c < 4; c++) {                               /* :201-209 */
                    const float x = node[((size_t)b * N + i) * Cn + (Cn - 4) + c];
                    out_bbox[((size_t)b * N + i) * 4 + c] = f[i] ? x * 0.5f + 0.5f : 0.0f;
                }
        }
    }
    return 0;
}

/* Number of OpenMP threads the oracle's loops use from now on (bench.py's cpu_baseline on a multi-rank run: the launcher starts the
 * ranks with OMP_NUM_THREADS=1, the baseline leg wants the host's cores).  Returns the number in effect. */
int dsgref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* The other two attribute encodings of the post-decode (`--node_encoding` / `--edge_encoding` = 'one_hot' | 'ddpm'):
 * _decode_node / _decode_adj (R/runner/sampler/sampler_node_adj.py:222-285) hand the clamped sample to
 * attribute_converter(in_encoding=..., out_encoding='int', flag_in_ddpm_range=True) (R/utils/attribute_code.py:13-58).
 *   one_hot  :225 / :245 threshold to +-1, mask -> attribute_one_hot_to_int (:212-237): (x+1)/2, mask, argmax over the channels
 *            (torch.argmax: the first maximal value), mask.
 *   ddpm     attribute_ddpm_to_int (:121-177): _get_intervals builds, with Python floats (doubles), L = 2.0/(k-1),
 *            center_i = -1.0 + i*L, min_i = center_i - 0.5*L (i = 0: -inf), max_i = center_i + L*0.5 (i = k-1: +inf);
 *            _assign_integers fills -1 and, for i ascending, sets i where (x > min_i) & (x <= max_i) -- torch compares an fp32
 *            tensor with a Python scalar in fp32, i.e. the thresholds are rounded to float first; then mask_nodes / mask_adjs.
 * enc: 0 'bits' (above), 1 'one_hot', 2 'ddpm'.  The adjacency diagonal is zeroed last (:279-283).  The threshold arithmetic goes
 * through volatile doubles, so center_i is a rounded product plus a rounded sum like CPython's (no FMA contraction).  Pinned by tests/golden/decode_enc.npz (the imported attribute_converter itself). */
static int dsgref_ddpm_class(float x, int k) {
    if (x != x) return -1;                                       /* NaN lies in no interval: the fill value stays */
    x = fminf(fmaxf(x, -1.0f), 1.0f);                            /* sampler_node_adj.py:223 / :243 */
    const double L = 2.0 / (double)(k - 1);
    int out = -1;
    for (int i = 0; i < k; i++) {
        volatile double prod = (double)i * L;                    /* (volatile: one rounding per Python operation) */
        volatile double center = -1.0 + prod;
        volatile double half = L * 0.5;
        volatile double dlo = center - half, dhi = center + half;
        const float lo = i == 0 ? -INFINITY : (float)dlo, hi = i == k - 1 ? INFINITY : (float)dhi;
        if (x > lo && x <= hi) out = i;
    }
    return out;
}
int dsgref_decode(dsgref *h, int B, const float *adj, const float *node, const uint8_t *flags, int enc_adj, int enc_node, int n_adj_type,
                  int n_node_type, int node_chans, int32_t *out_adj, int32_t *out_node, float *out_bbox) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node;
    if (enc_adj == 0 && enc_node == 0)
        return dsgref_decode_bits(h, B, adj, node, flags, n_adj_type, n_node_type, node_chans, out_adj, out_node, out_bbox);
    if (n_adj_type < 2 || n_node_type < 2) return -1;
    /* bits on one side only: decode both as bits first, then overwrite the other side below */
    if (enc_adj == 0 || enc_node == 0) {
        int32_t *ta = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * N * N), *tn = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * N);
        dsgref_decode_bits(h, B, adj, node, flags, n_adj_type, n_node_type, enc_node == 0 ? node_chans : 1, ta, tn, NULL);
        if (enc_adj == 0) memcpy(out_adj, ta, sizeof(int32_t) * (size_t)B * N * N);
        if (enc_node == 0) memcpy(out_node, tn, sizeof(int32_t) * (size_t)B * N);
        free(ta); free(tn);
    }
    for (int b = 0; b < B; b++) {
        const uint8_t *f = flags + (size_t)b * N;
        if (enc_adj != 0)
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) {
                    const int ok = f[i] && f[j];
                    int v = 0;
                    if (enc_adj == 1) {
                        float best = -1.0f;
                        for (int c = 0; c < Ca; c++) {
                            float x = adj[(((size_t)b * Ca + c) * N + i) * N + j];
                            x = fminf(fmaxf(x, -1.0f), 1.0f);                       /* :243 */
                            float y = x > 0.0f ? 1.0f : -1.0f;                       /* :245-246 */
                            if (!ok) y = 0.0f;                                       /* :247 mask_adjs */
                            y = (y + 1.0f) / 2.0f;                                   /* attribute_code.py:225 */
                            if (!ok) y = 0.0f;                                       /* :226 */
                            if (y > best) { best = y; v = c; }                       /* :230 argmax, first maximum */
                        }
                    } else v = dsgref_ddpm_class(adj[(((size_t)b * Ca) * N + i) * N + j], n_adj_type);
                    if (!ok) v = 0;                                                  /* :232 / :172-173 mask */
                    if (i == j) v = 0;                                               /* sampler_node_adj.py:281 */
                    out_adj[((size_t)b * N + i) * N + j] = v;
                }
        for (int i = 0; i < N; i++) {
            if (enc_node != 0) {
                int v = 0;
                if (enc_node == 1) {
                    float best = -1.0f;
                    for (int c = 0; c < node_chans; c++) {
                        float x = node[((size_t)b * N + i) * Cn + c];
                        x = fminf(fmaxf(x, -1.0f), 1.0f);                           /* :223 */
                        float y = x > 0.0f ? 1.0f : -1.0f;                           /* :225 */
                        if (!f[i]) y = 0.0f;                                         /* :226 mask_nodes */
                        y = (y + 1.0f) / 2.0f;
                        if (!f[i]) y = 0.0f;
                        if (y > best) { best = y; v = c; }
                    }
                } else v = dsgref_ddpm_class(node[((size_t)b * N + i) * Cn], n_node_type);
                if (!f[i]) v = 0;
                out_node[(size_t)b * N + i] = v;
            }
            if (out_bbox)
                for (int c = 0; c < 4; c++) {                                       /* :201-209 */
                    const float x = node[((size_t)b * N + i) * Cn + (Cn - 4) + c];
                    out_bbox[((size_t)b * N + i) * 4 + c] = f[i] ? x * 0.5f + 0.5f : 0.0f;
                }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* training-time objective and loss, forward only (first half)                   */
/* ------------------------------------------------------------------------------------------ */
/* NodeAdjEDMObjectiveGenerator.get_input_output (R/runner/objectives/edm.py:160-180, :239-281) with sigma_dist = precond =
 * 'edm' and symmetric_noise = False (learning_utils.py:25-29):
 *   sigma_b  = exp(rnd_b * P_std + P_mean),  P_mean = -1.2, P_std = 1.2          (:176-177)
 *   weight_b = (sigma^2 + sigma_data^2) / (sigma * sigma_data)^2, sigma_data = .5 (:178)
 *   noisy_adj  = mask_adjs(clean_adj + eps_adj * sigma_b)   (graph_utils.add_sym_normal_noise, non_symmetric=True: :133-148)
 *   noisy_node = clean_node + mask_nodes(eps_node * sigma_b)                      (:246-254)
 * rnd [B], eps_adj, eps_node are the N(0,1) draws in the reference's draw order (sigma, adjacency, node). */
NO_FMA int dsgref_train_inputs(dsgref *h, int B, const float *clean_adj, const float *clean_node, const uint8_t *flags,
                               const float *rnd, const float *eps_adj, const float *eps_node, float *sigmas, float *weights,
                               float *noisy_adj, float *noisy_node) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node;
    for (int b = 0; b < B; b++) {
        const float s = expf(rnd[b] * 1.2f + (-1.2f));
        sigmas[b] = s;
        const float sd = s * 0.5f;
        weights[b] = (s * s + 0.25f) / (sd * sd);
        const uint8_t *f = flags + (size_t)b * N;
        for (int c = 0; c < Ca; c++)
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) {
                    const size_t k = (((size_t)b * Ca + c) * N + i) * N + j;
                    noisy_adj[k] = (f[i] && f[j]) ? clean_adj[k] + eps_adj[k] * s : 0.0f;
                }
        for (int i = 0; i < N; i++)
            for (int c = 0; c < Cn; c++) {
                const size_t k = ((size_t)b * N + i) * Cn + c;
                noisy_node[k] = clean_node[k] + (f[i] ? eps_node[k] * s : 0.0f);
            }
    }
    return 0;
}

/* The bounding-box term of ONE node, every iou_loss_type of the trainer (R/runner/trainer/trainer_node_adj.py:130-153):
 *   net_output_x_bbox = (x[..., -4:] + 1) / 2; box_convert(cxcywh -> xyxy).clamp(0, 1); then
 *   0 'iou'          -(box_iou(a, b).diag())^2            torchvision.ops.boxes.box_iou: wh = (min(rb) - max(lt)).clamp(min=0),
 *                                                         inter = w h, iou = inter / (area_a + area_b - inter)
 *   1 'giou'         generalized_box_iou_loss(reduction='none')
 *   2 'giou_squared' that ** 2
 *   3 'diou'         distance_box_iou_loss(reduction='none')
 *   4 'ciou'         complete_box_iou_loss(reduction='none')
 * torchvision is an un-vendored dependency of the reference (setup/requirements.txt pins torch 2.3.0, whose companion is torchvision
 * 0.18) and is absent from this image: types 1-4 restate torchvision/ops/{giou,diou,ciou}_loss.py and _utils._loss_inter_union as
 * published (eps = 1e-7):
 *   _loss_inter_union: xkis1 = max(x1, x1g) ..., intsct = (xkis2 - xkis1)(ykis2 - ykis1) where (ykis2 > ykis1) & (xkis2 > xkis1) else 0,
 *                      union = (x2 - x1)(y2 - y1) + (x2g - x1g)(y2g - y1g) - intsct
 *   giou: iou = intsct / (union + eps); enclosing box (xc1, yc1, xc2, yc2); area_c = (xc2 - xc1)(yc2 - yc1);
 *         loss = 1 - (iou - (area_c - union) / (area_c + eps))
 *   diou: loss = 1 - iou + ((x_p - x_g)^2 + (y_p - y_g)^2) / ((xc2 - xc1)^2 + (yc2 - yc1)^2 + eps), centres x_p = (x2 + x1)/2 ...
 *   ciou: v = 4/pi^2 (atan(w_pred / h_pred) - atan(w_gt / h_gt))^2; alpha = v / (1 - iou + v + eps) under no_grad; loss = diou + alpha v
 * Parity for these four is therefore "unpinned" against torchvision itself; the fixtures (tests/golden/iou_losses.npz) come from a
 * torch restatement of the same published code under the reference's autograd engine (tools/gen_golden.py::gen_iou_losses).
 * grad (optional, 4 doubles): d loss / d the four raw node channels x[-4:], as autograd derives it -- clamp passes the gradient where
 * min <= v <= max, maximum/minimum to the larger/smaller argument and half to each on a tie, the masked intersection only where open. */
static float box_term(const float *pred4, const float *tgt4, int type, double *grad) {
    float bx[2][4], raw[4];
    for (int q = 0; q < 2; q++) {
        const float *src = q ? tgt4 : pred4;
        const float cx = (src[0] + 1.0f) / 2.0f, cy = (src[1] + 1.0f) / 2.0f, bw = (src[2] + 1.0f) / 2.0f, bh = (src[3] + 1.0f) / 2.0f;
        const float v[4] = {cx - 0.5f * bw, cy - 0.5f * bh, cx + 0.5f * bw, cy + 0.5f * bh};
        for (int t = 0; t < 4; t++) { bx[q][t] = fminf(fmaxf(v[t], 0.0f), 1.0f); if (!q) raw[t] = v[t]; }
    }
    const float x1 = bx[0][0], y1 = bx[0][1], x2 = bx[0][2], y2 = bx[0][3];
    const float x1g = bx[1][0], y1g = bx[1][1], x2g = bx[1][2], y2g = bx[1][3];
    double gc[4] = {0, 0, 0, 0};   /* d loss / d (x1, y1, x2, y2) after the clamp */
    float loss;
    if (type == 0) {
        const float a0 = (x2 - x1) * (y2 - y1), a1 = (x2g - x1g) * (y2g - y1g);
        const float dwf = fminf(x2, x2g) - fmaxf(x1, x1g), dhf = fminf(y2, y2g) - fmaxf(y1, y1g);
        const float iw = fmaxf(dwf, 0.0f), ih = fmaxf(dhf, 0.0f);
        const float inter = iw * ih, iou = inter / (a0 + a1 - inter);
        loss = -(iou * iou);
        if (grad) {
            const double aw = x2 - x1, ah = y2 - y1, dw = dwf, dh = dhf;
            const double I = (double)iw * ih, U = (double)a0 + (double)a1 - I, io = I / U;
            const double g_iou = -2.0 * io;                              /* d/d iou of -(iou^2) */
            const double g_inter = g_iou * (U + I) / (U * U);            /* iou = I/U, dU/dI = -1 */
            const double g_area = g_iou * (-I) / (U * U);
            gc[0] = g_inter * ((dw >= 0 && x1 > x1g) ? -(double)ih : 0.0) + g_area * (-ah);
            gc[1] = g_inter * ((dh >= 0 && y1 > y1g) ? -(double)iw : 0.0) + g_area * (-aw);
            gc[2] = g_inter * ((dw >= 0 && x2 < x2g) ? (double)ih : 0.0) + g_area * ah;
            gc[3] = g_inter * ((dh >= 0 && y2 < y2g) ? (double)iw : 0.0) + g_area * aw;
        }
    } else {
        const float eps = 1e-7f;
        /* _loss_inter_union */
        const float xkis1 = fmaxf(x1, x1g), ykis1 = fmaxf(y1, y1g), xkis2 = fminf(x2, x2g), ykis2 = fminf(y2, y2g);
        const int mask = (ykis2 > ykis1) && (xkis2 > xkis1);
        const float intsct = mask ? (xkis2 - xkis1) * (ykis2 - ykis1) : 0.0f;
        const float uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsct;
        const float iou = intsct / (uni + eps);
        const float xc1 = fminf(x1, x1g), yc1 = fminf(y1, y1g), xc2 = fmaxf(x2, x2g), yc2 = fmaxf(y2, y2g);
        /* derivative pieces with respect to (x1, y1, x2, y2), in double */
#define SEL(a, b) ((a) > (b) ? 1.0 : ((a) == (b) ? 0.5 : 0.0))   /* share of maximum(a, b)'s gradient that goes to a */
        double dI[4] = {0, 0, 0, 0}, dU[4], dIoU[4];
        if (mask) {
            dI[0] = -SEL(x1, x1g) * (double)(ykis2 - ykis1); dI[1] = -SEL(y1, y1g) * (double)(xkis2 - xkis1);
            dI[2] = SEL(x2g, x2) * (double)(ykis2 - ykis1);  dI[3] = SEL(y2g, y2) * (double)(xkis2 - xkis1);
        }
        const double dA[4] = {-(double)(y2 - y1), -(double)(x2 - x1), (double)(y2 - y1), (double)(x2 - x1)};
        const double Ue = (double)uni + (double)eps;
        for (int t = 0; t < 4; t++) { dU[t] = dA[t] - dI[t]; dIoU[t] = (dI[t] * Ue - (double)intsct * dU[t]) / (Ue * Ue); }
        const double cw = xc2 - xc1, ch = yc2 - yc1;
        const double dcw[4] = {-SEL(x1g, x1), 0.0, SEL(x2, x2g), 0.0};   /* xc1 = minimum(x1, x1g): to x1 when x1 < x1g */
        const double dch[4] = {0.0, -SEL(y1g, y1), 0.0, SEL(y2, y2g)};
#undef SEL
        if (type == 1 || type == 2) {
            const float area_c = (xc2 - xc1) * (yc2 - yc1);
            const float miouk = iou - ((area_c - uni) / (area_c + eps));
            const float l = 1.0f - miouk;
            loss = type == 2 ? l * l : l;
            if (grad) {
                const double Ace = (double)area_c + (double)eps, scale = type == 2 ? 2.0 * (double)l : 1.0;
                for (int t = 0; t < 4; t++) {
                    const double dAc = dcw[t] * ch + dch[t] * cw;
                    const double dfrac = ((dAc - dU[t]) * Ace - ((double)area_c - (double)uni) * dAc) / (Ace * Ace);
                    gc[t] = scale * (-dIoU[t] + dfrac);
                }
            }
        } else {
            const float diag2 = ((xc2 - xc1) * (xc2 - xc1)) + ((yc2 - yc1) * (yc2 - yc1)) + eps;
            const float x_p = (x2 + x1) / 2.0f, y_p = (y2 + y1) / 2.0f, x_g = (x1g + x2g) / 2.0f, y_g = (y1g + y2g) / 2.0f;
            const float cd2 = ((x_p - x_g) * (x_p - x_g)) + ((y_p - y_g) * (y_p - y_g));
            float l = 1.0f - iou + (cd2 / diag2);
            float v = 0.0f, alpha = 0.0f, dat = 0.0f;
            const float w_pred = x2 - x1, h_pred = y2 - y1, w_gt = x2g - x1g, h_gt = y2g - y1g;
            if (type == 4) {
                dat = atanf(w_pred / h_pred) - atanf(w_gt / h_gt);
                v = (4.0f / (3.14159265358979323846f * 3.14159265358979323846f)) * (dat * dat);
                alpha = v / (1.0f - iou + v + eps);
                l = l + alpha * v;
            }
            loss = l;
            if (grad) {
                const double D2 = diag2, dx = (double)x_p - (double)x_g, dy = (double)y_p - (double)y_g;
                const double dcd2[4] = {dx, dy, dx, dy};   /* 2 (x_p - x_g) * 1/2 */
                /* dv/dw = 8/pi^2 dat h / (h^2 + w^2), dv/dh = -8/pi^2 dat w / (h^2 + w^2); w = x2 - x1, h = y2 - y1 */
                const double kv = type == 4 ? (double)alpha * (8.0 / (M_PI * M_PI)) * (double)dat / ((double)h_pred * h_pred + (double)w_pred * w_pred) : 0.0;
                const double dv[4] = {-kv * h_pred, kv * w_pred, kv * h_pred, -kv * w_pred};
                for (int t = 0; t < 4; t++) {
                    const double dD2 = 2.0 * cw * dcw[t] + 2.0 * ch * dch[t];
                    gc[t] = -dIoU[t] + (dcd2[t] * D2 - (double)cd2 * dD2) / (D2 * D2) + dv[t];
                }
            }
        }
    }
    if (grad) {
        for (int t = 0; t < 4; t++) if (!(raw[t] >= 0.0f && raw[t] <= 1.0f)) gc[t] = 0.0;   /* clamp(min=0, max=1) */
        /* corners = (cx - w/2, cy - h/2, cx + w/2, cy + h/2), (cx, cy, w, h) = (x + 1) / 2 */
        grad[0] = 0.5 * (gc[0] + gc[2]);
        grad[1] = 0.5 * (gc[1] + gc[3]);
        grad[2] = 0.5 * 0.5 * (gc[2] - gc[0]);
        grad[3] = 0.5 * 0.5 * (gc[3] - gc[1]);
    }
    return loss;
}

/* NodeAdjRainbowLoss.forward(..., reduction='none') (R/loss/rainbow_loss.py:37-101) for [B,C,N,N] / [B,N,C] tensors, plus the
 * bounding-box term of the trainer (R/runner/trainer/trainer_node_adj.py:130-159; iou_type as box_term above):
 *   loss_adj_b  = sum(mask * w_b * (pred - target)^2) / n_b^2 / C_adj  * edge_loss_weight
 *   loss_node_b = sum(mask * w_b * (pred - target)^2) / n_b   / C_node * node_loss_weight
 *                 + iou_w * w_b * sum_i f_i * box_term_i / n_total        (n_total = valid nodes of the WHOLE batch: the
 *                   reference divides by node_flags.view(-1).sum(), trainer_node_adj.py:158)
 * Accumulation in double, so that the fixture comparison measures the device kernel's summation, not this one's. */
int dsgref_rainbow_loss(dsgref *h, int B, const float *pred_adj, const float *pred_node, const float *tgt_adj, const float *tgt_node,
                        const uint8_t *flags, const float *w, float edge_w, float node_w, float iou_w, int iou_type, float *loss_adj,
                        float *loss_node) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node;
    long n_total = 0;
    for (size_t k = 0; k < (size_t)B * N; k++) n_total += flags[k] ? 1 : 0;
    for (int b = 0; b < B; b++) {
        const uint8_t *f = flags + (size_t)b * N;
        int n = 0;
        for (int i = 0; i < N; i++) n += f[i] ? 1 : 0;
        const float wb = w ? w[b] : 1.0f;
        double sa = 0.0, sn = 0.0, si = 0.0;
        for (int c = 0; c < Ca; c++)
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++)
                    if (f[i] && f[j]) {
                        const size_t k = (((size_t)b * Ca + c) * N + i) * N + j;
                        const float d = pred_adj[k] - tgt_adj[k];
                        sa += (double)(d * d * wb);
                    }
        for (int i = 0; i < N; i++)
            if (f[i]) {
                for (int c = 0; c < Cn; c++) {
                    const size_t k = ((size_t)b * N + i) * Cn + c;
                    const float d = pred_node[k] - tgt_node[k];
                    sn += (double)(d * d * wb);
                }
                if (iou_w != 0.0f) { /* trainer_node_adj.py:131-156 */
                    const size_t o = ((size_t)b * N + i) * Cn + (Cn - 4);
                    si += (double)box_term(pred_node + o, tgt_node + o, iou_type, NULL);
                }
            }
        loss_adj[b] = (float)(sa / ((double)n * (double)n) / (double)Ca) * edge_w;
        loss_node[b] = (float)(sn / (double)n / (double)Cn) * node_w + (iou_w != 0.0f ? iou_w * (float)(si / (double)n_total) * wb : 0.0f);
    }
    return 0;
}

/* Backward of  loss = mean_b(loss_adj[b]) + mean_b(loss_node[b])  (trainer_node_adj.py:163, loss terms as in dsgref_rainbow_loss)
 * with respect to the preconditioned outputs D (grad_adj, grad_node), and -- when sigmas is given -- with respect to the raw
 * network outputs:  D = mask(c_skip x + c_out F)  (model/precond/precond.py:101-104)  =>  dL/dF = c_out(sigma_b) * dL/dD on valid
 * entries, 0 elsewhere;  c_out = sigma sigma_data / sqrt(sigma^2 + sigma_data^2)  (runner/objectives/edm.py:122-126).
 *   d loss_adj[b] / d D_adj  = edge_w * 2 w_b (D - target) / (n_b^2 C_adj)          on f_i f_j
 *   d loss_node[b] / d D_node = node_w * 2 w_b (D - target) / (n_b C_node)          on f_i
 *   bbox term  iou_w w_b sum_i f_i box_term_i / n_total:  autograd through  (x+1)/2 -> cxcywh->xyxy -> clamp[0,1] -> the loss of
 *   iou_type (box_term: clamp passes the gradient inside [0,1] inclusive; max/min to the selected argument).
 * Everything carries the 1/B of the batch mean. */
int dsgref_rainbow_loss_backward(dsgref *h, int B, const float *pred_adj, const float *pred_node, const float *tgt_adj,
                                 const float *tgt_node, const uint8_t *flags, const float *w, float edge_w, float node_w, float iou_w,
                                 int iou_type, const float *sigmas, float *grad_adj, float *grad_node, float *grad_F_adj, float *grad_F_node) {
    const int N = h->N, Ca = h->c_adj, Cn = h->c_node;
    long n_total = 0;
    for (size_t k = 0; k < (size_t)B * N; k++) n_total += flags[k] ? 1 : 0;
    for (int b = 0; b < B; b++) {
        const uint8_t *f = flags + (size_t)b * N;
        int n = 0;
        for (int i = 0; i < N; i++) n += f[i] ? 1 : 0;
        const float wb = w ? w[b] : 1.0f;
        const double ka = (double)edge_w * 2.0 * (double)wb / ((double)n * (double)n * (double)Ca) / (double)B;
        const double kn = (double)node_w * 2.0 * (double)wb / ((double)n * (double)Cn) / (double)B;
        const double ki = (double)iou_w * (double)wb / (double)n_total / (double)B;
        float c_out = 1.0f;
        if (sigmas) { const float sg = sigmas[b], sd = 0.5f; c_out = sg * sd / sqrtf(sg * sg + sd * sd); }
        for (int c = 0; c < Ca; c++)
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) {
                    const size_t k = (((size_t)b * Ca + c) * N + i) * N + j;
                    const float g = (f[i] && f[j]) ? (float)(ka * (double)(pred_adj[k] - tgt_adj[k])) : 0.0f;
                    grad_adj[k] = g;
                    if (grad_F_adj) grad_F_adj[k] = c_out * g;
                }
        for (int i = 0; i < N; i++) {
            double gb[4] = {0, 0, 0, 0};
            if (f[i] && iou_w != 0.0f) {
                const size_t o = ((size_t)b * N + i) * Cn + (Cn - 4);
                (void)box_term(pred_node + o, tgt_node + o, iou_type, gb);
                for (int t = 0; t < 4; t++) gb[t] *= ki;
            }
            for (int c = 0; c < Cn; c++) {
                const size_t k = ((size_t)b * N + i) * Cn + c;
                double g = f[i] ? kn * (double)(pred_node[k] - tgt_node[k]) : 0.0;
                if (f[i] && c >= Cn - 4) g += gb[c - (Cn - 4)];
                grad_node[k] = (float)g;
                if (grad_F_node) grad_F_node[k] = c_out * (float)g;
            }
        }
    }
    return 0;
}
