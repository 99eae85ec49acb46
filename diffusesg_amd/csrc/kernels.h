// kernels.h -- launch wrappers of the gfx950 kernels (kernels.hip), used by dsg_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsg {

// Plan validation (dsg_api.cpp: validate_plan): while this is set on the calling thread, every launcher of the forward path checks
// its arguments and picks its kernel as usual but launches nothing.  A launcher that does not cover a shape returns false -- it never
// terminates the process; the host turns that into DSG_ERR_INVALID at plan time (first use of a batch size / dsg_set_option).
extern thread_local bool g_dry_run;

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2,
           // training-form products of the fp32 GEMM only (launch_gemm): GELU whose pre-activation is kept in C2 (the backward needs it);
           // the product multiplied by GELU'(res) -- `res` is that kept pre-activation (the backward through fc2 and the GELU in one pass)
           ACT_GELU_KEEP = 3, ACT_DGELU = 4 };

// geometry of one Swin block's windows
struct WinGeom {
    int res;      // tokens per side
    int ws;       // window side
    int shift;    // cyclic shift (0 or ws/2)
    int heads;
    int C;
};

// C[M,N] = act( pro(A)[M,K] * W[N,K]^T + bias ) (+ res), optionally stored twice.
struct GemmArgs {
    const float *A = nullptr;  int lda = 0;      // rows of K1 floats (k <  K1)
    const float *A2 = nullptr; int lda2 = 0;     // optional second source (k >= K1): concat along K
    int K1 = 0;                                  // == K when A2 is null
    const float *W = nullptr;                    // [N, K] row-major (torch Linear layout)
    const void *Wb = nullptr;                    // the same weight as bf16 [N, K]: non-null selects the bf16-MFMA kernel
    const void *Ws3 = nullptr;                   // the same weight split into three bf16 planes [3][N][K]: split-bf16 kernel
    const float *bias = nullptr;                 // [N] or null
    const float *ln_stats = nullptr;             // [M,2] (mean, rstd): A is replaced by (A-mean)*rstd on the way in; the
                                                 // LayerNorm's gamma/beta must already be folded into W / bias
    const float *res = nullptr; int ldres = 0;   // residual added after the activation, or null
    float *C = nullptr;  int ldc = 0;
    float *C2 = nullptr; int ldc2 = 0;           // optional second destination
    int M = 0, N = 0, K = 0;                     // K % 32 == 0
    int act = ACT_NONE;
    // LayerNorm statistics from per-column-tile partial sums written by the PRODUCING GEMM's epilogue (stats_out below):
    // ln_part [M][ln_nparts][2] = (sum, sum of squares) over 96 columns each; mean/rstd are formed in the prologue
    const float *ln_part = nullptr; int ln_nparts = 0;
    // epilogue extensions of the fp32 kernel (act == ACT_NONE only): the next Swin block's modulate+SiLU
    // x <- silu(shift + x*(1+scale)) applied to the value stored to C (C2 keeps the un-modulated value), with
    // (scale,shift) = mod_aff[b*mod_ld + mod_off + {n, N+n}], b = row / mod_T (mod_ld == 0: one row for the whole batch);
    // and the row statistics of what was stored to C: stats_out [M][ceil(N/96)][2] partial (sum, sumsq) per column tile
    const float *mod_aff = nullptr; int mod_ld = 0, mod_off = 0, mod_T = 1;
    float *stats_out = nullptr;
    // fused QKV projection + window attention (8x8 windows; launch_gemm_qkv_attn): W = qkv weight [3C,K] (q rows pre-scaled),
    // bias = qkv bias [3C]; a block computes q|k|v of ONE head for TWO windows (rows gathered through the window partition /
    // cyclic shift), runs softmax(q k^T + attn_bias) v from LDS and stores the head's 32 output columns to C [M, wg.C]
    // PatchMerging gather in the A path (diffusesg.py:323-332): logical A row m = (b, i, j) of the coarse grid is
    // cat[x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)] of the fine activation g.A [B, a4_res^2, K/4]; no copy is ever
    // materialised.  LayerNorm(4C) statistics come from the four source rows' partials (ln_part, ln_nparts per source row).
    int a4_res = 0;                              // > 0 selects the gather; needs ln_part, no A2
    const float *attn_bias = nullptr;            // [nWt][heads][64][64] key-major, log2(e)-scaled (build_bias_table)
    WinGeom wg{0, 0, 0, 0, 0};
    int attn_batch = 0;
    int batch = 1;                               // > 1: that many independent products in one launch (fp32 kernel, AMODE 2: split-K slices)
    size_t batch_strideA = 0, batch_strideW = 0, batch_strideC = 0;   // floats between consecutive products' operands / outputs
    int a_bf16 = 0, c_bf16 = 0;                  // bf16 mode only: A / C are bf16 tensors (lda / ldc in elements); see kernels_lp.hip
    const float *gelu_tab = nullptr;             // filled in by launch_gemm (table-driven GELU of the split kernel)
    unsigned long long *prof = nullptr;          // measurement mode: {min block start, max block end} in 100 MHz ticks
};
bool launch_gemm(const GemmArgs &g, hipStream_t s);   // false: the argument combination is not built (nothing launched)
// fused LN1 -> QKV -> window attention for 64-token windows (fp32 kernel); returns false if the geometry is not supported
bool launch_gemm_qkv_attn(const GemmArgs &g, hipStream_t s);
void launch_f32_to_bf16(const float *src, void *dst, size_t n, hipStream_t s);

// ---- bf16 block pipeline (kernels_bx.hip; "gemm_bf16" mode): C = epilogue(A[M,K] . W[N,K]^T), A and W bf16 in HBM ----
struct BxGemm {
    const void *A = nullptr;  int lda = 0;        // bf16 [M, K1], leading dimension in elements (multiple of 8)
    const void *A2 = nullptr; int lda2 = 0;       // optional second source for k >= K1 (concat along K); K1 a multiple of the k chunk
    int K1 = 0;
    const void *W = nullptr;                      // bf16 [N, K]
    const float *bias = nullptr;                  // [N] or null
    const float *res = nullptr; int ldres = 0;    // fp32 residual, added after the activation
    float *C = nullptr; int ldc = 0;              // fp32 destination (optional; may alias res)
    void *Cb = nullptr; int ldcb = 0;             // bf16 destination (optional): the stored value, or with ln_out its LayerNorm (no affine)
    void *C2b = nullptr; int ldc2b = 0;           // bf16 copy of the value BEFORE the modulation below (skip connection)
    int M = 0, N = 0, K = 0;                      // K % 8 == 0, N % 4 == 0
    int act = ACT_NONE;                           // ACT_NONE | ACT_GELU
    // the next Swin block's modulate+SiLU applied to the stored value (GemmArgs::mod_* semantics)
    const float *mod_aff = nullptr; int mod_ld = 0, mod_off = 0, mod_T = 1;
    int ln_out = 0;                               // Cb = (v - mean(v)) * rstd(v) over the whole row: N must be 96, 192 or 384 (one tile)
    unsigned long long *dbg = nullptr;            // DSG_BX_EXP == 4 builds only: per-block phase clocks [grid][8] (tools/bx_exp.sh)
};
bool launch_gemm_bx(const BxGemm &g, hipStream_t s);   // false: shape not covered (nothing launched)
// x <- [modulate_next] (x + fc2(GELU(fc1(xn)))) in one kernel (mlp_ratio 4, C in {96, 192, 384}); xn = LayerNorm-2 of x as bf16
// (gamma / beta folded into W1 / b1), W1 [4C, C] and W2 [C, 4C] bf16; xn_out (optional) receives the LayerNorm (out_mode 1) or the
// plain bf16 copy (out_mode 2) of the stored row; mod_* as BxGemm.  false: width not covered.
struct BxMlp {
    const void *xn = nullptr; float *x = nullptr;
    const void *W1 = nullptr; const float *b1 = nullptr; const void *W2 = nullptr; const float *b2 = nullptr;
    void *xn_out = nullptr; int out_mode = 0;
    const float *mod_aff = nullptr; int mod_ld = 0, mod_off = 0, mod_T = 1;
    int M = 0, C = 0;
    int wide8 = 1;   // C = 384: 1 the eight-wave LDS-DMA kernel (needs img), 2 round 3's eight-wave kernel, 0 mlp_bx_kernel<384> (one wave per SIMD)
    unsigned long long *dbg = nullptr;   // measurement runs of the debug entry: [grid][8 waves][8] s_memtime stamps (mlp384d_bx_kernel)
    const void *img96 = nullptr; // C = 96 with proj, no modulate: W1 | W2 | Wp in fragment order (launch_mlp96r_image) -> the LDS-resident kernel
    const void *img2 = nullptr;  // C = 384, wide8 = 3: the chunk-major image of mlp384s_bx_kernel (launch_mlp384s_images; same byte count as img)
    int skew = 0;                // mlp384d_bx_kernel: stagger step of the first-round blocks in shader clocks (0: all start together)
    const void *img = nullptr;   // C = 384: W1 | W2 | Wp pre-arranged for mlp384d_bx_kernel (launch_mlp384_images; mlp384_image_bytes())
    // the attention half's tail in front (att != null; xn is then unused): x <- x + att Wp^T + bp first, its LayerNorm feeds fc1
    const void *att = nullptr, *Wp = nullptr; const float *bp = nullptr;
};
bool launch_mlp_bx(const BxMlp &g, hipStream_t s);
// the C = 384 weights in the order mlp384d_bx_kernel's LDS-DMA ring and fragment reads want them: W1b [1536, 384], W2b [384, 1536],
// Wpb [384, 384] (may be null: no proj stage) row-major bf16 -> img (mlp384_image_bytes() bytes)
size_t mlp384_image_bytes();
void launch_mlp384_images(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s);
size_t mlp96r_image_bytes();
void launch_mlp96r_image(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s);
void launch_mlp384s_images(const void *W1b, const void *W2b, const void *Wpb, void *img, hipStream_t s);
// x fp32 [B*T, C] -> optional in-place modulate+SiLU (aff != null) -> xn bf16: LayerNorm without affine (ln) or the plain copy
void launch_ln_bx(float *x, const float *aff, int aff_ld, int aff_off, void *xn, int B, int T, int C, bool ln, hipStream_t s);
// window attention on bf16 qkv [B*T, 3C] -> bf16 out [B*T, C]; biasT as launch_window_attn; false: window size not covered
bool launch_attn_bx(const void *qkv, const float *biasT, void *out, int B, const WinGeom &g, hipStream_t s);
// QKV projection + window attention fused (no qkv tensor): xn bf16 [B*T, C] (LayerNorm-1 without affine), W bf16 [3C, C] and bias [3C]
// (gamma / beta and the q scale folded in), biasP = the block's bias tiles in accumulator order as fp16 (launch_bias_permute_bx of the
// [nWt][heads][Wp][Wp] table), out bf16 [B*T, C].  false: window size / width not covered.
struct BxQkvAttn { const void *xn = nullptr, *W = nullptr; const float *bias = nullptr; const void *biasP = nullptr; void *out = nullptr; int B = 0; WinGeom g{0, 0, 0, 0, 0};
                   int variant = 0; const void *Wimg = nullptr; const float *biasF = nullptr; unsigned long long *dbg = nullptr; };   // Wimg: launch_qkv_image of W (the wave-per-unit kernel streams it)
void launch_qkv_image(const void *Wb, void *img, int C, int heads, hipStream_t s);
void launch_bias_permute_f32(const float *biasT, float *out, int n_tiles, int Wp, hipStream_t s);   // biasF: the tiles in accumulator order, fp32   // 10 x 10 windows: 0 one wave per (window, head) (qkv_attn_wx_kernel), 1 one block per (window, head) (qkv_attn_bx_kernel)
bool launch_qkv_attn_bx(const BxQkvAttn &a, hipStream_t s);
void launch_bias_permute_bx(const float *biasT, void *out_fp16, int n_tiles, int Wp, hipStream_t s);
void launch_f32_split3(const float *src, void *dst, size_t n, hipStream_t s);
void launch_bf16_to_f32(const void *src, float *dst, size_t n, hipStream_t s);
// device table of the fused MLP's table-driven GELU; must be called once (outside any stream capture) before the first launch
const float *gelu_table();

// x <- x + fc2(GELU(fc1(LN(x)))) with fragment-major packed weights (pack_mlp_weights in dsg_api.cpp); C in {96,192}
// stats_out (optional): [M][1][2] (sum, sumsq) of the rows written back, in the GEMM epilogue's partial format
void launch_fused_mlp(float *x, const float *gam, const float *bet, const float *W1p, const float *b1, const float *W2p,
                      const float *b2, int M, int C, float *stats_out, hipStream_t s);

// input assembly + 1x1 conv + LayerNorm + modulate+SiLU in one kernel (E = 96, in_chans <= 64); false if unsupported
bool launch_fused_patch_embed96(const float *adj, const float *node, const float *sc_adj, const float *sc_node, const int *has_sc,
                                const uint8_t *flags, const float *Wp, const float *bias, const float *gam, const float *bet,
                                const float *aff, int aff_ld, int aff_off, int aff_off2, float *x, int B, int N, int Ca, int Cn,
                                int self_cond, int Kp, hipStream_t s, void *xn = nullptr);   // aff_off2 >= 0: also apply that block's
                                // modulate+SiLU; xn (bf16 [B*N*N, 96]): also the LayerNorm (no affine) of the stored row (bf16 pipeline)
// final LN + folded read_out/adj-head chain + masked adjacency output, and the LN(x) pooling for the node head (E = 96)
// pool_part [B*N][readout_pool_segments(N)][96]: per-tile partial sums, reduced in fixed order into pool_ext [B*N,128]
int readout_pool_segments(int N);
void launch_fused_readout96(const float *x, const float *gam, const float *bet, const float *Wfp, const float *fa, const float *W2p,
                            const float *f2, const uint8_t *flags, float *out_adj, float *pool_part, float *pool_ext, int B, int N,
                            int Ca, hipStream_t s, bool bf16_frags = false);   // bf16_frags: Wfp / W2p are bf16 fragments (dsg_api.cpp: ro_fapb / ro_f2pb)
// whole attention half of a C=96 Swin block in one kernel (modulate+SiLU, LN1, QKV, window attention, proj, residual);
// windows of at most 64 tokens; packed weights from pack_attn_weights in dsg_api.cpp
void launch_fused_attn96(float *x, const float *aff, int aff_ld, int aff_off, const float *gam, const float *bet, const float *Wqp,
                         const float *bqkv, const float *biasT, const float *Wpp, const float *bproj, int B, const WinGeom &g,
                         bool premod, hipStream_t s);   // premod: x is already modulated by the producing kernel
// qkv [B*T, 3C] token order -> out [B*T, C] token order; biasT [nWt][heads][Wp][Wp] (key-major)
bool launch_window_attn(const float *qkv, const float *biasT, float *out, int B, const WinGeom &g, hipStream_t s, bool out_bf16 = false, bool in_bf16 = false);   // false: not built (nothing launched)

// x <- silu(shift + x*(1+scale)), (scale,shift) = aff[b][off .. off+2C); stats of the new rows
void launch_mod_stats(float *x, const float *aff, int aff_ld, int aff_off, float *stats, int B, int T, int C, hipStream_t s);
// stats[m] = (mean, rstd) of row m
void launch_ln_stats(const float *x, float *stats, int M, int C, hipStream_t s);
// y = silu(shift + LN(x)*(1+scale)) (PatchEmbed tail)
void launch_ln_mod(const float *x, const float *g, const float *b, const float *aff, int aff_ld, int aff_off,
                   float *y, int B, int T, int C, hipStream_t s);
// PatchMerging gather + LN(4C): x [B,res*res,C] -> y [B,(res/2)^2,4C]
void launch_merge_ln(const float *x, const float *g, const float *b, float *y, int B, int res, int C, hipStream_t s, bool out_bf16 = false);
// PatchBreakup middle: LN(D) -> 4 chunks scattered 2x2 -> LN(D/4): y [B,res*res,D] -> z [B,(2res)^2,D/4]
void launch_breakup_ln(const float *y, const float *g, const float *b, const float *pg, const float *pb, float *z,
                       int B, int res, int D, hipStream_t s, bool out_bf16 = false);   // out_bf16: y / z are bf16 tensors
// positional embedding of the noise label: pe [B,E]
void launch_noise_pe(const float *c_noise, float *pe, int B, int E, hipStream_t s);
// input assembly: token-major [B*N*N, Kp] (zero padded), channel order of diffusesg.py:792-802
void launch_assemble(const float *adj, const float *node, const float *sc_adj, const float *sc_node, const int *has_sc,
                     const uint8_t *flags, float *out, int B, int N, int Ca, int Cn, int self_cond, int Kp, hipStream_t s);
// adjacency head tail: h [B*N*N, E] -> out [B,Ca,N,N] = mask(h @ W^T + b)
void launch_head_adj(const float *h, const float *W, const float *bias, const uint8_t *flags, float *out,
                     int B, int N, int E, int Ca, hipStream_t s);
// masked mean pooling over j (divided by N): rep [B*N*N, E] -> pool [B*N, E]
void launch_pool(const float *rep, const uint8_t *flags, float *pool, int B, int N, int E, hipStream_t s);
// node head tail: h [B*N, E] -> out [B,N,Cn] = mask(h @ W^T + b)
void launch_head_node(const float *h, const float *W, const float *bias, const uint8_t *flags, float *out,
                      int B, int N, int E, int Cn, hipStream_t s);

// ---- preconditioning / sampler elementwise kernels (adj and node parts handled in one launch) ----
struct StatePtrs { float *adj; float *node; };
struct CStatePtrs { const float *adj; const float *node; };
struct Dims { int B, N, Ca, Cn; };

// c_noise[i] = ln(sigma[i])/4
void launch_cnoise(const float *sigmas, float *c_noise, int n, hipStream_t s);
// in = c_in(sigma[b]) * x ; c_noise[b] = ln(sigma[b])/4
void launch_precond_in(CStatePtrs x, const float *sigmas, StatePtrs in, float *c_noise, Dims d, hipStream_t s);
// D = mask(c_skip*x + c_out*F); optionally stored to a second destination
void launch_precond_out(CStatePtrs x, CStatePtrs F, const float *sigmas, const uint8_t *flags, StatePtrs D, StatePtrs D2,
                        Dims d, hipStream_t s);
// x0 = mask(eps) * scale (gen_init_sample + initial scaling); eps from `init` or Philox(seed, stream): stream 0 is the
// initial sample, stream i+1 the churn noise of step i (launch_churn)
void launch_init(CStatePtrs init, float scale, uint64_t seed, uint32_t stream, const uint8_t *flags, StatePtrs x, Dims d, hipStream_t s);

// ---- reverse-loop kernels driven by a DEVICE step counter, so that one captured step body can be replayed for every step ----
// Per-step scalars, computed on the host up front exactly as before (dsg_sigma_schedule) and uploaded once per sample() call.
struct StepRow { float noise_coef, sigma, inv_t, inv_tp, h; int pad[3]; };
// Per-run control block at a fixed device address: the step counter the kernels index StepRow[] / the noise streams with.
struct RunCtl { int step; int pad; unsigned long long seed; const float *noise_adj; const float *noise_node; };
// x_hat = mask(x + coef[step]*eps), eps = recorded noise[step] (ctl->noise_*) or Philox(seed, step+1)
void launch_churn_tab(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs xhat, Dims d, hipStream_t s);
// in = c_in(sigma[step]) * x
void launch_precond_in_tab(CStatePtrs x, const StepRow *tab, const RunCtl *ctl, StatePtrs in, Dims d, hipStream_t s);
// D = mask(c_skip*x + c_out*F) at sigma[step]
void launch_precond_out_tab(CStatePtrs x, CStatePtrs F, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs D, Dims d,
                            hipStream_t s);
void launch_euler_tab(CStatePtrs xhat, CStatePtrs D, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags, StatePtrs x, Dims d,
                      hipStream_t s);
void launch_heun_tab(CStatePtrs xhat, CStatePtrs D1, CStatePtrs D2, const StepRow *tab, const RunCtl *ctl, const uint8_t *flags,
                     StatePtrs x, Dims d, hipStream_t s);
// dst[0..n) = table[step][0..n)  (the step's (scale,shift) row), and ctl->step += 1
void launch_step_row(const float *table, int n, const RunCtl *ctl, float *dst, hipStream_t s);
void launch_step_advance(RunCtl *ctl, hipStream_t s);
// ---- training-time objective and loss, forward only ----
// sigma_b = exp(rnd_b*1.2 - 1.2), weight_b = (sigma^2 + .25)/(sigma*.5)^2, noisy = clean + mask(eps*sigma) (adjacency: the sum
// is masked); rnd / eps: given tensors, or Philox(seed) streams when null (objectives/edm.py:160-180, :239-281)
void launch_train_inputs(CStatePtrs clean, const float *rnd, CStatePtrs eps, uint64_t seed, const uint8_t *flags, float *sigmas,
                         float *weights, StatePtrs noisy, Dims d, hipStream_t s);
// NodeAdjRainbowLoss(reduction='none') + the trainer's bbox IoU term: per-sample losses [B] (rainbow_loss.py:37-101,
// trainer_node_adj.py:130-159); one block per sample, fixed-order reduction
// iou_type: 0 'iou', 1 'giou', 2 'giou_squared', 3 'diou', 4 'ciou' (trainer_node_adj.py:138-153; DSG_IOU_* of dsg.h)
void launch_rainbow_loss(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w, float edge_w, float node_w, float iou_w,
                         int iou_type, float *loss_adj, float *loss_node, Dims d, hipStream_t s);
// ---- training-time block (train_kernels.hip; correctness-first kernels, not on the sampling path) ----
struct TrainBlockParams {   // the 15 parameter tensors of a SwinTransformerBlock, reference layouts ([out, in] linears)
    float *aff_w, *aff_b, *n1_w, *n1_b, *rpb, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2_w, *n2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct TrainBlockArgs {
    int B, res, ws, shift, heads, C, hidden;
    TrainBlockParams W, G;                 // weights (read) and their gradients (written)
    const float *x_in, *emb, *grad_out;    // [B*T, C], [B, 512], [B*T, C] or null (forward only)
    float *x_out, *grad_in, *grad_emb;     // [B*T, C], [B*T, C], [B, 512]
    // saved forward tensors and scratch (caller-allocated): aff / d_aff [B, 2C]; x_mod, xn1, att, x1, xn2, d_x1, t_mc, t_mc2 [M, C];
    // stats1, stats2 [M, 2]; qkv, t_m3c [M, 3C]; pre, hid, t_mh [M, hidden]
    float *aff, *d_aff, *x_mod, *xn1, *att, *x1, *xn2, *d_x1, *t_mc, *t_mc2, *stats1, *stats2, *qkv, *t_m3c, *pre, *hid, *t_mh;
    // whole-network step: a.aff is already filled (one grouped launch for all blocks) and the affine linear's own backward
    // (dW, db, d_emb from a.d_aff) is left to grouped launches at the end
    bool aff_grouped = false;
};
// a group of small products in one launch (t_gemm_grouped / t_colsum_grouped): C = op(A) op(B) (+ bias)
struct TGemmProb { const float *A, *B, *bias; float *C; int lda, ldb, ldc, M, N, K; };
constexpr int T_GROUP_MAX = 32;
struct TGemmGroup { TGemmProb p[T_GROUP_MAX]; int n = 0; };
void t_gemm_grouped(bool ta, bool tb, bool sum, const TGemmGroup &g, hipStream_t s);   // sum: ONE output = the sum of the products (shared M, N, C)
void t_sum_grouped(const TGemmGroup &g, float *out, int n, hipStream_t s);           // out[i] = sum_z p[z].C[i], i < n (contiguous outputs)
void t_colsum_grouped(const TGemmGroup &g, hipStream_t s);                             // p.C[n] = sum_m p.A[m][n] per problem
bool train_block(const TrainBlockArgs &a, hipStream_t s);            // forward; + backward when a.grad_out is set
bool train_block_backward(const TrainBlockArgs &a, hipStream_t s);   // backward alone, from the tensors the forward left in `a`
// building blocks of the whole-network training step (same file): C (+)= op(A) op(B) (+ bias), column sums, elementwise / row ops
// a_colsum (weight-gradient products, ta && !tb): also out[m] = sum_k A[k][m], the bias gradient that goes with dW = dy^T x
void t_gemm(bool ta, bool tb, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K,
            bool accumulate, hipStream_t s, float *a_colsum = nullptr, const float *res = nullptr, int act = ACT_NONE, float *c2 = nullptr);
// res: C = res + product (same pitch as C); act = ACT_GELU_KEEP: c2 = product + bias, C = GELU(c2); act = ACT_DGELU: C = product * GELU'(res)
// scratch of the training kernels is kept per stream (train_kernels.hip); a failed allocation is reported here, once
bool t_scratch_failed(hipStream_t s, bool clear);
void t_scratch_release();
void t_colsum(const float *X, int ld, float *out, int M, int N, hipStream_t s);
void t_silu(const float *x, const float *dy, float *out, size_t n, bool bwd, hipStream_t s);
void t_gelu(const float *x, const float *dy, float *out, size_t n, bool bwd, hipStream_t s);
void t_add(float *a, const float *b, size_t n, hipStream_t s);
void t_ln_fwd(const float *x, const float *gam, const float *bet, float *y, float *stats, int M, int C, hipStream_t s);
// dx_out = (dx_in ? dx_in : 0) + LayerNorm backward of dy; d_gamma / d_beta (either may be null) = the affine parameters' gradients
void t_ln_bwd(const float *x, const float *gam, const float *stats, const float *dy, const float *dx_in, float *dx_out, float *d_gamma, float *d_beta,
              int M, int C, hipStream_t s);
void t_modulate(const float *x, const float *aff, const float *dy, float *out, float *d_aff, int B, int T, int C, bool bwd, hipStream_t s);
void t_regroup(const float *src, float *dst, int B, int res, int C, bool gather, hipStream_t s);
void t_concat(const float *x, const float *skip, float *cat, size_t M, int C, hipStream_t s);
void t_split(const float *dcat, float *dx, float *dskip_acc, size_t M, int C, hipStream_t s);
void t_adj_out(const float *oa, const uint8_t *flags, float *F, const float *dF, float *d_oa, int B, int N, int Ca, bool bwd, hipStream_t s);
void t_pool_bwd(const float *d_pool, const uint8_t *flags, float *d_rep_acc, int B, int N, int E, hipStream_t s);
void t_rowmask(const float *x, const uint8_t *flags, float *y, size_t M, int C, hipStream_t s);
bool t_adam_step(int n, float *const *params, float *const *grads, float *const *m, float *const *v, const int64_t *numel, int step, float lr,
                 float b1, float b2, float eps, float wd, float max_norm, float *host_total_norm, hipStream_t s);
bool t_ema_update(int n, float *const *ema, const float *const *params, const int64_t *numel, float decay, hipStream_t s);

void launch_rainbow_loss_backward(CStatePtrs pred, CStatePtrs tgt, const uint8_t *flags, const float *w, float edge_w, float node_w,
                                  float iou_w, int iou_type, const float *sigmas, StatePtrs grad, StatePtrs gradF, Dims d, hipStream_t s);
// post-decode of the samples; enc_*: 0 'bits', 1 'one_hot', 2 'ddpm' (DSG_ENC_* of dsg.h); node_chans = attribute channels of a node row
void launch_decode(const float *adj, const float *node, const uint8_t *flags, int enc_adj, int enc_node, int n_adj_type, int n_node_type,
                   int node_chans, int32_t *out_adj, int32_t *out_node, float *out_bbox, Dims d, hipStream_t s);

}  // namespace dsg
