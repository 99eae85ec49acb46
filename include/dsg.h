/*
 * dsg.h -- C ABI of libdsg.so: the MI355X (gfx950) scene-graph diffusion sampler.
 *
 * The reference (ubc-vision/DiffuseSG) has no FFI layer; its seam for this path is three
 * Python callables.  Each entry point below names the reference interface it
 * replaces (R/ = DiffuseSG/ in the reference tree):
 *
 *   dsg_denoise  <->  DiffuseSG.forward                R/model/diffusesg/diffusesg.py:765
 *   dsg_precond  <->  NodeAdjPrecond.forward           R/model/precond/precond.py:65
 *   dsg_sample   <->  NodeAdjEDMSampler.sample         R/runner/mcmc_sampler/edm.py:291
 *   dsg_set_weight / dsg_finalize_weights <-> load_model(strict=True)  R/utils/sampling_utils.py:34
 *   dsg_create   <->  get_network's DiffuseSG(...) ctor kwargs         R/utils/learning_utils.py:47-64
 *
 * Conventions
 *   - return 0 on success, a negative dsg_status otherwise; dsg_last_error(h) has the message.
 *   - plain pointers and sizes only.  All tensor arguments are DEVICE pointers to contiguous
 *     fp32 (flags: uint8) in the reference's layouts:
 *        adj   [B, C_adj, N, N]     node  [B, N, C_node]     flags [B, N] (1 = valid node)
 *     (the reference squeezes singleton channel dims; the memory layout is unchanged).
 *   - the caller owns every I/O buffer; the library owns packed weights and its workspace and
 *     never mutates an input.  One handle per device, not re-entrant.  All work is enqueued on
 *     the caller's stream (`stream` is a hipStream_t passed as void*; NULL = default stream).
 *     Training entries: the saved-activation arena and backward scratch belong to the handle; the small scratch of the
 *     handle-less kernels (split-K partials, column sums, the optimiser's tables) is kept per STREAM, so two streams never
 *     share a buffer; an allocation failure anywhere is reported as DSG_ERR_HIP, never by terminating the process.
 *   - there is no CPU fallback: every entry point fails with DSG_ERR_HIP if the device is missing.
 */
#ifndef DSG_H
#define DSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSG_MAX_LAYERS 8
/* Generation of this header's argument lists; dsg_abi_version() returns the one the library was built with.  A binding compares the two
 * at load time (diffusesg_amd/lib.py does): round 3 inserted `iou_loss_type` into three entries, and a caller built against the older
 * header would otherwise pass shifted arguments without any error.  History: 1-3 = rounds 1-3 (unversioned), 4 = round 4
 * (dsg_decode with an encoding argument, dsg_abi_version, dsg_last_error(NULL)). */
#define DSG_ABI_VERSION 4

typedef enum {
    DSG_OK = 0,
    DSG_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    DSG_ERR_WEIGHTS = -2,   /* unknown key, shape mismatch, or missing tensor at finalize */
    DSG_ERR_HIP = -3,       /* HIP runtime error (no device, out of memory, launch failure) */
    DSG_ERR_STATE = -4      /* call order violated (e.g. denoise before finalize_weights) */
} dsg_status;

typedef struct dsg_handle_s *dsg_handle;

/* Mirrors the kwargs of DiffuseSG(...) as passed by get_network (learning_utils.py:47-64). */
typedef struct dsg_config {
    int32_t max_node_num;              /* img_size (N); patch_size is 1 */
    int32_t c_adj;                     /* out_chans_adj */
    int32_t c_node;                    /* out_chans_node */
    int32_t embed_dim;                 /* feature_dims[-1], 96 */
    int32_t num_layers;                /* len(depths) */
    int32_t depths[DSG_MAX_LAYERS];
    int32_t num_heads[DSG_MAX_LAYERS]; /* [3,6,12,24]; head_dim must be 32 */
    int32_t window_size;
    int32_t mlp_ratio;                 /* 4 */
    int32_t self_condition;            /* train.self_cond */
} dsg_config;

/* Mirrors NodeAdjEDMSampler.__init__ (edm.py:236-255) for discretization='edm', schedule='linear',
 * scaling='none' (the only combination get_mc_sampler builds, sampling_utils.py:15-23). */
typedef struct dsg_sampler_cfg {
    int32_t num_steps;
    int32_t heun;                      /* 1: solver='heun', 0: 'euler' */
    float S_churn, S_min, S_max, S_noise;
    double sigma_min, sigma_max, rho;  /* 0.002, 80, 7 */
    int32_t use_graph;                 /* 1: replay captured hipGraphs: whole step bodies of the loop (churn, preconditioning, the
                                          network forwards, the Euler/Heun update; per-step scalars come from a device table
                                          indexed by a device step counter), or with option "loop_graph" = 0 only the network
                                          forward; 0: every kernel is launched eagerly */
    int32_t reserved;
} dsg_sampler_cfg;

/* Counters of the last dsg_sample call. */
typedef struct dsg_sample_stats {
    int64_t precond_calls;             /* 2T-1 (heun) or T (euler) */
    int64_t net_forwards;              /* precond_calls + number of coins that fired */
    int64_t graph_replays;             /* network forwards that ran from replayed graphs (== net_forwards with use_graph) */
} dsg_sample_stats;

int dsg_create(const dsg_config *cfg, dsg_handle *out);
void dsg_destroy(dsg_handle h);
/* h == NULL: the reason the last dsg_create on the calling thread failed (there is no handle to ask then) */
const char *dsg_last_error(dsg_handle h);
const char *dsg_version(void);
int32_t dsg_abi_version(void);   /* DSG_ABI_VERSION of the header the library was built from */

/* `key` is the reference state-dict name, with or without the 'model.' prefix of the precond
 * wrapper (precond.py:15) and/or the 'module.' prefix of DDP (sampling_utils.py:47-53).
 * `data` is fp32 (int64 for relative_position_index, which is validated and dropped);
 * is_device != 0 means `data` is a device pointer. */
int dsg_set_weight(dsg_handle h, const char *key, const void *data, const int64_t *shape, int32_t ndim,
                   int32_t is_device);
/* Packs derived tables; fails with DSG_ERR_WEIGHTS naming the first missing tensor (strict=True). */
int dsg_finalize_weights(dsg_handle h);
/* Number of state-dict keys expected / names, to let a binding iterate them. */
int dsg_num_weight_keys(dsg_handle h);
const char *dsg_weight_key(dsg_handle h, int32_t i);

/* Device bytes the library holds for batch size B (activations + sampler state). */
size_t dsg_workspace_bytes(dsg_handle h, int32_t B);

/* DiffuseSG.forward: noise_labels[B] = c_noise; sc_adj / sc_node may be NULL (zeros). */
int dsg_denoise(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags,
                const float *noise_labels, const float *sc_adj, const float *sc_node,
                float *out_adj, float *out_node, void *stream);

/* NodeAdjPrecond.forward with precond='edm'.  `coin` is the outcome of the reference's
 * `np.random.rand() < 0.5` (precond.py:90), drawn by the caller. */
int dsg_precond(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags,
                const float *sigmas, const float *sc_adj, const float *sc_node, int32_t coin,
                float *out_adj, float *out_node, void *stream);

/* NodeAdjEDMSampler.sample.
 *   init_adj/init_node  NULL: drawn on device (Philox, `seed`), masked (gen_init_sample, edm.py:257-289)
 *   noise_adj [T,B,C_adj,N,N], noise_node [T,B,N,C_node]  NULL: churn noise drawn on device
 *   coins [precond calls] host pointer, one byte per preconditioned call in call order;
 *         NULL: Bernoulli(0.5) from a host generator seeded with `seed`
 *   gt_adj/gt_node non-NULL: sanity-check mode (edm.py:372-377), the denoiser is bypassed
 *   snap_steps/snap_adj/snap_node: optional interim snapshots (edm.py:429-432): after step
 *         snap_steps[k] the state is copied to slot k of snap_adj / snap_node (device) */
int dsg_sample(dsg_handle h, const dsg_sampler_cfg *cfg, int32_t B, const uint8_t *flags,
               const float *init_adj, const float *init_node,
               const float *noise_adj, const float *noise_node,
               const uint8_t *coins, uint64_t seed,
               const float *gt_adj, const float *gt_node,
               const int32_t *snap_steps, int32_t n_snap, float *snap_adj, float *snap_node,
               float *out_adj, float *out_node, dsg_sample_stats *stats, void *stream);

/* The library's device noise streams (Philox4x32-10 keyed by `seed`, Box-Muller): what dsg_sample draws when it is handed
 * NULL init / NULL churn noise.  noise_stream 0 = the initial sample of gen_init_sample (edm.py:257-289: randn, rows and
 * columns of padded nodes zeroed, NOT yet scaled by sigma(t0)); noise_stream i+1 = the churn noise of step i (the
 * randn_like draws of edm.py:361-364).  Writes mask(eps) in the layouts above.  dsg_sample(init = this output) is
 * bit-identical to dsg_sample(init = NULL) with the same seed, so a caller that needs the unscaled init back
 * (nodes_ls[0] of edm.py:326-337) draws it here first. */
int dsg_gen_noise(dsg_handle h, int32_t B, const uint8_t *flags, uint64_t seed, uint32_t noise_stream,
                  float *out_adj, float *out_node, void *stream);

/* sigma_steps (fp64, edm.py:84-88) and the fp32 per-step scalars the loop uses; out arrays of
 * length num_steps.  Host-only helper, exposed so bindings/tests can inspect the schedule. */
int dsg_sigma_schedule(const dsg_sampler_cfg *cfg, double *sigma_steps, float *t_hat, float *noise_coef,
                       float *h_step);

/* Kernel selection.  The narrow levels (C = 96 / 192) have register-resident fused kernels; each can be switched off to
 * fall back to the generic GEMM + attention + row-kernel path (all combinations are parity-tested):
 *   "fused_attn" (C=96 attention block), "fused_mlp", "fused_mlp_maxc" (96|192), "fused_readout", "fused_patch_embed",
 *   "fused_rowstats" (modulate+SiLU and LayerNorm statistics in the producing GEMM's epilogue instead of row kernels),
 *   "fused_qkv_attn" (8x8 / 10x10 windows: QKV projection + window attention in one kernel, q/k/v never reach HBM),
 *   "fused_merge" (PatchMerging's 2x2 gather + LayerNorm(4C) inside the reduction GEMM's A path; 1: where it pays (>= 8192 merged
 *   rows), 2: at every size, 0: merge_ln kernel).
 * Reverse loop: "loop_graph" = 1 (default): dsg_sample replays one captured hipGraph per step (a handful of distinct step bodies:
 *   first / steady / last step x the two self-conditioning coins); 0: the round-1 scheme, only the network forward is a graph.
 * Precision modes (default: exact fp32 MFMA everywhere):
 *   "gemm_split" = 1: every GEMM as six bf16-MFMA partial products of hi/mid/lo (3 x bf16 = 24-bit) operand splits with
 *       fp32 accumulation -- fp32-level accuracy, the 1e-4 parity bar still holds; 1.3-1.6x faster GEMMs (power-bound).
 *   "gemm_bf16" = 1: the block/merge/breakup GEMMs on bf16 MFMA with operands rounded to bf16 (activations, LayerNorm,
 *       softmax and the sampler stay fp32) -- BASELINE config 5; parity against the fp32 oracle then holds to 1.5e-2 RMS /
 *       5e-2 max-abs of the output scale, not 1e-4.  "gemm_split" takes precedence if both are set.
 *   "bf16_act" (0 / 1 / 2, default 2; acts only with "gemm_bf16"): 1 -- tensors whose only consumer is the bf16 GEMM (the MLP's
 *       hidden activations, the attention output) are stored as bf16 by their producer; the consumer rounds its A operand to bf16
 *       anyway, so results are bit-identical to 0, with half the bytes and no conversion in the consumer.  2 -- q, k, v are
 *       stored as bf16 as well; the attention arithmetic stays fp32 on the widened values, results move within the mode's bar.
 *   "bf16_pipe" (default 1; acts only with "gemm_bf16"): the bf16 block pipeline -- a Swin block is two kernels with bf16 tensors between
 *       them; its kernel choices (all parity-tested, same bar): "bf16_qkv_attn" 1 QKV projection + window attention in one kernel (10 x 10
 *       windows: one wave per (window, head)), 2 the block-per-head kernel everywhere, 3 the former only where a block of four units lies in
 *       one window, 0 GEMM + attention kernel; "bf16_mlp" 1 fused fc1-GELU-fc2 (C = 384: eight waves, weight images streamed by LDS-DMA),
 *       4 round 3's eight-wave kernel, 5 the one-wave-per-SIMD LDS-DMA kernel, 6 = 1 with level 0 on the LDS-resident kernel, 2 four waves at every width, 3 GEMM pair at C = 384,
 *       0 GEMM pairs; "bf16_proj_mlp" 1 proj + residual + LayerNorm-2 in front of the MLP kernel; "bf16_readout" 1 the read-out on the bf16 pipe. */
int dsg_set_option(dsg_handle h, const char *name, int32_t value);
/* The value an option currently has on this handle (what the next forward will run with), whichever way it was set
 * (dsg_set_option or a DSG_* environment default): measurement code reports the precision mode from here. */
int dsg_get_option(dsg_handle h, const char *name, int32_t *value);

/* Measurement: runs n_iters eager network forwards on the batch-B workspace (whatever inputs the last call left
 * there) with HIP events bracketing every kernel launch on `stream`, and accumulates per kernel class
 * (0 = MFMA GEMM, 1 = window attention, 2 = row kernels (LayerNorm/modulate/heads), 3 = elementwise, 4 = fused
 * narrow-level blocks (attention / MLP / read-out / patch-embed)):
 * total milliseconds, number of launches, algorithmic FLOPs (2*M*N*K; 4*T*W*C for attention).  Arrays of length 5.
 * An event bracket around a single launch includes ~10-15 us of dispatch latency that back-to-back (graph) launches do
 * not pay, so for the dominant kernel (class 0) the kernel also stamps first-block-start / last-block-end on the GPU's
 * 100 MHz constant clock: *gemm_inkernel_ms receives the sum of those in-kernel durations (agrees with rocprofv3). */
int dsg_profile_forward(dsg_handle h, int32_t B, int32_t n_iters, double *ms_by_kind, int64_t *launches_by_kind,
                        double *flops_by_kind, double *gemm_inkernel_ms, void *stream);

/* The shader clock the chip held during the GEMM launches of the last dsg_profile_forward call, in GHz: median over the launches of
 * (block 0's lifetime in s_memtime shader cycles) / (the same lifetime in 100 MHz s_memrealtime ticks).  Dense f32-MFMA streams are
 * power-limited on MI355X: the clock-limited ceiling of a perfect kernel is this clock x 1024 SIMDs x 64 FLOP/clk, not 157.3 TFLOP/s. */
double dsg_profile_clock_ghz(dsg_handle h);

/* Debug: copy the named stage's activation (e.g. "down0.block0") of the next dsg_denoise call to
 * `dst` (device, capacity in floats).  Token-major [B, T, C]. */
int dsg_debug_tap(dsg_handle h, const char *stage, float *dst, int64_t capacity);
void dsg_debug_clear_taps(dsg_handle h);

/* On-device post-decode of the samples (sampler_node_adj.py:222-285) for the reference's three attribute encodings
 * (`--edge_encoding` / `--node_encoding`; R/utils/attribute_code.py:13 attribute_converter(..., out_encoding='int')):
 *   DSG_ENC_BITS     clamp(-1,1) -> > 0 -> MSB-first integer (bin2dec, :319) -> clamp to [0, n_type-1]
 *   DSG_ENC_ONE_HOT  clamp -> +-1 threshold at 0 -> argmax over the channels (:212-237): the first positive channel, 0 if none
 *   DSG_ENC_DDPM     clamp -> the class whose interval (lo_i, hi_i] of width 2/(n_type-1) around -1 + 2i/(n_type-1) holds the
 *                    value (:121-177; the reference's double-precision thresholds compared in fp32), -1 for NaN
 * Rows / columns of padded nodes and the adjacency diagonal are 0.  adj [B,C_adj,N,N] (C_adj = bits | n_adj_type | 1), node
 * [B,N,C_node]: the first node_chans channels are the attribute (bits | n_node_type | 1), the last four the bounding box when
 * out_bbox != NULL (-> *0.5+0.5, masked; :201-209).  Needs the handle only for N / C_adj / C_node (no weights). */
enum { DSG_ENC_BITS = 0, DSG_ENC_ONE_HOT = 1, DSG_ENC_DDPM = 2 };
int dsg_decode(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags, int32_t edge_encoding,
               int32_t node_encoding, int32_t n_adj_type, int32_t n_node_type, int32_t node_chans,
               int32_t *out_adj /*[B,N,N]*/, int32_t *out_node /*[B,N]*/, float *out_bbox /*[B,N,4] or NULL*/, void *stream);
/* dsg_decode with both encodings 'bits' (the README's recipe) */
int dsg_decode_bits(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags,
                    int32_t n_adj_type, int32_t n_node_type, int32_t node_bits,
                    int32_t *out_adj /*[B,N,N]*/, int32_t *out_node /*[B,N]*/, float *out_bbox /*[B,N,4] or NULL*/,
                    void *stream);

/* Debug / verification: one GEMM of the library, C[M,N] = act(LN?(A)[M,K] . W[N,K]^T + bias) (+ res), in a chosen arithmetic
 * (mode 0: fp32 MFMA; 1: bf16 operands, fp32 accumulate -- "gemm_bf16"; 2: three-way split bf16 -- "gemm_split"); act 0 none,
 * 1 GELU, 2 SiLU; ln_stats [M,2] (mean, rstd) or NULL.  Device pointers, K % 32 == 0, synchronises `stream`.  Lets a test diff
 * WHOLE output matrices between the arithmetic modes (a rare wrong 16-lane group is invisible to sampled checks). */
int dsg_debug_gemm(int32_t M, int32_t N, int32_t K, const float *A, const float *W, const float *bias, const float *ln_stats,
                   const float *res, int32_t act, int32_t mode, float *C, void *stream);

/* The bf16 block pipeline's kernels on their own (test hooks, csrc/kernels_bx.hip; device pointers, synchronise `stream`).
 * dsg_debug_gemm_bx: A [M,K], W [N,K] given as fp32 and rounded to bf16 inside; epilogue pieces as in the forward: bias [N], fp32
 *   residual res [M,N], act (0 | 1 GELU), mod = (scale [N] | shift [N]) of a batch-uniform modulate+SiLU, ln_out (LayerNorm of the
 *   stored row into out_Cb; N in {96,192,384}).  out_C [M,N] fp32 store; out_Cb / out_C2b [M,N]: the bf16 stores widened to fp32
 *   (any may be NULL, not both of out_C / out_Cb).  DSG_ERR_INVALID when the shape is not covered.
 * dsg_debug_attn_bx: window attention on qkv [B*res*res, 3*32*heads] (rounded to bf16 inside; q pre-scaled) with the key-major
 *   log2(e)-scaled bias table [nW|1][heads][Wp][Wp] -> out [B*res*res, 32*heads] (the bf16 result widened).
 * time_iters > 0 with out_ms (host): also the mean HIP-event time of that many back-to-back launches of the kernel (tools/bx_bench.py). */
int dsg_debug_gemm_bx(int32_t M, int32_t N, int32_t K, const float *A, const float *W, const float *bias, const float *res, int32_t act,
                      const float *mod, int32_t ln_out, float *out_C, float *out_Cb, float *out_C2b, int32_t time_iters, float *out_ms,
                      void *stream);
int dsg_debug_attn_bx(int32_t B, int32_t res, int32_t ws, int32_t shift, int32_t heads, const float *qkv, const float *biasT, float *out,
                      int32_t time_iters, float *out_ms, void *stream);
/* the fused QKV projection + window attention kernel on caller-provided operands: xn [B*res*res, C] (C = 32 heads), W [3C, C] (q rows
 * pre-scaled by d^-1/2 log2 e), bias [3C], biasT as above; operands are rounded to bf16 on the way in, out is the bf16 result as fp32.
 * Kernel selection for 10 x 10 windows: the wave-per-(window, head) kernel; shift + 1000 selects the block-per-head kernel instead. */
int dsg_debug_qkv_attn_bx(int32_t B, int32_t res, int32_t ws, int32_t shift, int32_t heads, const float *xn, const float *W, const float *bias,
                          const float *biasT, float *out, int32_t time_iters, float *out_ms, void *stream);
/* the fused MLP kernel with the attention half's tail in front: x <- [modulate] (x1 + fc2(GELU(fc1(LN(x1))))), x1 = x + att Wp^T + bp;
 * att [M, C], Wp [C, C], W1 [4C, C], W2 [C, 4C] are rounded to bf16 on the way in; x [M, C] fp32 in place; out_xn as dsg_debug_mlp_bx */
int dsg_debug_projmlp_bx(int32_t M, int32_t C, const float *att, float *x, const float *Wp, const float *bp, const float *W1, const float *b1,
                         const float *W2, const float *b2, const float *mod, int32_t out_mode, float *out_xn, int32_t time_iters, float *out_ms,
                         void *stream);
/* dsg_debug_mlp_bx: the fused MLP half of a block, x [M,C] <- [modulate] (x + fc2(GELU(fc1(xn)))) in place, with xn [M,C], W1 [4C,C],
 *   W2 [C,4C] given as fp32 and rounded to bf16 inside; mod = (scale [C] | shift [C]) or NULL; out_mode 0 none, 1 LayerNorm of the
 *   stored row, 2 its plain copy -> out_xn [M,C] (the bf16 store widened).  C in {96, 192, 384}.
 *   Kernel selection at C = 384 (both entries): out_mode as is = the LDS-DMA kernel on pre-arranged weight images; + 16 the four-wave
 *   kernel of the narrower levels (dsg_debug_mlp_bx only); + 32 round 3's eight-wave kernel; + 64 the one-wave-per-SIMD LDS-DMA kernel.
 *   dsg_debug_projmlp_bx at C = 96 without modulate: + 128 the LDS-resident persistent kernel (mlp96r_bx_kernel). */
int dsg_debug_mlp_bx(int32_t M, int32_t C, const float *xn, float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *mod, int32_t out_mode, float *out_xn, int32_t time_iters, float *out_ms, void *stream);

/* The noise-conditioning path on its own (test / inspection hook for SURVEY fixture G1): PositionalEmbedding -> map_layer0/1 with
 * SiLU (R/model/diffusesg/diffusesg.py:507-513, :768-771) -> every `affine` linear (:238, :574) for `rows` noise labels c_noise
 * (device, [rows]).  out_pe [rows, embed_dim], out_emb [rows, 512], out_aff [rows, dsg_affine_width(h)] (any may be NULL):
 * (scale | shift) of patch_embed, then of down_layers[l].blocks[j], then of up_layers[i].blocks[j] -- the table dsg_sample builds
 * once per call for all its steps. */
int dsg_noise_embed(dsg_handle h, int32_t rows, const float *c_noise, float *out_pe, float *out_emb, float *out_aff, void *stream);
int32_t dsg_affine_width(dsg_handle h);

/* ---- a training iteration: objective, loss, loss backward (no handle: these only need the tensor dimensions), then
 * the network in training form with its backward, the optimiser step and the EMA update further down.
 * Return DSG_OK / DSG_ERR_INVALID / DSG_ERR_HIP.
 *
 * dsg_train_inputs <-> NodeAdjEDMObjectiveGenerator.get_input_output   R/runner/objectives/edm.py:160-180, :239-281
 *   (precond = sigma_dist = 'edm', symmetric_noise = False: learning_utils.py:25-29)
 *   sigma_b = exp(rnd_b*1.2 - 1.2); weight_b = (sigma^2 + 0.25)/(0.5 sigma)^2;
 *   noisy_adj = mask(clean_adj + eps_adj*sigma_b); noisy_node = clean_node + mask(eps_node*sigma_b)
 *   rnd_sigma [B] / eps_adj / eps_node: the N(0,1) draws (device pointers) or NULL = the library's Philox streams of `seed`. */
int dsg_train_inputs(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *clean_adj, const float *clean_node,
                     const uint8_t *flags, const float *rnd_sigma, const float *eps_adj, const float *eps_node, uint64_t seed,
                     float *out_sigmas, float *out_weights, float *out_noisy_adj, float *out_noisy_node, void *stream);
/* iou_loss_type of the trainer's bounding-box term (R/runner/trainer/trainer_node_adj.py:138-153; `--iou_loss_type`, the reference's
 * README trains with 'giou', its YAMLs default to 'iou'): 'iou' = -(torchvision.ops.box_iou)^2; the others are torchvision.ops'
 * generalized_ / distance_ / complete_box_iou_loss(reduction='none') ('giou_squared' squares the first).  torchvision is an
 * un-vendored dependency of the reference and absent from this image: those four are restated from its published algorithm. */
enum { DSG_IOU_IOU = 0, DSG_IOU_GIOU = 1, DSG_IOU_GIOU_SQUARED = 2, DSG_IOU_DIOU = 3, DSG_IOU_CIOU = 4 };
/* dsg_rainbow_loss <-> NodeAdjRainbowLoss.forward(reduction='none')    R/loss/rainbow_loss.py:37-101
 *   plus the trainer's bounding-box term (iou_loss_type: DSG_IOU_*)    R/runner/trainer/trainer_node_adj.py:130-159
 *   (last four node channels; weight 0 switches it off).  Outputs: per-sample losses [B]; the step's scalar loss is
 *   mean(out_loss_adj) + mean(out_loss_node) (trainer_node_adj.py:167). */
int dsg_rainbow_loss(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *pred_adj, const float *pred_node,
                     const float *target_adj, const float *target_node, const uint8_t *flags, const float *loss_weight,
                     float edge_loss_weight, float node_loss_weight, float iou_loss_weight, int32_t iou_loss_type, float *out_loss_adj,
                     float *out_loss_node, void *stream);
/* dsg_rainbow_loss_backward: first stage of loss.backward() of a training step   R/runner/trainer/trainer_node_adj.py:163-170
 *   loss = mean_b(loss_adj) + mean_b(loss_node) with the terms of dsg_rainbow_loss (IoU term through autograd's clamp / max / min
 *   rules).  out_grad_* = dL/d(preconditioned outputs), layouts of pred_*.  With sigmas [B] (may be NULL) also
 *   out_grad_F_* = dL/d(raw network outputs F) = c_out(sigma_b) * out_grad_*   (D = mask(c_skip x + c_out F), precond.py:101-104).
 *   (dsg_train_step_grads chains this with the network's own backward.) */
int dsg_rainbow_loss_backward(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *pred_adj, const float *pred_node,
                              const float *target_adj, const float *target_node, const uint8_t *flags, const float *loss_weight,
                              float edge_loss_weight, float node_loss_weight, float iou_loss_weight, int32_t iou_loss_type,
                              const float *sigmas, float *out_grad_adj, float *out_grad_node, float *out_grad_F_adj, float *out_grad_F_node,
                              void *stream);

/* One SwinTransformerBlock in training form: forward x_out = block(x_in, emb) (R/model/diffusesg/diffusesg.py:232-277 with
 * WindowAttention :108-139 and Mlp :19-25) and, when grad_out != NULL, its backward as torch.autograd derives it -- the first
 * building block of the network backward.  Kernels: csrc/train_kernels.hip -- products, weight gradients and the
 * attention forward / backward on fp32 MFMA, row passes HBM-bound; pinned to the reference's autograd by tests/golden/block_backward.npz.
 *   block: state-dict prefix of the block, e.g. "down_layers.0.blocks.1".  x_in, x_out, grad_out, grad_in: [B, T, C] token-major
 *   (the reference's [B, L, C]); emb, grad_emb: [B, 512] (the mapped noise embedding, dsg_noise_embed).  names[i] (relative to the
 *   prefix: "affine.weight", "affine.bias", "norm1.weight", "norm1.bias", "attn.relative_position_bias_table", "attn.qkv.weight",
 *   "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
 *   "mlp.fc2.weight", "mlp.fc2.bias") -> grad_params[i] (device, the parameter's shape); all 15 are required with grad_out. */
int dsg_block_train(dsg_handle h, const char *block, int32_t B, const float *x_in, const float *emb, const float *grad_out, float *x_out,
                    float *grad_in, float *grad_emb, int32_t n_params, const char *const *names, float *const *grad_params, void *stream);

/* The whole network in training form: F = DiffuseSG.forward(in_adj, in_node, flags, c_noise, sc_adj, sc_node)
 * (R/model/diffusesg/diffusesg.py:765-830) and, when grad_F_adj != NULL, the gradient of every parameter for the upstream gradients
 * dL/dF (e.g. from dsg_rainbow_loss_backward) -- what loss.backward() of a training step leaves in the parameters' .grad
 * (R/runner/trainer/trainer_node_adj.py:163-170), pinned by tests/golden/train_backward.npz.  MFMA products, weight gradients
 * and attention since round 3 (csrc/train_kernels.hip; measured rates: DESIGN.md §4).  dsg_finalize_weights must have run once; weights
 * set afterwards (an optimiser step) are used as they are -- the training form reads the raw tensors only.
 *   in_adj [B,C_adj,N,N], in_node [B,N,C_node]: the preconditioned inputs c_in(sigma) * noisy (precond.py:100); c_noise [B];
 *   sc_*: the self-conditioning inputs (constants: the reference detaches them) or NULL; out_F_*: the raw network outputs;
 *   names[i] (state-dict keys of all parameters) -> grad_params[i] (device buffers of the parameters' shapes, overwritten).
 *   Like every entry that takes a stream, the work is only ENQUEUED when the call returns (round 2 synchronised at the end of this one). */
int dsg_train_grads(dsg_handle h, int32_t B, const float *in_adj, const float *in_node, const uint8_t *flags, const float *c_noise,
                    const float *sc_adj, const float *sc_node, const float *grad_F_adj, const float *grad_F_node, float *out_F_adj,
                    float *out_F_node, int32_t n_params, const char *const *names, float *const *grad_params, void *stream);

/* One training iteration up to and including loss.backward() (R/runner/trainer/trainer_node_adj.py:96-170, 'edm' objective):
 * D = NodeAdjPrecond(noisy, sigmas) with the network in training form, per-sample losses as dsg_rainbow_loss, and -- when the gradient
 * buffers are given -- the gradient of  loss_adj.mean() + loss_node.mean()  for every parameter.  sc_*: the detached self-conditioning
 * inputs (precond.py:90-98; the Python mirror draws the coin and computes them with dsg_precond) or NULL.  The caller supplies the
 * objective's tensors (dsg_train_inputs); dsg_adam_step / dsg_ema_update below finish the iteration, gradient averaging across
 * ranks is the host's (diffusesg_amd.dist.all_reduce_mean over RCCL). */
int dsg_train_step_grads(dsg_handle h, int32_t B, const float *noisy_adj, const float *noisy_node, const uint8_t *flags, const float *sigmas,
                         const float *sc_adj, const float *sc_node, const float *target_adj, const float *target_node,
                         const float *loss_weight, float edge_loss_weight, float node_loss_weight, float iou_loss_weight,
                         int32_t iou_loss_type, float *out_D_adj, float *out_D_node, float *out_loss_adj, float *out_loss_node,
                         int32_t n_params, const char *const *names, float *const *grad_params, void *stream);

/* The no-grad pass of a self-conditioning training step (precond.py:92-98): D = NodeAdjPrecond(noisy, sigmas) with NO self-conditioning
 * input, computed by the network in TRAINING form -- i.e. from the raw parameters (dsg_train_bind_params / dsg_set_weight), without the
 * sampling path's derived weights, which an optimiser step invalidates (dsg_finalize_weights costs several forwards).  out_sc_* are the
 * masked D tensors the caller then passes to dsg_train_step_grads as sc_adj / sc_node. */
int dsg_train_self_cond(dsg_handle h, int32_t B, const float *noisy_adj, const float *noisy_node, const uint8_t *flags, const float *sigmas,
                        float *out_sc_adj, float *out_sc_node, void *stream);

/* Let the training-form entries (dsg_train_grads, dsg_train_step_grads, dsg_train_self_cond) read these parameters IN PLACE: names[i]
 * (state-dict key) -> params[i] (device tensor of the parameter's shape, e.g. the tensor the optimiser updates), instead of the
 * handle's own copies -- no upload per iteration.  The caller keeps the tensors alive and unchanged in address until the next call of
 * this function; n_params = 0 clears the binding.  Keys not bound fall back to the handle's copy.  The sampling-path entries are not
 * affected: they use what dsg_set_weight + dsg_finalize_weights installed. */
int dsg_train_bind_params(dsg_handle h, int32_t n_params, const char *const *names, const float *const *params);

/* The rest of a training iteration (R/runner/trainer/trainer_node_adj.py:170-175), on caller-owned device tensors, no handle:
 * dsg_adam_step <-> nn.utils.clip_grad_norm_(parameters, max_norm) followed by torch.optim.Adam.step()  (utils/learning_utils.py:137-140:
 *   betas (0.9, 0.999), eps 1e-8, L2 weight_decay; gradients are scaled in place by the clip coefficient like the reference;
 *   `step` counts from 1; max_grad_norm <= 0 switches clipping off).  out_total_norm (host, may be NULL): the norm before clipping.
 * dsg_ema_update <-> ema_pytorch's EMA.update_moving_average: ema <- ema + (param - ema)(1 - decay).  ema_pytorch is an un-vendored,
 *   unpinned dependency of the reference (setup/requirements.txt:20) and absent here: parity unpinned; the decay schedule is restated
 *   from its published algorithm in diffusesg_amd/train.py::EMAHip. */
int dsg_adam_step(int32_t n_tensors, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                  const int64_t *numel, int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                  float *out_total_norm, void *stream);
int dsg_ema_update(int32_t n_tensors, float *const *ema, const float *const *params, const int64_t *numel, float decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSG_H */
