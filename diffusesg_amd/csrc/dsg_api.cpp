// dsg_api.cpp -- host side of libdsg.so: weight registry, the static launch plan of one network
// forward, preconditioning, and the EDM reverse loop.  Implements include/dsg.h.
//
// Reference mapping (R/ = DiffuseSG/):
//   forward plan      R/model/diffusesg/diffusesg.py:739-830
//   preconditioning   R/model/precond/precond.py:65-110, R/runner/objectives/edm.py:122-126
//   reverse loop      R/runner/mcmc_sampler/edm.py:291-445
#include "../../include/dsg.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <functional>
#include <unordered_map>
#include <string>
#include <tuple>
#include <vector>

#include "kernels.h"

using namespace dsg;

#define NOISE_EMB 512
// attention scores are formed directly in the exp2 domain: q carries head_dim^-0.5 * log2(e), the bias/mask tables log2(e)
static const double kLog2e = 1.4426950408889634;
static const double kQScale = 0.17677669529663687 * 1.4426950408889634;

namespace {

struct DevTensor {
    float *p = nullptr;
    std::vector<int64_t> shape;
    int64_t numel = 0;
};

struct WSpec {
    std::string key;
    std::vector<int64_t> shape;
    bool is_index = false;  // relative_position_index: int64 constant, validated and dropped
    bool is_mask = false;   // attn_mask: fp32 constant, accepted and dropped (re-derived)
};

struct BlockPlan {
    std::string prefix;
    int lvl, C, res, ws, shift, heads;
    int aff_off;        // offset of this block's (scale,shift) in the concatenated affine output
    float *biasT = nullptr;  // [nWt][heads][Wp][Wp]
    void *biasP = nullptr;   // the same tiles as fp16 in score-accumulator order (qkv_attn_bx_kernel)
    float *biasF = nullptr;  // ... and as fp32 (qkv_attn_wx_kernel: 10 x 10 windows)
    float *w1p = nullptr, *w2p = nullptr;  // fragment-major packed MLP weights (fused_mlp_kernel), narrow levels only
    float *wqp = nullptr, *wpp = nullptr;  // fragment-major packed qkv / proj weights (fused_attn96_kernel), C == 96 only
    float *bqkv_s = nullptr;               // qkv bias with the q part pre-scaled (fused_attn96_kernel)
    // LayerNorm folded into the consuming linear: W' = W.diag(gamma), b' = b + W.beta (the GEMM's A path then only
    // applies (x-mean)*rstd)
    float *qkv_wf = nullptr, *qkv_bf = nullptr, *fc1_wf = nullptr, *fc1_bf = nullptr;
};

struct Workspace {
    int B = 0;
    size_t bytes = 0;
    std::vector<void *> allocs;
    // network forward (fixed addresses: the captured graph reads/writes these)
    float *in_adj, *in_node, *sc_adj, *sc_node, *c_noise, *f_adj, *f_node;
    uint8_t *flags;
    int *has_sc;
    // self-conditioning input the next forward_fixed reads: the fixed buffers + device flag above (dsg_denoise, dsg_precond, the
    // forward-only graph), or -- inside a step body of the reverse loop -- the denoised buffer itself / null, with no flag and
    // no copy, so that a captured step body consists of kernel nodes only
    const float *cur_sc_adj = nullptr, *cur_sc_node = nullptr;
    const int *cur_has_sc = nullptr;
    float *pe, *emb0, *emb, *aff, *tok_in, *x, *y, *qkv, *att, *hid, *stats, *pool, *hn, *pool_ext, *pool_part;
    void *xn = nullptr;   // bf16 [B*T, C]: the normalised input of the next QKV / fc1 GEMM (bf16 block pipeline, kernels_bx.hip)
    float *skips[DSG_MAX_LAYERS];
    // sampler state
    float *x_adj, *x_node, *xh_adj, *xh_node, *sig;
    float *d_adj[3], *d_node[3];
    // batch-uniform noise level (the sampler): every sample shares one (scale,shift) row, taken from a table that
    // dsg_sample computes once for all steps; aff_ld == 0 broadcasts row 0 of `aff`
    bool uniform = false;
    int aff_ld = 0;
    hipGraphExec_t graph = nullptr, graph_uniform = nullptr;
    hipStream_t cap_stream = nullptr;
    // reverse loop: control block (device step counter, seed, recorded-noise pointers) and the captured step bodies, one
    // per (self-cond slot, output slots, Euler/Heun update, coin of stage 1, coin of stage 2) combination that occurs
    RunCtl *ctl = nullptr;
    std::map<int, std::pair<hipGraphExec_t, int>> step_graphs;   // key -> (exec, network forwards inside)
    long plan_gen = -1;   // h->plan_gen this workspace's launch plan was last validated against (validate_plan)
};

struct Tap { std::string name; float *dst; int64_t cap; };

}  // namespace

struct dsg_handle_s {
    dsg_config cfg;
    int N, Ca, Cn, E, L, Cin, Kp;
    std::vector<WSpec> specs;
    std::map<std::string, DevTensor> w;
    bool finalized = false;
    bool ever_finalized = false;                                  // block plans exist and every weight has been set at least once (training entries read raw weights only)
    std::string err;
    // first launch of the current forward whose shape / argument combination no kernel covers ("" = none).  Filled by the launch
    // wrappers below instead of terminating the process; validate_plan() walks a whole forward in a dry run (kernels.h: g_dry_run)
    // when a batch size is first used and whenever an option changes, so the entry points return DSG_ERR_INVALID before any work
    // is enqueued or captured
    std::string plan_err;
    long plan_gen = 0;   // bumped by everything that changes which kernels a forward launches (options, weights, debug taps)
    // derived
    std::vector<BlockPlan> down[DSG_MAX_LAYERS], up[DSG_MAX_LAYERS];
    float *aff_w = nullptr, *aff_b = nullptr;  // concatenated affine linears [aff_n, 512]
    int aff_n = 0, pe_aff_off = 0;
    float *pe_w = nullptr;      // patch_embed.proj padded to [E, Kp]
    float *pe_wp = nullptr;     // the same, fragment-major packed [E/32][Kp/8][64][4] (fused_patch_embed96_kernel)
    float *ro0_w = nullptr;     // read_out.0 transposed to [out,in]
    // folded read-out (E = 96): Fa = F1.W2.W1.W0^T packed fragment-major, fa; F2 padded+packed; node: Gext [E,128]
    float *ro_fap = nullptr, *ro_fa = nullptr, *ro_f2p = nullptr, *ro_gext = nullptr;
    float *ro_fapb = nullptr, *ro_f2pb = nullptr;   // the same two matrices in the bf16 MFMA's fragment order (fp32 here; their bf16 copies are what runs)
    float *ro0_wf = nullptr, *ro0_bf = nullptr;  // read_out.0 ([out,in]) with the final norm's gamma/beta folded in
    float *merge_wf[DSG_MAX_LAYERS] = {}, *merge_bf[DSG_MAX_LAYERS] = {};   // PatchMerging reduction with its LayerNorm(4C) folded in
    std::vector<void *> derived_allocs;
    std::map<int, std::unique_ptr<Workspace>> ws;
    // kernel-selection options (dsg_set_option); defaults may be overridden once by DSG_* environment variables
    bool opt_fused_attn = true, opt_fused_mlp = true, opt_fused_readout = true, opt_fused_pe = true;
    bool opt_fused_merge_small = false;   // also fuse PatchMerging below the size where it pays (tests force it on)
    bool opt_fused_merge = true;      // PatchMerging: gather + LayerNorm(4C) inside the reduction GEMM's A path (no merge_ln kernel)
    bool opt_loop_graph = true;       // capture whole step bodies of the reverse loop (0: only the network forward is a graph)
    bool opt_fused_qkv_attn = true;   // QKV projection + 64-token window attention in one kernel (q, k, v never reach HBM)
    bool opt_fused_rowstats = true;   // modulate+SiLU and LayerNorm statistics in the producing GEMM's epilogue (fp32 kernel)
    bool opt_gemm_bf16 = false;                                   // bf16-MFMA GEMMs (fp32 accumulate), opt-in precision mode
    int opt_bf16_act = 2;                                         // in that mode: 1 hidden / attention-output tensors stored as bf16 (bit-identical), 2 also qkv
    bool opt_bf16_pipe = true;                                    // in that mode: the bf16 block pipeline of kernels_bx.hip (0: round 2's kernels_lp.hip path)
    bool opt_bf16_readout = true;                                 // in that pipeline: the read-out's two products on the bf16 matrix pipe
    bool opt_bf16_proj_mlp = true;                                // in that pipeline: proj + residual + LayerNorm-2 in front of the fused MLP kernel (0: the proj GEMM)
    int opt_bf16_qkv_attn = 1;                                    // in that pipeline: QKV projection + window attention in one kernel (0: GEMM + attn_bx_kernel through a bf16 qkv tensor)
    int opt_bf16_mlp = 1;                                         // in that pipeline: the fused fc1-GELU-fc2 kernels (0: two GEMMs with a bf16 hidden tensor; 1: C <= 192 on 4 waves, C = 384 on 8; 2: C = 384 on the 4-wave kernel too; 3: GEMM pair at C = 384)
    std::vector<std::pair<const float *, size_t>> gemm_weights;   // every fp32 GEMM weight (pointer, numel)
    std::map<const float *, void *> w_bf16;                       // bf16 copies, built when the mode is switched on
    std::map<const float *, void *> w_img96;                      // per C = 96 block (key: its fc1_wf): W1 | W2 | Wp in fragment order for mlp96r_bx_kernel
    std::map<const float *, void *> w_qimg;                       // per block (key: its qkv_wf): the QKV weight in qkv_attn_wx_kernel's streaming order
    std::map<const float *, void *> w_img2;                       // the same weights chunk-major for mlp384s_bx_kernel (bf16_mlp = 5)
    std::map<const float *, void *> w_img;                        // per C = 384 block (key: its fc1_wf): W1 | W2 | Wp pre-arranged for mlp384d_bx_kernel's LDS-DMA ring
    bool opt_gemm_split = false;                                  // split-bf16 GEMMs (3 planes, 6 products): fp32-accurate, opt-in
    std::map<const float *, void *> w_split;                      // [3][N][K] bf16 planes
    int opt_fused_mlp_maxc = 96;   // at C = 192 the plain GEMM pair is faster than the one-wave-per-SIMD fused kernel
    // per-step (scale,shift) table of the sampler (batch-uniform sigma): [cap][aff_n] and its staging buffers
    int tab_cap = 0;
    float *tab_sig = nullptr, *tab_cn = nullptr, *tab_pe = nullptr, *tab_e0 = nullptr, *tab_e1 = nullptr, *tab_aff = nullptr;
    StepRow *tab_step = nullptr;   // per-step scalars of the loop, read by the table-driven sampler kernels
    std::vector<Tap> taps;
    // training form (dsg_train_*): saved-activation arena and backward scratch, owned by the handle and kept between calls
    // (hipMalloc / hipFree of ~10 GB per iteration cost more than a tenth of it); they only grow
    float *train_arena = nullptr, *train_scr = nullptr, *train_io = nullptr, *train_mid = nullptr;
    size_t train_arena_cap = 0, train_scr_cap = 0, train_io_cap = 0, train_mid_cap = 0;
    int *train_const = nullptr;                                   // device {0, 1}: the "self-conditioning present" switch of launch_assemble
    // dsg_train_bind_params: parameters the training form reads in place (the optimiser's own device tensors) instead of the handle's copies
    std::unordered_map<std::string, const float *> train_params;
    dsg_sample_stats last_stats{};
    // per-kernel-class timing (dsg_profile_forward): HIP events bracketing every launch on the launch stream
    bool prof_on = false;          // HIP-event brackets around every launch
    bool prof_stamps = false;      // in-kernel start/end stamps of the GEMM launches (no events: launches stay back to back)
    std::vector<hipEvent_t> prof_events;
    size_t prof_used = 0;
    std::vector<int> prof_kind;
    std::vector<double> prof_flops;
    std::vector<std::string> prof_tag;
    // in-kernel stamps of the dominant kernel (GEMM): one {start,end} pair per launch
    unsigned long long *prof_gemm = nullptr;
    int prof_gemm_cap = 0, prof_gemm_used = 0;
    double prof_clock_ghz = 0.0;   // median over the GEMM launches of the last dsg_profile_forward: shader clock held by block 0
};

namespace {

int fail(dsg_handle h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

// a launcher declined its arguments: remember the first one of this forward (layer / shape in the message)
void plan_fail(dsg_handle h, const char *fmt, ...) {
    if (!h->plan_err.empty()) return;
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    h->plan_err = buf;
}

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(h, DSG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

enum ProfKind { PK_GEMM = 0, PK_ATTN = 1, PK_ROW = 2, PK_ELEM = 3, PK_FUSED = 4, PK_COUNT = 5 };

struct ProfScope {
    dsg_handle h; hipStream_t s;
    ProfScope(dsg_handle h_, hipStream_t s_, int kind, double flops, const char *tag = nullptr) : h(h_), s(s_) {
        if (!h->prof_on) return;
        while (h->prof_events.size() < h->prof_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { h->prof_on = false; return; }
            h->prof_events.push_back(e);
        }
        h->prof_kind.push_back(kind);
        h->prof_flops.push_back(flops);
        h->prof_tag.push_back(tag ? tag : "");
        (void)hipEventRecord(h->prof_events[h->prof_used], s);
    }
    ~ProfScope() {
        if (!h->prof_on) return;
        (void)hipEventRecord(h->prof_events[h->prof_used + 1], s);
        h->prof_used += 2;
    }
};

// captured graphs bake weight pointers, kernel selection and table addresses: drop them whenever one of those changes
void drop_graphs(dsg_handle h) {
    h->plan_gen++;   // ... and the launch plans are validated again before their next use (validate_plan)
    for (auto &kv : h->ws) {
        Workspace *w = kv.second.get();
        if (w->graph) { (void)hipGraphExecDestroy(w->graph); w->graph = nullptr; }
        if (w->graph_uniform) { (void)hipGraphExecDestroy(w->graph_uniform); w->graph_uniform = nullptr; }
        for (auto &g : w->step_graphs) (void)hipGraphExecDestroy(g.second.first);
        w->step_graphs.clear();
    }
}

int level_window(const dsg_config &c, int lvl) {
    const int r = c.max_node_num >> lvl;
    return r <= c.window_size ? r : c.window_size;
}
int block_shift(const dsg_config &c, int lvl, int j) {
    const int r = c.max_node_num >> lvl;
    if (r <= c.window_size) return 0;
    return (j % 2 == 0) ? 0 : c.window_size / 2;
}

void add_block_specs(std::vector<WSpec> &s, const std::string &p, int C, int heads, int ws, int res, int shift, int mr) {
    const int64_t W = (int64_t)ws * ws;
    s.push_back({p + ".affine.weight", {2 * C, NOISE_EMB}});
    s.push_back({p + ".affine.bias", {2 * C}});
    s.push_back({p + ".norm1.weight", {C}});
    s.push_back({p + ".norm1.bias", {C}});
    s.push_back({p + ".attn.relative_position_bias_table", {(2 * ws - 1) * (2 * ws - 1), heads}});
    s.push_back({p + ".attn.relative_position_index", {W, W}, true, false});
    s.push_back({p + ".attn.qkv.weight", {3 * C, C}});
    s.push_back({p + ".attn.qkv.bias", {3 * C}});
    s.push_back({p + ".attn.proj.weight", {C, C}});
    s.push_back({p + ".attn.proj.bias", {C}});
    s.push_back({p + ".norm2.weight", {C}});
    s.push_back({p + ".norm2.bias", {C}});
    s.push_back({p + ".mlp.fc1.weight", {mr * C, C}});
    s.push_back({p + ".mlp.fc1.bias", {mr * C}});
    s.push_back({p + ".mlp.fc2.weight", {C, mr * C}});
    s.push_back({p + ".mlp.fc2.bias", {C}});
    if (shift > 0) {
        const int64_t nW = (int64_t)(res / ws) * (res / ws);
        s.push_back({p + ".attn_mask", {nW, W, W}, false, true});
    }
}

// The key list of DiffuseSG(...).state_dict() (diffusesg.py:587-720), same derivation as diffusesg_amd/spec.py
void build_specs(dsg_handle h) {
    const dsg_config &c = h->cfg;
    const int E = c.embed_dim, L = c.num_layers, mr = c.mlp_ratio;
    auto &s = h->specs;
    s.push_back({"patch_embed.affine.weight", {2 * E, NOISE_EMB}});
    s.push_back({"patch_embed.affine.bias", {2 * E}});
    s.push_back({"patch_embed.proj.weight", {E, h->Cin, 1, 1}});
    s.push_back({"patch_embed.proj.bias", {E}});
    s.push_back({"patch_embed.norm.weight", {E}});
    s.push_back({"patch_embed.norm.bias", {E}});
    for (int l = 0; l < L; l++) {
        const int C = E << l, res = c.max_node_num >> l, ws = level_window(c, l);
        for (int j = 0; j < c.depths[l]; j++)
            add_block_specs(s, "down_layers." + std::to_string(l) + ".blocks." + std::to_string(j), C, c.num_heads[l], ws, res,
                            block_shift(c, l, j), mr);
        if (l < L - 1) {
            const std::string p = "down_layers." + std::to_string(l) + ".downsample";
            s.push_back({p + ".reduction.weight", {2 * C, 4 * C}});
            s.push_back({p + ".norm.weight", {4 * C}});
            s.push_back({p + ".norm.bias", {4 * C}});
        }
    }
    for (int i = 0; i < L; i++) {
        const int l = L - 1 - i;
        const int C = E << l, res = c.max_node_num >> l, ws = level_window(c, l);
        if (i > 0) {
            const int D = 4 * C;
            const std::string p = "up_layers." + std::to_string(i) + ".upsample";
            s.push_back({p + ".pre_linear.weight", {D, D}});
            s.push_back({p + ".norm.weight", {D}});
            s.push_back({p + ".norm.bias", {D}});
            s.push_back({p + ".post_linear.weight", {D / 4, D / 4}});
            s.push_back({p + ".post_norm.weight", {D / 4}});
            s.push_back({p + ".post_norm.bias", {D / 4}});
        }
        for (int j = 0; j < c.depths[l]; j++)
            add_block_specs(s, "up_layers." + std::to_string(i) + ".blocks." + std::to_string(j), C, c.num_heads[l], ws, res,
                            block_shift(c, l, j), mr);
    }
    for (int k = 0; k < 3; k++) {
        s.push_back({"read_out." + std::to_string(k) + ".weight", {E, E, 1, 1}});
        s.push_back({"read_out." + std::to_string(k) + ".bias", {E}});
    }
    s.push_back({"map_layer0.weight", {NOISE_EMB, E}});
    s.push_back({"map_layer0.bias", {NOISE_EMB}});
    s.push_back({"map_layer1.weight", {NOISE_EMB, NOISE_EMB}});
    s.push_back({"map_layer1.bias", {NOISE_EMB}});
    s.push_back({"norm.weight", {E}});
    s.push_back({"norm.bias", {E}});
    s.push_back({"readout_adj_mlp.fc1.weight", {E, E}});
    s.push_back({"readout_adj_mlp.fc1.bias", {E}});
    s.push_back({"readout_adj_mlp.fc2.weight", {c.c_adj, E}});
    s.push_back({"readout_adj_mlp.fc2.bias", {c.c_adj}});
    s.push_back({"readout_node_mlp.fc1.weight", {E, E}});
    s.push_back({"readout_node_mlp.fc1.bias", {E}});
    s.push_back({"readout_node_mlp.fc2.weight", {c.c_node, E}});
    s.push_back({"readout_node_mlp.fc2.bias", {c.c_node}});
}

std::string strip_prefix(const char *key) {
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);   // DDP wrapper (sampling_utils.py:47-53)
    if (k.rfind("model.", 0) == 0) k = k.substr(6);    // Precond.model (precond.py:15)
    return k;
}

const float *WT(dsg_handle h, const std::string &key) { return h->w.at(key).p; }

int dev_alloc(dsg_handle h, std::vector<void *> &pool, void **p, size_t bytes) {
    HIP_TRY(h, hipMalloc(p, bytes ? bytes : 16));
    pool.push_back(*p);
    return 0;
}

// dense, key-major (transposed) bias+mask table of one block: [nWt][heads][Wp][Wp], entry [key][query]
// bias: relative_position_bias_table[index[query][key]][head] (diffusesg.py:121-124)
// mask: 0 / -100 between different shift regions (diffusesg.py:207-226); padded key slots: -1e30
int build_bias_table(dsg_handle h, BlockPlan &bp) {
    const int ws = bp.ws, W = ws * ws, Wp = ((W + 31) / 32) * 32, heads = bp.heads, res = bp.res, shift = bp.shift;
    const int nwr = res / ws, nWt = shift > 0 ? nwr * nwr : 1;
    const DevTensor &tab = h->w.at(bp.prefix + ".attn.relative_position_bias_table");
    std::vector<float> table(tab.numel);
    HIP_TRY(h, hipMemcpy(table.data(), tab.p, sizeof(float) * tab.numel, hipMemcpyDeviceToHost));
    std::vector<float> out((size_t)nWt * heads * Wp * Wp, 0.f);
    for (int w = 0; w < nWt; w++) {
        const int wi = w / nwr, wj = w % nwr;
        std::vector<int> region(W, 0);
        if (shift > 0)
            for (int p = 0; p < W; p++) {
                const int si = wi * ws + p / ws, sj = wj * ws + p % ws;
                const int ri = si < res - ws ? 0 : (si < res - shift ? 1 : 2);
                const int rj = sj < res - ws ? 0 : (sj < res - shift ? 1 : 2);
                region[p] = 3 * ri + rj;
            }
        for (int hd = 0; hd < heads; hd++) {
            float *o = out.data() + ((size_t)w * heads + hd) * Wp * Wp;
            for (int key = 0; key < Wp; key++)
                for (int q = 0; q < Wp; q++) {
                    float v;
                    if (key >= W) v = -1.0e30f;
                    else if (q >= W) v = 0.f;
                    else {
                        const int qi = q / ws, qj = q % ws, ki = key / ws, kj = key % ws;
                        const int idx = (qi - ki + ws - 1) * (2 * ws - 1) + (qj - kj + ws - 1);
                        v = table[(size_t)idx * heads + hd];
                        if (shift > 0 && region[q] != region[key]) v += -100.0f;
                        v = (float)((double)v * kLog2e);
                    }
                    o[(size_t)key * Wp + q] = v;
                }
        }
    }
    void *p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * out.size())) return rc;
    HIP_TRY(h, hipMemcpy(p, out.data(), sizeof(float) * out.size(), hipMemcpyHostToDevice));
    bp.biasT = (float *)p;
    void *pp;
    if (int rc = dev_alloc(h, h->derived_allocs, &pp, sizeof(uint16_t) * out.size())) return rc;
    launch_bias_permute_bx(bp.biasT, pp, nWt * heads, Wp, nullptr);
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    bp.biasP = pp;
    if (ws == 10) {
        void *pf;
        if (int rc = dev_alloc(h, h->derived_allocs, &pf, sizeof(float) * out.size())) return rc;
        launch_bias_permute_f32(bp.biasT, (float *)pf, nWt * heads, Wp, nullptr);
        HIP_TRY(h, hipStreamSynchronize(nullptr));
        bp.biasF = (float *)pf;
    }
    return 0;
}

// Fragment-major packing for fused_mlp_kernel: every MFMA operand fetch becomes one coalesced 1-KiB wave load.
//   W1p[nt][s][lane][t]    = fc1.weight[32nt + (lane&31)][8s + 4(lane>>5) + t]
//   W2p[nt][ct][g][lane][t] = fc2.weight[32ct + (lane&31)][32nt + 8g + 4(lane>>5) + t]
int pack_mlp_weights(dsg_handle h, BlockPlan &bp) {
    const int C = bp.C, Hd = h->cfg.mlp_ratio * C;
    if (!(C == 96 || C == 192) || h->cfg.mlp_ratio != 4) return 0;
    const int S = C / 8, CT = C / 32, NT = Hd / 32;
    std::vector<float> w1((size_t)Hd * C), w2((size_t)C * Hd), p1((size_t)Hd * C), p2((size_t)C * Hd);
    HIP_TRY(h, hipMemcpy(w1.data(), WT(h, bp.prefix + ".mlp.fc1.weight"), sizeof(float) * w1.size(), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(w2.data(), WT(h, bp.prefix + ".mlp.fc2.weight"), sizeof(float) * w2.size(), hipMemcpyDeviceToHost));
    for (int nt = 0; nt < NT; nt++)
        for (int s = 0; s < S; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int t = 0; t < 4; t++)
                    p1[(((size_t)nt * S + s) * 64 + lane) * 4 + t] = w1[(size_t)(32 * nt + (lane & 31)) * C + 8 * s + 4 * (lane >> 5) + t];
    for (int nt = 0; nt < NT; nt++)
        for (int ct = 0; ct < CT; ct++)
            for (int g = 0; g < 4; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int t = 0; t < 4; t++)
                        p2[((((size_t)nt * CT + ct) * 4 + g) * 64 + lane) * 4 + t] =
                            w2[(size_t)(32 * ct + (lane & 31)) * Hd + 32 * nt + 8 * g + 4 * (lane >> 5) + t];
    void *p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * p1.size())) return rc;
    bp.w1p = (float *)p;
    HIP_TRY(h, hipMemcpy(bp.w1p, p1.data(), sizeof(float) * p1.size(), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * p2.size())) return rc;
    bp.w2p = (float *)p;
    HIP_TRY(h, hipMemcpy(bp.w2p, p2.data(), sizeof(float) * p2.size(), hipMemcpyHostToDevice));
    return 0;
}

// W' = W.diag(gamma), b' = b + W.beta for a linear that consumes LayerNorm output (W device [N,K], gamma/beta [K])
int fold_ln(dsg_handle h, const float *W, const float *bias, const float *gamma, const float *beta, int N, int K, float **Wf, float **bf,
            int scaled_rows = 0, double row_scale = 1.0) {
    std::vector<float> w((size_t)N * K), b(N, 0.f), g(K), be(K);
    HIP_TRY(h, hipMemcpy(w.data(), W, sizeof(float) * w.size(), hipMemcpyDeviceToHost));
    if (bias) HIP_TRY(h, hipMemcpy(b.data(), bias, sizeof(float) * N, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(g.data(), gamma, sizeof(float) * K, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(be.data(), beta, sizeof(float) * K, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; n++) {
        double acc = b[n];
        const double rs = n < scaled_rows ? row_scale : 1.0;
        for (int k = 0; k < K; k++) { acc += (double)w[(size_t)n * K + k] * be[k]; w[(size_t)n * K + k] = (float)((double)w[(size_t)n * K + k] * g[k] * rs); }
        b[n] = (float)(acc * rs);
    }
    void *p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * w.size())) return rc;
    *Wf = (float *)p;
    HIP_TRY(h, hipMemcpy(p, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * N)) return rc;
    *bf = (float *)p;
    HIP_TRY(h, hipMemcpy(p, b.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    return 0;
}

// Fragment-major packing for fused_attn96_kernel (same two patterns as the MLP):
//   Wqp[nt][s][lane][t]     = qkv.weight[32nt + (lane&31)][8s + 4(lane>>5) + t]      nt = {q,k,v} x head
//   Wpp[hd][ct][g][lane][t] = proj.weight[32ct + (lane&31)][32hd + 8g + 4(lane>>5) + t]
int pack_attn_weights(dsg_handle h, BlockPlan &bp) {
    const int C = bp.C;
    if (C != 96 || bp.ws * bp.ws > 64) return 0;
    const int S = C / 8, CT = C / 32, NT = 3 * C / 32, HD = C / 32;
    std::vector<float> wq((size_t)3 * C * C), wp((size_t)C * C), p1(wq.size()), p2(wp.size());
    HIP_TRY(h, hipMemcpy(wq.data(), WT(h, bp.prefix + ".attn.qkv.weight"), sizeof(float) * wq.size(), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(wp.data(), WT(h, bp.prefix + ".attn.proj.weight"), sizeof(float) * wp.size(), hipMemcpyDeviceToHost));
    for (int nt = 0; nt < NT; nt++)
        for (int s = 0; s < S; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int t = 0; t < 4; t++)
                    p1[(((size_t)nt * S + s) * 64 + lane) * 4 + t] =
                        (float)((double)wq[(size_t)(32 * nt + (lane & 31)) * C + 8 * s + 4 * (lane >> 5) + t] * (nt < HD ? kQScale : 1.0));
    std::vector<float> bq((size_t)3 * C);
    HIP_TRY(h, hipMemcpy(bq.data(), WT(h, bp.prefix + ".attn.qkv.bias"), sizeof(float) * bq.size(), hipMemcpyDeviceToHost));
    for (int n = 0; n < C; n++) bq[n] = (float)((double)bq[n] * kQScale);
    for (int hd = 0; hd < HD; hd++)
        for (int ct = 0; ct < CT; ct++)
            for (int g = 0; g < 4; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int t = 0; t < 4; t++)
                        p2[((((size_t)hd * CT + ct) * 4 + g) * 64 + lane) * 4 + t] =
                            wp[(size_t)(32 * ct + (lane & 31)) * C + 32 * hd + 8 * g + 4 * (lane >> 5) + t];
    void *p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * p1.size())) return rc;
    bp.wqp = (float *)p;
    HIP_TRY(h, hipMemcpy(bp.wqp, p1.data(), sizeof(float) * p1.size(), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * p2.size())) return rc;
    bp.wpp = (float *)p;
    HIP_TRY(h, hipMemcpy(bp.wpp, p2.data(), sizeof(float) * p2.size(), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * bq.size())) return rc;
    bp.bqkv_s = (float *)p;
    HIP_TRY(h, hipMemcpy(bp.bqkv_s, bq.data(), sizeof(float) * bq.size(), hipMemcpyHostToDevice));
    return 0;
}

const void *bf16_of(dsg_handle h, const float *W) {
    if (!h->opt_gemm_bf16) return nullptr;
    auto it = h->w_bf16.find(W);
    return it == h->w_bf16.end() ? nullptr : it->second;
}

int ensure_bf16_weights(dsg_handle h) {
    for (auto &pw : h->gemm_weights) {
        if (h->w_bf16.count(pw.first)) continue;
        void *q;
        HIP_TRY(h, hipMalloc(&q, pw.second * 2));
        launch_f32_to_bf16(pw.first, q, pw.second, nullptr);
        h->w_bf16[pw.first] = q;
    }
    // the C = 96 blocks' fc1 / fc2 / proj weights in fragment order for the LDS-resident MLP kernel (launch_mlp96r_image)
    if (h->cfg.mlp_ratio == 4)
        for (int l = 0; l < h->L; l++)
            for (auto *vec : {&h->down[l], &h->up[l]})
                for (auto &b : *vec) {
                    if (b.C != 96 || !b.fc1_wf || h->w_img96.count(b.fc1_wf)) continue;
                    const float *w2 = WT(h, b.prefix + ".mlp.fc2.weight"), *wp = WT(h, b.prefix + ".attn.proj.weight");
                    if (!h->w_bf16.count(b.fc1_wf) || !h->w_bf16.count(w2) || !h->w_bf16.count(wp)) continue;
                    void *q;
                    HIP_TRY(h, hipMalloc(&q, mlp96r_image_bytes()));
                    launch_mlp96r_image(h->w_bf16[b.fc1_wf], h->w_bf16[w2], h->w_bf16[wp], q, nullptr);
                    h->w_img96[b.fc1_wf] = q;
                }
    // every block's QKV weight once more in the fragment order qkv_attn_wx_kernel streams (launch_qkv_image)
    for (int l = 0; l < h->L; l++)
        for (auto *vec : {&h->down[l], &h->up[l]})
            for (auto &b : *vec) {
                if (!b.qkv_wf || b.ws != 10 || h->w_qimg.count(b.qkv_wf) || !h->w_bf16.count(b.qkv_wf)) continue;
                void *q;
                HIP_TRY(h, hipMalloc(&q, (size_t)3 * b.C * b.C * 2));
                launch_qkv_image(h->w_bf16[b.qkv_wf], q, b.C, b.heads, nullptr);
                h->w_qimg[b.qkv_wf] = q;
            }
    // the C = 384 blocks' fc1 / fc2 / proj weights once more, in the piece order the LDS-DMA MLP kernel streams and reads them
    if (h->cfg.mlp_ratio == 4)
        for (int l = 0; l < h->L; l++)
            for (auto *vec : {&h->down[l], &h->up[l]})
                for (auto &b : *vec) {
                    if (b.C != 384 || !b.fc1_wf || h->w_img.count(b.fc1_wf)) continue;
                    const float *w2 = WT(h, b.prefix + ".mlp.fc2.weight"), *wp = WT(h, b.prefix + ".attn.proj.weight");
                    if (!h->w_bf16.count(b.fc1_wf) || !h->w_bf16.count(w2) || !h->w_bf16.count(wp)) continue;
                    void *q;
                    HIP_TRY(h, hipMalloc(&q, mlp384_image_bytes()));
                    launch_mlp384_images(h->w_bf16[b.fc1_wf], h->w_bf16[w2], h->w_bf16[wp], q, nullptr);
                    h->w_img[b.fc1_wf] = q;
                    HIP_TRY(h, hipMalloc(&q, mlp384_image_bytes()));
                    launch_mlp384s_images(h->w_bf16[b.fc1_wf], h->w_bf16[w2], h->w_bf16[wp], q, nullptr);
                    h->w_img2[b.fc1_wf] = q;
                }
    HIP_TRY(h, hipDeviceSynchronize());
    return 0;
}
const void *img_of(dsg_handle h, const float *fc1_wf) {
    auto it = h->w_img.find(fc1_wf);
    return it == h->w_img.end() ? nullptr : it->second;
}
const void *img2_of(dsg_handle h, const float *fc1_wf) {
    auto it = h->w_img2.find(fc1_wf);
    return it == h->w_img2.end() ? nullptr : it->second;
}

const void *split_of(dsg_handle h, const float *W) {
    if (!h->opt_gemm_split) return nullptr;
    auto it = h->w_split.find(W);
    return it == h->w_split.end() ? nullptr : it->second;
}

int ensure_split_weights(dsg_handle h) {
    for (auto &pw : h->gemm_weights) {
        if (h->w_split.count(pw.first)) continue;
        void *q;
        HIP_TRY(h, hipMalloc(&q, pw.second * 6));
        launch_f32_split3(pw.first, q, pw.second, nullptr);
        h->w_split[pw.first] = q;
    }
    HIP_TRY(h, hipDeviceSynchronize());
    return 0;
}

bool env_on(const char *name, bool dflt) {
    const char *v = getenv(name);
    return v ? (v[0] != '0') : dflt;
}

}  // namespace

extern "C" {

const char *dsg_version(void) { return "dsg-gfx950 0.4 (fp32 MFMA; ABI 4)"; }
int32_t dsg_abi_version(void) { return DSG_ABI_VERSION; }
// dsg_last_error(NULL): why the last dsg_create on this thread failed (there is no handle to ask then)
static thread_local std::string g_create_err;
// live handles of the process: when the last one is destroyed the per-stream scratch of the handle-less training kernels is released too
// (train_kernels.hip keeps it per hipStream_t and it only grows; a destroyed stream's address may be reused by a later stream)
static std::atomic<int> g_live_handles{0};

int dsg_create(const dsg_config *cfg, dsg_handle *out) {
    if (!cfg || !out) { g_create_err = "dsg_create: null argument"; return DSG_ERR_INVALID; }
    *out = nullptr;
    g_create_err.clear();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "dsg_create: no HIP device (there is no CPU path)"; return DSG_ERR_HIP; }
    auto h = new dsg_handle_s();
    h->cfg = *cfg;
    h->N = cfg->max_node_num; h->Ca = cfg->c_adj; h->Cn = cfg->c_node; h->E = cfg->embed_dim; h->L = cfg->num_layers;
    auto bad = [&](const char *m) { delete h; g_create_err = std::string("dsg_create: ") + m; return DSG_ERR_INVALID; };
    if (h->L < 1 || h->L > DSG_MAX_LAYERS) return bad("num_layers");
    if (h->E % 32 != 0 || h->E < 64) return bad("embed_dim must be a multiple of 32");
    if (cfg->mlp_ratio < 1 || h->N < 1 || h->Ca < 1 || h->Cn < 1) return bad("sizes");
    if (h->N % (1 << (h->L - 1)) != 0) return bad("max_node_num not divisible by 2^(L-1)");
    for (int l = 0; l < h->L; l++) {
        const int C = h->E << l, res = h->N >> l, ws = level_window(*cfg, l);
        if (cfg->num_heads[l] * 32 != C) return bad("head_dim must be 32");
        if (res % ws != 0) return bad("resolution not divisible by window");
        if (!(ws == 2 || ws == 4 || ws == 5 || ws == 8 || ws == 10)) return bad("window side must be one of 2, 4, 5, 8, 10");
        if (C > 1536 || (l < h->L - 1 && 4 * C > 1536)) return bad("row wider than 1536 channels");
    }
    h->Cin = (cfg->self_condition ? 2 : 1) * (h->Ca + 2 * h->Cn);
    h->Kp = ((h->Cin + 31) / 32) * 32;
    build_specs(h);
    if (!gelu_table()) { delete h; return DSG_ERR_HIP; }
    h->opt_fused_attn = env_on("DSG_FUSED_ATTN", true);
    h->opt_fused_mlp = env_on("DSG_FUSED_MLP", true);
    h->opt_fused_readout = env_on("DSG_FUSED_READOUT", true);
    h->opt_fused_pe = env_on("DSG_FUSED_PE", true);
    h->opt_fused_rowstats = env_on("DSG_FUSED_ROWSTATS", true);
    h->opt_fused_qkv_attn = env_on("DSG_FUSED_QKV_ATTN", true);
    h->opt_loop_graph = env_on("DSG_LOOP_GRAPH", true);
    h->opt_fused_merge = env_on("DSG_FUSED_MERGE", true);
    if (getenv("DSG_FUSED_MLP_MAXC")) h->opt_fused_mlp_maxc = atoi(getenv("DSG_FUSED_MLP_MAXC"));
    h->opt_gemm_bf16 = env_on("DSG_GEMM_BF16", false);
    h->opt_bf16_pipe = env_on("DSG_BF16_PIPE", true);
    if (getenv("DSG_BF16_QA")) h->opt_bf16_qkv_attn = atoi(getenv("DSG_BF16_QA"));   // dev knob: 2 = the block-per-head QKV + attention kernel
    if (getenv("DSG_BF16_MLP")) h->opt_bf16_mlp = atoi(getenv("DSG_BF16_MLP"));   // dev knob: A/B of the C = 384 MLP kernels (1 LDS-DMA, 4 round 3's)
    h->opt_gemm_split = env_on("DSG_GEMM_SPLIT", false);
    *out = h;
    g_live_handles++;
    return DSG_OK;
}

void dsg_destroy(dsg_handle h) {
    if (!h) return;
    for (auto &kv : h->w) (void)hipFree(kv.second.p);
    for (auto &kv : h->w_bf16) (void)hipFree(kv.second);
    for (auto &kv : h->w_img) (void)hipFree(kv.second);
    for (auto &kv : h->w_img2) (void)hipFree(kv.second);
    for (auto &kv : h->w_qimg) (void)hipFree(kv.second);
    for (auto &kv : h->w_img96) (void)hipFree(kv.second);
    for (auto &kv : h->w_split) (void)hipFree(kv.second);
    for (void *p : h->derived_allocs) (void)hipFree(p);
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    if (h->prof_gemm) (void)hipFree(h->prof_gemm);
    for (float *q : {h->tab_sig, h->tab_cn, h->tab_pe, h->tab_e0, h->tab_e1, h->tab_aff}) if (q) (void)hipFree(q);
    drop_graphs(h);
    if (h->tab_step) (void)hipFree(h->tab_step);
    if (h->train_arena) (void)hipFree(h->train_arena);
    if (h->train_scr) (void)hipFree(h->train_scr);
    if (h->train_io) (void)hipFree(h->train_io);
    if (h->train_mid) (void)hipFree(h->train_mid);
    if (h->train_const) (void)hipFree(h->train_const);
    for (auto &kv : h->ws) {
        if (kv.second->cap_stream) (void)hipStreamDestroy(kv.second->cap_stream);
        for (void *p : kv.second->allocs) (void)hipFree(p);
    }
    delete h;
    if (--g_live_handles == 0) { (void)hipDeviceSynchronize(); t_scratch_release(); }
}

const char *dsg_last_error(dsg_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int dsg_num_weight_keys(dsg_handle h) { return h ? (int)h->specs.size() : 0; }
const char *dsg_weight_key(dsg_handle h, int32_t i) {
    if (!h || i < 0 || i >= (int)h->specs.size()) return nullptr;
    return h->specs[i].key.c_str();
}

int dsg_set_weight(dsg_handle h, const char *key, const void *data, const int64_t *shape, int32_t ndim, int32_t is_device) {
    if (!h || !key || !data) return DSG_ERR_INVALID;
    const std::string k = strip_prefix(key);
    const WSpec *sp = nullptr;
    for (auto &s : h->specs) if (s.key == k) { sp = &s; break; }
    if (!sp) return fail(h, DSG_ERR_WEIGHTS, "unexpected key '%s'", key);
    if ((int)sp->shape.size() != ndim) return fail(h, DSG_ERR_WEIGHTS, "rank mismatch for '%s'", key);
    int64_t numel = 1;
    for (int i = 0; i < ndim; i++) {
        if (shape[i] != sp->shape[i]) return fail(h, DSG_ERR_WEIGHTS, "shape mismatch for '%s' (dim %d: %lld vs %lld)", key, i,
                                                  (long long)shape[i], (long long)sp->shape[i]);
        numel *= shape[i];
    }
    DevTensor &t = h->w[k];
    if (sp->is_index || sp->is_mask) {  // constant buffers: re-derived by the library, only registered as present
        t.numel = numel; t.shape = sp->shape;
        return DSG_OK;
    }
    if (!t.p) HIP_TRY(h, hipMalloc((void **)&t.p, sizeof(float) * numel));
    HIP_TRY(h, hipMemcpy(t.p, data, sizeof(float) * numel, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    t.numel = numel; t.shape = sp->shape;
    h->finalized = false;
    return DSG_OK;
}

int dsg_finalize_weights(dsg_handle h) {
    if (!h) return DSG_ERR_INVALID;
    for (auto &s : h->specs)
        if (!h->w.count(s.key)) return fail(h, DSG_ERR_WEIGHTS, "missing key '%s' (strict load)", s.key.c_str());
    for (void *p : h->derived_allocs) (void)hipFree(p);
    h->derived_allocs.clear();
    for (int l = 0; l < DSG_MAX_LAYERS; l++) { h->down[l].clear(); h->up[l].clear(); }
    const dsg_config &c = h->cfg;
    const int E = h->E, L = h->L;
    // block plans + concatenated affine (all (scale,shift) linears of the net in one [aff_n,512] GEMM)
    std::vector<std::string> aff_prefixes;
    int off = 0;
    h->pe_aff_off = off; aff_prefixes.push_back("patch_embed"); off += 2 * E;
    auto mk = [&](const std::string &p, int l, int j) {
        BlockPlan b;
        b.prefix = p; b.lvl = l; b.C = E << l; b.res = c.max_node_num >> l; b.ws = level_window(c, l);
        b.shift = block_shift(c, l, j); b.heads = c.num_heads[l]; b.aff_off = off;
        aff_prefixes.push_back(p); off += 2 * b.C;
        return b;
    };
    for (int l = 0; l < L; l++)
        for (int j = 0; j < c.depths[l]; j++)
            h->down[l].push_back(mk("down_layers." + std::to_string(l) + ".blocks." + std::to_string(j), l, j));
    for (int i = 0; i < L; i++)
        for (int j = 0; j < c.depths[L - 1 - i]; j++)
            h->up[i].push_back(mk("up_layers." + std::to_string(i) + ".blocks." + std::to_string(j), L - 1 - i, j));
    h->aff_n = off;
    void *p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * (size_t)off * NOISE_EMB)) return rc;
    h->aff_w = (float *)p;
    if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * (size_t)off)) return rc;
    h->aff_b = (float *)p;
    {
        size_t o = 0;
        for (auto &pre : aff_prefixes) {
            const DevTensor &w = h->w.at(pre + ".affine.weight"), &b = h->w.at(pre + ".affine.bias");
            HIP_TRY(h, hipMemcpy(h->aff_w + o * NOISE_EMB, w.p, sizeof(float) * w.numel, hipMemcpyDeviceToDevice));
            HIP_TRY(h, hipMemcpy(h->aff_b + o, b.p, sizeof(float) * b.numel, hipMemcpyDeviceToDevice));
            o += b.numel;
        }
    }
    for (int l = 0; l < L; l++) {
        for (auto *vec : {&h->down[l], &h->up[l]})
            for (auto &b : *vec) {
                if (int rc = build_bias_table(h, b)) return rc;
                if (int rc = pack_mlp_weights(h, b)) return rc;
                if (int rc = pack_attn_weights(h, b)) return rc;
                const int C = b.C, Hd = h->cfg.mlp_ratio * C;
                if (int rc = fold_ln(h, WT(h, b.prefix + ".attn.qkv.weight"), WT(h, b.prefix + ".attn.qkv.bias"), WT(h, b.prefix + ".norm1.weight"),
                                     WT(h, b.prefix + ".norm1.bias"), 3 * C, C, &b.qkv_wf, &b.qkv_bf, C, kQScale)) return rc;
                if (int rc = fold_ln(h, WT(h, b.prefix + ".mlp.fc1.weight"), WT(h, b.prefix + ".mlp.fc1.bias"), WT(h, b.prefix + ".norm2.weight"),
                                     WT(h, b.prefix + ".norm2.bias"), Hd, C, &b.fc1_wf, &b.fc1_bf)) return rc;
            }
    }
    for (int l = 0; l + 1 < L; l++) {   // PatchMerging: norm(4C) folded into reduction (no bias in the reference: the folded bias is W.beta)
        const std::string pm = "down_layers." + std::to_string(l) + ".downsample";
        const int C = E << l;
        if (int rc = fold_ln(h, WT(h, pm + ".reduction.weight"), nullptr, WT(h, pm + ".norm.weight"), WT(h, pm + ".norm.bias"), 2 * C, 4 * C,
                             &h->merge_wf[l], &h->merge_bf[l])) return rc;
    }
    // patch_embed.proj [E,Cin,1,1] -> [E,Kp] zero padded
    {
        std::vector<float> src((size_t)E * h->Cin), dst((size_t)E * h->Kp, 0.f);
        HIP_TRY(h, hipMemcpy(src.data(), WT(h, "patch_embed.proj.weight"), sizeof(float) * src.size(), hipMemcpyDeviceToHost));
        for (int e = 0; e < E; e++)
            for (int k = 0; k < h->Cin; k++) dst[(size_t)e * h->Kp + k] = src[(size_t)e * h->Cin + k];
        if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * dst.size())) return rc;
        h->pe_w = (float *)p;
        HIP_TRY(h, hipMemcpy(h->pe_w, dst.data(), sizeof(float) * dst.size(), hipMemcpyHostToDevice));
        h->pe_wp = nullptr;
        if (E == 96 && h->Kp <= 64) {
            const int S = h->Kp / 8;
            std::vector<float> pk(dst.size());
            for (int nt = 0; nt < E / 32; nt++)
                for (int sx = 0; sx < S; sx++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int t = 0; t < 4; t++)
                            pk[(((size_t)nt * S + sx) * 64 + lane) * 4 + t] = dst[(size_t)(32 * nt + (lane & 31)) * h->Kp + 8 * sx + 4 * (lane >> 5) + t];
            if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * pk.size())) return rc;
            h->pe_wp = (float *)p;
            HIP_TRY(h, hipMemcpy(h->pe_wp, pk.data(), sizeof(float) * pk.size(), hipMemcpyHostToDevice));
        }
    }
    // read_out.0 is a ConvTranspose2d: weight [in,out,1,1] (diffusesg.py:706) -> [out,in]
    {
        std::vector<float> src((size_t)E * E), dst((size_t)E * E);
        HIP_TRY(h, hipMemcpy(src.data(), WT(h, "read_out.0.weight"), sizeof(float) * src.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < E; i++)
            for (int o = 0; o < E; o++) dst[(size_t)o * E + i] = src[(size_t)i * E + o];
        if (int rc = dev_alloc(h, h->derived_allocs, &p, sizeof(float) * dst.size())) return rc;
        h->ro0_w = (float *)p;
        HIP_TRY(h, hipMemcpy(h->ro0_w, dst.data(), sizeof(float) * dst.size(), hipMemcpyHostToDevice));
        if (int rc = fold_ln(h, h->ro0_w, WT(h, "read_out.0.bias"), WT(h, "norm.weight"), WT(h, "norm.bias"), E, E, &h->ro0_wf, &h->ro0_bf)) return rc;
    }
    // Folded read-out (E = 96, C_adj <= 32): final-LN output -> read_out.0/1/2 -> readout_adj_mlp.fc1 is affine up to the
    // GELU, and the node head's pooled shared_rep is an affine image of the pooled LN output.  Fold in double precision.
    h->ro_fap = h->ro_fa = h->ro_f2p = h->ro_gext = nullptr;
    h->ro_fapb = h->ro_f2pb = nullptr;
    if (E == 96 && h->Ca <= 32) {
        auto dl = [&](const char *k, size_t n) { std::vector<float> v(n); (void)hipMemcpy(v.data(), WT(h, k), sizeof(float) * n, hipMemcpyDeviceToHost); return v; };
        const std::vector<float> r0 = dl("read_out.0.weight", (size_t)E * E), r1 = dl("read_out.1.weight", (size_t)E * E),
                                 r2 = dl("read_out.2.weight", (size_t)E * E), b0 = dl("read_out.0.bias", E), b1 = dl("read_out.1.bias", E),
                                 b2 = dl("read_out.2.bias", E), F1 = dl("readout_adj_mlp.fc1.weight", (size_t)E * E),
                                 f1 = dl("readout_adj_mlp.fc1.bias", E), F2 = dl("readout_adj_mlp.fc2.weight", (size_t)h->Ca * E),
                                 G1 = dl("readout_node_mlp.fc1.weight", (size_t)E * E);
        typedef std::vector<double> dv;
        auto matmul = [&](const dv &X, const dv &Y) { dv Z((size_t)E * E, 0.0); for (int i = 0; i < E; i++) for (int k = 0; k < E; k++) { const double a = X[(size_t)i * E + k]; for (int j = 0; j < E; j++) Z[(size_t)i * E + j] += a * Y[(size_t)k * E + j]; } return Z; };
        auto matvec = [&](const dv &X, const dv &v) { dv z(E, 0.0); for (int i = 0; i < E; i++) for (int k = 0; k < E; k++) z[i] += X[(size_t)i * E + k] * v[k]; return z; };
        dv W0t((size_t)E * E), W1(r1.begin(), r1.end()), W2(r2.begin(), r2.end()), dF1(F1.begin(), F1.end()), dG1(G1.begin(), G1.end());
        for (int i = 0; i < E; i++) for (int o = 0; o < E; o++) W0t[(size_t)o * E + i] = r0[(size_t)i * E + o];  // ConvTranspose2d [in,out]
        const dv A = matmul(W2, matmul(W1, W0t));
        dv t1 = matvec(W1, dv(b0.begin(), b0.end()));
        for (int i = 0; i < E; i++) t1[i] += b1[i];
        dv a = matvec(W2, t1);
        for (int i = 0; i < E; i++) a[i] += b2[i];
        const dv Fa = matmul(dF1, A);
        dv fa = matvec(dF1, a);
        for (int i = 0; i < E; i++) fa[i] += f1[i];
        const dv Gn = matmul(dG1, A), ga = matvec(dG1, a);
        const int S = E / 8;
        std::vector<float> fap((size_t)E * E), f2p((size_t)3 * 4 * 64 * 4, 0.f), gext((size_t)E * 128, 0.f), faf(E);
        for (int nt = 0; nt < 3; nt++)
            for (int sx = 0; sx < S; sx++)
                for (int lane = 0; lane < 64; lane++)
                    for (int t = 0; t < 4; t++)
                        fap[(((size_t)nt * S + sx) * 64 + lane) * 4 + t] = (float)Fa[(size_t)(32 * nt + (lane & 31)) * E + 8 * sx + 4 * (lane >> 5) + t];
        for (int nt = 0; nt < 3; nt++)
            for (int g = 0; g < 4; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int t = 0; t < 4; t++) {
                        const int c = lane & 31;
                        f2p[(((size_t)nt * 4 + g) * 64 + lane) * 4 + t] = c < h->Ca ? F2[(size_t)c * E + 32 * nt + 8 * g + 4 * (lane >> 5) + t] : 0.f;
                    }
        for (int i = 0; i < E; i++) {
            faf[i] = (float)fa[i];
            for (int k = 0; k < E; k++) gext[(size_t)i * 128 + k] = (float)Gn[(size_t)i * E + k];
            gext[(size_t)i * 128 + 96] = (float)ga[i];
        }
        auto up = [&](const std::vector<float> &v, float **dst) -> int {
            void *q;
            if (int rc = dev_alloc(h, h->derived_allocs, &q, sizeof(float) * v.size())) return rc;
            *dst = (float *)q;
            HIP_TRY(h, hipMemcpy(q, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice));
            return 0;
        };
        if (int rc = up(fap, &h->ro_fap)) return rc;
        if (int rc = up(faf, &h->ro_fa)) return rc;
        if (int rc = up(f2p, &h->ro_f2p)) return rc;
        if (int rc = up(gext, &h->ro_gext)) return rc;
        // bf16-MFMA fragments (32x32x16: lane (row, half) holds 8 k values per k-step; k order 16 s + 8 (j >> 2) + 4 half + (j & 3), the order
        // in which fused_readout96_kernel<true>'s register pairs enumerate LN(x)'s channels / the hidden units)
        std::vector<float> fapb((size_t)3 * 6 * 64 * 8), f2pb((size_t)3 * 2 * 64 * 8, 0.f);
        for (int nt = 0; nt < 3; nt++)
            for (int sx = 0; sx < 6; sx++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++)
                        fapb[(((size_t)nt * 6 + sx) * 64 + lane) * 8 + j] =
                            (float)Fa[(size_t)(32 * nt + (lane & 31)) * E + 16 * sx + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)];
        for (int nt = 0; nt < 3; nt++)
            for (int g2 = 0; g2 < 2; g2++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int c = lane & 31;
                        f2pb[(((size_t)nt * 2 + g2) * 64 + lane) * 8 + j] =
                            c < h->Ca ? F2[(size_t)c * E + 32 * nt + 16 * g2 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)] : 0.f;
                    }
        if (int rc = up(fapb, &h->ro_fapb)) return rc;
        if (int rc = up(f2pb, &h->ro_f2pb)) return rc;
    }
    drop_graphs(h);   // captured graphs bake weight pointers
    // bf16 copies (opt-in mode): drop stale ones, list every GEMM weight, rebuild if the mode is on
    for (auto &kv : h->w_bf16) (void)hipFree(kv.second);
    h->w_bf16.clear();
    for (auto &kv : h->w_img) (void)hipFree(kv.second);
    h->w_img.clear();
    for (auto &kv : h->w_img2) (void)hipFree(kv.second);
    h->w_img2.clear();
    for (auto &kv : h->w_qimg) (void)hipFree(kv.second);
    h->w_qimg.clear();
    for (auto &kv : h->w_img96) (void)hipFree(kv.second);
    h->w_img96.clear();
    for (auto &kv : h->w_split) (void)hipFree(kv.second);
    h->w_split.clear();
    h->gemm_weights.clear();
    for (auto &kv : h->w)
        if (kv.second.p && kv.second.shape.size() >= 2) h->gemm_weights.push_back({kv.second.p, (size_t)kv.second.numel});
    for (int l = 0; l < L; l++)
        for (auto *vec : {&h->down[l], &h->up[l]})
            for (auto &b : *vec) {
                h->gemm_weights.push_back({b.qkv_wf, (size_t)3 * b.C * b.C});
                h->gemm_weights.push_back({b.fc1_wf, (size_t)c.mlp_ratio * b.C * b.C});
            }
    for (int l = 0; l + 1 < L; l++) h->gemm_weights.push_back({h->merge_wf[l], (size_t)8 * (E << l) * (E << l)});
    h->gemm_weights.push_back({h->aff_w, (size_t)h->aff_n * NOISE_EMB});
    h->gemm_weights.push_back({h->pe_w, (size_t)E * h->Kp});
    h->gemm_weights.push_back({h->ro0_wf, (size_t)E * E});
    if (h->ro_gext) h->gemm_weights.push_back({h->ro_gext, (size_t)E * 128});
    if (h->ro_fapb) h->gemm_weights.push_back({h->ro_fapb, (size_t)3 * 6 * 64 * 8});
    if (h->ro_f2pb) h->gemm_weights.push_back({h->ro_f2pb, (size_t)3 * 2 * 64 * 8});
    if (h->opt_gemm_bf16) if (int rc = ensure_bf16_weights(h)) return rc;
    if (h->opt_gemm_split) if (int rc = ensure_split_weights(h)) return rc;
    h->finalized = true;
    h->ever_finalized = true;
    return DSG_OK;
}

}  // extern "C"

namespace {

size_t per_sample_floats(dsg_handle h, std::vector<size_t> *parts = nullptr) {
    const size_t T0 = (size_t)h->N * h->N, E = h->E;
    const size_t sa = (size_t)h->Ca * h->N * h->N, sn = (size_t)h->N * h->Cn;
    std::vector<size_t> v = {
        sa, sn, sa, sn, sa, sn,                  // in, sc, f
        (size_t)E, NOISE_EMB, NOISE_EMB, (size_t)h->aff_n,   // pe, emb0, emb, aff
        T0 * h->Kp,                              // tok_in
        T0 * E, T0 * E, T0 * E / 2,              // x, y, xn (bf16)
        3 * T0 * E, T0 * E,                      // qkv, att
        (size_t)h->cfg.mlp_ratio * T0 * E,       // hid
        2 * T0,                                  // stats
        (size_t)h->N * E, (size_t)h->N * E,      // pool, hn
        (size_t)h->N * 128, (size_t)h->N * readout_pool_segments(h->N) * 96,   // pool_ext, pool_part
        sa, sn, sa, sn, 3 * sa, 3 * sn,          // sampler x, xhat, d[3]
    };
    size_t tot = 0;
    for (auto s : v) tot += s;
    for (int l = 0; l < h->L; l++) tot += (T0 * E) >> (l + 1 < h->L ? l + 1 : l);
    if (parts) *parts = v;
    return tot;
}

int validate_plan(dsg_handle h, Workspace *w);
int get_workspace(dsg_handle h, int B, Workspace **out) {
    auto it = h->ws.find(B);
    if (it != h->ws.end()) { *out = it->second.get(); return validate_plan(h, *out); }
    auto w = std::make_unique<Workspace>();
    w->B = B;
    const size_t T0 = (size_t)h->N * h->N, E = h->E;
    const size_t sa = (size_t)B * h->Ca * h->N * h->N, sn = (size_t)B * h->N * h->Cn;
    auto A = [&](float **p, size_t n) -> int {
        void *q;
        if (int rc = dev_alloc(h, w->allocs, &q, sizeof(float) * n)) return rc;
        *p = (float *)q; w->bytes += sizeof(float) * n;
        return 0;
    };
#define ALLOC(ptr, n) do { if (int rc = A(&(ptr), (n))) return rc; } while (0)
    ALLOC(w->in_adj, sa); ALLOC(w->in_node, sn); ALLOC(w->sc_adj, sa); ALLOC(w->sc_node, sn);
    ALLOC(w->f_adj, sa); ALLOC(w->f_node, sn); ALLOC(w->c_noise, B); ALLOC(w->sig, B);
    ALLOC(w->pe, (size_t)B * E); ALLOC(w->emb0, (size_t)B * NOISE_EMB); ALLOC(w->emb, (size_t)B * NOISE_EMB);
    ALLOC(w->aff, (size_t)B * h->aff_n);
    ALLOC(w->tok_in, (size_t)B * T0 * h->Kp);
    ALLOC(w->x, (size_t)B * T0 * E); ALLOC(w->y, (size_t)B * T0 * E);
    { float *xn_; ALLOC(xn_, (size_t)B * T0 * E / 2 + 64); w->xn = xn_; }
    ALLOC(w->qkv, (size_t)B * 3 * T0 * E); ALLOC(w->att, (size_t)B * T0 * E);
    ALLOC(w->hid, (size_t)B * h->cfg.mlp_ratio * T0 * E);
    ALLOC(w->stats, (size_t)B * 2 * T0);
    ALLOC(w->pool, (size_t)B * h->N * E); ALLOC(w->hn, (size_t)B * h->N * E); ALLOC(w->pool_ext, (size_t)B * h->N * 128);
    ALLOC(w->pool_part, (size_t)B * h->N * readout_pool_segments(h->N) * 96);
    for (int l = 0; l < h->L; l++) ALLOC(w->skips[l], ((size_t)B * T0 * E) >> (l + 1 < h->L ? l + 1 : l));
    ALLOC(w->x_adj, sa); ALLOC(w->x_node, sn); ALLOC(w->xh_adj, sa); ALLOC(w->xh_node, sn);
    for (int k = 0; k < 3; k++) { ALLOC(w->d_adj[k], sa); ALLOC(w->d_node[k], sn); }
#undef ALLOC
    void *q;
    if (int rc = dev_alloc(h, w->allocs, &q, (size_t)B * h->N)) return rc;
    w->flags = (uint8_t *)q;
    if (int rc = dev_alloc(h, w->allocs, &q, 16)) return rc;
    w->has_sc = (int *)q;
    if (int rc = dev_alloc(h, w->allocs, &q, sizeof(RunCtl))) return rc;
    w->ctl = (RunCtl *)q;
    *out = w.get();
    h->ws[B] = std::move(w);
    return validate_plan(h, *out);
}

void tap(dsg_handle h, const char *name, const float *src, size_t numel, hipStream_t s) {
    if (g_dry_run) return;
    for (auto &t : h->taps)
        if (t.name == name && (int64_t)numel <= t.cap)
            (void)hipMemcpyAsync(t.dst, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, s);
}

// P_GEMM: always exact fp32 (noise embedding / modulation parameters, patch embed, read-out and heads: small, and their
// error would enter every block).  P_GEMM_LP: the Swin-block, PatchMerging and PatchBreakup linears, which the opt-in
// "gemm_bf16" mode runs on bf16 MFMA.
// "gemm_split" (fp32-accurate split-bf16 products) applies to every GEMM and takes precedence over "gemm_bf16".
#define P_GEMM_LP(g) do { (g).Ws3 = split_of(h, (g).W); (g).Wb = (g).Ws3 ? nullptr : bf16_of(h, (g).W); P_GEMM_(g); } while (0)
#define P_GEMM(g) do { (g).Ws3 = split_of(h, (g).W); (g).Wb = nullptr; P_GEMM_(g); } while (0)
#define P_GEMM_(g) do { char tg_[96]; if (h->prof_stamps && h->prof_gemm && h->prof_gemm_used < h->prof_gemm_cap) (g).prof = h->prof_gemm + 4 * (h->prof_gemm_used++); else (g).prof = nullptr; \
    if (h->prof_on) snprintf(tg_, sizeof(tg_), "gemm M=%d N=%d K=%d ln=%d act=%d res=%d", (g).M, (g).N, (g).K, ((g).ln_stats != nullptr) + 2 * ((g).ln_part != nullptr) + 4 * ((g).stats_out != nullptr) + 8 * ((g).mod_aff != nullptr), (g).act, (g).res != nullptr); \
    ProfScope ps_(h, s, PK_GEMM, 2.0 * (double)(g).M * (double)(g).N * (double)(g).K, tg_); \
    if (!launch_gemm((g), s)) plan_fail(h, "gemm: argument combination not built (M=%d N=%d K=%d ln=%d act=%d res=%d bf16 A/C=%d/%d)", (g).M, (g).N, (g).K, ((g).ln_stats != nullptr) + 2 * ((g).ln_part != nullptr), (g).act, (g).res != nullptr, (g).a_bf16, (g).c_bf16); } while (0)
#define P_KERN(kind, flops, call) do { ProfScope ps_(h, s, (kind), (flops), #call); call; } while (0)

// Row-kernel fusion (fp32 GEMM kernel only; off while debug taps want the un-modulated block outputs): the GEMM that produces
// a block's input also applies that block's modulate+SiLU and leaves per-column-tile (sum, sumsq) partials of the stored rows
// in w->stats, from which the consuming GEMM forms the LayerNorm statistics -- mod_stats / ln_stats launches disappear.
bool rowstats_on(dsg_handle h) { return h->opt_fused_rowstats && !h->opt_gemm_split && h->taps.empty(); }   // fp32 and bf16 GEMM kernels
// does block `nb` take its input pre-modulated?  Generic blocks also read the LN1 partials the producer leaves; the C = 96 fused
// attention kernel only skips its own modulate+SiLU (twice: prologue and shortcut) and keeps computing LN1 itself.
bool wants_premod(dsg_handle h, const BlockPlan *nb) { return nb && rowstats_on(h); }
// attach "modulate for block nb + row statistics" to the GEMM that writes nb's input (M rows = B * T tokens of nb's level)
void attach_premod(dsg_handle h, Workspace *w, GemmArgs &g, const BlockPlan *nb) {
    if (!wants_premod(h, nb)) return;
    g.stats_out = w->stats;
    g.mod_aff = w->aff; g.mod_ld = w->aff_ld; g.mod_off = nb->aff_off; g.mod_T = nb->res * nb->res;
}

// One Swin block (diffusesg.py:232-277) on x [B*T, C] in place.  premod: x is already modulated and w->stats holds the LN1
// partials (attach_premod on the producer).  next: the block that consumes this block's output directly (same width), or null.
// want_stats: leave (sum, sumsq) partials of the block's OUTPUT rows in w->stats even without a next block (PatchMerging
// reads them).  Returns what the producer left behind: .premod -- next's input is pre-modulated (+ its LN1 partials);
// .stats_parts -- > 0: partials of the un-modulated output rows with that many pairs per row.
struct BlockOut { bool premod; int stats_parts; };
BlockOut run_block(dsg_handle h, Workspace *w, const BlockPlan &b, bool premod, const BlockPlan *next, bool want_stats, hipStream_t s) {
    const int B = w->B, T = b.res * b.res, C = b.C, M = B * T, Hd = h->cfg.mlp_ratio * C;
    const std::string &p = b.prefix;
    const bool fuse = rowstats_on(h);
    const bool mlp_fused = h->opt_fused_mlp && b.w1p && C <= h->opt_fused_mlp_maxc;
    // bf16 mode: a tensor whose only consumer is the bf16 GEMM (which rounds its A operand to bf16 on the way into LDS) is stored as
    // bf16 by its producer -- same values bit for bit, half the bytes, no conversion in the consumer (kernels_lp.hip ABF / CBF)
    auto bf16_tensor_ok = [&](const float *W, int K) {
        return h->opt_gemm_bf16 && !h->opt_gemm_split && h->opt_bf16_act && K % 64 == 0 && h->taps.empty() && bf16_of(h, W) != nullptr;
    };
    GemmArgs g;
    if (h->opt_fused_attn && b.wqp) {
        // modulate+SiLU, LN1, QKV, window attention, proj and the residual in one register-resident kernel
        WinGeom wg{b.res, b.ws, b.shift, b.heads, C};
        P_KERN(PK_FUSED, 2.0 * (double)M * C * 4.0 * C + 4.0 * (double)M * (double)(b.ws * b.ws) * (double)C,
               launch_fused_attn96(w->x, w->aff, w->aff_ld, b.aff_off, WT(h, p + ".norm1.weight"), WT(h, p + ".norm1.bias"), b.wqp,
                                   b.bqkv_s, b.biasT, b.wpp, WT(h, p + ".attn.proj.bias"), B, wg, premod, s));
    } else {
        // x <- silu(shift + x*(1+scale)) (also the shortcut), LayerNorm-1 statistics
        if (!premod) P_KERN(PK_ROW, 0.0, launch_mod_stats(w->x, w->aff, w->aff_ld, b.aff_off, w->stats, B, T, C, s));
        g.A = w->x; g.lda = C; g.K1 = C; g.K = C; g.M = M;
        if (premod) { g.ln_part = w->stats; g.ln_nparts = (C + 95) / 96; }
        else g.ln_stats = w->stats;   // gamma/beta of norm1 are folded into qkv_wf / qkv_bf
        g.W = b.qkv_wf; g.bias = b.qkv_bf; g.N = 3 * C;
        WinGeom wg{b.res, b.ws, b.shift, b.heads, C};
        bool attn_done = false, att_bf16 = false;
        if (h->opt_fused_qkv_attn && (b.ws == 8 || b.ws == 10) && !h->opt_gemm_bf16 && !h->opt_gemm_split) {
            // LN1 -> QKV -> softmax(q k^T + bias) v in one kernel: q, k, v of (two windows, one head) stay in LDS
            g.attn_bias = b.biasT; g.wg = wg; g.attn_batch = B; g.C = w->att; g.ldc = C;
            char tg_[96];
            if (h->prof_stamps && h->prof_gemm && h->prof_gemm_used < h->prof_gemm_cap) g.prof = h->prof_gemm + 4 * (h->prof_gemm_used++);
            if (h->prof_on) snprintf(tg_, sizeof(tg_), "gemm+attn M=%d N=%d K=%d ln=%d heads=%d", g.M, g.N, g.K, g.ln_part ? 2 : 1, b.heads);
            ProfScope ps_(h, s, PK_GEMM, 2.0 * (double)M * 3.0 * C * C + 4.0 * (double)M * (double)(b.ws * b.ws) * (double)C, tg_);
            attn_done = launch_gemm_qkv_attn(g, s);
        }
        if (!attn_done) {
            g.attn_bias = nullptr; g.prof = nullptr;
            g.C = w->qkv; g.ldc = 3 * C;
            att_bf16 = bf16_tensor_ok(WT(h, p + ".attn.proj.weight"), C);
            // level 2: q, k, v leave the QKV GEMM as bf16 too (the attention math stays fp32 on the widened values; this one
            // changes results -- within the bf16 mode's stated bar -- because the fp32 attention kernel is the consumer)
            const bool qkv_bf16 = att_bf16 && h->opt_bf16_act >= 2 && bf16_of(h, b.qkv_wf) != nullptr;
            g.c_bf16 = qkv_bf16;
            P_GEMM_LP(g);
            P_KERN(PK_ATTN, 4.0 * (double)M * (double)(b.ws * b.ws) * (double)C,
                   if (!launch_window_attn(w->qkv, b.biasT, w->att, B, wg, s, att_bf16, qkv_bf16))
                       plan_fail(h, "%s: window attention not built (window %d, C=%d, bf16 in/out=%d/%d)", p.c_str(), b.ws, C, qkv_bf16, att_bf16));
        }
        g = GemmArgs();
        g.A = w->att; g.lda = C; g.K1 = C; g.K = C; g.M = M; g.N = C;
        g.a_bf16 = att_bf16;
        g.W = WT(h, p + ".attn.proj.weight"); g.bias = WT(h, p + ".attn.proj.bias");
        g.res = w->x; g.ldres = C; g.C = w->x; g.ldc = C;
        if (fuse && !mlp_fused) g.stats_out = w->stats;   // LN2 partials of x + proj(...)
        P_GEMM_LP(g);
    }
    if (mlp_fused) {
        // LN2 + fc1 + GELU + fc2 + residual in one kernel, hidden activations never leave the register file
        const bool st = want_stats && fuse;
        P_KERN(PK_FUSED, 4.0 * (double)M * (double)C * (double)Hd,
               launch_fused_mlp(w->x, WT(h, p + ".norm2.weight"), WT(h, p + ".norm2.bias"), b.w1p, WT(h, p + ".mlp.fc1.bias"), b.w2p,
                                WT(h, p + ".mlp.fc2.bias"), M, C, st ? w->stats : nullptr, s));
        return BlockOut{false, st ? 1 : 0};   // the fused MLP has no modulate epilogue: the next block runs its own mod_stats
    }
    const bool ln2_part = fuse && !(h->opt_fused_attn && b.wqp);   // the proj GEMM above left the partials
    if (!ln2_part) P_KERN(PK_ROW, 0.0, launch_ln_stats(w->x, w->stats, M, C, s));
    g = GemmArgs();
    g.A = w->x; g.lda = C; g.K1 = C; g.K = C; g.M = M; g.N = Hd;
    if (ln2_part) { g.ln_part = w->stats; g.ln_nparts = (C + 95) / 96; }
    else g.ln_stats = w->stats;   // gamma/beta of norm2 are folded into fc1_wf / fc1_bf
    g.W = b.fc1_wf; g.bias = b.fc1_bf; g.act = ACT_GELU;
    g.C = w->hid; g.ldc = Hd;
    // measurement only (DSG_HID_EXP=1, wrong results): a leading dimension of 0 gives the kernels an empty buffer range -- fc1's stores of the
    // 4C-wide hidden tensor are dropped and fc2's loads of it return zeros without memory traffic, every instruction still issues: the
    // upper bound of what a fused fc1 -> GELU -> fc2 kernel could gain by keeping the hidden tensor on the chip (profiles/r4/fp32_upper_bounds.txt)
    static const bool hid_exp = getenv("DSG_HID_EXP") != nullptr;
    if (hid_exp) g.ldc = 0;
    const bool hid_bf16 = bf16_tensor_ok(WT(h, p + ".mlp.fc2.weight"), Hd) && bf16_of(h, b.fc1_wf) != nullptr;
    g.c_bf16 = hid_bf16;
    P_GEMM_LP(g);
    g = GemmArgs();
    g.A = w->hid; g.lda = hid_exp ? 0 : Hd; g.K1 = Hd; g.K = Hd; g.M = M; g.N = C;
    g.a_bf16 = hid_bf16;
    g.W = WT(h, p + ".mlp.fc2.weight"); g.bias = WT(h, p + ".mlp.fc2.bias");
    g.res = w->x; g.ldres = C; g.C = w->x; g.ldc = C;
    attach_premod(h, w, g, next);
    const bool st = !g.stats_out && want_stats && fuse;
    if (st) g.stats_out = w->stats;   // plain row statistics of the output (EPI 1)
    P_GEMM_LP(g);
    return BlockOut{g.mod_aff != nullptr, st ? (C + 95) / 96 : 0};
}

// PositionalEmbedding + map_layer0/1 + all affine linears for `rows` noise labels (diffusesg.py:768-771, :238, :574)
void embed_rows(dsg_handle h, const float *c_noise, int rows, float *pe, float *emb0, float *emb, float *aff, hipStream_t s) {
    const int E = h->E;
    P_KERN(PK_ELEM, 0.0, launch_noise_pe(c_noise, pe, rows, E, s));
    GemmArgs g;
    g.A = pe; g.lda = E; g.K1 = E; g.K = E; g.M = rows; g.N = NOISE_EMB; g.act = ACT_SILU;
    g.W = WT(h, "map_layer0.weight"); g.bias = WT(h, "map_layer0.bias"); g.C = emb0; g.ldc = NOISE_EMB;
    P_GEMM(g);
    g.A = emb0; g.lda = NOISE_EMB; g.K1 = NOISE_EMB; g.K = NOISE_EMB;
    g.W = WT(h, "map_layer1.weight"); g.bias = WT(h, "map_layer1.bias"); g.C = emb;
    P_GEMM(g);
    g = GemmArgs();
    g.A = emb; g.lda = NOISE_EMB; g.K1 = NOISE_EMB; g.K = NOISE_EMB; g.M = rows; g.N = h->aff_n;
    g.W = h->aff_w; g.bias = h->aff_b; g.C = aff; g.ldc = h->aff_n;
    P_GEMM(g);
}

// input assembly + PatchEmbed (diffusesg.py:784-802, 562-577) -> w->x.  Returns true when the first block's modulate+SiLU rode
// along (fused kernel only).  fp32 path: only when that block is the fused C = 96 attention kernel (which then skips it);
// any_first_block: whatever the first block is (the bf16 block pipeline normalises the modulated tensor in its own row pass).
bool patch_embed_stage(dsg_handle h, Workspace *w, bool fp32_rule, hipStream_t s, bool *xn_done = nullptr) {
    const dsg_config &c = h->cfg;
    const int B = w->B, N = h->N, E = h->E, T0 = N * N;
    GemmArgs g;
    bool pe_done = false, pe_premod = false;
    if (h->opt_fused_pe && h->pe_wp) {
        ProfScope ps_(h, s, PK_FUSED, 2.0 * (double)B * T0 * E * h->Cin, "launch_fused_patch_embed96");
        // the first block's modulate+SiLU rides along when that block is the fused C = 96 attention kernel (which then skips it)
        const BlockPlan *b0 = h->down[0].empty() ? nullptr : &h->down[0][0];
        pe_premod = wants_premod(h, b0) && (fp32_rule ? (h->opt_fused_attn && b0->wqp) : true);
        pe_done = launch_fused_patch_embed96(w->in_adj, w->in_node, w->cur_sc_adj, w->cur_sc_node, w->cur_has_sc, w->flags, h->pe_wp,
                                             WT(h, "patch_embed.proj.bias"), WT(h, "patch_embed.norm.weight"),
                                             WT(h, "patch_embed.norm.bias"), w->aff, w->aff_ld, h->pe_aff_off, pe_premod ? b0->aff_off : -1,
                                             w->x, B, N, h->Ca, h->Cn, c.self_condition, h->Kp, s, (xn_done && pe_premod) ? w->xn : nullptr);
        pe_premod = pe_premod && pe_done;
        if (xn_done) *xn_done = pe_premod;
    }
    if (!pe_done) {
        P_KERN(PK_ELEM, 0.0, launch_assemble(w->in_adj, w->in_node, w->cur_sc_adj, w->cur_sc_node, w->cur_has_sc, w->flags, w->tok_in, B, N, h->Ca, h->Cn,
                        c.self_condition, h->Kp, s));
        g = GemmArgs();
        g.A = w->tok_in; g.lda = h->Kp; g.K1 = h->Kp; g.K = h->Kp; g.M = B * T0; g.N = E;
        g.W = h->pe_w; g.bias = WT(h, "patch_embed.proj.bias"); g.C = w->y; g.ldc = E;
        P_KERN(PK_GEMM, 2.0 * (double)g.M * (double)g.N * (double)h->Cin,   // padded K is not algorithmic work
               if (!launch_gemm(g, s)) plan_fail(h, "patch_embed.proj: GEMM not built (M=%d N=%d K=%d)", g.M, g.N, g.K));
        P_KERN(PK_ROW, 0.0, launch_ln_mod(w->y, WT(h, "patch_embed.norm.weight"), WT(h, "patch_embed.norm.bias"), w->aff, w->aff_ld, h->pe_aff_off,
                      w->x, B, T0, E, s));
    }
    return pe_premod;
}

// final norm + read_out + heads (diffusesg.py:758-761, :806-825): w->x -> (f_adj, f_node)
bool bx_on(dsg_handle h);
void readout_stage(dsg_handle h, Workspace *w, hipStream_t s) {
    const int B = w->B, N = h->N, E = h->E, T0 = N * N;
    GemmArgs g;
    const int M0 = B * T0;
    if (h->opt_fused_readout && h->ro_fap && h->taps.empty()) {
        // one pass over x: LN, folded read_out+fc1, GELU, fc2, masked adjacency store; pooled LN(x) for the node head
        // (the bf16 block pipeline runs the two products of the read-out on the bf16 matrix pipe as well: option bf16_readout)
        const float *fap_b = (bx_on(h) && h->opt_bf16_readout && h->ro_fapb) ? (const float *)bf16_of(h, h->ro_fapb) : nullptr;
        const float *f2p_b = fap_b ? (const float *)bf16_of(h, h->ro_f2pb) : nullptr;
        const bool ro_bf = fap_b && f2p_b;
        P_KERN(PK_FUSED, 2.0 * (double)M0 * E * (E + 32.0),
               launch_fused_readout96(w->x, WT(h, "norm.weight"), WT(h, "norm.bias"), ro_bf ? fap_b : h->ro_fap, h->ro_fa, ro_bf ? f2p_b : h->ro_f2p,
                                      WT(h, "readout_adj_mlp.fc2.bias"), w->flags, w->f_adj, w->pool_part, w->pool_ext, B, N, h->Ca, s, ro_bf));
        g = GemmArgs();
        g.A = w->pool_ext; g.lda = 128; g.K1 = 128; g.K = 128; g.M = B * N; g.N = E; g.act = ACT_GELU;
        g.W = h->ro_gext; g.bias = WT(h, "readout_node_mlp.fc1.bias"); g.C = w->hn; g.ldc = E;
        P_GEMM(g);
        P_KERN(PK_ROW, 0.0, launch_head_node(w->hn, WT(h, "readout_node_mlp.fc2.weight"), WT(h, "readout_node_mlp.fc2.bias"), w->flags,
                                             w->f_node, B, N, E, h->Cn, s));
        return;
    }
    // faithful chain (diffusesg.py:758-761): final norm, three 1x1 convs
    P_KERN(PK_ROW, 0.0, launch_ln_stats(w->x, w->stats, M0, E, s));
    g = GemmArgs();
    g.A = w->x; g.lda = E; g.K1 = E; g.K = E; g.M = M0; g.N = E;
    g.ln_stats = w->stats;   // final norm's gamma/beta folded into ro0_wf / ro0_bf
    g.W = h->ro0_wf; g.bias = h->ro0_bf; g.C = w->y; g.ldc = E;
    P_GEMM(g);
    g = GemmArgs();
    g.A = w->y; g.lda = E; g.K1 = E; g.K = E; g.M = M0; g.N = E;
    g.W = WT(h, "read_out.1.weight"); g.bias = WT(h, "read_out.1.bias"); g.C = w->att; g.ldc = E;
    P_GEMM(g);
    g.A = w->att; g.W = WT(h, "read_out.2.weight"); g.bias = WT(h, "read_out.2.bias"); g.C = w->y;
    P_GEMM(g);  // y = shared_rep, token-major
    tap(h, "read_out", w->y, (size_t)M0 * E, s);
    // adjacency head (diffusesg.py:806-809, :825)
    g.A = w->y; g.W = WT(h, "readout_adj_mlp.fc1.weight"); g.bias = WT(h, "readout_adj_mlp.fc1.bias"); g.act = ACT_GELU;
    g.C = w->att;
    P_GEMM(g);
    P_KERN(PK_ROW, 0.0, launch_head_adj(w->att, WT(h, "readout_adj_mlp.fc2.weight"), WT(h, "readout_adj_mlp.fc2.bias"), w->flags, w->f_adj, B, N, E,
                    h->Ca, s));
    // node head (diffusesg.py:812-822)
    P_KERN(PK_ELEM, 0.0, launch_pool(w->y, w->flags, w->pool, B, N, E, s));
    g = GemmArgs();
    g.A = w->pool; g.lda = E; g.K1 = E; g.K = E; g.M = B * N; g.N = E; g.act = ACT_GELU;
    g.W = WT(h, "readout_node_mlp.fc1.weight"); g.bias = WT(h, "readout_node_mlp.fc1.bias"); g.C = w->hn; g.ldc = E;
    P_GEMM(g);
    P_KERN(PK_ROW, 0.0, launch_head_node(w->hn, WT(h, "readout_node_mlp.fc2.weight"), WT(h, "readout_node_mlp.fc2.bias"), w->flags, w->f_node, B, N,
                     E, h->Cn, s));
}

// ================= the bf16 block pipeline ("gemm_bf16" mode; kernels_bx.hip) =================
// Every tensor a GEMM reads is bf16 in HBM; the residual stream x stays fp32.  State of a level between two launches:
//   BX_RAW   x holds the un-modulated activation
//   BX_MOD   x holds silu(shift + x (1 + scale)) of the block about to run (its shortcut), w->xn is not valid yet
//   BX_READY x as BX_MOD and w->xn = LayerNorm-1 of it without affine (gamma / beta are folded into qkv_wf / qkv_bf)
enum BxState { BX_RAW = 0, BX_MOD = 1, BX_READY = 2 };
// (the pipeline's kernels are built and tested for embed_dim = 96 -- every configuration of the reference; any other width keeps
// round 2's kernels_lp.hip path instead of meeting an uncovered shape half-way through a forward, and get_option reports that)
bool bx_on(dsg_handle h) { return h->opt_gemm_bf16 && !h->opt_gemm_split && h->opt_bf16_pipe && h->E == 96; }

#define P_BX(g, tag)                                                                                                           \
    do {                                                                                                                       \
        char tg_[96];                                                                                                          \
        if (h->prof_on) snprintf(tg_, sizeof(tg_), "gemm_bx %s M=%d N=%d K=%d ln=%d", tag, (g).M, (g).N, (g).K, (g).ln_out);   \
        ProfScope ps_(h, s, PK_GEMM, 2.0 * (double)(g).M * (double)(g).N * (double)(g).K, tg_);                               \
        if (!launch_gemm_bx((g), s)) plan_fail(h, "bf16 pipeline: gemm_bx shape not covered (%s M=%d N=%d K=%d K1=%d ln=%d)", tag, (g).M, (g).N, (g).K, (g).K1, (g).ln_out); \
    } while (0)

bool bx_full_row(int C) { return C == 96 || C == 192 || C == 384; }   // widths a single GEMM tile spans: LayerNorm in the epilogue

// the "produce the next consumer's input" part of a GEMM that writes a level's activation x [M, C]: the next block's modulate+SiLU
// and its LayerNorm-1 as bf16 (full-row widths), or the plain bf16 copy (PatchBreakup reads the un-normalised tensor).  Returns the
// state x / xn are in after the launch plus whatever row pass finish_bx still has to run.
BxState attach_bx_out(dsg_handle h, Workspace *w, BxGemm &g, const BlockPlan *next, bool want_copy) {
    const bool fuse = h->taps.empty();   // taps want the un-modulated block outputs
    if (next && fuse) {
        g.mod_aff = w->aff; g.mod_ld = w->aff_ld; g.mod_off = next->aff_off; g.mod_T = next->res * next->res;
        if (bx_full_row(g.N)) { g.ln_out = 1; g.Cb = w->xn; g.ldcb = g.N; return BX_READY; }
        return BX_MOD;
    }
    if (!next && want_copy) { g.Cb = w->xn; g.ldcb = g.N; }
    return BX_RAW;
}

// One Swin block (diffusesg.py:232-277) on x [B*T, C] in place; `st`: what the producer left (above).  next / want_copy as run_block.
BxState run_block_bx(dsg_handle h, Workspace *w, const BlockPlan &b, BxState st, const BlockPlan *next, bool want_copy, hipStream_t s) {
    const int B = w->B, T = b.res * b.res, C = b.C, M = B * T, Hd = h->cfg.mlp_ratio * C;
    const std::string &p = b.prefix;
    // x <- silu(shift + x (1 + scale)) (also the shortcut) and LayerNorm-1 -> xn, whatever the producer has not done
    if (st == BX_RAW) P_KERN(PK_ROW, 0.0, launch_ln_bx(w->x, w->aff, w->aff_ld, b.aff_off, w->xn, B, T, C, true, s));
    else if (st == BX_MOD) P_KERN(PK_ROW, 0.0, launch_ln_bx(w->x, nullptr, 0, 0, w->xn, B, T, C, true, s));
    BxGemm g;
    WinGeom wg{b.res, b.ws, b.shift, b.heads, C};
    bool fused_qa = false;
    if (h->opt_bf16_qkv_attn && h->taps.empty()) {   // (the qkv tap wants the tensor)
        BxQkvAttn qa;
        qa.xn = w->xn; qa.W = bf16_of(h, b.qkv_wf); qa.bias = b.qkv_bf; qa.biasP = b.biasP; qa.out = w->att; qa.B = B; qa.g = wg;
        { auto it = h->w_qimg.find(b.qkv_wf); qa.Wimg = it == h->w_qimg.end() ? nullptr : it->second; }
        qa.biasF = b.biasF;
        // 10 x 10 windows: 1 the wave-per-unit kernel, 2 the block-per-head kernel (its A/B), 3 wave-per-unit only where a block lies in one window (heads % 4 == 0)
        qa.variant = (h->opt_bf16_qkv_attn == 2 || (h->opt_bf16_qkv_attn == 3 && b.heads % 4 != 0)) ? 1 : 0;
        ProfScope ps_(h, s, PK_ATTN, 2.0 * (double)M * 3.0 * C * C + 4.0 * (double)M * (double)(b.ws * b.ws) * (double)C, "qkv_attn_bx");
        fused_qa = launch_qkv_attn_bx(qa, s);
    }
    if (!fused_qa) {
        g.A = w->xn; g.lda = C; g.K = C; g.M = M; g.N = 3 * C;
        g.W = bf16_of(h, b.qkv_wf); g.bias = b.qkv_bf; g.Cb = w->qkv; g.ldcb = 3 * C;
        P_BX(g, "qkv");
        ProfScope ps_(h, s, PK_ATTN, 4.0 * (double)M * (double)(b.ws * b.ws) * (double)C, "attn_bx");
        if (!launch_attn_bx(w->qkv, b.biasT, w->att, B, wg, s) && !launch_window_attn(w->qkv, b.biasT, w->att, B, wg, s, true, true))
            plan_fail(h, "bf16 pipeline: %s: window attention not built (window %d, C=%d)", p.c_str(), b.ws, C);
    }
    // the fused MLP kernel can take the proj linear, the residual and LayerNorm-2 in front (x + proj(att) never goes to HBM)
    const bool mlp_fused = h->opt_bf16_mlp && h->cfg.mlp_ratio == 4 && (C == 96 || C == 192 || (C == 384 && h->opt_bf16_mlp != 3));
    const bool proj_in_mlp = mlp_fused && h->opt_bf16_proj_mlp && h->taps.empty() && (C == 96 || C == 192 || (C == 384 && (h->opt_bf16_mlp == 1 || h->opt_bf16_mlp == 4 || h->opt_bf16_mlp == 5 || h->opt_bf16_mlp == 6)));
    const bool full = bx_full_row(C);
    if (!proj_in_mlp) {
        g = BxGemm();
        g.A = w->att; g.lda = C; g.K = C; g.M = M; g.N = C;
        g.W = bf16_of(h, WT(h, p + ".attn.proj.weight")); g.bias = WT(h, p + ".attn.proj.bias");
        g.res = w->x; g.ldres = C; g.C = w->x; g.ldc = C;
        if (full) { g.ln_out = 1; g.Cb = w->xn; g.ldcb = C; }   // LayerNorm-2 of x + proj(...) (gamma / beta folded into fc1_wf / fc1_bf)
        P_BX(g, "proj");
        if (!full) P_KERN(PK_ROW, 0.0, launch_ln_bx(w->x, nullptr, 0, 0, w->xn, B, T, C, true, s));
    }
    // fc1 -> GELU -> fc2 -> + residual (-> the next block's modulate / LayerNorm-1) in one kernel, no hidden tensor (measured at COCO
    // B = 512, tools/bx_bench.py: 324 us against 302 + 325 for the GEMM pair at C = 96, 270 against 193 + 196 at C = 192, and at C = 384
    // the eight-wave kernel mlp384_bx 199 against 111 + 152; the four-wave kernel at C = 384 -- option value 2, tests only -- needs
    // 230 + 192 registers, runs one wave per SIMD and loses to the pair: 312 us).  C = 768 (VG's deepest level) keeps the GEMM pair.
    if (mlp_fused) {
        BxMlp m;
        // C = 384: 1 -> the LDS-DMA kernel on pre-arranged weight images (round 4), 4 -> round 3's eight-wave kernel, 2 -> the four-wave one
        m.wide8 = h->opt_bf16_mlp == 2 ? 0 : (h->opt_bf16_mlp == 4 ? 2 : (h->opt_bf16_mlp == 5 ? 3 : 1));
        m.img = C == 384 ? img_of(h, b.fc1_wf) : nullptr;
        m.img2 = C == 384 ? img2_of(h, b.fc1_wf) : nullptr;
        if (C == 96 && h->opt_bf16_mlp == 6) { auto it = h->w_img96.find(b.fc1_wf); m.img96 = it == h->w_img96.end() ? nullptr : it->second; }   // (6: level 0 on the LDS-resident kernel -- measured at par, not the default)
        if (proj_in_mlp) { m.att = w->att; m.Wp = bf16_of(h, WT(h, p + ".attn.proj.weight")); m.bp = WT(h, p + ".attn.proj.bias"); }
        m.xn = w->xn; m.x = w->x; m.W1 = bf16_of(h, b.fc1_wf); m.b1 = b.fc1_bf;
        m.W2 = bf16_of(h, WT(h, p + ".mlp.fc2.weight")); m.b2 = WT(h, p + ".mlp.fc2.bias"); m.M = M; m.C = C;
        BxState out = BX_RAW;
        if (next && h->taps.empty()) {
            m.mod_aff = w->aff; m.mod_ld = w->aff_ld; m.mod_off = next->aff_off; m.mod_T = next->res * next->res;
            m.xn_out = w->xn; m.out_mode = 1; out = BX_READY;
        } else if (!next && want_copy) { m.xn_out = w->xn; m.out_mode = 2; }
        ProfScope ps_(h, s, PK_FUSED, 4.0 * (double)M * (double)C * (double)Hd + (proj_in_mlp ? 2.0 * (double)M * C * C : 0.0), proj_in_mlp ? "proj_mlp_bx" : "mlp_bx");
        if (!launch_mlp_bx(m, s)) plan_fail(h, "bf16 pipeline: %s: fused MLP shape not covered (M=%d C=%d proj=%d)", p.c_str(), M, C, (int)proj_in_mlp);
        return out;
    }
    g = BxGemm();
    g.A = w->xn; g.lda = C; g.K = C; g.M = M; g.N = Hd;
    g.W = bf16_of(h, b.fc1_wf); g.bias = b.fc1_bf; g.act = ACT_GELU; g.Cb = w->hid; g.ldcb = Hd;
    P_BX(g, "fc1");
    g = BxGemm();
    g.A = w->hid; g.lda = Hd; g.K = Hd; g.M = M; g.N = C;
    g.W = bf16_of(h, WT(h, p + ".mlp.fc2.weight")); g.bias = WT(h, p + ".mlp.fc2.bias");
    g.res = w->x; g.ldres = C; g.C = w->x; g.ldc = C;
    const BxState out = attach_bx_out(h, w, g, next, want_copy);
    P_BX(g, "fc2");
    return out;
}

// DiffuseSG.forward in the bf16 block pipeline; same structure (and the same fp32 noise embedding, PatchEmbed, read-out and heads)
// as forward_fixed below
void forward_fixed_bx(dsg_handle h, Workspace *w, hipStream_t s) {
    const int B = w->B, N = h->N, E = h->E, L = h->L, T0 = N * N;
    char name[64];
    BxGemm g;
    w->aff_ld = w->uniform ? 0 : h->aff_n;
    if (!w->uniform) embed_rows(h, w->c_noise, B, w->pe, w->emb0, w->emb, w->aff, s);
    bool xn_done = false;   // the fused PatchEmbed kernel also leaves block 0's LayerNorm-1 input when it applies that block's modulate
    BxState st = patch_embed_stage(h, w, false, s, &xn_done) ? (xn_done ? BX_READY : BX_MOD) : BX_RAW;
    tap(h, "patch_embed", w->x, (size_t)B * T0 * E, s);
    // encoder (diffusesg.py:745-748); skips are stored as bf16 (their only reader is PatchBreakup's pre_linear)
    for (int l = 0; l < L; l++) {
        const int C = E << l, res = N >> l, T = res * res;
        for (size_t j = 0; j < h->down[l].size(); j++) {
            const BlockPlan *next = j + 1 < h->down[l].size() ? &h->down[l][j + 1] : (l == L - 1 && !h->up[0].empty() ? &h->up[0][0] : nullptr);
            st = run_block_bx(h, w, h->down[l][j], st, next, false, s);   // (PatchMerging gathers from the fp32 x: no bf16 copy needed)
            snprintf(name, sizeof(name), "down%d.block%d", l, (int)j);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
        if (l < L - 1) {
            const std::string p = "down_layers." + std::to_string(l) + ".downsample";
            // 2x2 gather + LayerNorm(4C) -> bf16 [B*T/4, 4C] (in w->hid), then the reduction; its epilogue stores the skip copy and
            // prepares the next level's first block
            P_KERN(PK_ROW, 0.0, launch_merge_ln(w->x, WT(h, p + ".norm.weight"), WT(h, p + ".norm.bias"), w->hid, B, res, C, s, true));
            g = BxGemm();
            g.A = w->hid; g.lda = 4 * C; g.K = 4 * C; g.M = B * T / 4; g.N = 2 * C;
            g.W = bf16_of(h, WT(h, p + ".reduction.weight")); g.C = w->x; g.ldc = 2 * C;
            g.C2b = w->skips[l]; g.ldc2b = 2 * C;
            st = attach_bx_out(h, w, g, h->down[l + 1].empty() ? nullptr : &h->down[l + 1][0], false);
            P_BX(g, "merge");
        }
        snprintf(name, sizeof(name), "down%d", l);
        tap(h, name, w->x, l < L - 1 ? (size_t)B * (T / 4) * 2 * C : (size_t)B * T * C, s);
    }
    // decoder (diffusesg.py:751-756)
    for (int i = 0; i < L; i++) {
        const int l = L - 1 - i, C = E << l, res = N >> l, T = res * res;
        if (i > 0) {
            const std::string p = "up_layers." + std::to_string(i) + ".upsample";
            const int D = 4 * C, Tc = T / 4;
            // pre_linear on cat([x, skip]): x's bf16 copy was left in w->xn by the previous level's last block, the skip is bf16
            g = BxGemm();
            g.A = w->xn; g.lda = D / 2; g.K1 = D / 2; g.A2 = w->skips[l]; g.lda2 = D / 2; g.K = D; g.M = B * Tc; g.N = D;
            g.W = bf16_of(h, WT(h, p + ".pre_linear.weight")); g.C = w->hid; g.ldc = D;
            P_BX(g, "pre_linear");
            P_KERN(PK_ROW, 0.0, launch_breakup_ln(w->hid, WT(h, p + ".norm.weight"), WT(h, p + ".norm.bias"), WT(h, p + ".post_norm.weight"),
                              WT(h, p + ".post_norm.bias"), w->y, B, res / 2, D, s, true));
            g = BxGemm();
            g.A = w->y; g.lda = C; g.K = C; g.M = B * T; g.N = C;
            g.W = bf16_of(h, WT(h, p + ".post_linear.weight")); g.C = w->x; g.ldc = C;
            st = attach_bx_out(h, w, g, h->up[i].empty() ? nullptr : &h->up[i][0], false);
            P_BX(g, "post_linear");
            snprintf(name, sizeof(name), "up%d.upsample", i);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
        for (size_t j = 0; j < h->up[i].size(); j++) {
            const BlockPlan *next = j + 1 < h->up[i].size() ? &h->up[i][j + 1] : nullptr;   // then PatchBreakup / the read-out
            st = run_block_bx(h, w, h->up[i][j], st, next, /*want_copy=*/i + 1 < L, s);
            snprintf(name, sizeof(name), "up%d.block%d", i, (int)j);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
        if (h->up[i].empty() && i + 1 < L)   // a level without blocks: PatchBreakup still needs x's bf16 copy
            P_KERN(PK_ROW, 0.0, launch_ln_bx(w->x, nullptr, 0, 0, w->xn, B, T, C, false, s));
    }
    readout_stage(h, w, s);
}

// DiffuseSG.forward on the workspace's fixed buffers: (in_adj,in_node,sc_*,flags,c_noise) -> (f_adj,f_node)
void forward_fixed(dsg_handle h, Workspace *w, hipStream_t s) {
    if (bx_on(h)) { forward_fixed_bx(h, w, s); return; }
    // the fused PatchMerging writes the coarser level into the other activation buffer and swaps the two names; put them back
    // on every exit so that each forward (and each captured graph) starts from the same assignment
    struct SwapGuard { Workspace *w; float *x, *y; ~SwapGuard() { w->x = x; w->y = y; } } swap_guard{w, w->x, w->y};
    const int B = w->B, N = h->N, E = h->E, L = h->L, T0 = N * N;
    char name[64];
    GemmArgs g;
    w->aff_ld = w->uniform ? 0 : h->aff_n;
    if (!w->uniform) {
        // noise embedding (diffusesg.py:768-771) and every block's (scale,shift) in one GEMM
        embed_rows(h, w->c_noise, B, w->pe, w->emb0, w->emb, w->aff, s);
    }
    const bool pe_premod = patch_embed_stage(h, w, true, s);
    tap(h, "patch_embed", w->x, (size_t)B * T0 * E, s);
    // encoder (diffusesg.py:745-748)
    bool premod = pe_premod;   // is w->x already modulated for the next block (generic blocks: with LN1 partials in w->stats)?
    int merge_parts = 0;       // > 0: w->stats holds partials of the level's final output rows (for the fused PatchMerging)
    for (int l = 0; l < L; l++) {
        const int C = E << l, res = N >> l, T = res * res;
        for (size_t j = 0; j < h->down[l].size(); j++) {
            // the consumer of this block's output: the next block of the level; after the deepest level the first decoder
            // block (no upsample there); otherwise PatchMerging, which takes the un-modulated tensor
            const BlockPlan *next = j + 1 < h->down[l].size() ? &h->down[l][j + 1] : (l == L - 1 && !h->up[0].empty() ? &h->up[0][0] : nullptr);
            // the level's last block leaves row statistics for the fused PatchMerging
            // (below ~8k merged rows the reduction GEMM is a single partial wave of tiles and the gather + LayerNorm FMA in its
            // long K loop costs more than the small merge_ln launch it replaces: measured 103 vs 81 + 13 us at M = 4096, K = 1536)
            const bool for_merge = !next && l < L - 1 && h->opt_fused_merge && rowstats_on(h) && !h->opt_gemm_bf16 && C % 32 == 0 &&
                                   (B * T / 4 >= 8192 || h->opt_fused_merge_small);
            const BlockOut bo = run_block(h, w, h->down[l][j], premod, next, for_merge, s);
            premod = bo.premod; merge_parts = bo.stats_parts;
            snprintf(name, sizeof(name), "down%d.block%d", l, (int)j);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
        if (l < L - 1) {
            const std::string p = "down_layers." + std::to_string(l) + ".downsample";
            g = GemmArgs();
            g.K1 = 4 * C; g.K = 4 * C; g.M = B * T / 4; g.N = 2 * C;
            if (merge_parts > 0) {
                // 2x2 gather, LayerNorm(4C) (statistics from the four source rows' partials, gamma/beta folded into the weight)
                // and the reduction in one GEMM: the merged [B*T/4, 4C] tensor is never written
                g.A = w->x; g.lda = C; g.a4_res = res; g.ln_part = w->stats; g.ln_nparts = merge_parts;
                g.W = h->merge_wf[l]; g.bias = h->merge_bf[l];
                g.C = w->y; g.ldc = 2 * C;   // cannot overwrite x while other tiles still gather from it: write y, then swap
            } else {
                P_KERN(PK_ROW, 0.0, launch_merge_ln(w->x, WT(h, p + ".norm.weight"), WT(h, p + ".norm.bias"), w->y, B, res, C, s));
                g.A = w->y; g.lda = 4 * C;
                g.W = WT(h, p + ".reduction.weight"); g.C = w->x; g.ldc = 2 * C;
            }
            g.C2 = w->skips[l]; g.ldc2 = 2 * C;
            attach_premod(h, w, g, h->down[l + 1].empty() ? nullptr : &h->down[l + 1][0]);   // the skip copy (C2) stays un-modulated
            P_GEMM_LP(g);
            premod = g.mod_aff != nullptr;
            if (merge_parts > 0) std::swap(w->x, w->y);   // the new level's activation lives in the other buffer from here on
            merge_parts = 0;
        }
        snprintf(name, sizeof(name), "down%d", l);
        tap(h, name, w->x, l < L - 1 ? (size_t)B * (T / 4) * 2 * C : (size_t)B * T * C, s);
        // the deepest level's skip is popped and discarded by the first up layer (diffusesg.py:754-755)
    }
    // decoder (diffusesg.py:751-756)
    for (int i = 0; i < L; i++) {
        const int l = L - 1 - i, C = E << l, res = N >> l, T = res * res;
        if (i > 0) {
            const std::string p = "up_layers." + std::to_string(i) + ".upsample";
            const int D = 4 * C, Tc = T / 4;  // coarse tokens, concatenated width
            g = GemmArgs();                   // pre_linear on cat([x, skip]) without materialising the concat
            g.A = w->x; g.lda = D / 2; g.K1 = D / 2; g.A2 = w->skips[l]; g.lda2 = D / 2; g.K = D; g.M = B * Tc; g.N = D;
            g.W = WT(h, p + ".pre_linear.weight"); g.C = w->hid; g.ldc = D;
            P_GEMM_LP(g);
            P_KERN(PK_ROW, 0.0, launch_breakup_ln(w->hid, WT(h, p + ".norm.weight"), WT(h, p + ".norm.bias"), WT(h, p + ".post_norm.weight"),
                              WT(h, p + ".post_norm.bias"), w->y, B, res / 2, D, s));
            g = GemmArgs();
            g.A = w->y; g.lda = C; g.K1 = C; g.K = C; g.M = B * T; g.N = C;
            g.W = WT(h, p + ".post_linear.weight"); g.C = w->x; g.ldc = C;
            attach_premod(h, w, g, h->up[i].empty() ? nullptr : &h->up[i][0]);
            P_GEMM_LP(g);
            premod = g.mod_aff != nullptr;
            snprintf(name, sizeof(name), "up%d.upsample", i);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
        for (size_t j = 0; j < h->up[i].size(); j++) {
            const BlockPlan *next = j + 1 < h->up[i].size() ? &h->up[i][j + 1] : nullptr;   // then PatchBreakup / the read-out
            premod = run_block(h, w, h->up[i][j], premod, next, false, s).premod;
            snprintf(name, sizeof(name), "up%d.block%d", i, (int)j);
            tap(h, name, w->x, (size_t)B * T * C, s);
        }
    }
    readout_stage(h, w, s);
}

// DSG_ERR_INVALID (and the message) if a launcher declined its arguments since the last check
int plan_status(dsg_handle h, Workspace *w) {
    if (h->plan_err.empty()) return 0;
    const std::string m = h->plan_err;
    h->plan_err.clear();
    return fail(h, DSG_ERR_INVALID, "unsupported configuration at batch %d: %s", w->B, m.c_str());
}

// Plan-time shape coverage: walk the whole forward of this workspace's batch size in a dry run -- every launcher checks its arguments
// and picks its instantiation, nothing is enqueued (kernels.h: g_dry_run) -- in both noise-label forms (per-sample rows: dsg_denoise /
// dsg_precond; the sampler's batch-uniform row) and with / without a self-conditioning input.  Runs when a batch size is first used and
// again after anything that changes the kernel selection (dsg_set_option, dsg_finalize_weights, debug taps): an uncovered shape is
// reported as DSG_ERR_INVALID by the entry point before any work is enqueued or captured -- never by terminating the process.
int validate_plan(dsg_handle h, Workspace *w) {
    if (w->plan_gen == h->plan_gen) return 0;
    const bool uni = w->uniform, prof_on = h->prof_on, prof_stamps = h->prof_stamps;
    const float *sca = w->cur_sc_adj, *scn = w->cur_sc_node; const int *hs = w->cur_has_sc;
    h->prof_on = false; h->prof_stamps = false;
    h->plan_err.clear();
    g_dry_run = true;
    for (int v = 0; v < 4; v++) {
        w->uniform = (v & 1) != 0;
        w->cur_sc_adj = (v & 2) ? w->sc_adj : nullptr; w->cur_sc_node = (v & 2) ? w->sc_node : nullptr; w->cur_has_sc = (v & 1) ? nullptr : w->has_sc;
        forward_fixed(h, w, nullptr);
    }
    g_dry_run = false;
    w->uniform = uni; h->prof_on = prof_on; h->prof_stamps = prof_stamps;
    w->cur_sc_adj = sca; w->cur_sc_node = scn; w->cur_has_sc = hs;
    if (int rc = plan_status(h, w)) return rc;
    w->plan_gen = h->plan_gen;
    return 0;
}

// run forward_fixed either eagerly or by replaying a captured graph
int run_forward(dsg_handle h, Workspace *w, bool use_graph, hipStream_t s) {
    // (plan_err: validate_plan has already walked this forward in a dry run, so these checks are the second line of defence)
    if (!use_graph || !h->taps.empty()) { forward_fixed(h, w, s); return plan_status(h, w); }
    hipGraphExec_t &exec = w->uniform ? w->graph_uniform : w->graph;
    if (!exec) {
        if (!w->cap_stream) HIP_TRY(h, hipStreamCreateWithFlags(&w->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph;
        HIP_TRY(h, hipStreamBeginCapture(w->cap_stream, hipStreamCaptureModeThreadLocal));
        forward_fixed(h, w, w->cap_stream);
        HIP_TRY(h, hipStreamEndCapture(w->cap_stream, &graph));
        if (int rc = plan_status(h, w)) { (void)hipGraphDestroy(graph); return rc; }
        hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { exec = nullptr; return fail(h, DSG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipGraphLaunch(exec, s));
    h->last_stats.graph_replays++;
    return 0;
}

Dims dims_of(dsg_handle h, int B) { return Dims{B, h->N, h->Ca, h->Cn}; }

int check_ready(dsg_handle h, int B) {
    if (!h) return DSG_ERR_INVALID;
    if (!h->finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (B < 1) return fail(h, DSG_ERR_INVALID, "batch must be >= 1");
    return 0;
}

// stage caller tensors into the fixed forward inputs
int stage_inputs(dsg_handle h, Workspace *w, const float *adj, const float *node, const uint8_t *flags, const float *sc_adj,
                 const float *sc_node, hipStream_t s) {
    const size_t sa = sizeof(float) * (size_t)w->B * h->Ca * h->N * h->N, sn = sizeof(float) * (size_t)w->B * h->N * h->Cn;
    if (adj) HIP_TRY(h, hipMemcpyAsync(w->in_adj, adj, sa, hipMemcpyDeviceToDevice, s));
    if (node) HIP_TRY(h, hipMemcpyAsync(w->in_node, node, sn, hipMemcpyDeviceToDevice, s));
    if (flags) HIP_TRY(h, hipMemcpyAsync(w->flags, flags, (size_t)w->B * h->N, hipMemcpyDeviceToDevice, s));
    const int has = (sc_adj && sc_node && h->cfg.self_condition) ? 1 : 0;
    if (has) {
        if (sc_adj != w->sc_adj) HIP_TRY(h, hipMemcpyAsync(w->sc_adj, sc_adj, sa, hipMemcpyDeviceToDevice, s));
        if (sc_node != w->sc_node) HIP_TRY(h, hipMemcpyAsync(w->sc_node, sc_node, sn, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(h, hipMemsetAsync(w->has_sc, 0, 16, s));
    if (has) HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)w->has_sc, 1, 1, s));
    w->cur_sc_adj = w->sc_adj; w->cur_sc_node = w->sc_node; w->cur_has_sc = w->has_sc;
    return 0;
}

// NodeAdjPrecond.forward on workspace state: x (xh_adj/xh_node), sigmas (w->sig) -> D (dst), optionally
// mirrored into the fixed self-cond input of the next forward.  sc: current self-cond or null.
int precond_core(dsg_handle h, Workspace *w, CStatePtrs x, const float *sc_adj, const float *sc_node, bool coin, StatePtrs dst,
                 bool use_graph, hipStream_t s, int64_t *nfe, const float *aff_row = nullptr) {
    const Dims d = dims_of(h, w->B);
    w->uniform = aff_row != nullptr;
    if (aff_row) HIP_TRY(h, hipMemcpyAsync(w->aff, aff_row, sizeof(float) * h->aff_n, hipMemcpyDeviceToDevice, s));
    launch_precond_in(x, w->sig, StatePtrs{w->in_adj, w->in_node}, w->c_noise, d, s);
    if (int rc = stage_inputs(h, w, nullptr, nullptr, nullptr, sc_adj, sc_node, s)) return rc;
    if (h->cfg.self_condition && coin) {  // precond.py:90-98
        if (int rc = run_forward(h, w, use_graph, s)) return rc;
        (*nfe)++;
        // the D of the extra pass becomes the self-cond input (written straight into the fixed sc buffers)
        launch_precond_out(x, CStatePtrs{w->f_adj, w->f_node}, w->sig, w->flags, StatePtrs{w->sc_adj, w->sc_node},
                           StatePtrs{nullptr, nullptr}, d, s);
        HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)w->has_sc, 1, 1, s));
    }
    if (int rc = run_forward(h, w, use_graph, s)) return rc;
    (*nfe)++;
    launch_precond_out(x, CStatePtrs{w->f_adj, w->f_node}, w->sig, w->flags, dst, StatePtrs{nullptr, nullptr}, d, s);
    return 0;
}

// ---- one step of the reverse loop (edm.py:350-427) as a static launch sequence ------------------------------------------
// Everything that changes from step to step is read on the device from StepRow[ctl->step] / the step's (scale,shift) row, so
// the sequence below depends only on the StepPlan and can be captured once per plan and replayed for every step it fits.
struct StepPlan {
    int sc_slot;          // d_* buffer holding the self-conditioning input of stage 1, -1 = None
    int s1, s2;           // d_* buffers receiving the stage-1 / stage-2 denoised outputs
    bool euler;           // Euler update (solver 'euler', or the last step: edm.py:394-396), else Heun
    bool coin1, coin2;    // outcome of the self-conditioning coin of each preconditioned call (precond.py:90)
    int key() const { return (sc_slot + 1) | (s1 << 2) | (s2 << 4) | ((int)euler << 6) | ((int)coin1 << 7) | ((int)coin2 << 8); }
};

// NodeAdjPrecond.forward at sigma[step] on workspace state x -> dst; forwards run inline on `s` (capturable)
// fwd_graph: replay the captured network forward (round-1 scheme: only the forward is a graph, the loop is host-enqueued)
int precond_tab(dsg_handle h, Workspace *w, CStatePtrs x, const float *sc_adj, const float *sc_node, bool coin, StatePtrs dst,
                hipStream_t s, int *nfe, bool fwd_graph) {
    const Dims d = dims_of(h, w->B);
    w->uniform = true;
    launch_precond_in_tab(x, h->tab_step, w->ctl, StatePtrs{w->in_adj, w->in_node}, d, s);
    const bool has = sc_adj && sc_node && h->cfg.self_condition;
    if (fwd_graph) {   // the forward-only graph reads the fixed self-cond buffers and the device flag
        if (int rc = stage_inputs(h, w, nullptr, nullptr, nullptr, sc_adj, sc_node, s)) return rc;
    } else {           // kernels only: the forward reads the denoised buffer in place (or nothing)
        w->cur_sc_adj = has ? sc_adj : nullptr; w->cur_sc_node = has ? sc_node : nullptr; w->cur_has_sc = nullptr;
    }
    if (h->cfg.self_condition && coin) {  // precond.py:90-98: the D of the extra pass becomes the self-cond input
        if (int rc = run_forward(h, w, fwd_graph, s)) return rc;
        (*nfe)++;
        launch_precond_out_tab(x, CStatePtrs{w->f_adj, w->f_node}, h->tab_step, w->ctl, w->flags, StatePtrs{w->sc_adj, w->sc_node}, d, s);
        if (fwd_graph) HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)w->has_sc, 1, 1, s));
        else { w->cur_sc_adj = w->sc_adj; w->cur_sc_node = w->sc_node; }
    }
    if (int rc = run_forward(h, w, fwd_graph, s)) return rc;
    (*nfe)++;
    launch_precond_out_tab(x, CStatePtrs{w->f_adj, w->f_node}, h->tab_step, w->ctl, w->flags, dst, d, s);
    return 0;
}

int enqueue_step(dsg_handle h, Workspace *w, const StepPlan &p, const float *gt_adj, const float *gt_node, hipStream_t s, int *nfe,
                 bool fwd_graph = false) {
    const Dims d = dims_of(h, w->B);
    const CStatePtrs xh{w->xh_adj, w->xh_node};
    launch_churn_tab(CStatePtrs{w->x_adj, w->x_node}, h->tab_step, w->ctl, w->flags, StatePtrs{w->xh_adj, w->xh_node}, d, s);   // edm.py:355-366
    CStatePtrs D1{gt_adj, gt_node};   // sanity-check mode (edm.py:372-377): the denoiser is bypassed
    if (!gt_adj) {
        launch_step_row(h->tab_aff, h->aff_n, w->ctl, w->aff, s);   // this step's (scale,shift) row for every block
        if (int rc = precond_tab(h, w, xh, p.sc_slot >= 0 ? w->d_adj[p.sc_slot] : nullptr, p.sc_slot >= 0 ? w->d_node[p.sc_slot] : nullptr,
                                 p.coin1, StatePtrs{w->d_adj[p.s1], w->d_node[p.s1]}, s, nfe, fwd_graph)) return rc;
        D1 = CStatePtrs{w->d_adj[p.s1], w->d_node[p.s1]};
    }
    if (p.euler) {
        launch_euler_tab(xh, D1, h->tab_step, w->ctl, w->flags, StatePtrs{w->x_adj, w->x_node}, d, s);
    } else {
        CStatePtrs D2 = D1;
        if (!gt_adj) {   // stage 2 re-evaluates at (x_hat, sigma(t_hat)) with self-cond = D1 (edm.py:400-405)
            const bool sc = h->cfg.self_condition;
            if (int rc = precond_tab(h, w, xh, sc ? w->d_adj[p.s1] : nullptr, sc ? w->d_node[p.s1] : nullptr, p.coin2,
                                     StatePtrs{w->d_adj[p.s2], w->d_node[p.s2]}, s, nfe, fwd_graph)) return rc;
            D2 = CStatePtrs{w->d_adj[p.s2], w->d_node[p.s2]};
        }
        launch_heun_tab(xh, D1, D2, h->tab_step, w->ctl, w->flags, StatePtrs{w->x_adj, w->x_node}, d, s);
    }
    launch_step_advance(w->ctl, s);
    return 0;
}

// capture + instantiate the step body of plan p (no-op if it exists).  Called for every plan of a run BEFORE the run's first
// launch: instantiating a new graph while earlier graph launches are still in flight on the caller's stream gave wrong
// trajectories on ROCm 7.2 (tools/loop_graph_ab.py history in DESIGN.md), so nothing is captured mid-flight.
int ensure_step_graph(dsg_handle h, Workspace *w, const StepPlan &p) {
    if (w->step_graphs.count(p.key())) return 0;
    if (!w->cap_stream) HIP_TRY(h, hipStreamCreateWithFlags(&w->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph;
    int n = 0;
    HIP_TRY(h, hipStreamBeginCapture(w->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step(h, w, p, nullptr, nullptr, w->cap_stream, &n);
    const hipError_t ec = hipStreamEndCapture(w->cap_stream, &graph);
    if (rc) { if (ec == hipSuccess) (void)hipGraphDestroy(graph); return rc; }
    if (ec != hipSuccess) return fail(h, DSG_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ec));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, DSG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    w->step_graphs.emplace(p.key(), std::make_pair(exec, n));
    if (getenv("DSG_GRAPH_VERBOSE")) fprintf(stderr, "[dsg-graph] captured step body key=%d (sc %d s1 %d s2 %d euler %d coins %d%d): %d forwards\n",
                                             p.key(), p.sc_slot, p.s1, p.s2, (int)p.euler, (int)p.coin1, (int)p.coin2, n);
    return 0;
}

// replay the step body of plan p on the caller's stream
int replay_step(dsg_handle h, Workspace *w, const StepPlan &p, hipStream_t s, int *nfe) {
    auto it = w->step_graphs.find(p.key());
    if (it == w->step_graphs.end()) return fail(h, DSG_ERR_STATE, "step body %d was not captured", p.key());
    HIP_TRY(h, hipGraphLaunch(it->second.first, s));
    *nfe += it->second.second;
    h->last_stats.graph_replays += it->second.second;
    return 0;
}

}  // namespace

extern "C" {

size_t dsg_workspace_bytes(dsg_handle h, int32_t B) {
    if (!h || !h->finalized || B < 1) return 0;
    return sizeof(float) * per_sample_floats(h) * (size_t)B + (size_t)B * h->N + 16;
}

int dsg_set_option(dsg_handle h, const char *name, int32_t value) {
    if (!h || !name) return DSG_ERR_INVALID;
    const std::string n(name);
    // the option set as it stands: restored if the new selection meets a shape no kernel covers (below)
    auto opts = [h]() {
        return std::tie(h->opt_fused_attn, h->opt_fused_mlp, h->opt_fused_mlp_maxc, h->opt_fused_readout, h->opt_fused_pe, h->opt_fused_rowstats,
                        h->opt_fused_qkv_attn, h->opt_loop_graph, h->opt_bf16_act, h->opt_bf16_pipe, h->opt_bf16_mlp, h->opt_bf16_qkv_attn,
                        h->opt_bf16_proj_mlp, h->opt_bf16_readout, h->opt_fused_merge, h->opt_fused_merge_small, h->opt_gemm_bf16, h->opt_gemm_split);
    };
    const std::tuple<bool, bool, int, bool, bool, bool, bool, bool, int, bool, int, bool, bool, bool, bool, bool, bool, bool> saved = opts();
    if (n == "fused_attn") h->opt_fused_attn = value != 0;
    else if (n == "fused_mlp") h->opt_fused_mlp = value != 0;
    else if (n == "fused_mlp_maxc") h->opt_fused_mlp_maxc = value;
    else if (n == "fused_readout") h->opt_fused_readout = value != 0;
    else if (n == "fused_patch_embed") h->opt_fused_pe = value != 0;
    else if (n == "fused_rowstats") h->opt_fused_rowstats = value != 0;
    else if (n == "fused_qkv_attn") h->opt_fused_qkv_attn = value != 0;
    else if (n == "loop_graph") h->opt_loop_graph = value != 0;
    else if (n == "bf16_act") h->opt_bf16_act = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "bf16_pipe") h->opt_bf16_pipe = value != 0;
    else if (n == "bf16_mlp") h->opt_bf16_mlp = value < 0 ? 0 : (value > 6 ? 6 : value);
    else if (n == "bf16_qkv_attn") h->opt_bf16_qkv_attn = value < 0 ? 0 : (value > 3 ? 3 : value);
    else if (n == "bf16_proj_mlp") h->opt_bf16_proj_mlp = value != 0;
    else if (n == "bf16_readout") h->opt_bf16_readout = value != 0;
    else if (n == "fused_merge") { h->opt_fused_merge = value != 0; h->opt_fused_merge_small = value > 1; }   // 2: at every size
    else if (n == "gemm_bf16") {
        h->opt_gemm_bf16 = value != 0;
        if (h->opt_gemm_bf16 && h->finalized) if (int rc = ensure_bf16_weights(h)) return rc;
    }
    else if (n == "gemm_split") {
        h->opt_gemm_split = value != 0;
        if (h->opt_gemm_split && h->finalized) if (int rc = ensure_split_weights(h)) return rc;
    }
    else return fail(h, DSG_ERR_INVALID, "unknown option '%s'", name);
    drop_graphs(h);   // captured graphs bake the kernel selection
    // plan time: every batch size in use must still be covered by the kernels the new selection launches; if not, the option keeps
    // its old value and the caller gets DSG_ERR_INVALID with the layer / shape in dsg_last_error
    if (h->finalized)
        for (auto &kv : h->ws)
            if (int rc = validate_plan(h, kv.second.get())) {
                const std::string why = h->err;
                opts() = saved;
                drop_graphs(h);
                return fail(h, rc, "option '%s' = %d rejected: %s", name, value, why.c_str());
            }
    return DSG_OK;
}

int dsg_get_option(dsg_handle h, const char *name, int32_t *value) {
    if (!h || !name || !value) return DSG_ERR_INVALID;
    const std::string n(name);
    if (n == "fused_attn") *value = h->opt_fused_attn;
    else if (n == "fused_mlp") *value = h->opt_fused_mlp;
    else if (n == "fused_mlp_maxc") *value = h->opt_fused_mlp_maxc;
    else if (n == "fused_readout") *value = h->opt_fused_readout;
    else if (n == "fused_patch_embed") *value = h->opt_fused_pe;
    else if (n == "fused_rowstats") *value = h->opt_fused_rowstats;
    else if (n == "fused_qkv_attn") *value = h->opt_fused_qkv_attn;
    else if (n == "loop_graph") *value = h->opt_loop_graph;
    else if (n == "bf16_act") *value = (h->opt_gemm_bf16 && !h->opt_gemm_split && !bx_on(h)) ? h->opt_bf16_act : 0;   // acts whenever round 2's bf16 path is what runs (pipeline off, or a width it does not take)
    else if (n == "bf16_pipe") *value = bx_on(h);   // the bf16 block pipeline runs (bf16 mode only)
    else if (n == "bf16_mlp") *value = bx_on(h) ? h->opt_bf16_mlp : 0;
    else if (n == "bf16_qkv_attn") *value = bx_on(h) ? h->opt_bf16_qkv_attn : 0;
    else if (n == "bf16_proj_mlp") *value = (bx_on(h) && h->opt_bf16_mlp && h->opt_bf16_proj_mlp) ? 1 : 0;
    else if (n == "bf16_readout") *value = (bx_on(h) && h->opt_bf16_readout && h->opt_fused_readout) ? 1 : 0;
    else if (n == "fused_merge") *value = h->opt_fused_merge ? (h->opt_fused_merge_small ? 2 : 1) : 0;
    else if (n == "gemm_bf16") *value = h->opt_gemm_bf16 && !h->opt_gemm_split;   // "gemm_split" takes precedence
    else if (n == "gemm_split") *value = h->opt_gemm_split;
    else return fail(h, DSG_ERR_INVALID, "unknown option '%s'", name);
    return DSG_OK;
}

int dsg_gen_noise(dsg_handle h, int32_t B, const uint8_t *flags, uint64_t seed, uint32_t noise_stream, float *out_adj,
                  float *out_node, void *stream) {
    if (!h || B < 1) return fail(h, DSG_ERR_INVALID, "batch must be >= 1");
    if (!flags || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null tensor");
    launch_init(CStatePtrs{nullptr, nullptr}, 1.0f, seed, noise_stream, flags, StatePtrs{out_adj, out_node}, dims_of(h, B),
                (hipStream_t)stream);
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}

int dsg_debug_tap(dsg_handle h, const char *stage, float *dst, int64_t capacity) {
    if (!h || !stage || !dst) return DSG_ERR_INVALID;
    h->taps.push_back({stage, dst, capacity});
    h->plan_gen++;   // taps switch the row-kernel fusions off: the launch plan changes
    return DSG_OK;
}
void dsg_debug_clear_taps(dsg_handle h) { if (h) { h->taps.clear(); h->plan_gen++; } }

int dsg_denoise(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags, const float *noise_labels,
                const float *sc_adj, const float *sc_node, float *out_adj, float *out_node, void *stream) {
    if (int rc = check_ready(h, B)) return rc;
    if (!adj || !node || !flags || !noise_labels || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    Workspace *w;
    if (int rc = get_workspace(h, B, &w)) return rc;
    if (int rc = stage_inputs(h, w, adj, node, flags, sc_adj, sc_node, s)) return rc;
    w->uniform = false;
    HIP_TRY(h, hipMemcpyAsync(w->c_noise, noise_labels, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (int rc = run_forward(h, w, false, s)) return rc;
    const size_t sa = sizeof(float) * (size_t)B * h->Ca * h->N * h->N, sn = sizeof(float) * (size_t)B * h->N * h->Cn;
    HIP_TRY(h, hipMemcpyAsync(out_adj, w->f_adj, sa, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(out_node, w->f_node, sn, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}

int dsg_precond(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags, const float *sigmas,
                const float *sc_adj, const float *sc_node, int32_t coin, float *out_adj, float *out_node, void *stream) {
    if (int rc = check_ready(h, B)) return rc;
    if (!adj || !node || !flags || !sigmas || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    Workspace *w;
    if (int rc = get_workspace(h, B, &w)) return rc;
    HIP_TRY(h, hipMemcpyAsync(w->flags, flags, (size_t)B * h->N, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(w->sig, sigmas, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    int64_t nfe = 0;
    if (int rc = precond_core(h, w, CStatePtrs{adj, node}, sc_adj, sc_node, coin != 0, StatePtrs{out_adj, out_node}, false, s, &nfe))
        return rc;
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}

// Host-side schedule.  Mirrors the reference's fp32 scalar-tensor arithmetic op by op (edm.py:318-323,
// :355-369); this file is compiled with -ffp-contract=off so no product is fused into a subtraction.
int dsg_sigma_schedule(const dsg_sampler_cfg *c, double *sigma_steps, float *t_hat, float *noise_coef, float *h_step) {
    if (!c || c->num_steps < 1) return DSG_ERR_INVALID;
    const int T = c->num_steps;
    std::vector<float> t(T + 1);
    const double a = std::pow(c->sigma_max, 1.0 / c->rho), b = std::pow(c->sigma_min, 1.0 / c->rho);
    for (int i = 0; i < T; i++) {
        const double frac = T > 1 ? (double)i / (double)(T - 1) : 0.0;
        const double sg = std::pow(a + frac * (b - a), c->rho);
        if (sigma_steps) sigma_steps[i] = sg;
        t[i] = (float)sg;
    }
    t[T] = 0.f;
    // python: min(S_churn / num_steps, np.sqrt(2) - 1) in double, cast to fp32 when multiplied with t_cur
    const float gamma_on = (float)std::fmin((double)c->S_churn / (double)T, std::sqrt(2.0) - 1.0);
    for (int i = 0; i < T; i++) {
        const float tc = t[i];
        const float gamma = (c->S_min <= tc && tc <= c->S_max) ? gamma_on : 0.f;
        volatile float gt = gamma * tc;
        volatile float th = tc + gt;
        volatile float th2 = th * th, tc2 = tc * tc;
        volatile float diff = th2 - tc2;
        volatile float rt = std::sqrt(diff > 0.f ? diff : 0.f);
        volatile float nz = rt * c->S_noise;
        if (t_hat) t_hat[i] = th;
        if (noise_coef) noise_coef[i] = nz;
        if (h_step) h_step[i] = t[i + 1] - th;
    }
    return DSG_OK;
}

int dsg_sample(dsg_handle h, const dsg_sampler_cfg *cfg, int32_t B, const uint8_t *flags, const float *init_adj,
               const float *init_node, const float *noise_adj, const float *noise_node, const uint8_t *coins, uint64_t seed,
               const float *gt_adj, const float *gt_node, const int32_t *snap_steps, int32_t n_snap, float *snap_adj,
               float *snap_node, float *out_adj, float *out_node, dsg_sample_stats *stats, void *stream) {
    if (int rc = check_ready(h, B)) return rc;
    if (!cfg || !flags || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null argument");
    if ((init_adj == nullptr) != (init_node == nullptr)) return fail(h, DSG_ERR_INVALID, "init_adj/init_node must both be given");
    if ((noise_adj == nullptr) != (noise_node == nullptr)) return fail(h, DSG_ERR_INVALID, "noise_adj/noise_node must both be given");
    if ((gt_adj == nullptr) != (gt_node == nullptr)) return fail(h, DSG_ERR_INVALID, "gt_adj/gt_node must both be given");
    const int T = cfg->num_steps;
    if (T < 1) return fail(h, DSG_ERR_INVALID, "num_steps < 1");
    hipStream_t s = (hipStream_t)stream;
    Workspace *w;
    if (int rc = get_workspace(h, B, &w)) return rc;
    const Dims d = dims_of(h, B);
    const size_t sa = (size_t)B * h->Ca * h->N * h->N, sn = (size_t)B * h->N * h->Cn;
    std::vector<float> t_hat(T), nz(T), hs(T);
    dsg_sigma_schedule(cfg, nullptr, t_hat.data(), nz.data(), hs.data());
    std::vector<float> t_steps(T + 1, 0.f);
    {
        std::vector<double> sg(T);
        dsg_sigma_schedule(cfg, sg.data(), nullptr, nullptr, nullptr);
        for (int i = 0; i < T; i++) t_steps[i] = (float)sg[i];
    }
    const int ncalls = cfg->heun ? 2 * T - 1 : T;
    std::vector<uint8_t> coin_buf(ncalls, 0);
    if (coins) memcpy(coin_buf.data(), coins, ncalls);
    else if (h->cfg.self_condition) {
        std::mt19937_64 gen(seed ^ 0x9E3779B97F4A7C15ull);
        for (int i = 0; i < ncalls; i++) coin_buf[i] = (gen() >> 63) & 1;
    }
    h->last_stats = dsg_sample_stats{};
    HIP_TRY(h, hipMemcpyAsync(w->flags, flags, (size_t)B * h->N, hipMemcpyDeviceToDevice, s));
    // x0 = init * sigma(t0) (edm.py:326, :346-347)
    launch_init(CStatePtrs{init_adj, init_node}, t_steps[0], seed, 0u, w->flags, StatePtrs{w->x_adj, w->x_node}, d, s);
    const bool use_graph = cfg->use_graph != 0 && !gt_adj && h->taps.empty();
    // per-step tables on the device: the loop's scalars (StepRow) and, since sigma is batch-uniform (edm.py:371), one noise
    // embedding + (scale,shift) row per step for all blocks, computed once for all steps in three GEMMs with M = T
    if (h->tab_cap < T) {
        drop_graphs(h);   // the captured step bodies bake the table addresses
        for (float **q : {&h->tab_sig, &h->tab_cn, &h->tab_pe, &h->tab_e0, &h->tab_e1, &h->tab_aff}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        if (h->tab_step) { (void)hipFree(h->tab_step); h->tab_step = nullptr; }
        h->tab_cap = 0;
        HIP_TRY(h, hipMalloc((void **)&h->tab_sig, sizeof(float) * T));
        HIP_TRY(h, hipMalloc((void **)&h->tab_cn, sizeof(float) * T));
        HIP_TRY(h, hipMalloc((void **)&h->tab_pe, sizeof(float) * (size_t)T * h->E));
        HIP_TRY(h, hipMalloc((void **)&h->tab_e0, sizeof(float) * (size_t)T * NOISE_EMB));
        HIP_TRY(h, hipMalloc((void **)&h->tab_e1, sizeof(float) * (size_t)T * NOISE_EMB));
        HIP_TRY(h, hipMalloc((void **)&h->tab_aff, sizeof(float) * (size_t)T * h->aff_n));
        HIP_TRY(h, hipMalloc((void **)&h->tab_step, sizeof(StepRow) * (size_t)T));
        h->tab_cap = T;
    }
    std::vector<StepRow> rows(T);
    for (int i = 0; i < T; i++) {
        volatile float t_prime = t_hat[i] + hs[i];  // alpha = 1 (edm.py:391)
        rows[i] = StepRow{nz[i], t_hat[i], 1.0f / t_hat[i], 1.0f / t_prime, hs[i], {0, 0, 0}};
    }
    RunCtl ctl_host{0, 0, (unsigned long long)seed, noise_adj, noise_node};
    HIP_TRY(h, hipMemcpyAsync(h->tab_step, rows.data(), sizeof(StepRow) * T, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(w->ctl, &ctl_host, sizeof(RunCtl), hipMemcpyHostToDevice, s));
    if (!gt_adj) {
        HIP_TRY(h, hipMemcpyAsync(h->tab_sig, t_hat.data(), sizeof(float) * T, hipMemcpyHostToDevice, s));
        launch_cnoise(h->tab_sig, h->tab_cn, T, s);
        embed_rows(h, h->tab_cn, T, h->tab_pe, h->tab_e0, h->tab_e1, h->tab_aff, s);
    }
    HIP_TRY(h, hipStreamSynchronize(s));  // the host vectors above must outlive their async copies; once per sample() call
    // the static plan of every step (edm.py:350-427): buffer rotation, Euler/Heun update, the pre-drawn coins
    std::vector<StepPlan> plans(T);
    int call = 0, snap_k = 0, nfe = 0;
    {
        int sc_slot = -1;  // which d_* buffer holds the current self-cond, -1 = None
        auto free_slot = [&](int a, int b2) { for (int k = 0; k < 3; k++) if (k != a && k != b2) return k; return 0; };
        for (int i = 0; i < T; i++) {
            StepPlan &p = plans[i];
            p.sc_slot = sc_slot;
            p.s1 = free_slot(sc_slot, -1);
            p.s2 = free_slot(p.s1, -1);
            p.euler = !cfg->heun || i == T - 1;   // edm.py:394-396
            p.coin1 = !gt_adj && coin_buf[call] != 0;
            p.coin2 = !gt_adj && !p.euler && coin_buf[call + 1] != 0;
            if (!gt_adj) call += p.euler ? 1 : 2;
            if (!gt_adj && h->cfg.self_condition) sc_slot = p.euler ? p.s1 : p.s2;   // edm.py:423-424: sc <- last denoised
        }
    }
    const bool step_graphs = use_graph && h->opt_loop_graph;
    if (step_graphs)
        for (int i = 0; i < T; i++) if (int rc = ensure_step_graph(h, w, plans[i])) return rc;   // at most ten distinct bodies
    for (int i = 0; i < T; i++) {
        if (step_graphs) { if (int rc = replay_step(h, w, plans[i], s, &nfe)) return rc; }
        else if (int rc = enqueue_step(h, w, plans[i], gt_adj, gt_node, s, &nfe, use_graph)) return rc;
        // interim snapshots (edm.py:429-432)
        while (snap_k < n_snap && snap_steps && snap_steps[snap_k] == i) {
            if (snap_adj) HIP_TRY(h, hipMemcpyAsync(snap_adj + (size_t)snap_k * sa, w->x_adj, sizeof(float) * sa, hipMemcpyDeviceToDevice, s));
            if (snap_node) HIP_TRY(h, hipMemcpyAsync(snap_node + (size_t)snap_k * sn, w->x_node, sizeof(float) * sn, hipMemcpyDeviceToDevice, s));
            snap_k++;
        }
    }
    HIP_TRY(h, hipMemcpyAsync(out_adj, w->x_adj, sizeof(float) * sa, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(out_node, w->x_node, sizeof(float) * sn, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipGetLastError());
    h->last_stats.precond_calls = call;
    h->last_stats.net_forwards = nfe;
    if (stats) *stats = h->last_stats;
    return DSG_OK;
}

int dsg_profile_forward(dsg_handle h, int32_t B, int32_t n_iters, double *ms_by_kind, int64_t *launches_by_kind,
                        double *flops_by_kind, double *gemm_inkernel_ms_out, void *stream) {
    if (int rc = check_ready(h, B)) return rc;
    if (!ms_by_kind || !launches_by_kind || !flops_by_kind || n_iters < 1) return fail(h, DSG_ERR_INVALID, "null argument");
    auto it = h->ws.find(B);
    if (it == h->ws.end()) return fail(h, DSG_ERR_STATE, "no workspace for batch %d: run dsg_denoise/dsg_sample first", B);
    hipStream_t s = (hipStream_t)stream;
    Workspace *w = it->second.get();
    for (int k = 0; k < PK_COUNT; k++) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; flops_by_kind[k] = 0.0; }
    if (!h->prof_gemm) {
        h->prof_gemm_cap = 512;
        HIP_TRY(h, hipMalloc((void **)&h->prof_gemm, sizeof(unsigned long long) * 4 * h->prof_gemm_cap));
    }
    std::vector<unsigned long long> stamps(4 * h->prof_gemm_cap);   // per launch: {min start, max end, block-0 shader cycles, block-0 ticks}
    std::vector<double> clocks;
    double gemm_inkernel_ms = 0.0;
    // pass A: in-kernel stamps only -- the launches run back to back exactly as in the sampler loop
    for (int iter = 0; iter < n_iters; iter++) {
        for (int i = 0; i < h->prof_gemm_cap; i++) { stamps[4 * i] = ~0ull; stamps[4 * i + 1] = 0ull; stamps[4 * i + 2] = 0ull; stamps[4 * i + 3] = 0ull; }
        HIP_TRY(h, hipMemcpyAsync(h->prof_gemm, stamps.data(), sizeof(unsigned long long) * stamps.size(), hipMemcpyHostToDevice, s));
        h->prof_gemm_used = 0;
        h->prof_stamps = true;
        forward_fixed(h, w, s);
        h->prof_stamps = false;
        HIP_TRY(h, hipStreamSynchronize(s));
        HIP_TRY(h, hipMemcpy(stamps.data(), h->prof_gemm, sizeof(unsigned long long) * stamps.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->prof_gemm_used; i++) {
            if (stamps[4 * i + 1] > stamps[4 * i]) gemm_inkernel_ms += (double)(stamps[4 * i + 1] - stamps[4 * i]) * 1e-5;  // 100 MHz ticks
            if (stamps[4 * i + 3] > 200) clocks.push_back((double)stamps[4 * i + 2] / (double)stamps[4 * i + 3] * 0.1);   // GHz, blocks > 2 us
            if (iter == n_iters - 1 && getenv("DSG_PROFILE_VERBOSE") && stamps[4 * i + 3] > 0)
                fprintf(stderr, "[dsg-clock] gemm launch %2d: block 0 held %.3f GHz over %.1f us\n", i,
                        (double)stamps[4 * i + 2] / (double)stamps[4 * i + 3] * 0.1, (double)stamps[4 * i + 3] * 0.01);
        }
    }
    std::sort(clocks.begin(), clocks.end());
    h->prof_clock_ghz = clocks.empty() ? 0.0 : clocks[clocks.size() / 2];
    // pass B: HIP-event brackets around every launch (per-class breakdown; each bracket includes dispatch latency)
    for (int iter = 0; iter < n_iters; iter++) {
        h->prof_on = true; h->prof_used = 0; h->prof_kind.clear(); h->prof_flops.clear(); h->prof_tag.clear();
        forward_fixed(h, w, s);
        const bool ok = h->prof_on;
        h->prof_on = false;
        HIP_TRY(h, hipStreamSynchronize(s));
        if (!ok) return fail(h, DSG_ERR_HIP, "hipEventCreate failed");
        for (size_t i = 0; i < h->prof_kind.size(); i++) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_events[2 * i], h->prof_events[2 * i + 1]));
            if (iter == n_iters - 1 && getenv("DSG_PROFILE_VERBOSE"))
                fprintf(stderr, "[dsg-prof] %8.1f us %7.2f TF  %.60s\n", ms * 1e3, h->prof_flops[i] / (ms * 1e-3) / 1e12, h->prof_tag[i].c_str());
            ms_by_kind[h->prof_kind[i]] += ms;
            launches_by_kind[h->prof_kind[i]] += 1;
            flops_by_kind[h->prof_kind[i]] += h->prof_flops[i];
        }
    }
    if (gemm_inkernel_ms_out) *gemm_inkernel_ms_out = gemm_inkernel_ms;
    return DSG_OK;
}

int32_t dsg_affine_width(dsg_handle h) { return (h && h->finalized) ? h->aff_n : 0; }

int dsg_noise_embed(dsg_handle h, int32_t rows, const float *c_noise, float *out_pe, float *out_emb, float *out_aff, void *stream) {
    if (!h || !h->finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (rows < 1 || !c_noise) return fail(h, DSG_ERR_INVALID, "rows / c_noise");
    hipStream_t s = (hipStream_t)stream;
    float *tmp = nullptr;   // pe | emb0 | emb | aff
    const size_t n_pe = (size_t)rows * h->E, n_e = (size_t)rows * NOISE_EMB, n_a = (size_t)rows * h->aff_n;
    HIP_TRY(h, hipMalloc((void **)&tmp, sizeof(float) * (n_pe + 2 * n_e + n_a)));
    float *pe = tmp, *emb0 = pe + n_pe, *emb = emb0 + n_e, *aff = emb + n_e;
    embed_rows(h, c_noise, rows, pe, emb0, emb, aff, s);
    hipError_t e = hipSuccess;
    if (out_pe) e = hipMemcpyAsync(out_pe, pe, sizeof(float) * n_pe, hipMemcpyDeviceToDevice, s);
    if (out_emb && e == hipSuccess) e = hipMemcpyAsync(out_emb, emb, sizeof(float) * n_e, hipMemcpyDeviceToDevice, s);
    if (out_aff && e == hipSuccess) e = hipMemcpyAsync(out_aff, aff, sizeof(float) * n_a, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    HIP_TRY(h, e);
    return DSG_OK;
}

int dsg_debug_gemm(int32_t M, int32_t N, int32_t K, const float *A, const float *W, const float *bias, const float *ln_stats,
                   const float *res, int32_t act, int32_t mode, float *C, void *stream) {
    if (M < 1 || N < 1 || K < 32 || K % 32 != 0 || !A || !W || !C || act < 0 || act > 2 || mode < 0 || mode > 2) return DSG_ERR_INVALID;
    if (!gelu_table()) return DSG_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    void *wlp = nullptr;
    if (mode == 1) { if (hipMalloc(&wlp, (size_t)N * K * 2) != hipSuccess) return DSG_ERR_HIP; launch_f32_to_bf16(W, wlp, (size_t)N * K, s); }
    if (mode == 2) { if (hipMalloc(&wlp, (size_t)N * K * 6) != hipSuccess) return DSG_ERR_HIP; launch_f32_split3(W, wlp, (size_t)N * K, s); }
    GemmArgs g;
    g.A = A; g.lda = K; g.K1 = K; g.K = K; g.M = M; g.N = N; g.W = W; g.bias = bias; g.ln_stats = ln_stats; g.act = act;
    if (res) { g.res = res; g.ldres = N; }
    g.C = C; g.ldc = N;
    if (mode == 1) g.Wb = wlp;
    if (mode == 2) g.Ws3 = wlp;
    const bool built = launch_gemm(g, s);
    const hipError_t e = hipStreamSynchronize(s);
    if (wlp) (void)hipFree(wlp);
    if (!built) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

static float time_launches(hipStream_t s, int iters, const std::function<void()> &launch) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int i = 0; i < 3; i++) launch();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; i++) launch();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / (float)iters;
}

int dsg_debug_gemm_bx(int32_t M, int32_t N, int32_t K, const float *A, const float *W, const float *bias, const float *res, int32_t act,
                      const float *mod, int32_t ln_out, float *out_C, float *out_Cb, float *out_C2b, int32_t time_iters, float *out_ms,
                      void *stream) {
    if (M < 1 || N < 1 || K < 8 || !A || !W || (!out_C && !out_Cb)) return DSG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    void *ab = nullptr, *wb = nullptr, *cb = nullptr, *c2b = nullptr;
    auto cleanup = [&]() { (void)hipFree(ab); (void)hipFree(wb); (void)hipFree(cb); (void)hipFree(c2b); };
    if (hipMalloc(&ab, (size_t)M * K * 2) != hipSuccess || hipMalloc(&wb, (size_t)N * K * 2) != hipSuccess ||
        hipMalloc(&cb, (size_t)M * N * 2) != hipSuccess || hipMalloc(&c2b, (size_t)M * N * 2) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
    launch_f32_to_bf16(A, ab, (size_t)M * K, s);
    launch_f32_to_bf16(W, wb, (size_t)N * K, s);
    BxGemm g;
    g.A = ab; g.lda = K; g.K = K; g.M = M; g.N = N; g.W = wb; g.bias = bias; g.act = act;
    if (res) { g.res = res; g.ldres = N; }
    if (out_C) { g.C = out_C; g.ldc = N; }
    if (out_Cb) { g.Cb = cb; g.ldcb = N; g.ln_out = ln_out; }
    if (out_C2b) { g.C2b = c2b; g.ldc2b = N; }
    if (mod) { g.mod_aff = mod; g.mod_ld = 0; g.mod_off = 0; g.mod_T = 1; }
    const bool ok = launch_gemm_bx(g, s);
    if (ok && out_Cb) launch_bf16_to_f32(cb, out_Cb, (size_t)M * N, s);
    if (ok && out_C2b) launch_bf16_to_f32(c2b, out_C2b, (size_t)M * N, s);
    if (ok && time_iters > 0 && out_ms) {   // timing: the residual stream is updated in place like in the forward (C aliases res)
        if (g.res && g.C) { g.C = const_cast<float *>(g.res); g.ldc = g.ldres; }
        *out_ms = time_launches(s, time_iters, [&]() { (void)launch_gemm_bx(g, s); });
        if (getenv("DSG_BX_DBG")) {   // phase clocks of a DSG_BX_EXP=4 build (tools/bx_exp.sh): mean over blocks, per tile
            const int nb = 4096;
            unsigned long long *dbg = nullptr;
            if (hipMalloc((void **)&dbg, sizeof(unsigned long long) * 8 * nb) == hipSuccess) {
                (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 8 * nb, s);
                g.dbg = dbg;
                (void)launch_gemm_bx(g, s);
                std::vector<unsigned long long> hbuf(8 * nb);
                (void)hipMemcpyAsync(hbuf.data(), dbg, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                double sum[7] = {0, 0, 0, 0, 0, 0, 0};
                int cnt = 0;
                for (int b = 0; b < nb; b++) if (hbuf[8 * b + 6]) { cnt++; for (int i = 0; i < 7; i++) sum[i] += (double)hbuf[8 * b + i]; }
                if (cnt) {
                    const double tiles = sum[5] / cnt;
                    fprintf(stderr, "   bx phases (kclk per tile, mean of %d blocks, %.1f tiles each): mfma+issue %.2f | barrier1 %.2f | wait+refill %.2f | barrier2 %.2f | "
                                    "epilogue %.2f | block total %.2f kclk\n", cnt, tiles, sum[0] / cnt / tiles / 1e3, sum[1] / cnt / tiles / 1e3, sum[2] / cnt / tiles / 1e3,
                            sum[3] / cnt / tiles / 1e3, sum[4] / cnt / tiles / 1e3, sum[6] / cnt / 1e3);
                }
                (void)hipFree(dbg);
                g.dbg = nullptr;
            }
        }
    }
    const hipError_t e = hipStreamSynchronize(s);
    cleanup();
    if (!ok) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_debug_mlp_bx(int32_t M, int32_t C, const float *xn, float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *mod, int32_t out_mode, float *out_xn, int32_t time_iters, float *out_ms, void *stream) {
    const int narrow384 = (out_mode >> 4) & 1;   // + 16: C = 384 on the one-wave-per-SIMD kernel instead of the eight-wave ones
    const int old384 = (out_mode >> 5) & 1;      // + 32: C = 384 on round 3's eight-wave kernel (register-staged weights) instead of the LDS-DMA one
    const int solo384 = (out_mode >> 6) & 1;     // + 64: C = 384 on mlp384s_bx_kernel (four waves, chunk-major LDS-DMA ring)
    out_mode &= 15;
    if (M < 1 || !xn || !x || !W1 || !b1 || !W2 || !b2 || (out_mode && !out_xn)) return DSG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    void *xb = nullptr, *w1b = nullptr, *w2b = nullptr, *ob = nullptr, *imgb = nullptr;
    auto cleanup = [&]() { (void)hipFree(xb); (void)hipFree(w1b); (void)hipFree(w2b); (void)hipFree(ob); (void)hipFree(imgb); };
    if (hipMalloc(&xb, (size_t)M * C * 2) != hipSuccess || hipMalloc(&w1b, (size_t)4 * C * C * 2) != hipSuccess ||
        hipMalloc(&w2b, (size_t)4 * C * C * 2) != hipSuccess || hipMalloc(&ob, (size_t)M * C * 2) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
    launch_f32_to_bf16(xn, xb, (size_t)M * C, s);
    launch_f32_to_bf16(W1, w1b, (size_t)4 * C * C, s);
    launch_f32_to_bf16(W2, w2b, (size_t)4 * C * C, s);
    BxMlp g;
    g.xn = xb; g.x = x; g.W1 = w1b; g.b1 = b1; g.W2 = w2b; g.b2 = b2; g.M = M; g.C = C;
    if (out_mode) { g.xn_out = ob; g.out_mode = out_mode; }
    g.wide8 = narrow384 ? 0 : (old384 ? 2 : (solo384 ? 3 : 1));
    if (C == 384 && (g.wide8 == 1 || g.wide8 == 3)) {
        if (hipMalloc(&imgb, mlp384_image_bytes()) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
        if (g.wide8 == 1) { launch_mlp384_images(w1b, w2b, nullptr, imgb, s); g.img = imgb; }
        else { launch_mlp384s_images(w1b, w2b, nullptr, imgb, s); g.img2 = imgb; }
    }
    if (mod) { g.mod_aff = mod; g.mod_ld = 0; g.mod_off = 0; g.mod_T = 1; }
    const bool ok = launch_mlp_bx(g, s);
    if (ok && out_mode) launch_bf16_to_f32(ob, out_xn, (size_t)M * C, s);
    if (ok && time_iters > 0 && out_ms) *out_ms = time_launches(s, time_iters, [&]() { (void)launch_mlp_bx(g, s); });   // (x keeps accumulating: timing only)
    const hipError_t e = hipStreamSynchronize(s);
    cleanup();
    if (!ok) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_debug_projmlp_bx(int32_t M, int32_t C, const float *att, float *x, const float *Wp, const float *bp, const float *W1, const float *b1,
                         const float *W2, const float *b2, const float *mod, int32_t out_mode, float *out_xn, int32_t time_iters, float *out_ms,
                         void *stream) {
    const int old384 = (out_mode >> 5) & 1;      // + 32: C = 384 on round 3's eight-wave kernel instead of the LDS-DMA one
    const int solo384 = (out_mode >> 6) & 1;     // + 64: C = 384 on mlp384s_bx_kernel
    const int staged96 = !((out_mode >> 7) & 1); // + 128: C = 96 without modulate on the LDS-resident kernel (mlp96r_bx_kernel) instead of the staged one
    out_mode &= 15;
    if (M < 1 || !att || !x || !Wp || !bp || !W1 || !b1 || !W2 || !b2 || (out_mode && !out_xn)) return DSG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    void *ab = nullptr, *wpb = nullptr, *w1b = nullptr, *w2b = nullptr, *ob = nullptr, *imgb = nullptr;
    auto cleanup = [&]() { (void)hipFree(ab); (void)hipFree(wpb); (void)hipFree(w1b); (void)hipFree(w2b); (void)hipFree(ob); (void)hipFree(imgb); };
    if (hipMalloc(&ab, (size_t)M * C * 2) != hipSuccess || hipMalloc(&wpb, (size_t)C * C * 2) != hipSuccess || hipMalloc(&w1b, (size_t)4 * C * C * 2) != hipSuccess ||
        hipMalloc(&w2b, (size_t)4 * C * C * 2) != hipSuccess || hipMalloc(&ob, (size_t)M * C * 2) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
    launch_f32_to_bf16(att, ab, (size_t)M * C, s);
    launch_f32_to_bf16(Wp, wpb, (size_t)C * C, s);
    launch_f32_to_bf16(W1, w1b, (size_t)4 * C * C, s);
    launch_f32_to_bf16(W2, w2b, (size_t)4 * C * C, s);
    BxMlp g;
    g.att = ab; g.Wp = wpb; g.bp = bp; g.x = x; g.W1 = w1b; g.b1 = b1; g.W2 = w2b; g.b2 = b2; g.M = M; g.C = C;
    if (out_mode) { g.xn_out = ob; g.out_mode = out_mode; }
    g.wide8 = old384 ? 2 : (solo384 ? 3 : 1);
    if (C == 384 && (g.wide8 == 1 || g.wide8 == 3)) {
        if (hipMalloc(&imgb, mlp384_image_bytes()) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
        if (g.wide8 == 1) { launch_mlp384_images(w1b, w2b, wpb, imgb, s); g.img = imgb; }
        else { launch_mlp384s_images(w1b, w2b, wpb, imgb, s); g.img2 = imgb; }
    }
    if (C == 96 && !mod && !staged96) {
        if (hipMalloc(&imgb, mlp96r_image_bytes()) != hipSuccess) { cleanup(); return DSG_ERR_HIP; }
        launch_mlp96r_image(w1b, w2b, wpb, imgb, s);
        g.img96 = imgb;
    }
    if (mod) { g.mod_aff = mod; g.mod_ld = 0; g.mod_off = 0; g.mod_T = 1; }
    const bool ok = launch_mlp_bx(g, s);
    if (ok && out_mode) launch_bf16_to_f32(ob, out_xn, (size_t)M * C, s);
    if (ok && time_iters > 0 && out_ms) *out_ms = time_launches(s, time_iters, [&]() { (void)launch_mlp_bx(g, s); });   // (x keeps accumulating: timing only)
    if (ok && C == 384 && g.wide8 == 1 && getenv("DSG_M384_CLK")) {   // dev measurement: phase clocks of one more launch (tools/bx_bench.py)
        const int nb = (M + 127) / 128;
        unsigned long long *dbg = nullptr;
        if (hipMalloc((void **)&dbg, sizeof(unsigned long long) * nb * 128) == hipSuccess) {
            (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * nb * 128, s);
            g.dbg = dbg;
            (void)launch_mlp_bx(g, s);
            std::vector<unsigned long long> hb((size_t)nb * 128);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hb.data(), dbg, sizeof(unsigned long long) * hb.size(), hipMemcpyDeviceToHost);
            double ph[4] = {0, 0, 0, 0}, tot = 0, iv[2][6] = {{0}};
            for (int b = 0; b < nb; b++)
                for (int w = 0; w < 8; w++) {
                    const unsigned long long *t = hb.data() + ((size_t)b * 8 + w) * 16;
                    for (int k = 0; k < 4; k++) ph[k] += (double)(t[k + 1] - t[k]);
                    tot += (double)(t[4] - t[0]);
                    for (int k = 0; k < 6; k++) iv[w >> 2][k] += (double)(t[6 + k] - t[5 + k]);
                }
            const double n = (double)nb * 8;
            fprintf(stderr, "   mlp384d phases (kclk per wave, mean of %d blocks): proj %.1f | LN + exchange %.1f | chunk-pair loop %.1f | epilogue %.1f | block %.1f\n",
                    nb, ph[0] / n / 1e3, ph[1] / n / 1e3, ph[2] / n / 1e3, ph[3] / n / 1e3, tot / n / 1e3);
            for (int tm = 0; tm < 2; tm++)
                fprintf(stderr, "      pair 8, %s half (clk): fc1 %.0f | wait + barrier %.0f | GELU + exchange %.0f | wait + barrier %.0f | fc2 %.0f | wait + barrier %.0f\n",
                        tm ? "second" : "first", iv[tm][0] / (n / 2), iv[tm][1] / (n / 2), iv[tm][2] / (n / 2), iv[tm][3] / (n / 2), iv[tm][4] / (n / 2), iv[tm][5] / (n / 2));
            (void)hipFree(dbg);
            g.dbg = nullptr;
        }
    }
    if (ok && C == 384 && g.wide8 == 3 && getenv("DSG_M384_CLK")) {   // dev measurement: phase clocks of one more launch of mlp384s_bx_kernel
        const int nb = (M + 127) / 128;
        unsigned long long *dbg = nullptr;
        if (hipMalloc((void **)&dbg, sizeof(unsigned long long) * nb * 128) == hipSuccess) {
            (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * nb * 128, s);
            g.dbg = dbg;
            (void)launch_mlp_bx(g, s);
            std::vector<unsigned long long> hb((size_t)nb * 128);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hb.data(), dbg, sizeof(unsigned long long) * hb.size(), hipMemcpyDeviceToHost);
            double ph[4] = {0, 0, 0, 0}, tot = 0, iv[3] = {0, 0, 0};
            for (int b = 0; b < nb; b++)
                for (int w = 0; w < 4; w++) {
                    const unsigned long long *t = hb.data() + ((size_t)b * 8 + w) * 16;
                    for (int k = 0; k < 4; k++) ph[k] += (double)(t[k + 1] - t[k]);
                    tot += (double)(t[4] - t[0]);
                    for (int k = 0; k < 3; k++) iv[k] += (double)(t[6 + k] - t[5 + k]);
                }
            const double n = (double)nb * 4;
            fprintf(stderr, "   mlp384s phases (kclk per wave, mean of %d blocks): prologue %.1f | wait %.1f | chunk loop %.1f | epilogue %.1f | block %.1f\n",
                    nb, ph[0] / n / 1e3, ph[1] / n / 1e3, ph[2] / n / 1e3, ph[3] / n / 1e3, tot / n / 1e3);
            fprintf(stderr, "      chunk 8 (clk): wait + barrier %.0f | fc1(c+1) + GELU(c) %.0f | fc2(c) %.0f\n", iv[0] / n, iv[1] / n, iv[2] / n);
            (void)hipFree(dbg);
            g.dbg = nullptr;
        }
    }
    const hipError_t e = hipStreamSynchronize(s);
    cleanup();
    if (!ok) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_debug_attn_bx(int32_t B, int32_t res, int32_t ws, int32_t shift, int32_t heads, const float *qkv, const float *biasT, float *out,
                      int32_t time_iters, float *out_ms, void *stream) {
    if (B < 1 || res < 1 || ws < 1 || res % ws != 0 || heads < 1 || !qkv || !biasT || !out) return DSG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int C = 32 * heads;
    const size_t M = (size_t)B * res * res;
    void *qb = nullptr, *ob = nullptr;
    if (hipMalloc(&qb, M * 3 * C * 2) != hipSuccess || hipMalloc(&ob, M * C * 2) != hipSuccess) { (void)hipFree(qb); (void)hipFree(ob); return DSG_ERR_HIP; }
    launch_f32_to_bf16(qkv, qb, M * 3 * C, s);
    const bool ok = launch_attn_bx(qb, biasT, ob, B, WinGeom{res, ws, shift, heads, C}, s);
    if (ok) launch_bf16_to_f32(ob, out, M * C, s);
    if (ok && time_iters > 0 && out_ms)
        *out_ms = time_launches(s, time_iters, [&]() { (void)launch_attn_bx(qb, biasT, ob, B, WinGeom{res, ws, shift, heads, C}, s); });
    const hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(qb); (void)hipFree(ob);
    if (!ok) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_debug_qkv_attn_bx(int32_t B, int32_t res, int32_t ws, int32_t shift, int32_t heads, const float *xn, const float *W, const float *bias,
                          const float *biasT, float *out, int32_t time_iters, float *out_ms, void *stream) {
    const int variant = shift >= 1000 ? 1 : 0;   // shift + 1000: 10 x 10 windows on the block-per-head kernel instead of the wave-per-unit one
    shift %= 1000;
    if (B < 1 || res < 1 || ws < 1 || res % ws != 0 || heads < 1 || !xn || !W || !bias || !biasT || !out) return DSG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int C = 32 * heads, Wp = (ws * ws + 31) / 32 * 32, nW = (res / ws) * (res / ws), nWt = shift > 0 ? nW : 1;
    const size_t M = (size_t)B * res * res, nb = (size_t)nWt * heads * Wp * Wp;
    void *xb = nullptr, *wb = nullptr, *ob = nullptr, *bp = nullptr;
    if (hipMalloc(&xb, M * C * 2) != hipSuccess || hipMalloc(&wb, (size_t)3 * C * C * 2) != hipSuccess || hipMalloc(&ob, M * C * 2) != hipSuccess ||
        hipMalloc(&bp, nb * 2) != hipSuccess) { (void)hipFree(xb); (void)hipFree(wb); (void)hipFree(ob); (void)hipFree(bp); return DSG_ERR_HIP; }
    launch_f32_to_bf16(xn, xb, M * C, s);
    launch_f32_to_bf16(W, wb, (size_t)3 * C * C, s);
    launch_bias_permute_bx(biasT, bp, nWt * heads, Wp, s);
    BxQkvAttn qa;
    qa.xn = xb; qa.W = wb; qa.bias = bias; qa.biasP = bp; qa.out = ob; qa.B = B; qa.g = WinGeom{res, ws, shift, heads, C};
    qa.variant = variant;
    void *qimg = nullptr;
    float *bfp = nullptr;
    if (ws == 10 && variant == 0) {
        if (hipMalloc(&qimg, (size_t)3 * C * C * 2) != hipSuccess) { (void)hipFree(xb); (void)hipFree(wb); (void)hipFree(ob); (void)hipFree(bp); return DSG_ERR_HIP; }
        launch_qkv_image(wb, qimg, C, heads, s);
        qa.Wimg = qimg;
        if (hipMalloc((void **)&bfp, nb * 4) != hipSuccess) { (void)hipFree(xb); (void)hipFree(wb); (void)hipFree(ob); (void)hipFree(bp); (void)hipFree(qimg); (void)hipFree(bfp); return DSG_ERR_HIP; }
        launch_bias_permute_f32(biasT, bfp, nWt * heads, Wp, s);
        qa.biasF = bfp;
    }
    const bool ok = launch_qkv_attn_bx(qa, s);
    if (ok) launch_bf16_to_f32(ob, out, M * C, s);
    if (ok && time_iters > 0 && out_ms) *out_ms = time_launches(s, time_iters, [&]() { (void)launch_qkv_attn_bx(qa, s); });
    if (ok && ws == 10 && variant == 0 && getenv("DSG_WX_CLK")) {   // dev measurement: phase clocks of one more launch of qkv_attn_wx_kernel
        const int nunits = B * nW * heads, nb = (nunits + 3) / 4;
        unsigned long long *dbg = nullptr;
        if (hipMalloc((void **)&dbg, sizeof(unsigned long long) * nb * 32) == hipSuccess) {
            (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * nb * 32, s);
            qa.dbg = dbg;
            (void)launch_qkv_attn_bx(qa, s);
            qa.dbg = nullptr;
            std::vector<unsigned long long> hb((size_t)nb * 32);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hb.data(), dbg, sizeof(unsigned long long) * hb.size(), hipMemcpyDeviceToHost);
            double ph[4] = {0, 0, 0, 0};
            for (int b = 0; b < nb; b++)
                for (int w = 0; w < 4; w++)
                    for (int k = 0; k < 4; k++) ph[k] += (double)(hb[((size_t)b * 4 + w) * 8 + k + 1] - hb[((size_t)b * 4 + w) * 8 + k]);
            const double n = (double)nb * 4;
            fprintf(stderr, "   qkv_attn_wx phases (kclk per wave, mean of %d blocks): setup + first DMA %.1f | K loop %.1f | q/k/v conversion %.1f | attention + stores %.1f\n",
                    nb, ph[0] / n / 1e3, ph[1] / n / 1e3, ph[2] / n / 1e3, ph[3] / n / 1e3);
            (void)hipFree(dbg);
        }
    }
    const hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(xb); (void)hipFree(wb); (void)hipFree(ob); (void)hipFree(bp); (void)hipFree(qimg); (void)hipFree(bfp);
    if (!ok) return DSG_ERR_INVALID;
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_train_inputs(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *clean_adj, const float *clean_node,
                     const uint8_t *flags, const float *rnd_sigma, const float *eps_adj, const float *eps_node, uint64_t seed,
                     float *out_sigmas, float *out_weights, float *out_noisy_adj, float *out_noisy_node, void *stream) {
    if (B < 1 || N < 1 || c_adj < 1 || c_node < 1 || !clean_adj || !clean_node || !flags || !out_sigmas || !out_weights ||
        !out_noisy_adj || !out_noisy_node || ((eps_adj == nullptr) != (eps_node == nullptr)))
        return DSG_ERR_INVALID;
    launch_train_inputs(CStatePtrs{clean_adj, clean_node}, rnd_sigma, CStatePtrs{eps_adj, eps_node}, seed, flags, out_sigmas, out_weights,
                        StatePtrs{out_noisy_adj, out_noisy_node}, Dims{B, N, c_adj, c_node}, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DSG_OK : DSG_ERR_HIP;
}

int dsg_rainbow_loss(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *pred_adj, const float *pred_node,
                     const float *target_adj, const float *target_node, const uint8_t *flags, const float *loss_weight,
                     float edge_loss_weight, float node_loss_weight, float iou_loss_weight, int32_t iou_loss_type, float *out_loss_adj,
                     float *out_loss_node, void *stream) {
    if (B < 1 || N < 1 || c_adj < 1 || c_node < 1 || !pred_adj || !pred_node || !target_adj || !target_node || !flags || !out_loss_adj ||
        !out_loss_node || (iou_loss_weight != 0.f && c_node < 4) || iou_loss_type < DSG_IOU_IOU || iou_loss_type > DSG_IOU_CIOU)
        return DSG_ERR_INVALID;
    launch_rainbow_loss(CStatePtrs{pred_adj, pred_node}, CStatePtrs{target_adj, target_node}, flags, loss_weight, edge_loss_weight,
                        node_loss_weight, iou_loss_weight, iou_loss_type, out_loss_adj, out_loss_node, Dims{B, N, c_adj, c_node},
                        (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DSG_OK : DSG_ERR_HIP;
}

int dsg_rainbow_loss_backward(int32_t B, int32_t N, int32_t c_adj, int32_t c_node, const float *pred_adj, const float *pred_node,
                              const float *target_adj, const float *target_node, const uint8_t *flags, const float *loss_weight,
                              float edge_loss_weight, float node_loss_weight, float iou_loss_weight, int32_t iou_loss_type,
                              const float *sigmas, float *out_grad_adj, float *out_grad_node, float *out_grad_F_adj, float *out_grad_F_node,
                              void *stream) {
    if (B < 1 || N < 1 || c_adj < 1 || c_node < 1 || !pred_adj || !pred_node || !target_adj || !target_node || !flags || !out_grad_adj ||
        !out_grad_node || (iou_loss_weight != 0.f && c_node < 4) || ((out_grad_F_adj || out_grad_F_node) && !sigmas) ||
        iou_loss_type < DSG_IOU_IOU || iou_loss_type > DSG_IOU_CIOU)
        return DSG_ERR_INVALID;
    launch_rainbow_loss_backward(CStatePtrs{pred_adj, pred_node}, CStatePtrs{target_adj, target_node}, flags, loss_weight,
                                 edge_loss_weight, node_loss_weight, iou_loss_weight, iou_loss_type, sigmas, StatePtrs{out_grad_adj, out_grad_node},
                                 StatePtrs{out_grad_F_adj, out_grad_F_node}, Dims{B, N, c_adj, c_node}, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DSG_OK : DSG_ERR_HIP;
}

int dsg_block_train(dsg_handle h, const char *block, int32_t B, const float *x_in, const float *emb, const float *grad_out, float *x_out,
                    float *grad_in, float *grad_emb, int32_t n_params, const char *const *names, float *const *grad_params, void *stream) {
    if (!h || !h->ever_finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (!block || B < 1 || !x_in || !emb || !x_out) return fail(h, DSG_ERR_INVALID, "null argument");
    const BlockPlan *bp = nullptr;
    for (int l = 0; l < h->L; l++)
        for (auto *vec : {&h->down[l], &h->up[l]})
            for (auto &b : *vec) if (b.prefix == block) bp = &b;
    if (!bp) return fail(h, DSG_ERR_INVALID, "no block '%s'", block);
    static const char *kNames[15] = {"affine.weight", "affine.bias", "norm1.weight", "norm1.bias", "attn.relative_position_bias_table",
                                     "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight",
                                     "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"};
    TrainBlockArgs a{};
    float **Wp[15] = {&a.W.aff_w, &a.W.aff_b, &a.W.n1_w, &a.W.n1_b, &a.W.rpb, &a.W.qkv_w, &a.W.qkv_b, &a.W.proj_w, &a.W.proj_b,
                      &a.W.n2_w, &a.W.n2_b, &a.W.fc1_w, &a.W.fc1_b, &a.W.fc2_w, &a.W.fc2_b};
    float **Gp[15] = {&a.G.aff_w, &a.G.aff_b, &a.G.n1_w, &a.G.n1_b, &a.G.rpb, &a.G.qkv_w, &a.G.qkv_b, &a.G.proj_w, &a.G.proj_b,
                      &a.G.n2_w, &a.G.n2_b, &a.G.fc1_w, &a.G.fc1_b, &a.G.fc2_w, &a.G.fc2_b};
    for (int i = 0; i < 15; i++) {
        auto it = h->w.find(bp->prefix + "." + kNames[i]);
        if (it == h->w.end()) return fail(h, DSG_ERR_STATE, "weight '%s.%s' is not loaded", block, kNames[i]);
        *Wp[i] = it->second.p;
    }
    if (grad_out) {
        if (!grad_in || !grad_emb || !names || !grad_params) return fail(h, DSG_ERR_INVALID, "backward needs grad_in, grad_emb and the parameter gradient buffers");
        for (int i = 0; i < 15; i++) {
            float *dst = nullptr;
            for (int k = 0; k < n_params; k++) if (names[k] && !strcmp(names[k], kNames[i])) dst = grad_params[k];
            if (!dst) return fail(h, DSG_ERR_INVALID, "no gradient buffer for '%s'", kNames[i]);
            *Gp[i] = dst;
        }
    }
    a.B = B; a.res = bp->res; a.ws = bp->ws; a.shift = bp->shift; a.heads = bp->heads; a.C = bp->C; a.hidden = h->cfg.mlp_ratio * bp->C;
    a.x_in = x_in; a.emb = emb; a.grad_out = grad_out; a.x_out = x_out; a.grad_in = grad_in; a.grad_emb = grad_emb;
    const size_t M = (size_t)B * a.res * a.res, C = a.C, H = a.hidden;
    const size_t sizes[17] = {(size_t)B * 2 * C, (size_t)B * 2 * C, M * C, M * C, M * C, M * C, M * C, M * C, M * C, M * C, M * 2, M * 2,
                              M * 3 * C, M * 3 * C, M * H, M * H, M * H};
    float **slots[17] = {&a.aff, &a.d_aff, &a.x_mod, &a.xn1, &a.att, &a.x1, &a.xn2, &a.d_x1, &a.t_mc, &a.t_mc2, &a.stats1, &a.stats2,
                         &a.qkv, &a.t_m3c, &a.pre, &a.hid, &a.t_mh};
    size_t total = 0;
    for (size_t v : sizes) total += (v + 63) / 64 * 64;
    float *scratch = nullptr;
    HIP_TRY(h, hipMalloc((void **)&scratch, sizeof(float) * total));
    size_t off = 0;
    for (int i = 0; i < 17; i++) { *slots[i] = scratch + off; off += (sizes[i] + 63) / 64 * 64; }
    const bool ok = train_block(a, (hipStream_t)stream);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(scratch);
    HIP_TRY(h, e);
    if (t_scratch_failed((hipStream_t)stream, true)) return fail(h, DSG_ERR_HIP, "out of memory (training scratch of this stream)");
    if (!ok) return fail(h, DSG_ERR_HIP, "train_block failed (window larger than 128 tokens or a launch error)");
    return DSG_OK;
}

// ---- whole-network training step: forward in training form + backward to every parameter (correctness-first kernels) ----
namespace {
struct TArena {   // bump allocator over one hipMalloc
    float *base = nullptr; size_t cap = 0, off = 0; bool dry = true;
    float *get(size_t n) { n = (n + 63) / 64 * 64; float *p = dry ? nullptr : base + off; off += n; return p; }
};
struct TLin { const float *x; float *y; };   // nothing else: linears keep their input pointer in the stage structs below
}  // namespace

// a handle-owned device buffer of at least `need` floats (grown after draining the stream that may still read the old one)
static float *train_buf(dsg_handle h, float *&buf, size_t &cap, size_t need, hipStream_t s) {
    if (need > cap) {
        if (buf) { (void)hipStreamSynchronize(s); (void)hipFree(buf); }
        buf = nullptr; cap = 0;
        if (hipMalloc((void **)&buf, sizeof(float) * need) != hipSuccess) { buf = nullptr; return nullptr; }
        cap = need;
    }
    return buf;
}

// `mid` (optional) runs between forward and backward on the same stream and fills dL/dF from the forward's outputs
typedef std::function<int(const float *F_adj, const float *F_node, float *dF_adj, float *dF_node)> TrainMid;
static int train_grads_core(dsg_handle h, int32_t B, const float *in_adj, const float *in_node, const uint8_t *flags, const float *c_noise,
                            const float *sc_adj, const float *sc_node, const float *grad_F_adj, const float *grad_F_node, const TrainMid *mid,
                            float *out_F_adj, float *out_F_node, int32_t n_params, const char *const *names, float *const *grad_params,
                            void *stream) {
    // the training form reads the raw weights only, so weights re-uploaded after an optimiser step need no re-packing here
    // (dsg_finalize_weights must have run once: block plans); the sampling-path entries still insist on `finalized`
    if (!h || !h->ever_finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (B < 1 || !in_adj || !in_node || !flags || !c_noise || !out_F_adj || !out_F_node) return fail(h, DSG_ERR_INVALID, "null argument");
    const bool bwd = grad_F_adj != nullptr || mid != nullptr;
    if (bwd && ((!mid && !grad_F_node) || !names || !grad_params)) return fail(h, DSG_ERR_INVALID, "backward needs both output gradients and the parameter gradient buffers");
    if (h->cfg.self_condition && ((sc_adj == nullptr) != (sc_node == nullptr))) return fail(h, DSG_ERR_INVALID, "self-conditioning needs both tensors or none");
    hipStream_t s = (hipStream_t)stream;
    const int N = h->N, E = h->E, L = h->L, Ca = h->Ca, Cn = h->Cn, Cin = h->Cin, T0 = N * N;
    const size_t M0 = (size_t)B * T0;
    std::unordered_map<std::string, float *> gmap;
    if (bwd) for (int k = 0; k < n_params; k++) if (names[k] && grad_params[k]) gmap[names[k]] = grad_params[k];
    std::string missing;
    auto Wt = [&](const std::string &k) -> float * {
        if (!h->train_params.empty()) { auto bt = h->train_params.find(k); if (bt != h->train_params.end()) return const_cast<float *>(bt->second); }
        auto it = h->w.find(k); if (it == h->w.end()) { if (missing.empty()) missing = k; return nullptr; } return it->second.p; };
    auto Gd = [&](const std::string &k) -> float * { if (!bwd) return nullptr; auto it = gmap.find(k); if (it == gmap.end()) { if (missing.empty()) missing = "grad:" + k; return nullptr; } return it->second; };
    // collect the blocks in execution order
    struct Stage { const BlockPlan *bp; TrainBlockArgs a; float *x_in, *x_out, *d_emb; };
    std::vector<Stage> enc[8], dec[8];
    TArena A;
    const int has_sc_host = (h->cfg.self_condition && sc_adj) ? 1 : 0;
    if (!h->train_const) {
        const int zo[2] = {0, 1};
        HIP_TRY(h, hipMalloc((void **)&h->train_const, sizeof(zo)));
        HIP_TRY(h, hipMemcpy(h->train_const, zo, sizeof(zo), hipMemcpyHostToDevice));
    }
    const int *has_sc_dev = h->train_const + has_sc_host;
    // everything below runs twice: a dry pass that only sizes the arena, then the real pass
    struct Bufs {
        float *pe, *m0, *s0, *m1, *emb, *d_emb, *pe_demb, *tok, *pe_lin, *pe_ln, *pe_stats, *pe_aff, *pe_daff, *x0;
        float *mcat[8], *mnrm[8], *mstats[8], *mout[8];                       // PatchMerging per level
        float *ucat[8], *uy[8], *un[8], *ustats[8], *usc[8], *upn[8], *upstats[8], *uout[8];   // PatchBreakup per up layer
        float *fy, *fstats, *z1, *z2, *rep, *ha_pre, *ha, *oa, *pool, *hn_pre, *hn, *on;
        float *d_skip[8];
    } Bf{};
    const float *skip_ptr[8] = {};
    int skip_res[8] = {}, skip_C[8] = {};
    auto fill_block = [&](const BlockPlan &b, Stage &st) -> bool {
        static const char *kNames[15] = {"affine.weight", "affine.bias", "norm1.weight", "norm1.bias", "attn.relative_position_bias_table",
                                         "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight",
                                         "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"};
        TrainBlockArgs &a = st.a;
        float **Wp[15] = {&a.W.aff_w, &a.W.aff_b, &a.W.n1_w, &a.W.n1_b, &a.W.rpb, &a.W.qkv_w, &a.W.qkv_b, &a.W.proj_w, &a.W.proj_b,
                          &a.W.n2_w, &a.W.n2_b, &a.W.fc1_w, &a.W.fc1_b, &a.W.fc2_w, &a.W.fc2_b};
        float **Gp[15] = {&a.G.aff_w, &a.G.aff_b, &a.G.n1_w, &a.G.n1_b, &a.G.rpb, &a.G.qkv_w, &a.G.qkv_b, &a.G.proj_w, &a.G.proj_b,
                          &a.G.n2_w, &a.G.n2_b, &a.G.fc1_w, &a.G.fc1_b, &a.G.fc2_w, &a.G.fc2_b};
        for (int i = 0; i < 15; i++) { *Wp[i] = Wt(b.prefix + "." + kNames[i]); *Gp[i] = Gd(b.prefix + "." + kNames[i]); }
        a.B = B; a.res = b.res; a.ws = b.ws; a.shift = b.shift; a.heads = b.heads; a.C = b.C; a.hidden = h->cfg.mlp_ratio * b.C;
        const size_t M = (size_t)B * b.res * b.res, C = b.C, H = a.hidden;
        a.aff = A.get((size_t)B * 2 * C); a.d_aff = A.get((size_t)B * 2 * C);
        a.x_mod = A.get(M * C); a.xn1 = A.get(M * C); a.att = A.get(M * C); a.x1 = A.get(M * C); a.xn2 = A.get(M * C);
        a.stats1 = A.get(M * 2); a.stats2 = A.get(M * 2); a.qkv = A.get(M * 3 * C); a.pre = A.get(M * H); a.hid = A.get(M * H);
        st.x_out = A.get(M * C); st.d_emb = A.get((size_t)B * NOISE_EMB);
        return true;
    };
    for (int pass = 0; pass < 2; pass++) {
        A.off = 0; A.dry = pass == 0;
        for (int l = 0; l < 8; l++) { enc[l].clear(); dec[l].clear(); }
        Bf.pe = A.get((size_t)B * E); Bf.m0 = A.get((size_t)B * NOISE_EMB); Bf.s0 = A.get((size_t)B * NOISE_EMB);
        Bf.m1 = A.get((size_t)B * NOISE_EMB); Bf.emb = A.get((size_t)B * NOISE_EMB); Bf.d_emb = A.get((size_t)B * NOISE_EMB);
        Bf.pe_demb = A.get((size_t)B * NOISE_EMB);
        Bf.tok = A.get(M0 * Cin); Bf.pe_lin = A.get(M0 * E); Bf.pe_ln = A.get(M0 * E); Bf.pe_stats = A.get(M0 * 2);
        Bf.pe_aff = A.get((size_t)B * 2 * E); Bf.pe_daff = A.get((size_t)B * 2 * E); Bf.x0 = A.get(M0 * E);
        int res = N, C = E;
        for (int l = 0; l < L; l++) {
            for (auto &b : h->down[l]) { Stage st{}; st.bp = &b; fill_block(b, st); enc[l].push_back(st); }
            if (l < L - 1) {
                const size_t M2 = (size_t)B * (res / 2) * (res / 2);
                Bf.mcat[l] = A.get(M2 * 4 * C); Bf.mnrm[l] = A.get(M2 * 4 * C); Bf.mstats[l] = A.get(M2 * 2); Bf.mout[l] = A.get(M2 * 2 * C);
                res /= 2; C *= 2;
            }
            skip_res[l] = res; skip_C[l] = C;
            Bf.d_skip[l] = A.get((size_t)B * res * res * C);
        }
        for (int i = 0; i < L; i++) {
            const int lvl = L - 1 - i;
            if (i > 0) {
                const size_t Mi = (size_t)B * res * res; const int D = 2 * C, Co = D / 4;
                Bf.ucat[i] = A.get(Mi * D); Bf.uy[i] = A.get(Mi * D); Bf.un[i] = A.get(Mi * D); Bf.ustats[i] = A.get(Mi * 2);
                Bf.usc[i] = A.get(Mi * 4 * Co); Bf.upn[i] = A.get(Mi * 4 * Co); Bf.upstats[i] = A.get(Mi * 4 * 2); Bf.uout[i] = A.get(Mi * 4 * Co);
                res *= 2; C /= 2;
            }
            for (auto &b : h->up[i]) { Stage st{}; st.bp = &b; fill_block(b, st); dec[i].push_back(st); }
            (void)lvl;
        }
        Bf.fy = A.get(M0 * E); Bf.fstats = A.get(M0 * 2); Bf.z1 = A.get(M0 * E); Bf.z2 = A.get(M0 * E); Bf.rep = A.get(M0 * E);
        Bf.ha_pre = A.get(M0 * E); Bf.ha = A.get(M0 * E); Bf.oa = A.get(M0 * Ca);
        Bf.pool = A.get((size_t)B * N * E); Bf.hn_pre = A.get((size_t)B * N * E); Bf.hn = A.get((size_t)B * N * E); Bf.on = A.get((size_t)B * N * Cn);
        if (pass == 0) {
            if (!missing.empty()) return fail(h, DSG_ERR_INVALID, "missing tensor '%s'", missing.c_str());
            // scratch for the backward: the widest [M, 4C] tensors of any stage, three of them, + block scratch
            A.cap = A.off;
            // the saved-activation arena and the backward scratch belong to the handle and are kept between calls
            if (A.cap + 16 > h->train_arena_cap) {
                if (h->train_arena) { (void)hipStreamSynchronize(s); (void)hipFree(h->train_arena); }
                h->train_arena = nullptr; h->train_arena_cap = 0;
                HIP_TRY(h, hipMalloc((void **)&h->train_arena, sizeof(float) * (A.cap + 16)));
                h->train_arena_cap = A.cap + 16;
            }
            A.base = h->train_arena;
        }
    }
    // widest scratch tensors (backward): sized for level 0's [M0, 4E]; every level has M C constant up to the 2x of merging
    size_t wide = M0 * (size_t)(4 * E);
    const size_t scr_need = wide * 3 + M0 * (size_t)E * 8 + 4096;
    if (scr_need > h->train_scr_cap) {
        if (h->train_scr) { (void)hipStreamSynchronize(s); (void)hipFree(h->train_scr); }
        h->train_scr = nullptr; h->train_scr_cap = 0;
        if (hipMalloc((void **)&h->train_scr, sizeof(float) * scr_need) != hipSuccess) return fail(h, DSG_ERR_HIP, "out of memory");
        h->train_scr_cap = scr_need;
    }
    float *scr = h->train_scr;
    float *t_mh = scr, *t_m3c = scr + wide, *t_w = scr + 2 * wide, *t_mc = scr + 3 * wide, *t_mc2 = t_mc + M0 * E * 2, *d_x = t_mc2 + M0 * E * 2,
          *d_y = d_x + M0 * E * 2;
    auto lin_fwd = [&](const float *x, const float *Wm, const float *bias, float *y, size_t M, int in, int out) {
        t_gemm(false, true, x, in, Wm, in, bias, y, out, (int)M, out, in, false, s);
    };
    auto lin_bwd = [&](const float *x, const float *Wm, const float *dy, float *dx, float *dW, float *db, size_t M, int in, int out) {
        t_gemm(true, false, dy, out, x, in, nullptr, dW, in, out, in, (int)M, false, s, db);       // dW [out,in] = dy^T x, db = colsum(dy)
        if (dx) t_gemm(false, false, dy, out, Wm, in, nullptr, dx, in, (int)M, in, out, false, s);   // dx = dy W
    };
    bool ok = true;
    // ================================ forward ================================
    launch_noise_pe(c_noise, Bf.pe, B, E, s);
    lin_fwd(Bf.pe, Wt("map_layer0.weight"), Wt("map_layer0.bias"), Bf.m0, B, E, NOISE_EMB);
    t_silu(Bf.m0, nullptr, Bf.s0, (size_t)B * NOISE_EMB, false, s);
    lin_fwd(Bf.s0, Wt("map_layer1.weight"), Wt("map_layer1.bias"), Bf.m1, B, NOISE_EMB, NOISE_EMB);
    t_silu(Bf.m1, nullptr, Bf.emb, (size_t)B * NOISE_EMB, false, s);
    launch_assemble(in_adj, in_node, sc_adj, sc_node, has_sc_dev, flags, Bf.tok, B, N, Ca, Cn, h->cfg.self_condition, Cin, s);
    lin_fwd(Bf.tok, Wt("patch_embed.proj.weight"), Wt("patch_embed.proj.bias"), Bf.pe_lin, M0, Cin, E);
    t_ln_fwd(Bf.pe_lin, Wt("patch_embed.norm.weight"), Wt("patch_embed.norm.bias"), Bf.pe_ln, Bf.pe_stats, (int)M0, E, s);
    // every `affine` linear of the network (PatchEmbed + all blocks) hangs off emb: one grouped launch
    std::vector<Stage *> all_stages;
    for (int l = 0; l < L; l++) for (auto &st : enc[l]) all_stages.push_back(&st);
    for (int i = 0; i < L; i++) for (auto &st : dec[i]) all_stages.push_back(&st);
    const bool grouped = (int)all_stages.size() + 1 <= T_GROUP_MAX;
    if (grouped) {
        TGemmGroup gf;
        gf.p[gf.n++] = TGemmProb{Bf.emb, Wt("patch_embed.affine.weight"), Wt("patch_embed.affine.bias"), Bf.pe_aff, NOISE_EMB, NOISE_EMB, 2 * E, B, 2 * E, NOISE_EMB};
        for (Stage *st : all_stages) {
            st->a.aff_grouped = true;
            gf.p[gf.n++] = TGemmProb{Bf.emb, st->a.W.aff_w, st->a.W.aff_b, st->a.aff, NOISE_EMB, NOISE_EMB, 2 * st->a.C, B, 2 * st->a.C, NOISE_EMB};
        }
        t_gemm_grouped(false, true, false, gf, s);
    } else {
        lin_fwd(Bf.emb, Wt("patch_embed.affine.weight"), Wt("patch_embed.affine.bias"), Bf.pe_aff, B, NOISE_EMB, 2 * E);
    }
    t_modulate(Bf.pe_ln, Bf.pe_aff, nullptr, Bf.x0, nullptr, B, T0, E, false, s);
    const float *x = Bf.x0;
    int res = N, C = E;
    auto run_blocks_fwd = [&](std::vector<Stage> &v) {
        for (auto &st : v) {
            st.x_in = const_cast<float *>(x);
            st.a.x_in = x; st.a.emb = Bf.emb; st.a.x_out = st.x_out; st.a.grad_out = nullptr;
            ok = ok && train_block(st.a, s);
            x = st.x_out;
        }
    };
    for (int l = 0; l < L; l++) {
        run_blocks_fwd(enc[l]);
        if (l < L - 1) {
            const std::string p = "down_layers." + std::to_string(l) + ".downsample";
            const size_t M2 = (size_t)B * (res / 2) * (res / 2);
            t_regroup(x, Bf.mcat[l], B, res, C, true, s);
            t_ln_fwd(Bf.mcat[l], Wt(p + ".norm.weight"), Wt(p + ".norm.bias"), Bf.mnrm[l], Bf.mstats[l], (int)M2, 4 * C, s);
            lin_fwd(Bf.mnrm[l], Wt(p + ".reduction.weight"), nullptr, Bf.mout[l], M2, 4 * C, 2 * C);
            x = Bf.mout[l]; res /= 2; C *= 2;
        }
        skip_ptr[l] = x;
    }
    for (int i = 0; i < L; i++) {
        const int lvl = L - 1 - i;
        if (i > 0) {
            const std::string p = "up_layers." + std::to_string(i) + ".upsample";
            const size_t Mi = (size_t)B * res * res; const int D = 2 * C, Co = D / 4;
            t_concat(x, skip_ptr[lvl], Bf.ucat[i], Mi, C, s);
            lin_fwd(Bf.ucat[i], Wt(p + ".pre_linear.weight"), nullptr, Bf.uy[i], Mi, D, D);
            t_ln_fwd(Bf.uy[i], Wt(p + ".norm.weight"), Wt(p + ".norm.bias"), Bf.un[i], Bf.ustats[i], (int)Mi, D, s);
            t_regroup(Bf.un[i], Bf.usc[i], B, 2 * res, Co, false, s);   // scatter: fine [4 Mi, Co] <- coarse [Mi, 4 Co]
            t_ln_fwd(Bf.usc[i], Wt(p + ".post_norm.weight"), Wt(p + ".post_norm.bias"), Bf.upn[i], Bf.upstats[i], (int)(4 * Mi), Co, s);
            lin_fwd(Bf.upn[i], Wt(p + ".post_linear.weight"), nullptr, Bf.uout[i], 4 * Mi, Co, Co);
            x = Bf.uout[i]; res *= 2; C /= 2;
        }
        run_blocks_fwd(dec[i]);
    }
    const float *xL = x;
    t_ln_fwd(xL, Wt("norm.weight"), Wt("norm.bias"), Bf.fy, Bf.fstats, (int)M0, E, s);
    t_gemm(false, false, Bf.fy, E, Wt("read_out.0.weight"), E, Wt("read_out.0.bias"), Bf.z1, E, (int)M0, E, E, false, s);   // ConvTranspose2d: [in, out]
    lin_fwd(Bf.z1, Wt("read_out.1.weight"), Wt("read_out.1.bias"), Bf.z2, M0, E, E);
    lin_fwd(Bf.z2, Wt("read_out.2.weight"), Wt("read_out.2.bias"), Bf.rep, M0, E, E);
    lin_fwd(Bf.rep, Wt("readout_adj_mlp.fc1.weight"), Wt("readout_adj_mlp.fc1.bias"), Bf.ha_pre, M0, E, E);
    t_gelu(Bf.ha_pre, nullptr, Bf.ha, M0 * E, false, s);
    lin_fwd(Bf.ha, Wt("readout_adj_mlp.fc2.weight"), Wt("readout_adj_mlp.fc2.bias"), Bf.oa, M0, E, Ca);
    t_adj_out(Bf.oa, flags, out_F_adj, nullptr, nullptr, B, N, Ca, false, s);
    launch_pool(Bf.rep, flags, Bf.pool, B, N, E, s);
    lin_fwd(Bf.pool, Wt("readout_node_mlp.fc1.weight"), Wt("readout_node_mlp.fc1.bias"), Bf.hn_pre, (size_t)B * N, E, E);
    t_gelu(Bf.hn_pre, nullptr, Bf.hn, (size_t)B * N * E, false, s);
    lin_fwd(Bf.hn, Wt("readout_node_mlp.fc2.weight"), Wt("readout_node_mlp.fc2.bias"), Bf.on, (size_t)B * N, E, Cn);
    t_rowmask(Bf.on, flags, out_F_node, (size_t)B * N, Cn, s);
    // ================================ backward ================================
    if (bwd && ok && mid) {
        const size_t na = (size_t)B * Ca * N * N, nn = (size_t)B * N * Cn;
        float *mid_buf = train_buf(h, h->train_mid, h->train_mid_cap, na + nn, s);
        if (!mid_buf) ok = false;
        else { ok = (*mid)(out_F_adj, out_F_node, mid_buf, mid_buf + na) == DSG_OK; grad_F_adj = mid_buf; grad_F_node = mid_buf + na; }
    }
    if (bwd && ok) {
        hipError_t e0 = grouped ? hipSuccess : hipMemsetAsync(Bf.d_emb, 0, sizeof(float) * (size_t)B * NOISE_EMB, s);
        for (int l = 0; l < L && e0 == hipSuccess; l++) e0 = hipMemsetAsync(Bf.d_skip[l], 0, sizeof(float) * (size_t)B * skip_res[l] * skip_res[l] * skip_C[l], s);
        ok = ok && e0 == hipSuccess;
        // heads
        float *d_on = t_mc, *d_hn = t_mc2, *d_pool = d_x, *d_rep = d_y;
        t_rowmask(grad_F_node, flags, d_on, (size_t)B * N, Cn, s);
        lin_bwd(Bf.hn, Wt("readout_node_mlp.fc2.weight"), d_on, d_hn, Gd("readout_node_mlp.fc2.weight"), Gd("readout_node_mlp.fc2.bias"), (size_t)B * N, E, Cn);
        t_gelu(Bf.hn_pre, d_hn, d_hn, (size_t)B * N * E, true, s);
        lin_bwd(Bf.pool, Wt("readout_node_mlp.fc1.weight"), d_hn, d_pool, Gd("readout_node_mlp.fc1.weight"), Gd("readout_node_mlp.fc1.bias"), (size_t)B * N, E, E);
        float *d_oa = t_m3c, *d_ha = t_mh;
        t_adj_out(nullptr, flags, nullptr, grad_F_adj, d_oa, B, N, Ca, true, s);
        lin_bwd(Bf.ha, Wt("readout_adj_mlp.fc2.weight"), d_oa, d_ha, Gd("readout_adj_mlp.fc2.weight"), Gd("readout_adj_mlp.fc2.bias"), M0, E, Ca);
        t_gelu(Bf.ha_pre, d_ha, d_ha, M0 * E, true, s);
        lin_bwd(Bf.rep, Wt("readout_adj_mlp.fc1.weight"), d_ha, d_rep, Gd("readout_adj_mlp.fc1.weight"), Gd("readout_adj_mlp.fc1.bias"), M0, E, E);
        t_pool_bwd(d_pool, flags, d_rep, B, N, E, s);
        // read_out.2, .1 (Conv2d [out,in]) and .0 (ConvTranspose2d [in,out]), final norm
        float *d_z2 = t_mh, *d_z1 = t_m3c, *d_fy = t_w;
        lin_bwd(Bf.z2, Wt("read_out.2.weight"), d_rep, d_z2, Gd("read_out.2.weight"), Gd("read_out.2.bias"), M0, E, E);
        lin_bwd(Bf.z1, Wt("read_out.1.weight"), d_z2, d_z1, Gd("read_out.1.weight"), Gd("read_out.1.bias"), M0, E, E);
        t_gemm(true, false, Bf.fy, E, d_z1, E, nullptr, Gd("read_out.0.weight"), E, E, E, (int)M0, false, s);        // dW [in,out] = y^T dz
        t_colsum(d_z1, E, Gd("read_out.0.bias"), (int)M0, E, s);
        t_gemm(false, true, d_z1, E, Wt("read_out.0.weight"), E, nullptr, d_fy, E, (int)M0, E, E, false, s);        // dy = dz W^T
        float *dx = d_x;   // running gradient wrt the current activation x (size up to M0*E*2)
        t_ln_bwd(xL, Wt("norm.weight"), Bf.fstats, d_fy, nullptr, dx, Gd("norm.weight"), Gd("norm.bias"), (int)M0, E, s);
        // blocks / up / down in reverse
        float *dcur = dx, *dnext = d_y;   // ping-pong
        auto run_blocks_bwd = [&](std::vector<Stage> &v) {
            for (int k = (int)v.size() - 1; k >= 0; k--) {
                Stage &st = v[k];
                TrainBlockArgs &a = st.a;
                const size_t M = (size_t)B * a.res * a.res, Cc = a.C, H = a.hidden;
                a.grad_out = dcur; a.grad_in = dnext; a.grad_emb = st.d_emb;
                a.d_x1 = t_w; a.t_mc = t_mc; a.t_mc2 = t_mc2; a.t_m3c = t_m3c; a.t_mh = t_mh;
                (void)M; (void)Cc; (void)H;
                ok = ok && train_block_backward(a, s);
                if (!a.aff_grouped) t_add(Bf.d_emb, st.d_emb, (size_t)B * NOISE_EMB, s);
                std::swap(dcur, dnext);
            }
        };
        res = N; C = E;
        for (int i = L - 1; i >= 0; i--) {
            const int lvl = L - 1 - i;
            run_blocks_bwd(dec[i]);
            if (i > 0) {
                // here res, C are the FINE values (after the breakup); the coarse input had res/2, 2C per source
                const std::string p = "up_layers." + std::to_string(i) + ".upsample";
                const int rc = res / 2, Cc = 2 * C; const size_t Mi = (size_t)B * rc * rc; const int D = 2 * Cc, Co = D / 4;
                float *d_upn = t_mh, *d_usc = t_m3c, *d_un = t_w, *d_uy = t_mh, *d_cat = t_m3c;
                lin_bwd(Bf.upn[i], Wt(p + ".post_linear.weight"), dcur, d_upn, Gd(p + ".post_linear.weight"), nullptr, 4 * Mi, Co, Co);
                t_ln_bwd(Bf.usc[i], Wt(p + ".post_norm.weight"), Bf.upstats[i], d_upn, nullptr, d_usc, Gd(p + ".post_norm.weight"), Gd(p + ".post_norm.bias"),
                         (int)(4 * Mi), Co, s);
                t_regroup(d_usc, d_un, B, res, Co, true, s);   // transpose of the scatter = gather
                t_ln_bwd(Bf.uy[i], Wt(p + ".norm.weight"), Bf.ustats[i], d_un, nullptr, d_uy, Gd(p + ".norm.weight"), Gd(p + ".norm.bias"), (int)Mi, D, s);
                lin_bwd(Bf.ucat[i], Wt(p + ".pre_linear.weight"), d_uy, d_cat, Gd(p + ".pre_linear.weight"), nullptr, Mi, D, D);
                t_split(d_cat, dnext, Bf.d_skip[lvl], Mi, Cc, s);
                std::swap(dcur, dnext);
                res = rc; C = Cc;
            }
        }
        for (int l = L - 1; l >= 0; l--) {
            // the tensor at the end of encoder level l (after its merging, if any) also fed the decoder as skip[l]
            if (l < L - 1 || true) t_add(dcur, Bf.d_skip[l], (size_t)B * skip_res[l] * skip_res[l] * skip_C[l], s);
            if (l < L - 1) {
                const std::string p = "down_layers." + std::to_string(l) + ".downsample";
                const int rf = res * 2, Cf = C / 2; const size_t M2 = (size_t)B * res * res;
                float *d_nrm = t_mh, *d_cat = t_m3c;
                lin_bwd(Bf.mnrm[l], Wt(p + ".reduction.weight"), dcur, d_nrm, Gd(p + ".reduction.weight"), nullptr, M2, 4 * Cf, 2 * Cf);
                t_ln_bwd(Bf.mcat[l], Wt(p + ".norm.weight"), Bf.mstats[l], d_nrm, nullptr, d_cat, Gd(p + ".norm.weight"), Gd(p + ".norm.bias"), (int)M2, 4 * Cf, s);
                t_regroup(d_cat, dnext, B, rf, Cf, false, s);   // transpose of the gather = scatter back to the fine grid
                std::swap(dcur, dnext);
                res = rf; C = Cf;
            }
            run_blocks_bwd(enc[l]);
        }
        // PatchEmbed: modulate <- LN <- 1x1 conv (the inputs are leaves)
        float *d_pe_ln = dnext, *d_pe_lin = t_mh;
        t_modulate(Bf.pe_ln, Bf.pe_aff, dcur, d_pe_ln, Bf.pe_daff, B, T0, E, true, s);
        if (grouped) {
            // the affine linears' own backward, all at once: dWa_z = d_aff_z^T emb, d_ba_z = colsum(d_aff_z), d_emb = sum_z d_aff_z Wa_z
            TGemmGroup gw, gb, ge;
            // (d_emb: every product into its own [B, 512] buffer -- 16 x 2 tiles per product fill the chip, one shared output would
            // leave it to 32 blocks -- then one ordered sum)
            auto add = [&](const float *d_aff, const float *Wa, float *dWa, float *dba, float *d_emb_z, int C2) {
                gw.p[gw.n++] = TGemmProb{d_aff, Bf.emb, nullptr, dWa, C2, NOISE_EMB, NOISE_EMB, C2, NOISE_EMB, B};
                gb.p[gb.n++] = TGemmProb{d_aff, nullptr, nullptr, dba, C2, 0, 0, B, C2, 0};
                ge.p[ge.n++] = TGemmProb{d_aff, Wa, nullptr, d_emb_z, C2, NOISE_EMB, NOISE_EMB, B, NOISE_EMB, C2};
            };
            add(Bf.pe_daff, Wt("patch_embed.affine.weight"), Gd("patch_embed.affine.weight"), Gd("patch_embed.affine.bias"), Bf.pe_demb, 2 * E);
            for (Stage *st : all_stages) add(st->a.d_aff, st->a.W.aff_w, st->a.G.aff_w, st->a.G.aff_b, st->d_emb, 2 * st->a.C);
            t_gemm_grouped(true, false, false, gw, s);
            t_colsum_grouped(gb, s);
            t_gemm_grouped(false, false, false, ge, s);
            t_sum_grouped(ge, Bf.d_emb, B * NOISE_EMB, s);
        } else {
            t_gemm(true, false, Bf.pe_daff, 2 * E, Bf.emb, NOISE_EMB, nullptr, Gd("patch_embed.affine.weight"), NOISE_EMB, 2 * E, NOISE_EMB, B, false, s);
            t_colsum(Bf.pe_daff, 2 * E, Gd("patch_embed.affine.bias"), B, 2 * E, s);
            t_gemm(false, false, Bf.pe_daff, 2 * E, Wt("patch_embed.affine.weight"), NOISE_EMB, nullptr, Bf.d_emb, NOISE_EMB, B, NOISE_EMB, 2 * E, true, s);
        }
        t_ln_bwd(Bf.pe_lin, Wt("patch_embed.norm.weight"), Bf.pe_stats, d_pe_ln, nullptr, d_pe_lin, Gd("patch_embed.norm.weight"), Gd("patch_embed.norm.bias"),
                 (int)M0, E, s);
        lin_bwd(Bf.tok, Wt("patch_embed.proj.weight"), d_pe_lin, nullptr, Gd("patch_embed.proj.weight"), Gd("patch_embed.proj.bias"), M0, Cin, E);
        // noise embedding: emb = silu(map1(silu(map0(pe))))
        float *d_m1 = t_mc, *d_s0 = t_mc2;
        t_silu(Bf.m1, Bf.d_emb, d_m1, (size_t)B * NOISE_EMB, true, s);
        lin_bwd(Bf.s0, Wt("map_layer1.weight"), d_m1, d_s0, Gd("map_layer1.weight"), Gd("map_layer1.bias"), B, NOISE_EMB, NOISE_EMB);
        t_silu(Bf.m0, d_s0, d_s0, (size_t)B * NOISE_EMB, true, s);
        lin_bwd(Bf.pe, Wt("map_layer0.weight"), d_s0, nullptr, Gd("map_layer0.weight"), Gd("map_layer0.bias"), B, E, NOISE_EMB);
    }
    // everything is queued on the caller's stream and every buffer involved outlives the call (caller's or the handle's): no
    // synchronisation here -- the host runs ahead into the optimiser step's launches
    HIP_TRY(h, hipGetLastError());
    if (!ok) return fail(h, DSG_ERR_HIP, "a training kernel failed to launch");
    if (t_scratch_failed(s, true)) return fail(h, DSG_ERR_HIP, "out of memory (training scratch of this stream)");
    return DSG_OK;
}

int dsg_train_grads(dsg_handle h, int32_t B, const float *in_adj, const float *in_node, const uint8_t *flags, const float *c_noise,
                    const float *sc_adj, const float *sc_node, const float *grad_F_adj, const float *grad_F_node, float *out_F_adj,
                    float *out_F_node, int32_t n_params, const char *const *names, float *const *grad_params, void *stream) {
    return train_grads_core(h, B, in_adj, in_node, flags, c_noise, sc_adj, sc_node, grad_F_adj, grad_F_node, nullptr, out_F_adj, out_F_node,
                            n_params, names, grad_params, stream);
}

int dsg_train_step_grads(dsg_handle h, int32_t B, const float *noisy_adj, const float *noisy_node, const uint8_t *flags, const float *sigmas,
                         const float *sc_adj, const float *sc_node, const float *target_adj, const float *target_node,
                         const float *loss_weight, float edge_loss_weight, float node_loss_weight, float iou_loss_weight,
                         int32_t iou_loss_type, float *out_D_adj, float *out_D_node, float *out_loss_adj, float *out_loss_node,
                         int32_t n_params, const char *const *names, float *const *grad_params, void *stream) {
    if (!h || !h->ever_finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (iou_loss_type < DSG_IOU_IOU || iou_loss_type > DSG_IOU_CIOU) return fail(h, DSG_ERR_INVALID, "iou_loss_type %d", iou_loss_type);
    if (B < 1 || !noisy_adj || !noisy_node || !flags || !sigmas || !target_adj || !target_node || !out_D_adj || !out_D_node || !out_loss_adj ||
        !out_loss_node)
        return fail(h, DSG_ERR_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const Dims d = dims_of(h, B);
    const size_t na = (size_t)B * h->Ca * h->N * h->N, nn = (size_t)B * h->N * h->Cn;
    // in_adj | in_node | c_noise | F_adj | F_node | dL/dD (adj, node): handle-owned, kept between calls
    float *buf = train_buf(h, h->train_io, h->train_io_cap, 3 * na + 3 * nn + (size_t)B + 64, s);
    if (!buf) return fail(h, DSG_ERR_HIP, "out of memory");
    float *in_a = buf, *in_n = in_a + na, *cn = in_n + nn, *F_a = cn + ((B + 63) / 64) * 64, *F_n = F_a + na, *tmp = F_n + nn;
    launch_precond_in(CStatePtrs{noisy_adj, noisy_node}, sigmas, StatePtrs{in_a, in_n}, cn, d, s);   // c_in * x, c_noise = ln(sigma)/4
    const TrainMid mid = [&](const float *Fa, const float *Fn, float *dFa, float *dFn) -> int {
        // D = mask(c_skip x + c_out F) (precond.py:101-104); per-sample losses; dL/dD; dL/dF = c_out dL/dD
        launch_precond_out(CStatePtrs{noisy_adj, noisy_node}, CStatePtrs{Fa, Fn}, sigmas, flags, StatePtrs{out_D_adj, out_D_node},
                           StatePtrs{nullptr, nullptr}, d, s);
        launch_rainbow_loss(CStatePtrs{out_D_adj, out_D_node}, CStatePtrs{target_adj, target_node}, flags, loss_weight, edge_loss_weight,
                            node_loss_weight, iou_loss_weight, iou_loss_type, out_loss_adj, out_loss_node, d, s);
        launch_rainbow_loss_backward(CStatePtrs{out_D_adj, out_D_node}, CStatePtrs{target_adj, target_node}, flags, loss_weight,
                                     edge_loss_weight, node_loss_weight, iou_loss_weight, iou_loss_type, sigmas, StatePtrs{tmp, tmp + na},
                                     StatePtrs{dFa, dFn}, d, s);
        return hipGetLastError() == hipSuccess ? DSG_OK : DSG_ERR_HIP;
    };
    const bool want_grads = names && grad_params && n_params > 0;
    int rc = train_grads_core(h, B, in_a, in_n, flags, cn, sc_adj, sc_node, nullptr, nullptr, want_grads ? &mid : nullptr, F_a, F_n, n_params, names,
                              grad_params, stream);
    if (rc == DSG_OK && !want_grads) rc = mid(F_a, F_n, in_a, in_n);   // forward only: still report D and the losses (gradients discarded)
    return rc;
}

int dsg_train_self_cond(dsg_handle h, int32_t B, const float *noisy_adj, const float *noisy_node, const uint8_t *flags, const float *sigmas,
                        float *out_sc_adj, float *out_sc_node, void *stream) {
    if (!h || !h->ever_finalized) return fail(h, DSG_ERR_STATE, "weights not finalized");
    if (B < 1 || !noisy_adj || !noisy_node || !flags || !sigmas || !out_sc_adj || !out_sc_node) return fail(h, DSG_ERR_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const Dims d = dims_of(h, B);
    const size_t na = (size_t)B * h->Ca * h->N * h->N, nn = (size_t)B * h->N * h->Cn;
    float *buf = train_buf(h, h->train_io, h->train_io_cap, 3 * na + 3 * nn + (size_t)B + 64, s);
    if (!buf) return fail(h, DSG_ERR_HIP, "out of memory");
    float *in_a = buf, *in_n = in_a + na, *cn = in_n + nn, *F_a = cn + ((B + 63) / 64) * 64, *F_n = F_a + na;
    launch_precond_in(CStatePtrs{noisy_adj, noisy_node}, sigmas, StatePtrs{in_a, in_n}, cn, d, s);
    const int rc = train_grads_core(h, B, in_a, in_n, flags, cn, nullptr, nullptr, nullptr, nullptr, nullptr, F_a, F_n, 0, nullptr, nullptr, stream);
    if (rc != DSG_OK) return rc;
    launch_precond_out(CStatePtrs{noisy_adj, noisy_node}, CStatePtrs{F_a, F_n}, sigmas, flags, StatePtrs{out_sc_adj, out_sc_node},
                       StatePtrs{nullptr, nullptr}, d, s);
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}

int dsg_train_bind_params(dsg_handle h, int32_t n_params, const char *const *names, const float *const *params) {
    if (!h || n_params < 0 || (n_params > 0 && (!names || !params))) return fail(h, DSG_ERR_INVALID, "null argument");
    h->train_params.clear();
    for (int k = 0; k < n_params; k++) {
        if (!names[k] || !params[k]) { h->train_params.clear(); return fail(h, DSG_ERR_INVALID, "null entry %d", k); }
        const std::string key = strip_prefix(names[k]);
        bool known = false;
        for (auto &sp : h->specs) if (sp.key == key) { known = !sp.is_index && !sp.is_mask; break; }
        if (!known) { h->train_params.clear(); return fail(h, DSG_ERR_WEIGHTS, "unexpected key '%s'", names[k]); }
        h->train_params[key] = params[k];
    }
    return DSG_OK;
}

int dsg_adam_step(int32_t n_tensors, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                  const int64_t *numel, int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                  float *out_total_norm, void *stream) {
    if (n_tensors < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || step < 1) return DSG_ERR_INVALID;
    for (int i = 0; i < n_tensors; i++) if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 1) return DSG_ERR_INVALID;
    return t_adam_step(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, step, lr, beta1, beta2, eps, weight_decay, max_grad_norm,
                       out_total_norm, (hipStream_t)stream) ? DSG_OK : DSG_ERR_HIP;
}

int dsg_ema_update(int32_t n_tensors, float *const *ema, const float *const *params, const int64_t *numel, float decay, void *stream) {
    if (n_tensors < 1 || !ema || !params || !numel) return DSG_ERR_INVALID;
    for (int i = 0; i < n_tensors; i++) if (!ema[i] || !params[i] || numel[i] < 1) return DSG_ERR_INVALID;
    return t_ema_update(n_tensors, ema, params, numel, decay, (hipStream_t)stream) ? DSG_OK : DSG_ERR_HIP;
}

double dsg_profile_clock_ghz(dsg_handle h) { return h ? h->prof_clock_ghz : 0.0; }

int dsg_decode(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags, int32_t edge_encoding,
               int32_t node_encoding, int32_t n_adj_type, int32_t n_node_type, int32_t node_chans, int32_t *out_adj, int32_t *out_node,
               float *out_bbox, void *stream) {
    if (!h || B < 1 || !adj || !node || !flags || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null tensor");
    if (edge_encoding < DSG_ENC_BITS || edge_encoding > DSG_ENC_DDPM || node_encoding < DSG_ENC_BITS || node_encoding > DSG_ENC_DDPM)
        return fail(h, DSG_ERR_INVALID, "encoding must be DSG_ENC_BITS, DSG_ENC_ONE_HOT or DSG_ENC_DDPM");
    if (n_adj_type < 2 || n_node_type < 2) return fail(h, DSG_ERR_INVALID, "fewer than two types");   // attribute_code.py:132
    if (node_chans < 1 || node_chans > h->Cn || (out_bbox && h->Cn < node_chans + 4)) return fail(h, DSG_ERR_INVALID, "node_chans");
    // the channel counts an encoding implies (sg_utils.py:348-409): one channel per type, or a single one
    if ((edge_encoding == DSG_ENC_ONE_HOT && h->Ca != n_adj_type) || (edge_encoding == DSG_ENC_DDPM && h->Ca != 1) ||
        (edge_encoding == DSG_ENC_BITS && (h->Ca > 30 || (1 << h->Ca) < n_adj_type)))
        return fail(h, DSG_ERR_INVALID, "edge encoding %d with %d types does not fit %d adjacency channels", edge_encoding, n_adj_type, h->Ca);
    if ((node_encoding == DSG_ENC_ONE_HOT && node_chans != n_node_type) || (node_encoding == DSG_ENC_DDPM && node_chans != 1) ||
        (node_encoding == DSG_ENC_BITS && (node_chans > 30 || (1 << node_chans) < n_node_type)))
        return fail(h, DSG_ERR_INVALID, "node encoding %d with %d types does not fit %d attribute channels", node_encoding, n_node_type, node_chans);
    launch_decode(adj, node, flags, edge_encoding, node_encoding, n_adj_type, n_node_type, node_chans, out_adj, out_node, out_bbox,
                  dims_of(h, B), (hipStream_t)stream);
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}

int dsg_decode_bits(dsg_handle h, int32_t B, const float *adj, const float *node, const uint8_t *flags, int32_t n_adj_type,
                    int32_t n_node_type, int32_t node_bits, int32_t *out_adj, int32_t *out_node, float *out_bbox, void *stream) {
    if (!h || B < 1 || !adj || !node || !flags || !out_adj || !out_node) return fail(h, DSG_ERR_INVALID, "null tensor");
    if (node_bits < 1 || node_bits > h->Cn || (out_bbox && h->Cn < node_bits + 4)) return fail(h, DSG_ERR_INVALID, "node_bits");
    launch_decode(adj, node, flags, DSG_ENC_BITS, DSG_ENC_BITS, n_adj_type, n_node_type, node_bits, out_adj, out_node, out_bbox, dims_of(h, B),
                  (hipStream_t)stream);
    HIP_TRY(h, hipGetLastError());
    return DSG_OK;
}


}  // extern "C"
