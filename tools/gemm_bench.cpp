// gemm_bench.cpp -- micro-benchmark of libdsg's kernels on random data (dev tool; not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_bench.cpp -o /tmp/gemm_bench.o &&
//        hipcc --offload-arch=gfx950 /tmp/gemm_bench.o diffusesg_amd/csrc/kernels.o diffusesg_amd/csrc/kernels_lp.o -o tools/bin/gemm_bench
// usage: gemm_bench [shape-index|-1] [iters] [mode: 0 fp32 MFMA, 1 bf16 MFMA, 2 split-bf16]; GB_FULLDIFF=1 adds a whole-matrix
//        comparison of mode 2 against the fp32 kernel (how the packed-f32 / bf16-MFMA hazard was found)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
#include "../diffusesg_amd/csrc/kernels.h"
using namespace dsg;
#ifdef DSG_CLOCK_DIAG
namespace dsg { extern __device__ unsigned long long *g_diag_buf; }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static float *dev_rand(size_t n, float scale, unsigned seed, std::vector<float> *keep = nullptr) {
    std::vector<float> h(n); std::mt19937 g(seed); std::normal_distribution<float> d(0.f, scale);
    for (auto &v : h) v = d(g);
    float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    if (keep) *keep = std::move(h);
    return p;
}
int main(int argc, char **argv) {
    struct S { int M, N, K, ln, act, res; };
    std::vector<S> shapes = {
        {4096, 1536, 1536, 0, 0, 0}, {16384, 384, 1536, 0, 0, 1}, {16384, 1536, 384, 1, 1, 0}, {16384, 1152, 384, 1, 0, 0},
        {65536, 768, 192, 1, 1, 0}, {65536, 192, 768, 0, 0, 1}, {262144, 384, 96, 1, 1, 0}, {262144, 96, 384, 0, 0, 1},
        {262144, 288, 96, 1, 0, 0}, {262144, 96, 96, 0, 0, 1}, {4096, 3072, 768, 1, 1, 0}, {4096, 768, 3072, 0, 0, 1},
    };
    int only = argc > 1 ? atoi(argv[1]) : -1, iters = argc > 2 ? atoi(argv[2]) : 20;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;   // 0 fp32 MFMA, 1 bf16 MFMA, 2 split-bf16 (3 planes, 6 products)
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s; CK(hipStreamCreate(&s));
    if (!gelu_table()) return 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t i = 0; i < shapes.size(); i++) {
        if (only >= 0 && (int)i != only) continue;
        S sh = shapes[i];
        std::vector<float> hA, hW, hb, hR;
        float *A = dev_rand((size_t)sh.M * sh.K, 1.f, 1, &hA), *W = dev_rand((size_t)sh.N * sh.K, 0.05f, 2, &hW);
        float *C = dev_rand((size_t)sh.M * sh.N, 1.f, 3), *bias = dev_rand(sh.N, 0.1f, 4, &hb);
        float *stats = dev_rand((size_t)sh.M * 2, 0.f, 5), *gam = dev_rand(sh.K, 1.f, 6), *bet = dev_rand(sh.K, 1.f, 7);
        {   // mean 0.1, rstd 0.9 for every row
            std::vector<float> hs((size_t)sh.M * 2);
            for (int m = 0; m < sh.M; m++) { hs[2 * m] = 0.1f; hs[2 * m + 1] = 0.9f; }
            CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        }
        float *R = sh.res ? dev_rand((size_t)sh.M * sh.N, 1.f, 8, &hR) : nullptr;
        void *Wlp = nullptr;
        if (mode == 1) { CK(hipMalloc(&Wlp, (size_t)sh.N * sh.K * 2)); launch_f32_to_bf16(W, Wlp, (size_t)sh.N * sh.K, s); }
        if (mode == 2) { CK(hipMalloc(&Wlp, (size_t)sh.N * sh.K * 6)); launch_f32_split3(W, Wlp, (size_t)sh.N * sh.K, s); }
        GemmArgs g; g.A = A; g.lda = sh.K; g.K1 = sh.K; g.K = sh.K; g.M = sh.M; g.N = sh.N; g.W = W; g.bias = bias;
        if (sh.ln) { g.ln_stats = stats; }
        if (mode == 1) g.Wb = Wlp;
        if (mode == 2) g.Ws3 = Wlp;
        g.act = sh.act; if (sh.res) { g.res = R; g.ldres = sh.N; } g.C = C; g.ldc = sh.N;
        for (int w = 0; w < 3; w++) launch_gemm(g, s);
        CK(hipGetLastError()); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < iters; w++) launch_gemm(g, s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        printf("gemm M=%6d N=%5d K=%5d ln=%d act=%d res=%d : %8.1f us  %6.1f TF", sh.M, sh.N, sh.K, sh.ln, sh.act, sh.res, ms * 1e3,
               2.0 * sh.M * sh.N * sh.K / (ms * 1e-3) / 1e12);
#ifdef DSG_CLOCK_DIAG
        {
            const int tmr = mode == 2 ? 256 : 128;
            const int nb = ((sh.M + tmr - 1) / tmr + 7) / 8 * 8 * ((sh.N + 95) / 96);
            unsigned long long *dbuf; CK(hipMalloc(&dbuf, sizeof(unsigned long long) * 2 * nb)); CK(hipMemset(dbuf, 0, sizeof(unsigned long long) * 2 * nb));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(dsg::g_diag_buf), &dbuf, sizeof(dbuf)));
            for (int w = 0; w < 10; w++) launch_gemm(g, s);
            CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> hb(2 * nb); CK(hipMemcpy(hb.data(), dbuf, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
            std::vector<double> clk, dur; for (int b = 0; b < nb; b++) if (hb[2*b+1]) { clk.push_back(100e6 * hb[2*b] / hb[2*b+1]); dur.push_back(hb[2*b]); }
            std::sort(clk.begin(), clk.end()); std::sort(dur.begin(), dur.end());
            if (!clk.empty()) printf("  | clock median %.3f GHz, block main-loop cycles median %.0f p10 %.0f p90 %.0f (MFMA-only %d)", clk[clk.size()/2] / 1e9, dur[dur.size()/2], dur[dur.size()/10], dur[dur.size()*9/10], mode == 2 ? sh.K / 32 * 2304 : sh.K / 32 * 6144);
            unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(dsg::g_diag_buf), &nul, sizeof(nul))); (void)hipFree(dbuf);
        }
#endif
        {   // accuracy on a sample of outputs against fp64
            std::vector<float> hC((size_t)sh.M * sh.N);
            CK(hipStreamSynchronize(s)); CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            std::mt19937 rg(99); double maxe = 0, sumsq = 0; const int ns = 20000; int nbad = 0;
            for (int t = 0; t < ns; t++) {
                const int m = rg() % sh.M, n = rg() % sh.N;
                double acc = hb[n];
                for (int k = 0; k < sh.K; k++) {
                    float a = hA[(size_t)m * sh.K + k];
                    if (sh.ln) a = fmaf(a, 0.9f, -0.1f * 0.9f);
                    acc += (double)a * (double)hW[(size_t)n * sh.K + k];
                }
                if (sh.act == 1) acc = 0.5 * acc * (1.0 + erf(acc / sqrt(2.0)));
                if (sh.res) acc += hR[(size_t)m * sh.N + n];
                const double e = fabs(acc - (double)hC[(size_t)m * sh.N + n]);
                if (e > 1e-3 && mode != 1 && nbad++ < 4) printf("\n   bad (m=%d n=%d) ref %.6f got %.6f", m, n, acc, hC[(size_t)m * sh.N + n]);
                maxe = std::max(maxe, e); sumsq += acc * acc;
            }
            printf("  | max err %.3g (rms out %.3g)", maxe, sqrt(sumsq / ns));
        }
        if (getenv("GB_DUMP")) {   // raw dump of C for offline whole-matrix comparisons between kernel variants
            std::vector<float> hd((size_t)sh.M * sh.N);
            CK(hipMemcpy(hd.data(), C, hd.size() * 4, hipMemcpyDeviceToHost));
            char fn[256]; snprintf(fn, sizeof(fn), "%s_%zu.bin", getenv("GB_DUMP"), i);
            FILE *f = fopen(fn, "wb"); if (f) { fwrite(hd.data(), 4, hd.size(), f); fclose(f); }
        }
        if (mode == 2 && getenv("GB_FULLDIFF")) {   // whole-matrix diff against the fp32 kernel
            std::vector<float> h2((size_t)sh.M * sh.N), h0((size_t)sh.M * sh.N);
            CK(hipMemcpy(h2.data(), C, h2.size() * 4, hipMemcpyDeviceToHost));
            GemmArgs g0 = g; g0.Ws3 = nullptr; launch_gemm(g0, s); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h0.data(), C, h0.size() * 4, hipMemcpyDeviceToHost));
            size_t nb = 0; int shown = 0;
            for (int m32 = 0; m32 < sh.M; m32 += 32) {   // one row block of 32 at a time
                unsigned mask = 0; int nmin = 1 << 30, nmax = -1, cnt = 0; double esum = 0;
                for (int r = 0; r < 32 && m32 + r < sh.M; r++)
                    for (int n = 0; n < sh.N; n++) {
                        const float d = fabsf(h0[(size_t)(m32 + r) * sh.N + n] - h2[(size_t)(m32 + r) * sh.N + n]);
                        if (d > 1e-3f) { mask |= 1u << r; nmin = std::min(nmin, n); nmax = std::max(nmax, n); cnt++; esum += d; }
                    }
                if (!cnt) continue;
                nb += cnt;
                if (shown++ < 30) printf("\n   bad block rows %d..%d (256-tile %d, wave %d, rb %d): row mask %08x, n %d..%d, %d entries, mean |err| %.4f",
                                         m32, m32 + 31, m32 / 256, (m32 % 256) / 64, (m32 % 64) / 32, mask, nmin, nmax, cnt, esum / cnt);
            }
            printf("\n   total bad %zu of %zu", nb, h0.size());
        }
#ifdef DSG_PHASE_DIAG
        if (mode == 0) {   // per-block phase stamps (100 MHz): entry -> first barrier -> main loop end -> stores acknowledged
            const int nb = ((sh.M + 127) / 128 + 7) / 8 * 8 * ((sh.N + 95) / 96);
            unsigned long long *pb; CK(hipMalloc(&pb, 32ull * nb)); CK(hipMemset(pb, 0, 32ull * nb));
            GemmArgs gp = g; gp.prof = pb; launch_gemm(gp, s); CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> hp(4ull * nb); CK(hipMemcpy(hp.data(), pb, 32ull * nb, hipMemcpyDeviceToHost));
            std::vector<double> a, b, c, d;
            for (int q = 0; q < nb; q++) if (hp[4*q+2]) { a.push_back(hp[4*q+1]/100.0); b.push_back(hp[4*q+2]/100.0); c.push_back(hp[4*q]/100.0); d.push_back(hp[4*q+3]/100.0); }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end()); std::sort(d.begin(), d.end());
            if (!a.empty()) printf("\n   phases (us, median/p90): prologue %.2f/%.2f  main loop %.2f/%.2f  epilogue issue %.2f/%.2f  store-ack wait %.2f/%.2f ; %zu blocks", a[a.size()/2], a[a.size()*9/10], b[b.size()/2], b[b.size()*9/10], c[c.size()/2], c[c.size()*9/10], d[d.size()/2], d[d.size()*9/10], a.size());
            (void)hipFree(pb);
        }
#endif
        printf("\n");
        if (Wlp) (void)hipFree(Wlp);
        for (float *q : {A, W, C, bias, stats, gam, bet, R}) if (q) (void)hipFree(q);
    }
    return 0;
}
