// gemm_bench.cpp -- micro-benchmark of libdsg's kernels on random data (dev tool; not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.cpp diffusesg_amd/csrc/kernels.o -o gpurun_out/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include "../diffusesg_amd/csrc/kernels.h"
using namespace dsg;
#ifdef DSG_CLOCK_DIAG
namespace dsg { extern __device__ unsigned long long *g_diag_buf; }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static float *dev_rand(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n); std::mt19937 g(seed); std::normal_distribution<float> d(0.f, scale);
    for (auto &v : h) v = d(g);
    float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
int main(int argc, char **argv) {
    struct S { int M, N, K, ln, act, res; };
    std::vector<S> shapes = {
        {4096, 1536, 1536, 0, 0, 0}, {16384, 384, 1536, 0, 0, 1}, {16384, 1536, 384, 1, 1, 0}, {16384, 1152, 384, 1, 0, 0},
        {65536, 768, 192, 1, 1, 0}, {65536, 192, 768, 0, 0, 1}, {262144, 384, 96, 1, 1, 0}, {262144, 96, 384, 0, 0, 1},
        {262144, 288, 96, 1, 0, 0}, {262144, 96, 96, 0, 0, 1}, {4096, 3072, 768, 1, 1, 0}, {4096, 768, 3072, 0, 0, 1},
    };
    int only = argc > 1 ? atoi(argv[1]) : -1, iters = argc > 2 ? atoi(argv[2]) : 20;
    hipStream_t s; CK(hipStreamCreate(&s));
    if (!gelu_table()) return 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t i = 0; i < shapes.size(); i++) {
        if (only >= 0 && (int)i != only) continue;
        S sh = shapes[i];
        float *A = dev_rand((size_t)sh.M * sh.K, 1.f, 1), *W = dev_rand((size_t)sh.N * sh.K, 0.05f, 2);
        float *C = dev_rand((size_t)sh.M * sh.N, 1.f, 3), *bias = dev_rand(sh.N, 0.1f, 4);
        float *stats = dev_rand((size_t)sh.M * 2, 0.f, 5), *gam = dev_rand(sh.K, 1.f, 6), *bet = dev_rand(sh.K, 1.f, 7);
        float *R = sh.res ? dev_rand((size_t)sh.M * sh.N, 1.f, 8) : nullptr;
        GemmArgs g; g.A = A; g.lda = sh.K; g.K1 = sh.K; g.K = sh.K; g.M = sh.M; g.N = sh.N; g.W = W; g.bias = bias;
        if (sh.ln) { g.ln_stats = stats; }
        g.act = sh.act; if (sh.res) { g.res = R; g.ldres = sh.N; } g.C = C; g.ldc = sh.N;
        for (int w = 0; w < 3; w++) launch_gemm(g, s);
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < iters; w++) launch_gemm(g, s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        printf("gemm M=%6d N=%5d K=%5d ln=%d act=%d res=%d : %8.1f us  %6.1f TF", sh.M, sh.N, sh.K, sh.ln, sh.act, sh.res, ms * 1e3,
               2.0 * sh.M * sh.N * sh.K / (ms * 1e-3) / 1e12);
#ifdef DSG_CLOCK_DIAG
        {
            const int nb = ((sh.M + 127) / 128 + 7) / 8 * 8 * ((sh.N + 95) / 96);
            unsigned long long *dbuf; CK(hipMalloc(&dbuf, sizeof(unsigned long long) * 2 * nb)); CK(hipMemset(dbuf, 0, sizeof(unsigned long long) * 2 * nb));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(dsg::g_diag_buf), &dbuf, sizeof(dbuf)));
            for (int w = 0; w < 10; w++) launch_gemm(g, s);
            CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> hb(2 * nb); CK(hipMemcpy(hb.data(), dbuf, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
            std::vector<double> clk, dur; for (int b = 0; b < nb; b++) if (hb[2*b+1]) { clk.push_back(100e6 * hb[2*b] / hb[2*b+1]); dur.push_back(hb[2*b]); }
            std::sort(clk.begin(), clk.end()); std::sort(dur.begin(), dur.end());
            if (!clk.empty()) printf("  | clock median %.3f GHz, block main-loop cycles median %.0f (ideal %d)", clk[clk.size()/2] / 1e9, dur[dur.size()/2], sh.K / 32 * 6144);
            unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(dsg::g_diag_buf), &nul, sizeof(nul))); (void)hipFree(dbuf);
        }
#endif
        printf("\n");
        for (float *q : {A, W, C, bias, stats, gam, bet, R}) if (q) (void)hipFree(q);
    }
    return 0;
}
