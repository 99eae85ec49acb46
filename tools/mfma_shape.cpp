// mfma_shape.cpp -- does the chip hold a different clock on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32?
// (MI355X_MICROARCH.md "DVFS give-back" item 7: for bf16 the 16x16x32 shape ran 1.15x the FLOP/s of 32x32x16 at equal cycles per
// FLOP because the chip held a higher clock.)  Bare MFMA loops on RANDOM operands held in registers (a pool of 8 A and 8 B values
// cycled per instruction so that consecutive MFMAs see different operands), one and two waves per SIMD, same FLOPs per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.cpp -o tools/bin/mfma_shape && tools/bin/mfma_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SHAPE>   // 0: 32x32x2 (3 accumulators of 16 regs), 1: 16x16x4 (12 accumulators of 4 regs): 48 accumulator registers either way
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ rnd, float *out, unsigned long long *cyc, int iters) {
    const int tid = threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = rnd[(blockIdx.x * 256 + tid) * 16 + i]; b[i] = rnd[(blockIdx.x * 256 + tid) * 16 + 8 + i]; }
    f32x16 acc32[3];
    f32x4 acc16[12];
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) acc32[j][r] = 0.f;
    for (int j = 0; j < 12; j++) for (int r = 0; r < 4; r++) acc16[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (SHAPE == 0) {
#pragma unroll
                for (int j = 0; j < 3; j++) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + j) & 7], acc32[j], 0, 0, 0);   // 3 x 4096 FLOP
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) acc16[j + 6 * (u & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + j) & 7], acc16[j + 6 * (u & 1)], 0, 0, 0);   // 6 x 2048 FLOP
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) s += acc32[j][r];
    for (int j = 0; j < 12; j++) for (int r = 0; r < 4; r++) s += acc16[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

// What pulls the clock down in the real GEMM?  The same 32x32x2 loop with the GEMM's memory-side activity added step by step, all on
// random data: MODE 1 = fragments re-read from LDS (4 ds_read_b128 per 12 MFMA, as gemm4_f32_kernel), MODE 2 = + the tile staging
// stores (7 ds_write_b128 per 48 MFMA and thread), MODE 3 = + the global loads that feed them (7 buffer-sized 16-B loads per 48 MFMA
// and thread from an L2-resident array), MODE 4 = two of those seven loads stream a 2 GB buffer instead (unique addresses per block
// and iteration: ~1.3 TB/s of real HBM reads, the GEMM class's measured average), MODE 5 = + one 16-B store per thread and
// iteration to a 1 GB buffer (~0.65 TB/s of HBM writes).
template <int MODE>
__global__ __launch_bounds__(256, 2) void kg(const float *__restrict__ rnd, float *out, unsigned long long *cyc, int iters,
                                             const f32x4 *__restrict__ big = nullptr, f32x4 *__restrict__ sink = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 224 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 224 * 36; i += 256) lds[i] = rnd[(blockIdx.x * 4096 + i) & 0xfffff];
    __syncthreads();
    f32x16 acc[3];
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const int a_off = (wave * 32 + (lane & 31)) * 36 + 4 * (lane >> 5), w_off = 128 * 36 + (lane & 31) * 36 + 4 * (lane >> 5);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(rnd);
    f32x4 st[7];
    for (int p = 0; p < 7; p++) st[p] = g4[(tid + 256 * p) & 0x3ffff];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        const int buf = (it & 1) * 224 * 36;
        if (MODE >= 3) {
#pragma unroll
            for (int p = 0; p < 7; p++) st[p] = g4[((it * 1792 + tid + 256 * p) & 0x3ffff)];   // 4 MB window: L2 resident
        }
        if (MODE >= 4) {
            const size_t base = ((size_t)it * gridDim.x + blockIdx.x) * 512;   // 8 KB per block and iteration, never re-read
#pragma unroll
            for (int p = 0; p < 2; p++) st[p] = big[(base + tid + 256 * p) & 0x7ffffff];   // 2 GB of f32x4
        }
        if (MODE >= 5) sink[(((size_t)it * gridDim.x + blockIdx.x) * 256 + tid) & 0x3ffffff] = st[6];   // 1 GB
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(lds + buf + a_off + 8 * s4);
            f32x4 b[3];
#pragma unroll
            for (int j = 0; j < 3; j++) b[j] = *reinterpret_cast<const f32x4 *>(lds + buf + w_off + 32 * j * 36 + 8 * s4);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 3; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j][t], acc[j], 0, 0, 0);
        }
        if (MODE >= 2) {
            float *dst = lds + (224 * 36 - buf) + ((tid >> 3) * 36 + 4 * (tid & 7));
#pragma unroll
            for (int p = 0; p < 7; p++) *reinterpret_cast<f32x4 *>(dst + 32 * p * 36) = st[p];
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
int rung(const char *name, const float *rnd, int iters, const f32x4 *big = nullptr, f32x4 *sink = nullptr) {
    const int blocks = 512;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)blocks * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 60; w++) hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, rnd, out, cyc, iters, big, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, rnd, out, cyc, iters, big, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> clk, cy;
    for (int b = 0; b < blocks; b++) if (h[2 * b + 1]) { clk.push_back(0.1 * (double)h[2 * b] / (double)h[2 * b + 1]); cy.push_back((double)h[2 * b] / ((double)iters * 48.0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cy.begin(), cy.end());
    const double flops = (double)blocks * 4 * iters * 48 * 4096.0;
    printf("%-44s 2 blocks/CU: %8.1f us  %6.1f TFLOP/s  clock median %.3f GHz  cycles/MFMA %.1f\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12,
           clk[clk.size() / 2], cy[cy.size() / 2]);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

template <int SHAPE>
int run(const char *name, int blocks_per_cu, const float *rnd, int iters) {
    const int blocks = 256 * blocks_per_cu;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)blocks * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 40; w++) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, rnd, out, cyc, iters);   // ~2 s of load first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, rnd, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < blocks; b++) if (h[2 * b + 1]) clk.push_back(0.1 * (double)h[2 * b] / (double)h[2 * b + 1]);
    std::sort(clk.begin(), clk.end());
    const double flops = (double)blocks * 4 * iters * 8 * 3 * 4096.0;   // per wave and iteration: 8 x 3 x 4096 (= 8 x 6 x 2048)
    printf("%-28s blocks/CU %d: %8.1f us  %6.1f TFLOP/s  clock median %.3f GHz\n", name, blocks_per_cu, ms * 1e3, flops / (ms * 1e-3) / 1e12, clk[clk.size() / 2]);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

int main() {
    const size_t n = (size_t)512 * 256 * 16;   // 8 MB
    std::vector<float> h(n); std::mt19937 g(7); std::normal_distribution<float> d(0.f, 1.f);
    for (auto &v : h) v = d(g);
    float *rnd; CK(hipMalloc(&rnd, n * 4)); CK(hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int bpc = 1; bpc <= 2; bpc++) {
        if (run<0>("v_mfma_f32_32x32x2_f32", bpc, rnd, iters)) return 1;
        if (run<1>("v_mfma_f32_16x16x4_f32", bpc, rnd, iters)) return 1;
    }
    if (rung<1>("32x32x2 + fragment reads from LDS", rnd, 3000)) return 1;
    if (rung<2>("  + tile staging stores + barrier", rnd, 3000)) return 1;
    if (rung<3>("  + global (L2) loads for the staging", rnd, 3000)) return 1;
    f32x4 *big, *sink;
    CK(hipMalloc(&big, (size_t)2 << 30)); CK(hipMalloc(&sink, (size_t)1 << 30));
    CK(hipMemset(big, 0x3c, (size_t)2 << 30)); CK(hipMemset(sink, 0, (size_t)1 << 30));
    if (rung<4>("  + 2 of 7 loads streaming HBM (~1.3 TB/s)", rnd, 3000, big, sink)) return 1;
    if (rung<5>("  + HBM stores (~0.65 TB/s)", rnd, 3000, big, sink)) return 1;
    return 0;
}
