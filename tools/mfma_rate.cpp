// mfma_rate.cpp -- dev microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate at 1 vs 2 waves per SIMD, with and without
// interleaved LDS reads / VALU work.  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.cpp -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0: pure MFMA, 1: + 4 ds_read_b128 per 12 MFMA, 2: + ds reads + 24 VALU per 12 MFMA
__global__ __launch_bounds__(256, 2) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x16 acc[3];
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b[3] = {{1.f, 1.f, 1.f, 1.f}, {2.f, 2.f, 2.f, 2.f}, {3.f, 3.f, 3.f, 3.f}};
    float v0 = tid, v1 = 1.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE >= 1) {
            const float *p = lds + ((tid * 36 + it * 4) & 8188 & ~3);
            a = *reinterpret_cast<const f32x4 *>(p);
            b[0] = *reinterpret_cast<const f32x4 *>(p + 1152);
            b[1] = *reinterpret_cast<const f32x4 *>(p + 2304);
            b[2] = *reinterpret_cast<const f32x4 *>(p + 3456);
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j][t], acc[j], 0, 0, 0);
                if (MODE >= 2) { v0 = v0 * 1.0001f + v1; v1 = v1 * 0.9999f + v0; }
            }
        if (MODE >= 2) {
#pragma unroll
            for (int q = 0; q < 12; q++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x3f6, 3, 0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1;
    for (int j = 0; j < 3; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> int run(int blocks_per_cu, const char *name) {
    const int nb = 256 * blocks_per_cu, iters = 4000;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 256 * 4)); CK(hipMalloc(&cyc, nb * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double mfma_per_wave = 12.0 * iters, flops = (double)nb * 4 * mfma_per_wave * 4096;
    printf("%-28s blocks/CU %d : %7.1f us  %6.1f TF | cycles/MFMA per SIMD %.1f (median wave), clock %.2f GHz\n", name, blocks_per_cu, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, (double)h[nb / 2] / (mfma_per_wave * blocks_per_cu), h[nb / 2] / (ms * 1e-3) / 1e9);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}
int main() {
    for (int bpc = 1; bpc <= 2; bpc++) {
        run<0>(bpc, "pure MFMA");
        run<1>(bpc, "MFMA + 4 ds_read_b128/12");
        run<2>(bpc, "MFMA + ds_read + 24 VALU/12");
    }
    return 0;
}
