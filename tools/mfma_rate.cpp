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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 32x32x16 MFMA with NV independent VALU FMAs per MFMA and NL ds_read_b128 per 12 MFMAs
template <int NV, int NL>
__global__ __launch_bounds__(256, 2) void kb(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x16 acc[6];
    for (int j = 0; j < 6; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    f32x4 af = {1.f, 2.f, 3.f, 4.f}, bf[3] = {{1.f, 1.f, 1.f, 1.f}, {2.f, 2.f, 2.f, 2.f}, {3.f, 3.f, 3.f, 3.f}};
    float v[8]; for (int q = 0; q < 8; q++) v[q] = tid + q;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (NL) {
            const float *p = lds + ((tid * 20 + it * 4) & 4092 & ~3);
#pragma unroll
            for (int q = 0; q < NL; q++) { const f32x4 t = *reinterpret_cast<const f32x4 *>(p + 256 * q); if (q & 1) bf[q % 3] = t; else af = t; }
        }
        const bf16x8 a = __builtin_bit_cast(bf16x8, af);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 6; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bf[j % 3]), acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; q++) v[(q + j) & 7] = fmaf(v[(q + j) & 7], 1.0001f, 0.5f);
            }
        if (NV) {
#pragma unroll
            for (int q = 0; q < 12; q++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int q = 0; q < 8; q++) s += v[q];
    for (int j = 0; j < 6; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
// dependency distance NACC (accumulators used round-robin), NV VALU per MFMA of a given MIX (0: fma, 1: and/sub/perm split mix)
template <int NACC, int NV, int MIX>
__global__ __launch_bounds__(256, 2) void kd(float *out, unsigned long long *cyc, int iters) {
    const int tid = threadIdx.x;
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    f32x4 af = {1.f, 2.f, 3.f, 4.f}, bf = {1.f, 1.f, 1.f, 1.f};
    float v[8]; for (int q = 0; q < 8; q++) v[q] = tid * 1.5f + q;
    unsigned u[8]; for (int q = 0; q < 8; q++) u[q] = tid + q;
    const bf16x8 a = __builtin_bit_cast(bf16x8, af), b = __builtin_bit_cast(bf16x8, bf);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 12; t++) {
            acc[t % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t % NACC], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const int i = (q + t) & 7;
                if (MIX == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                else if ((q + t) % 3 == 0) u[i] = __float_as_uint(v[i]) & 0xffff0000u;
                else if ((q + t) % 3 == 1) v[i] = v[i] - __uint_as_float(u[i]);
                else u[(i + 1) & 7] = __builtin_amdgcn_perm(u[i], __float_as_uint(v[i]), 0x07060302u);
            }
        }
        if (NV) {
#pragma unroll
            for (int q = 0; q < 12; q++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int q = 0; q < 8; q++) s += v[q] + u[q];
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int NV, int MIX> int rund(int blocks_per_cu) {
    const int nb = 256 * blocks_per_cu, iters = 4000;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 256 * 4)); CK(hipMalloc(&cyc, nb * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kd<NACC, NV, MIX>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((kd<NACC, NV, MIX>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double mfma_per_wave = 12.0 * iters;
    printf("bf16 32x32x16, %d accumulators round-robin, %d VALU/MFMA (%s)  blocks/CU %d : %7.1f us | cycles/MFMA per SIMD %.1f\n", NACC, NV,
           MIX ? "and/sub/perm" : "fma", blocks_per_cu, ms * 1e3, (double)h[nb / 2] / (mfma_per_wave * blocks_per_cu));
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

template <int NV, int NL> int runb(int blocks_per_cu) {
    const int nb = 256 * blocks_per_cu, iters = 4000;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 256 * 4)); CK(hipMalloc(&cyc, nb * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kb<NV, NL>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((kb<NV, NL>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nb); CK(hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double mfma_per_wave = 12.0 * iters, flops = (double)nb * 4 * mfma_per_wave * 32768;
    printf("bf16 32x32x16 + %d VALU/MFMA + %d ds_read_b128/12 MFMA  blocks/CU %d : %7.1f us  %7.1f TF | cycles/MFMA per SIMD %.1f, clock %.2f GHz\n", NV, NL,
           blocks_per_cu, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)h[nb / 2] / (mfma_per_wave * blocks_per_cu), h[nb / 2] / (ms * 1e-3) / 1e9);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
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
int main(int argc, char **argv) {
    if (argc > 1) {   // dependency-distance / VALU-mix study
        for (int bpc = 1; bpc <= 2; bpc++) {
            rund<1, 0, 0>(bpc); rund<2, 0, 0>(bpc); rund<3, 0, 0>(bpc); rund<6, 0, 0>(bpc);
            rund<3, 3, 0>(bpc); rund<3, 3, 1>(bpc); rund<6, 3, 0>(bpc); rund<6, 3, 1>(bpc); rund<3, 2, 1>(bpc); rund<6, 2, 1>(bpc);
        }
        return 0;
    }
    for (int bpc = 1; bpc <= 2; bpc++) {
        run<0>(bpc, "pure MFMA");
        run<1>(bpc, "MFMA + 4 ds_read_b128/12");
        run<2>(bpc, "MFMA + ds_read + 24 VALU/12");
    }
    for (int bpc = 1; bpc <= 2; bpc++) {
        runb<0, 0>(bpc); runb<2, 0>(bpc); runb<4, 0>(bpc); runb<6, 0>(bpc); runb<0, 6>(bpc); runb<0, 12>(bpc); runb<4, 6>(bpc); runb<4, 12>(bpc);
    }
    return 0;
}
