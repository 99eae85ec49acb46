// GPU probe: rate at which every CU can stream the SAME weight image from L2 into LDS by LDS-DMA (global_load_lds_dwordx4),
// as a function of the bytes kept in flight per CU and of whether the CUs walk the image in step or rotated against each other.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_stream_probe.cpp -o tools/bin/dma_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_vptr;
constexpr int STAGE = 49152;   // 48 KB = 48 pieces of 1 KiB, 6 per wave
template <int INFLIGHT>   // stages requested ahead (1..3)
__global__ __launch_bounds__(512, 1) void stream(const char *img, int n_stages_img, int iters, int rotate, unsigned long long *clk, float *sink) {
    __shared__ __attribute__((aligned(16))) char lds[3 * STAGE + 16384];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int start = rotate ? (blockIdx.x * 5) % n_stages_img : 0;
    auto req = [&](int it) {
        const char *src = img + (size_t)((start + it) % n_stages_img) * STAGE + wave * 6144 + lane * 16;
        char *dst = lds + (it % 3) * STAGE + wave * 6144;
#pragma unroll
        for (int i = 0; i < 6; i++) __builtin_amdgcn_global_load_lds(src + i * 1024, (lds_vptr)(dst + i * 1024), 16, 0, 0);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < INFLIGHT; k++) req(k);
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        // stage `it` must have landed: at most INFLIGHT - 1 younger requests outstanding
        if (INFLIGHT == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        if (INFLIGHT == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
        if (INFLIGHT == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        acc += *reinterpret_cast<const float *>(lds + (it % 3) * STAGE + threadIdx.x * 64);   // touch the stage
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        if (it + INFLIGHT < iters) req(it + INFLIGHT);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}
int main(int argc, char **argv) {
    const int n_stages = 54, iters = 54 * 8, blocks = argc > 1 ? atoi(argv[1]) : 256;
    char *img; unsigned long long *clk; float *sink;
    hipMalloc(&img, (size_t)n_stages * STAGE); hipMemset(img, 1, (size_t)n_stages * STAGE);
    hipMalloc(&clk, blocks * 8); hipMalloc(&sink, 4);
    for (int rotate = 0; rotate < 2; rotate++)
        for (int inflight = 1; inflight <= 3; inflight++) {
            for (int rep = 0; rep < 2; rep++) {
                if (inflight == 1) hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(512), 0, 0, img, n_stages, iters, rotate, clk, sink);
                if (inflight == 2) hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(512), 0, 0, img, n_stages, iters, rotate, clk, sink);
                if (inflight == 3) hipLaunchKernelGGL(stream<3>, dim3(blocks), dim3(512), 0, 0, img, n_stages, iters, rotate, clk, sink);
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += (double)v;
            const double per_stage = s / blocks / iters;
            printf("blocks %d, %s, %d stage(s) (%d KB) in flight: %.0f clk per 48-KB stage = %.1f B/clk/CU\n", blocks, rotate ? "rotated starts" : "in step", inflight,
                   48 * inflight, per_stage, STAGE / per_stage);
        }
    return 0;
}
