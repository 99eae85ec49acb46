// GPU probe: does global_load_lds_dwordx4 (LDS-DMA) reach LDS offsets beyond 64 KiB on gfx950, and is its LDS image lane-linear?
// build: hipcc --offload-arch=gfx950 -O3 tools/glds_probe.cpp -o tools/bin/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_vptr;
__global__ __launch_bounds__(512, 1) void probe(const unsigned *src, unsigned *out, int n_off, const int *offs) {
    __shared__ __attribute__((aligned(16))) char lds[163840];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) reinterpret_cast<unsigned *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    if (wave == 0)
        for (int k = 0; k < n_off; k++) {
            const int off = __builtin_amdgcn_readfirstlane(offs[k]);
            __builtin_amdgcn_global_load_lds((const char *)src + k * 1024 + lane * 16, (lds_vptr)(lds + off), 16, 0, 0);
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) out[i] = reinterpret_cast<unsigned *>(lds)[i];
}
int main() {
    std::vector<int> offs = {0, 1024, 60 * 1024, 65536, 70 * 1024, 98304, 130 * 1024, 147456, 162816};
    const int n = (int)offs.size();
    std::vector<unsigned> h(n * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0x10000000u + (unsigned)i;
    unsigned *d, *o; int *doff;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 163840); hipMalloc(&doff, n * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, offs.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d, o, n, doff);
    std::vector<unsigned> r(163840 / 4);
    if (hipMemcpy(r.data(), o, 163840, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
    int bad = 0;
    for (int k = 0; k < n; k++) {
        int ok = 1;
        for (int i = 0; i < 256; i++) if (r[offs[k] / 4 + i] != h[k * 256 + i]) ok = 0;
        printf("piece %d -> LDS offset %6d: %s (first word %08x, expected %08x)\n", k, offs[k], ok ? "OK" : "WRONG", r[offs[k] / 4], h[k * 256]);
        bad += !ok;
    }
    size_t touched = 0;
    for (size_t i = 0; i < r.size(); i++) touched += r[i] != 0xdeadbeefu;
    printf("words changed: %zu (expected %d)\n", touched, n * 256);
    return bad;
}
