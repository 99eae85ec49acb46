// store_pattern.cpp -- how fast does the chip absorb a GEMM epilogue's stores, as a function of the per-instruction footprint?
//
// The bf16 pipeline's GEMMs (csrc/kernels_bx.hip) compute transposed products: a lane owns ONE output row and four consecutive columns
// per accumulator group, so one wave-wide store instruction writes 32 rows x 16 B (bf16) or 32 rows x 32 B (fp32).  This program writes
// the L1 qkv output (M = 204800 rows x N = 576 bf16 = 236 MB) from registers with
//   mode 0: that footprint (8 B per lane, 32 rows x 16 B per instruction, 128 x 192 block tiles of four 64 x 96 wave tiles),
//   mode 1: the same tiles, each wave writing its 64 x 96 sub-tile as row-contiguous 16-B pieces (what an LDS transpose would give:
//           one instruction = 1 KB = 5.3 rows x 192 B),
//   mode 2: fully linear 16 B per lane over the whole matrix (upper bound),
//   mode 3 / 4: the fp32 forms of 0 / 1 (16 B per lane, 32 rows x 32 B per instruction | row-contiguous 384-B row pieces).
// No loads, no LDS, no MFMA: only the store path.
//
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern.cpp -o tools/bin/store_pattern && tools/bin/store_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void store_kernel(unsigned char *out, int M, int N, int tiles_m, int tiles_n, unsigned v) {
    const int ES = MODE >= 3 ? 4 : 2;                       // element size
    const size_t ld = (size_t)N * ES;                       // row pitch in bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == 2) {
        const size_t total = (size_t)M * ld / 16, stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) *(u32x4 *)(out + i * 16) = u32x4{v, v, v, v};
        return;
    }
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        const int row0 = tm * 128 + (wave >> 1) * 64, col0 = tn * 192 + (wave & 1) * 96;
        if (MODE == 0 || MODE == 3) {
            for (int mt = 0; mt < 2; mt++)
                for (int g = 0; g < 12; g++) {
                    const int row = row0 + mt * 32 + (lane & 31), col = col0 + g * 8 + (lane >> 5) * 4;
                    if (row < M) {
                        unsigned char *p = out + (size_t)row * ld + (size_t)col * ES;
                        if (MODE == 0) *(u32x2 *)p = u32x2{v, v + g}; else *(u32x4 *)p = u32x4{v, v + g, v, v};
                    }
                }
        } else {
            const int rowbytes = 96 * ES, per_row = rowbytes / 16;   // 16-B pieces per sub-tile row
            for (int i = lane; i < 64 * per_row; i += 64) {
                const int r = i / per_row, c = i % per_row, row = row0 + r;
                if (row < M) *(u32x4 *)(out + (size_t)row * ld + (size_t)col0 * ES + c * 16) = u32x4{v, v + r, v, v};
            }
        }
    }
}

template <int MODE>
static int run(unsigned char *buf, int M, int N, const char *what) {
    const int tiles_m = (M + 127) / 128, tiles_n = N / 192;
    const int ES = MODE >= 3 ? 4 : 2;
    const double bytes = (double)M * N * ES;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {512, 1024, tiles_m * tiles_n}) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, buf, M, N, tiles_m, tiles_n, 1u);
        CK(hipEventRecord(e0));
        const int it = 20;
        for (int w = 0; w < it; w++) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, buf, M, N, tiles_m, tiles_n, 2u + w);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d %-58s grid %6d: %7.1f us  %5.2f TB/s\n", MODE, what, grid, 1e3 * ms / it, bytes / (ms / it * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    const int M = 204800, N = 576;
    unsigned char *buf;
    CK(hipMalloc(&buf, (size_t)M * N * 4 + 4096));
    if (run<0>(buf, M, N, "bf16, 8 B/lane, 32 rows x 16 B per instruction (shipped)")) return 1;
    if (run<1>(buf, M, N, "bf16, 16 B/lane, row-contiguous 192-B pieces (LDS transpose)")) return 1;
    if (run<2>(buf, M, N, "bf16, linear 16 B/lane (upper bound)")) return 1;
    if (run<3>(buf, M, N, "fp32, 16 B/lane, 32 rows x 32 B per instruction (shipped)")) return 1;
    if (run<4>(buf, M, N, "fp32, 16 B/lane, row-contiguous 384-B pieces (LDS transpose)")) return 1;
    CK(hipFree(buf));
    return 0;
}
