// write_size_cal.cpp -- calibration of the WRITE_SIZE PMC counter for the store shapes the kernels use (MI355X_MICROARCH.md calibrates
// 16-B/lane stores only).  Three kernels write exactly 256 MiB each:  k_dword_row: 4 B per lane, a wave covers 256 contiguous bytes;
// k_dword_gemm: the GEMM epilogue's shape -- 4 B per lane, lanes 0..31 cover 128 contiguous bytes of one row, lanes 32..63 the same
// columns four rows further down (row stride 6 KiB); k_dwordx4: 16 B per lane.  tools/pmc_write_cal.sh runs this under
// `rocprofv3 --pmc WRITE_SIZE` and prints counter * 1024 / bytes written per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/write_size_cal.cpp -o tools/bin/write_size_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t BYTES = (size_t)256 << 20;

__global__ void k_dword_row(float *out) { out[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)threadIdx.x; }
__global__ void k_dword_gemm(float *out) {   // one block = 128 rows x 96 columns of a [M][1536] matrix, stored like gemm4_f32_kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int tiles_n = 16, tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    float *base = out + ((size_t)tm * 128 + wave * 32 + 4 * lhalf) * 1536 + tn * 96 + lrow;
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 16; r++) base[(size_t)((r & 3) + 8 * (r >> 2)) * 1536 + 32 * j] = (float)r;
}
__global__ void k_dwordx4(f32x4 *out) { out[(size_t)blockIdx.x * 256 + threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f}; }

int main() {
    // k_dword_gemm touches whole 128-row bands of a [rows][1536] matrix: size the buffer for the last band, not for the byte count
    const size_t tiles = BYTES / (128 * 96 * 4), bands = (tiles + 15) / 16, need = bands * 128 * 1536 * sizeof(float);
    const size_t alloc = need > BYTES ? need : BYTES;
    float *buf; CK(hipMalloc(&buf, alloc)); CK(hipMemset(buf, 0, alloc)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_dword_row, dim3(BYTES / 1024), dim3(256), 0, 0, buf);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_dword_gemm, dim3(BYTES / (128 * 96 * 4)), dim3(256), 0, 0, buf);   // 5461 tiles of 48 KiB (= 255.98 MiB)
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_dwordx4, dim3(BYTES / 4096), dim3(256), 0, 0, reinterpret_cast<f32x4 *>(buf));
    CK(hipDeviceSynchronize());
    printf("bytes written: k_dword_row %zu k_dword_gemm %zu k_dwordx4 %zu\n", BYTES, (BYTES / (128 * 96 * 4)) * (size_t)(128 * 96 * 4), BYTES);
    return 0;
}
