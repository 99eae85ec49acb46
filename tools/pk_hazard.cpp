// pk_hazard.cpp -- discriminating experiment for the round-1 "wrong 16-lane group" event of the bf16-MFMA GEMMs.
//
// Round 1 saw rare wrong lanes (one 16-lane group of one VGPR in ~1e3 wave tiles) in kernels whose A operands were built by
// packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 from the SLP vectoriser) or v_cvt_pk_bf16_f32 right next to
// v_mfma_f32_32x32x16_bf16 of the same wave, and removed them by banning those instructions.  That left two candidate causes:
//   (H) a VALU<->MFMA register hazard of the multi-pass packed instructions (a packed-f32 op runs in four 16-lane passes), or
//   (L) an ordering problem on the LDS path those kernels also used (double-buffered tile, one barrier per chunk).
// This program has NO LDS and NO global memory inside its loop: bf16 MFMAs whose A operand is (re)written by a packed-f32
// FMA / a packed bf16 conversion immediately before it is consumed, with operand values chosen so that every result is known
// exactly.  Any wrong lane here is (H); zero wrong lanes over >1e10 lane-results leaves (L).
//
//   hipcc --offload-arch=gfx950 -O3 tools/pk_hazard.cpp -o tools/bin/pk_hazard && tools/bin/pk_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: A operand refreshed by v_pk_fma_f32 (x*1+0 keeps the bit pattern: two bf16 1.0 per dword)
// MODE 1: A operand refreshed by scalar v_fma_f32 (control)
// MODE 2: A operand refreshed by v_cvt_pk_bf16_f32 of fp32 ones
// MODE 3: like 0, plus an independent packed-f32 chain whose own result is checked (x <- x*1 + d, exact integers)
// MODE 4/5 (lds_kernel below): the packed result goes through LDS first -- ds_write_b128 of the freshly written registers,
//           the registers are overwritten by the next packed op right behind the store (write-after-read on the store data),
//           the SAME wave reads it back (each lane its neighbour's slot; no other wave touches the slab: no barrier, no cross-wave
//           ordering involved) and feeds
//           it to the MFMAs.  4: v_pk_fma_f32, 5: v_cvt_pk_bf16_f32.
template <int MODE>
__global__ __launch_bounds__(256, 2) void hazard_kernel(float *acc_out, float *pk_out, float one, float zero, float d, int iters) {
    const int tid = threadIdx.x;
    f32x16 acc[3];
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    // A = eight bf16 values per lane, as four DISTINCT dwords (so that nothing is common-subexpression-eliminated):
    // [1,2] [2,1] [1,1] [2,2]  -> sum 12 per lane, 24 per output element and MFMA (K = 16 spans the two half-waves); B = ones
    f32x2 a01 = {__uint_as_float(0x40003F80u), __uint_as_float(0x3F804000u)};
    f32x2 a23 = {__uint_as_float(0x3F803F80u), __uint_as_float(0x40004000u)};
    const u32x4 bb = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 B = __builtin_bit_cast(bf16x8, bb);
    const f32x2 one2 = {one, one}, zero2 = {zero, zero}, d2 = {d, d};
    f32x2 chain = {(float)(tid & 7), (float)(tid & 3)};
    float f1 = 1.0f * one, f2 = 2.0f * one;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0 || MODE == 3) {
                a01 = __builtin_elementwise_fma(a01, one2, zero2);   // v_pk_fma_f32: x*1+0 keeps the bit pattern
                a23 = __builtin_elementwise_fma(a23, one2, zero2);
            } else if (MODE == 1) {
                float x0 = a01[0], x1 = a01[1], x2 = a23[0], x3 = a23[1];
                x0 = __builtin_fmaf(x0, one, zero); asm volatile("" : "+v"(x0));
                x1 = __builtin_fmaf(x1, one, zero); asm volatile("" : "+v"(x1));
                x2 = __builtin_fmaf(x2, one, zero); asm volatile("" : "+v"(x2));
                x3 = __builtin_fmaf(x3, one, zero); asm volatile("" : "+v"(x3));
                a01 = f32x2{x0, x1}; a23 = f32x2{x2, x3};
            } else if (MODE == 2) {
                f1 = __builtin_fmaf(f1, one, zero); f2 = __builtin_fmaf(f2, one, zero);
                const bf16x2 c12 = {(__bf16)f1, (__bf16)f2}, c21 = {(__bf16)f2, (__bf16)f1};   // v_cvt_pk_bf16_f32
                const bf16x2 c11 = {(__bf16)f1, (__bf16)f1}, c22 = {(__bf16)f2, (__bf16)f2};
                a01 = f32x2{__builtin_bit_cast(float, c12), __builtin_bit_cast(float, c21)};
                a23 = f32x2{__builtin_bit_cast(float, c11), __builtin_bit_cast(float, c22)};
            }
            if (MODE == 3) chain = __builtin_elementwise_fma(chain, one2, d2);
            const u32x4 ab = {__float_as_uint(a01[0]), __float_as_uint(a01[1]), __float_as_uint(a23[0]), __float_as_uint(a23[1])};
            const bf16x8 A = __builtin_bit_cast(bf16x8, ab);
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[j], 0, 0, 0);
        }
    }
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * 48;
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 16; r++) acc_out[base + j * 16 + r] = acc[j][r];
    pk_out[((size_t)blockIdx.x * 256 + tid) * 2] = chain[0];
    pk_out[((size_t)blockIdx.x * 256 + tid) * 2 + 1] = chain[1];
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void lds_kernel(float *acc_out, float *pk_out, float one, float zero, float d, int iters) {
    __shared__ __attribute__((aligned(16))) float slab[256 * 4 * 2];   // [2 buffers][256 threads] x 16 B, private per thread
    const int tid = threadIdx.x;
    f32x16 acc[3];
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    f32x2 a01 = {__uint_as_float(0x40003F80u), __uint_as_float(0x3F804000u)};
    f32x2 a23 = {__uint_as_float(0x3F803F80u), __uint_as_float(0x40004000u)};
    const u32x4 bb = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 B = __builtin_bit_cast(bf16x8, bb);
    const f32x2 one2 = {one, one}, zero2 = {zero, zero};
    float f1 = 1.0f * one, f2 = 2.0f * one;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // both buffers hold the pattern before the loop: the compiler may (and does) hoist a lane's read of its neighbour's slot
    // above its own write, so a read can see the previous pass's store -- same bits either way
    for (int q = 0; q < 2; q++) *reinterpret_cast<f32x4 *>(slab + (q * 256 + tid) * 4) = f32x4{a01[0], a01[1], a23[0], a23[1]};
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 4) {
                a01 = __builtin_elementwise_fma(a01, one2, zero2);
                a23 = __builtin_elementwise_fma(a23, one2, zero2);
            } else {
                f1 = __builtin_fmaf(f1, one, zero); f2 = __builtin_fmaf(f2, one, zero);
                const bf16x2 c12 = {(__bf16)f1, (__bf16)f2}, c21 = {(__bf16)f2, (__bf16)f1};
                const bf16x2 c11 = {(__bf16)f1, (__bf16)f1}, c22 = {(__bf16)f2, (__bf16)f2};
                a01 = f32x2{__builtin_bit_cast(float, c12), __builtin_bit_cast(float, c21)};
                a23 = f32x2{__builtin_bit_cast(float, c11), __builtin_bit_cast(float, c22)};
            }
            float *dst = slab + ((u & 1) * 256 + tid) * 4;
            const float *src = slab + ((u & 1) * 256 + (tid ^ 1)) * 4;                 // the neighbouring lane's slot (same wave)
            *reinterpret_cast<f32x4 *>(dst) = f32x4{a01[0], a01[1], a23[0], a23[1]};   // ds_write_b128 of the fresh registers
            const f32x4 back = *reinterpret_cast<const f32x4 *>(src);                  // ds_read_b128; LDS ops of a wave are in order
            const u32x4 ab = {__float_as_uint(back[0]), __float_as_uint(back[1]), __float_as_uint(back[2]), __float_as_uint(back[3])};
            const bf16x8 A = __builtin_bit_cast(bf16x8, ab);
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[j], 0, 0, 0);
        }
    }
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * 48;
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 16; r++) acc_out[base + j * 16 + r] = acc[j][r];
    pk_out[((size_t)blockIdx.x * 256 + tid) * 2] = 0.f;
    pk_out[((size_t)blockIdx.x * 256 + tid) * 2 + 1] = 0.f;
}

template <int MODE>
int run(const char *name, int blocks, int iters, int launches) {
    float *acc = nullptr, *pk = nullptr;
    const size_t n_acc = (size_t)blocks * 256 * 48, n_pk = (size_t)blocks * 256 * 2;
    CK(hipMalloc((void **)&acc, n_acc * 4));
    CK(hipMalloc((void **)&pk, n_pk * 4));
    std::vector<float> h_acc(n_acc), h_pk(n_pk);
    // every MFMA adds sum_k a_k*1 = 24 to every accumulator element: 4 MFMAs per accumulator per iteration
    const float expect = 24.0f * 4.0f * (float)iters;   // exact in fp32 while < 2^24
    const float dstep = 1.0f;
    unsigned long long bad_acc = 0, bad_pk = 0, checked = 0;
    int first_lane = -1, first_block = -1;
    for (int l = 0; l < launches; l++) {
        if (MODE >= 4) hipLaunchKernelGGL(lds_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, acc, pk, 1.0f, 0.0f, dstep, iters);
        else hipLaunchKernelGGL(hazard_kernel<(MODE < 4 ? MODE : 0)>, dim3(blocks), dim3(256), 0, 0, acc, pk, 1.0f, 0.0f, dstep, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_acc.data(), acc, n_acc * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_pk.data(), pk, n_pk * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n_acc; i++)
            if (h_acc[i] != expect) {
                if (!bad_acc) { first_lane = (int)((i / 48) % 64); first_block = (int)(i / 48 / 256); }
                bad_acc++;
            }
        if (MODE == 3)
            for (size_t t = 0; t < n_pk / 2; t++) {
                const int tid = (int)(t % 256);
                const float e0 = (float)(tid & 7) + 4.0f * iters * dstep, e1 = (float)(tid & 3) + 4.0f * iters * dstep;
                if (h_pk[2 * t] != e0 || h_pk[2 * t + 1] != e1) bad_pk++;
            }
        checked += n_acc;
    }
    printf("%-46s: %llu accumulator lane-values checked (%.2e MFMA lane-results), wrong accumulators %llu, wrong packed-chain values %llu",
           name, checked, (double)checked * 4.0 * iters, bad_acc, bad_pk);
    if (bad_acc) printf("  [first: block %d lane %d]", first_block, first_lane);
    printf("\n");
    (void)hipFree(acc); (void)hipFree(pk);
    return 0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;     // 96*iters must stay below 2^24
    const int launches = argc > 2 ? atoi(argv[2]) : 6;
    const int blocks = 256 * 2 * 4;                        // two resident blocks per CU (two waves per SIMD), four rounds
    if (run<1>("control: A operand rewritten by v_fma_f32", blocks, iters, launches)) return 1;
    if (run<0>("A operand rewritten by v_pk_fma_f32", blocks, iters, launches)) return 1;
    if (run<3>("v_pk_fma_f32 on A + independent packed chain", blocks, iters, launches)) return 1;
    if (run<2>("A operand rewritten by v_cvt_pk_bf16_f32", blocks, iters, launches)) return 1;
    if (run<4>("v_pk_fma_f32 -> ds_write_b128 -> ds_read -> MFMA", blocks, iters, launches)) return 1;
    if (run<5>("v_cvt_pk_bf16_f32 -> ds_write_b128 -> ds_read -> MFMA", blocks, iters, launches)) return 1;
    return 0;
}
