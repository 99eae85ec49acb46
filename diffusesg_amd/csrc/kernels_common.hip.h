// kernels_common.hip.h -- device helpers shared by kernels.hip (fp32 MFMA path) and kernels_lp.hip (bf16-MFMA GEMMs).
#pragma once
#include "kernels.h"

#include <math.h>
#include <stdlib.h>
#include <cmath>

// Every kernel launch of the forward path goes through this macro: under a plan-validation dry run (kernels.h: g_dry_run) the
// launchers run all their shape checks and select their instantiation, but enqueue nothing.
#define DSG_LAUNCH(...) do { if (!::dsg::g_dry_run) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace dsg {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LN_EPS 1e-5f

// The f32 MFMA shares the SIMD's vector ALUs, so every VALU instruction in an MFMA kernel costs matrix throughput
// (tools/mfma_rate.cpp).  An IEEE division expands to ~10 VALU ops and precise expf to ~8: inside the network kernels
// reciprocals, rsqrt and exp use the 1-ulp hardware instructions (v_rcp_f32 / v_rsq_f32 / v_exp_f32); the resulting
// relative error (~1e-7) is three orders of magnitude below the fp32 parity bar.  The sampler's scalar algebra keeps
// correctly rounded IEEE operations (precond/churn/euler/heun kernels below).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float silu_exact(float x) { return x * fast_rcp(1.0f + __expf(-x)); }
// exact-erf GELU (nn.GELU default).  With Phi(x) = 1 - Phi(-|x|) for x >= 0 and Phi(-|x|) for x < 0 both signs collapse into
//     gelu(x) = max(x, 0) - |x| * Phi(-|x|)
// -- no select, no cancellation in the negative tail.  log2 Phi(-a) is smooth on [0, 6] (-1 at 0, ~ -a^2/2 log2e - log2(a sqrt(2 pi))
// beyond); a degree-6 polynomial fitted to it under the weight a*Phi(-a) (the absolute error of the subtracted term; fit by
// tools/fit_gelu.py) gives the term to 5e-8 in exact arithmetic, and above a = 6 it is < 6e-9 * a/6 whatever the exponent says, so
// a is clamped there.  9 VALU + 1 transcendental per element (round 2's first form, A&S 7.1.26 with a reciprocal: 12 + 2 and
// constant moves; every VALU op in an f32-MFMA kernel is paid in matrix throughput and v_exp/v_rcp issue at quarter rate).
// Measured max abs error vs an fp64 GELU over [-12,12]: 2.8e-7 (an fp32 erff evaluation: 4e-7).
__device__ __forceinline__ float gelu_f(float x) {
    const float a = fminf(fabsf(x), 6.0f);
    float p = fmaf(a, 3.3159643839e-05f, -7.6972447974e-04f);
    p = fmaf(a, p, 8.0821445939e-03f);
    p = fmaf(a, p, -5.3413999628e-02f);
    p = fmaf(a, p, -4.5876976689e-01f);
    p = fmaf(a, p, -1.1512020345f);
    p = fmaf(a, p, -9.9999303260e-01f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), fmaxf(x, 0.0f));
}

// Two GELUs per call on the packed-fp32 instructions (v_pk_fma_f32: both halves of a register pair per issue, constants from SGPRs):
// the six polynomial steps and the closing FMA are one instruction per PAIR, |x| clamp and max(x, 0) are one v_med3 each (no NaN
// canonicalisation in front) -- 7 VALU per element instead of 11.5.  Same arithmetic as gelu_f (an IEEE FMA is an IEEE FMA) except that
// the closing product uses the clamped a instead of |x|: equal for |x| <= 6, and beyond the subtracted term is < 1e-8 either way.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_f2(float x0, float x1) {
    const f32x2_t a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x0), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x1), 0.0f, 6.0f)};
    f32x2_t p = __builtin_elementwise_fma(a, (f32x2_t){3.3159643839e-05f, 3.3159643839e-05f}, (f32x2_t){-7.6972447974e-04f, -7.6972447974e-04f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){8.0821445939e-03f, 8.0821445939e-03f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-5.3413999628e-02f, -5.3413999628e-02f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-4.5876976689e-01f, -4.5876976689e-01f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-1.1512020345f, -1.1512020345f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-9.9999303260e-01f, -9.9999303260e-01f});
    const f32x2_t e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    const f32x2_t r = {__builtin_amdgcn_fmed3f(x0, 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(x1, 0.0f, 3.0e38f)};
    return __builtin_elementwise_fma(-a, e, r);
}

// GELU'(x) = Phi(x) + x phi(x) for two values (training form: the fc2-backward GEMM's epilogue).  Phi(-|x|) is the same exp2(polynomial)
// as in gelu_f2 (absolute error <= 2.4e-6, at 0), Phi(x) = 1/2 + copysign(1/2 - Phi(-|x|), x), phi by one more exp2: 10 VALU and two
// transcendentals per element on the packed-fp32 instructions, against ~45 for the erff / expf form it replaces (round 4: that epilogue made
// the K = C products of the backward pass 1.5-2x as long as the same shapes without it).
__device__ __forceinline__ f32x2_t dgelu_f2(float x0, float x1) {
    const f32x2_t x = {x0, x1};
    const f32x2_t a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x0), 0.0f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x1), 0.0f, 6.0f)};
    f32x2_t p = __builtin_elementwise_fma(a, (f32x2_t){3.3159643839e-05f, 3.3159643839e-05f}, (f32x2_t){-7.6972447974e-04f, -7.6972447974e-04f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){8.0821445939e-03f, 8.0821445939e-03f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-5.3413999628e-02f, -5.3413999628e-02f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-4.5876976689e-01f, -4.5876976689e-01f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-1.1512020345f, -1.1512020345f});
    p = __builtin_elementwise_fma(a, p, (f32x2_t){-9.9999303260e-01f, -9.9999303260e-01f});
    const f32x2_t t = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};                   // Phi(-|x|)
    const f32x2_t h = (f32x2_t){0.5f, 0.5f} - t;                                                      // >= 0
    const f32x2_t Phi = (f32x2_t){0.5f, 0.5f} + (f32x2_t){__builtin_copysignf(h[0], x0), __builtin_copysignf(h[1], x1)};
    const f32x2_t q = (x * x) * (f32x2_t){-0.72134752044f, -0.72134752044f};                           // -x^2 / 2 * log2 e
    const f32x2_t ph = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    return __builtin_elementwise_fma(x * (f32x2_t){0.3989422804f, 0.3989422804f}, ph, Phi);
}

// The same form with a degree-4 fit (tools/fit_gelu.py, deg 4: max abs error 6.2e-6 against an fp64 GELU over [-12, 12]), 8 VALU instead of
// 10.  Tried in the bf16 block pipeline (round 4), whose hidden activations are rounded to bf16 right behind it -- NOT used: no measurable
// gain on COCO B = 512 (1119 graphs/s either way) and the C = 96 whole-matrix kernel test moved from inside to just outside its bar
// (5.26e-4 against 5e-4: the approximation flips a few more bf16 roundings of hidden values).  Kept for the record.
__device__ __forceinline__ float gelu_bx4(float x) {
    const float a = fminf(fabsf(x), 6.0f);
    float p = fmaf(a, 3.864195930e-03f, -4.407001343e-02f);
    p = fmaf(a, p, -4.680282015e-01f);
    p = fmaf(a, p, -1.147365234f);
    p = fmaf(a, p, -1.000480675f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), fmaxf(x, 0.0f));
}

// Table-driven GELU for the MFMA kernels (10 VALU + one ds_read_b128 instead of 14 VALU + 2 transcendentals).
// Phi(x) on [-6,6], nodes every 1/64 with (Phi, phi, -x*phi/2): second-order Taylor from the nearest node,
// |error| <= (1/128)^3/6 * max|Phi'''| = 3.2e-8; outside the range Phi is clamped (gelu error < 6e-9).
constexpr int GELU_NODES = 769;                 // 12 * 64 + 1
constexpr int GELU_TAB_FLOATS = GELU_NODES * 4; // float4 per node (16-B aligned LDS reads)
__device__ __forceinline__ float gelu_lut(float x, const float *tab /* LDS */) {
    const float t = __builtin_amdgcn_fmed3f(x + 6.0f, 0.0f, 12.0f);
    const float r = __builtin_rintf(t * 64.0f);
    const float d = fmaf(r, -0.015625f, t);
    const f32x4 c = *reinterpret_cast<const f32x4 *>(tab + ((int)r << 2));
    return x * fmaf(d, fmaf(d, c[2], c[1]), c[0]);
}
__device__ __forceinline__ void gelu_tab_to_lds(float *dst, const float *__restrict__ src, int tid, int nthreads) {
    for (int i = tid; i < GELU_NODES; i += nthreads)
        reinterpret_cast<f32x4 *>(dst)[i] = reinterpret_cast<const f32x4 *>(src)[i];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int GBM = 128, GBN = 96, GBK = 32, GLD = 36;   // GEMM block tile (see kernels.hip)

#ifdef DSG_CLOCK_DIAG
extern __device__ unsigned long long *g_diag_buf;
#endif

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// fp32 -> bf16, round to nearest even, on the integer pipe: bits [31:16] of the result are the bf16 value.  The one rounding the
// bf16 mode uses everywhere (A operands on their way into LDS, weights at pack time, bf16-stored activations), so that storing an
// activation as bf16 gives the consumer bit for bit what it would have rounded itself.
__device__ __forceinline__ unsigned bf16_rne_hi(float x) {
    const unsigned u = __float_as_uint(x);
    return u + 0x7fffu + ((u >> 16) & 1u);
}
__device__ __forceinline__ f32x4 buf_load4_bf16(rsrc_t r, unsigned voff, unsigned soff) {   // four consecutive bf16 -> fp32 (exact)
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    f32x4 v;
    v[0] = __uint_as_float(u[0] << 16); v[1] = __uint_as_float(u[0] & 0xffff0000u);
    v[2] = __uint_as_float(u[1] << 16); v[3] = __uint_as_float(u[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void buf_store_bf16(float v, rsrc_t r, unsigned voff, unsigned soff) {   // offsets in bytes (2 per element)
    __builtin_amdgcn_raw_buffer_store_b16((short)(bf16_rne_hi(v) >> 16), r, voff, soff, 0);
}


// fp32 -> bf16 (round to nearest even) two / four at a time with the hardware conversion (v_cvt_pk_bf16_f32): same values as
// bf16_rne_hi for every finite input
typedef unsigned u32x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // a in the low half
    typedef __bf16 bf16x2_c __attribute__((ext_vector_type(2)));
    typedef float f32x2_c __attribute__((ext_vector_type(2)));
    const f32x2_c v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_c));
}
__device__ __forceinline__ u32x2_c pack_bf16x4(const f32x4 v) { u32x2_c r; r[0] = pack_bf16(v[0], v[1]); r[1] = pack_bf16(v[2], v[3]); return r; }

// implemented in kernels_lp.hip: the bf16-MFMA GEMMs (g.Ws3: split-bf16, fp32-accurate; g.Wb: plain bf16 operands).
// Returns false if the arguments do not select / fit one of them (the caller then runs the fp32 kernel).
bool launch_gemm_lp(const GemmArgs &g, hipStream_t s);

}  // namespace dsg
