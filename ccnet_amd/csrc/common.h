// Shared device helpers for the gfx950 kernels of libccnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ccnet_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CCNET_LAUNCH_CHECK()                                  \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return CCNET_ELAUNCH; \
    } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
// round-to-nearest-even f32 -> bf16 via the hardware convert (NaN safe).
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN safe)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}
// x * sigmoid(x); v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division sequence
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. at the fp32 rounding level of the exact form):
// one v_rcp_f32, one v_exp_f32 and a degree-5 Horner chain, branch-free
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float g) { return 0.5f * g * (1.0f + erf_fast(g * 0.70710678118654752f)); }

// a * gelu_erf(g) on two lanes of packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 run two fp32 ops per lane per issue).
// erfc(|g| / sqrt 2) = exp2(|g| Q(|g|)) with Q a degree-5 fit (tools/fit_gelu.py: max abs error of gelu 3.1e-7 over
// all fp32 g, at the fp32 evaluation floor), so gelu(g) = max(g, 0) - |g|/2 * exp2(|g| Q(|g|)): one exponential per
// element, no reciprocal (the two quarter-rate instructions dominated the A&S 7.1.26 form in this epilogue).
__device__ __forceinline__ f32x2 geglu_pk(f32x2 a, f32x2 g) {
    const f32x2 ag = {fabsf(g[0]), fabsf(g[1])};
    f32x2 q = __builtin_elementwise_fma(ag, f32x2{2.9924784e-05f, 2.9924784e-05f}, f32x2{-0.0007398792f, -0.0007398792f});
    q = __builtin_elementwise_fma(q, ag, f32x2{0.007977478f, 0.007977478f});
    q = __builtin_elementwise_fma(q, ag, f32x2{-0.053238206f, -0.053238206f});
    q = __builtin_elementwise_fma(q, ag, f32x2{-0.45891568f, -0.45891568f});
    q = __builtin_elementwise_fma(q, ag, f32x2{-1.1511471f, -1.1511471f});
    q = q * ag;
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 relu = {fmaxf(g[0], 0.0f), fmaxf(g[1], 0.0f)};
    const f32x2 nh = ag * f32x2{-0.5f, -0.5f};
    return a * __builtin_elementwise_fma(nh, e, relu);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
