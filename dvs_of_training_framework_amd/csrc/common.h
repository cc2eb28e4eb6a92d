// Shared helpers for the gfx950 kernels of libdvsof_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dvsof.h"

#define DVSOF_HIP_TRY(expr)                         \
    do {                                            \
        hipError_t e_ = (expr);                     \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

// after a kernel launch: launch-configuration errors surface here
#define DVSOF_LAUNCH_CHECK() DVSOF_HIP_TRY(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

constexpr int kWave = 64;  // gfx950 wavefront

// Sum over the 64 lanes of a wave; result valid in lane 0.
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Charbonnier rho(d) = (d^2 + eps^2)^0.45 and rho'(d) from ONE log2 and ONE
// exp2 (reference: utils/loss.py:24-35, alpha = 0.45, epsilon = 1e-3):
// e = s^(alpha-1), rho = s * e, rho' = 2 alpha d e.  The transcendental pipe
// runs at a quarter of the VALU rate and the loss kernel is bound by it.
struct Charb {
    float val, der;
};
__device__ __forceinline__ Charb charbonnier(float d)
{
    const float s = fmaf(d, d, 1e-6f);
    const float e = __builtin_amdgcn_exp2f(-0.55f * __builtin_amdgcn_logf(s));
    Charb c;
    c.val = s * e;
    c.der = 0.9f * d * e;
    return c;
}
__device__ __forceinline__ float charb_val(float d)
{
    const float s = fmaf(d, d, 1e-6f);
    return s * __builtin_amdgcn_exp2f(-0.55f * __builtin_amdgcn_logf(s));
}

// Two Charbonnier evaluations at once (the two flow channels of one pair):
// the arithmetic around the two v_log / v_exp maps to packed-f32 VALU ops
// (v_pk_fma_f32 / v_pk_mul_f32: two floats per lane per issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Charb2 {
    f32x2 val, der;
};
__device__ __forceinline__ Charb2 charbonnier2(f32x2 d)
{
    const f32x2 eps = {1e-6f, 1e-6f};
    const f32x2 s = __builtin_elementwise_fma(d, d, eps);
    f32x2 l = {__builtin_amdgcn_logf(s.x), __builtin_amdgcn_logf(s.y)};
    l *= -0.55f;
    const f32x2 e = {__builtin_amdgcn_exp2f(l.x), __builtin_amdgcn_exp2f(l.y)};
    Charb2 c;
    c.val = s * e;
    c.der = (d * 0.9f) * e;
    return c;
}
