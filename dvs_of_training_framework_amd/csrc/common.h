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
