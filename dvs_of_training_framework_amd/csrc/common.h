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

// Timing probes (DVSOF_GCONV_DBG / DVSOF_LOSS_DBG: skip pieces of a kernel to
// see what they cost; results are wrong by construction) exist only in the
// PROBE build (`make probes` -> libdvsof_hip_probes.so, -DDVSOF_PROBES).  In
// the product library DVSOF_DBG(P) is the constant 0: no runtime wrong-result
// switch, no dead scalar tests in the hot loops.
#ifdef DVSOF_PROBES
#define DVSOF_DBG(P) ((P).dbg)
#else
#define DVSOF_DBG(P) 0
#endif

constexpr int kWave = 64;  // gfx950 wavefront

// Stream-ordered fill by a KERNEL.  hipMemsetAsync becomes a memset NODE in a
// captured hipGraph: the step executor (exec.hip) replays kernel nodes only,
// and re-launching a graph with a memset node while its previous replay was
// still in flight is the suspected cause of round 1's "write to a read-only
// page" fault.  Every zero-fill the library enqueues is therefore a kernel.
static __global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *__restrict__ p, uint32_t v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;    // words up to 16-byte alignment
    if (head > n) head = n;
    if (i < head) p[i] = v;
    uint4 *q = (uint4 *)(p + head);
    const size_t nq = (n - head) / 4;
    for (size_t k = i; k < nq; k += stride) q[k] = make_uint4(v, v, v, v);
    const size_t done = head + nq * 4;
    if (i < n - done) p[done + i] = v;
}
// nbytes: a multiple of 4 (float / int32 buffers)
static inline int fill_u32(void *p, uint32_t v, size_t nbytes, hipStream_t st)
{
    const size_t n = nbytes / 4;
    if (n == 0) return DVSOF_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)p, v, n);
    return (int)hipGetLastError();
}

// Sum over the 64 lanes of a wave; the result is valid in EVERY lane.
// 32-bit types: DPP row shifts + row broadcasts + one v_readlane, all VALU --
// ds_bpermute (what __shfl compiles to) goes through the LDS crossbar and
// measured ~12 us of a 97 us loss sweep for 42 of them per wave.
//   row_shr:n   0x110 + n   lane i of a 16-lane row reads lane i - n (0 outside)
//   row_bcast15 0x142       lane 15 of row r -> every lane of row r + 1
//   row_bcast31 0x143       lane 31 -> every lane of rows 2, 3
__device__ __forceinline__ int wave_sum_bits(int v, bool is_float)
{
#if defined(__HIP_DEVICE_COMPILE__)
    auto add = [&](int a, int b) -> int {
        return is_float ? __builtin_bit_cast(int, __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b))
                        : a + b;
    };
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // lane 15 of each row: row sum
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));  // rows 1, 3 += row 0, 2
    v = add(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));  // rows 2, 3 += lane 31
    return __builtin_amdgcn_readlane(v, 63);
#else
    return v;
#endif
}
__device__ __forceinline__ float wave_sum(float v)
{
    return __builtin_bit_cast(float, wave_sum_bits(__builtin_bit_cast(int, v), true));
}
__device__ __forceinline__ int wave_sum(int v) { return wave_sum_bits(v, false); }
// 64-bit: shuffle tree (only the few closing waves of a reduction use it); valid in lane 0
__device__ __forceinline__ double wave_sum(double v)
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
