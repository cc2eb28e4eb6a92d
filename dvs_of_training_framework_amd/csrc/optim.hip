// Fused multi-tensor AdamW (amsgrad) step: one launch per parameter group.
// Replaces torch.optim.AdamW(amsgrad=True).step() as constructed at
// train_flownet.py:57-75 and called at utils/training.py:164 (foreach /
// unfused ATen ops: ~10 launches per tensor).  HBM-bound: reads p,g,m,v,vmax,
// writes p,m,v,vmax = 36 B per parameter.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay;
    float step_size;  // lr / (1 - beta1^t)
    float bc2_sqrt;   // sqrt(1 - beta2^t)
    int amsgrad;
};

#ifndef DVSOF_ADAM_CHUNK
#define DVSOF_ADAM_CHUNK 1024   // (measured in the step: 1024 3150, 2048 3133, 4096 3118, 8192 3091 samples/s)
#endif
constexpr int CHUNK = DVSOF_ADAM_CHUNK;  // elements per workgroup

// op order of torch.optim.adam._single_tensor_adam (decoupled weight decay)
__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float &vm,
                                          const AdamArgs &a)
{
    p = p * (1.f - a.lr * a.weight_decay);
    m = m + (g - m) * (1.f - a.beta1);           // lerp_
    v = v * a.beta2 + (1.f - a.beta2) * g * g;   // mul_().addcmul_()
    float denom;
    if (a.amsgrad) {
        vm = fmaxf(vm, v);
        denom = sqrtf(vm) / a.bc2_sqrt + a.eps;
    } else {
        denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    }
    p = p - a.step_size * (m / denom);
}

// table: per tensor {p, g, m, v, vmax} pointers and element count; chunk
// table: (tensor id, chunk index) per workgroup.
// dyn (optional, device): {lr, lr / (1 - beta1^t), sqrt(1 - beta2^t)} of THIS
// step.  A step captured in a hipGraph bakes its kernel arguments in; what
// changes from step to step (learning-rate schedule, bias corrections) then
// comes from this table, which the host refreshes before every replay.
__global__ __launch_bounds__(256) void adamw_kernel(const uint64_t *__restrict__ ptrs,
                                                    const int64_t *__restrict__ sizes,
                                                    const int32_t *__restrict__ chunks,
                                                    const AdamArgs a_in,
                                                    const float *__restrict__ dyn)
{
    AdamArgs a = a_in;
    if (dyn) {
        a.lr = dyn[0];
        a.step_size = dyn[1];
        a.bc2_sqrt = dyn[2];
    }
    const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
    float *p = (float *)ptrs[5 * t + 0];
    const float *g = (const float *)ptrs[5 * t + 1];
    float *m = (float *)ptrs[5 * t + 2];
    float *v = (float *)ptrs[5 * t + 3];
    float *vm = (float *)ptrs[5 * t + 4];
    const int64_t n = sizes[t];
    const int64_t base = (int64_t)c * CHUNK;
#pragma unroll
    for (int it = 0; it < CHUNK / 1024; ++it) {
        const int64_t i = base + it * 1024 + threadIdx.x * 4;
        if (i + 3 < n) {
            f32x4 P = *(f32x4u *)(p + i), G = *(const f32x4u *)(g + i);
            f32x4 M = *(f32x4u *)(m + i), V = *(f32x4u *)(v + i);
            f32x4 X = a.amsgrad ? *(f32x4u *)(vm + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = P[j], mj = M[j], vj = V[j], xj = X[j];
                adam_elem(pj, G[j], mj, vj, xj, a);
                P[j] = pj; M[j] = mj; V[j] = vj; X[j] = xj;
            }
            *(f32x4u *)(p + i) = P;
            *(f32x4u *)(m + i) = M;
            *(f32x4u *)(v + i) = V;
            if (a.amsgrad) *(f32x4u *)(vm + i) = X;
        } else {
            for (int64_t j = i; j < n && j < i + 4; ++j) {
                float x = a.amsgrad ? vm[j] : 0.f;
                adam_elem(p[j], g[j], m[j], v[j], x, a);
                if (a.amsgrad) vm[j] = x;
            }
        }
    }
}

// The per-step table of a captured step ({lr, lr/bc1, sqrt(bc2)} per group) written from
// KERNEL ARGUMENTS: a 16-byte host-to-device copy is a blit kernel with system-scope fences
// (4 us + an 11 us bubble in front of the next kernel, at the head of every step)
constexpr int DYN_VALUES = 64;
struct DynValues {
    float v[DYN_VALUES];
};
__global__ void set_dynamic_kernel(float *dyn, DynValues vals, int n)
{
    if ((int)threadIdx.x < n) dyn[threadIdx.x] = vals.v[threadIdx.x];
}

}  // namespace

extern "C" {

int dvsof_adamw_chunk_elems(void) { return CHUNK; }

int dvsof_adamw_step(const uint64_t *ptrs, const int64_t *sizes, const int32_t *chunks,
                     int num_chunks, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, int amsgrad, void *stream)
{
    if (!ptrs || !sizes || !chunks || num_chunks < 0 || step < 1) return DVSOF_EINVAL;
    if (num_chunks == 0) return DVSOF_OK;
    AdamArgs a = {};
    a.lr = lr;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    // bias corrections in double like the Python reference (python floats)
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.amsgrad = amsgrad;
    hipLaunchKernelGGL(adamw_kernel, dim3(num_chunks), dim3(256), 0, as_stream(stream), ptrs, sizes,
                       chunks, a, (const float *)nullptr);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

void dvsof_adamw_dynamic(float lr, float beta1, float beta2, int step, float *host_out3)
{
    // the same double-precision bias corrections as dvsof_adamw_step
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    host_out3[0] = lr;
    host_out3[1] = (float)((double)lr / bc1);
    host_out3[2] = (float)sqrt(bc2);
}

int dvsof_adamw_set_dynamic(float *dyn, const float *host_values, int n, void *stream)
{
    if (!dyn || !host_values || n < 0) return DVSOF_EINVAL;
    for (int o = 0; o < n; o += DYN_VALUES) {   // the values travel as kernel arguments
        DynValues v;
        const int m = n - o < DYN_VALUES ? n - o : DYN_VALUES;
        for (int i = 0; i < DYN_VALUES; ++i) v.v[i] = i < m ? host_values[o + i] : 0.f;
        hipLaunchKernelGGL(set_dynamic_kernel, dim3(1), dim3(DYN_VALUES), 0, as_stream(stream), dyn + o, v, m);
        DVSOF_LAUNCH_CHECK();
    }
    return DVSOF_OK;
}

int dvsof_adamw_step_dyn(const uint64_t *ptrs, const int64_t *sizes, const int32_t *chunks,
                         int num_chunks, const float *dyn, float beta1, float beta2, float eps,
                         float weight_decay, int amsgrad, void *stream)
{
    if (!ptrs || !sizes || !chunks || !dyn || num_chunks < 0) return DVSOF_EINVAL;
    if (num_chunks == 0) return DVSOF_OK;
    AdamArgs a = {};
    a.lr = a.step_size = a.bc2_sqrt = 0.f;   // from dyn
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    a.amsgrad = amsgrad;
    hipLaunchKernelGGL(adamw_kernel, dim3(num_chunks), dim3(256), 0, as_stream(stream), ptrs, sizes,
                       chunks, a, dyn);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// RAdam (Liu et al., ICLR 2020) and Ranger (= RAdam + Lookahead k, alpha
// [+ gradient centralisation]).  Replace RAdam.radam.RAdam and ranger.Ranger
// as constructed at train_flownet.py:62-71 (un-vendored submodules upstream:
// arithmetic restated from the published algorithms, oracle/ref_optim.py).
// ---------------------------------------------------------------------------
namespace {

struct RAdamArgs {
    float lr, beta1, beta2, eps, weight_decay;
    float step_size;   // already divided by (1 - beta1^t)
    int rectified;     // N_sma above the threshold: adaptive step
    int lookahead;     // Ranger: this is a k-th step
    float la_alpha;
};

__device__ __forceinline__ void radam_elem(float &p, float g, float &m, float &v, float &slow,
                                           const RAdamArgs &a)
{
    v = v * a.beta2 + (1.f - a.beta2) * g * g;
    m = m * a.beta1 + (1.f - a.beta1) * g;
    if (a.weight_decay != 0.f) p = p - a.weight_decay * a.lr * p;
    if (a.rectified) p = p - a.step_size * a.lr * (m / (sqrtf(v) + a.eps));
    else if (a.step_size > 0.f) p = p - a.step_size * a.lr * m;
    if (a.lookahead) {
        slow = slow + a.la_alpha * (p - slow);
        p = slow;
    }
}

__global__ __launch_bounds__(256) void radam_kernel(const uint64_t *__restrict__ ptrs,
                                                    const int64_t *__restrict__ sizes,
                                                    const int32_t *__restrict__ chunks,
                                                    const RAdamArgs a, const int use_slow)
{
    const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
    float *p = (float *)ptrs[5 * t + 0];
    const float *g = (const float *)ptrs[5 * t + 1];
    float *m = (float *)ptrs[5 * t + 2];
    float *v = (float *)ptrs[5 * t + 3];
    float *sl = (float *)ptrs[5 * t + 4];
    const int64_t n = sizes[t];
    const int64_t base = (int64_t)c * CHUNK;
#pragma unroll
    for (int it = 0; it < CHUNK / 1024; ++it) {
        const int64_t i = base + it * 1024 + threadIdx.x * 4;
        if (i + 3 < n) {
            f32x4 P = *(f32x4u *)(p + i), G = *(const f32x4u *)(g + i);
            f32x4 M = *(f32x4u *)(m + i), V = *(f32x4u *)(v + i);
            f32x4 S = (use_slow && a.lookahead) ? *(f32x4u *)(sl + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = P[j], mj = M[j], vj = V[j], sj = S[j];
                radam_elem(pj, G[j], mj, vj, sj, a);
                P[j] = pj; M[j] = mj; V[j] = vj; S[j] = sj;
            }
            *(f32x4u *)(p + i) = P;
            *(f32x4u *)(m + i) = M;
            *(f32x4u *)(v + i) = V;
            if (use_slow && a.lookahead) *(f32x4u *)(sl + i) = S;
        } else {
            for (int64_t j = i; j < n && j < i + 4; ++j) {
                float s = (use_slow && a.lookahead) ? sl[j] : 0.f;
                radam_elem(p[j], g[j], m[j], v[j], s, a);
                if (use_slow && a.lookahead) sl[j] = s;
            }
        }
    }
}

// Gradient centralisation: g[r][:] -= mean(g[r][:]), one workgroup per row.
__global__ __launch_bounds__(256) void grad_centralize_kernel(float *g, int row_len)
{
    __shared__ double red[4];
    float *row = g + (size_t)blockIdx.x * row_len;
    double s = 0;
    for (int i = threadIdx.x; i < row_len; i += 256) s += (double)row[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)row_len);
    for (int i = threadIdx.x; i < row_len; i += 256) row[i] -= mean;
}

}  // namespace

extern "C" {

int dvsof_radam_step(const uint64_t *ptrs, const int64_t *sizes, const int32_t *chunks,
                     int num_chunks, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float nsma_threshold, int degenerate_to_sgd,
                     int lookahead_now, float lookahead_alpha, void *stream)
{
    if (!ptrs || !sizes || !chunks || num_chunks < 0 || step < 1) return DVSOF_EINVAL;
    if (num_chunks == 0) return DVSOF_OK;
    RAdamArgs a = {};
    a.lr = lr;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    // rectification term in double, like the Python references
    const double b2t = pow((double)beta2, (double)step);
    const double nmax = 2.0 / (1.0 - (double)beta2) - 1.0;
    const double nsma = nmax - 2.0 * step * b2t / (1.0 - b2t);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    a.rectified = (degenerate_to_sgd & 2) ? nsma >= (double)nsma_threshold   // RAdam: ">="
                                          : nsma > (double)nsma_threshold;   // Ranger: ">"
    if (a.rectified)
        a.step_size = (float)(sqrt((1.0 - b2t) * (nsma - 4.0) / (nmax - 4.0) * (nsma - 2.0) / nsma *
                                   nmax / (nmax - 2.0)) / bc1);
    else
        a.step_size = (degenerate_to_sgd & 1) ? (float)(1.0 / bc1) : -1.f;
    a.lookahead = lookahead_now;
    a.la_alpha = lookahead_alpha;
    hipLaunchKernelGGL(radam_kernel, dim3(num_chunks), dim3(256), 0, as_stream(stream), ptrs, sizes,
                       chunks, a, lookahead_alpha > 0.f ? 1 : 0);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_grad_centralize(float *grad, int rows, int row_len, void *stream)
{
    if (!grad || rows < 1 || row_len < 1) return DVSOF_EINVAL;
    hipLaunchKernelGGL(grad_centralize_kernel, dim3(rows), dim3(256), 0, as_stream(stream), grad,
                       row_len);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
