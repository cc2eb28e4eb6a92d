// Event stream -> per-pixel count image and polarity-signed voxel grid.
//
// Replaces (reference paths): get_count_image utils/data.py:120-136 and the
// EV_FlowNet quantization layer as called at utils/training.py:59-64 /
// scripts/quantize_preprocessed.py:87-91 (source absent upstream; arithmetic
// per docs/VOXEL_SPEC.md, restated on the CPU in oracle/dvsof_oracle.c).
//
// HBM-bound integer/byte work.  v1: one thread per event, coalesced reads of
// the five event columns, scatter with memory-side atomics into the grid that
// was zero-filled on the same stream.
#include "common.h"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void count_image_kernel(const int64_t *__restrict__ x,
                                                         const int64_t *__restrict__ y,
                                                         int64_t n, int H, int W,
                                                         uint32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t xi = x[i], yi = y[i];
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(&out[yi * W + xi], 1u);
    }
}

__global__ __launch_bounds__(NT) void voxelize_kernel(
    const int64_t *__restrict__ x, const int64_t *__restrict__ y, const float *__restrict__ t,
    const int64_t *__restrict__ pol, const int64_t *__restrict__ sample, int64_t n,
    const float *__restrict__ t0, const float *__restrict__ t1, int B, int C, int H, int W,
    float *__restrict__ out, int32_t *__restrict__ bin0, int64_t *__restrict__ lin0)
{
    const int64_t stride = (int64_t)gridDim.x * NT;
    const size_t plane = (size_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t b = sample[i], xi = x[i], yi = y[i];
        int c0 = -1;
        int64_t lin = -1;
        if (b >= 0 && b < B && xi >= 0 && xi < W && yi >= 0 && yi < H) {
            const float ts = t[i], lo = t0[b], hi = t1[b];
            if (ts >= lo && ts <= hi) {
                const float dt = hi - lo;
                const float tn = dt > 0.f ? ((ts - lo) / dt) * (float)(C - 1) : 0.f;
                c0 = min((int)floorf(tn), C - 1);
                const float f = tn - (float)c0;
                const float p = (float)pol[i];
                lin = (int64_t)((((size_t)b * C + c0) * H + (size_t)yi) * W + (size_t)xi);
                atomicAdd(&out[lin], p * (1.f - f));
                if (c0 + 1 < C) atomicAdd(&out[lin + plane], p * f);
            }
        }
        if (bin0) bin0[i] = c0;
        if (lin0) lin0[i] = lin;
    }
}

int grid_for(int64_t n) { return (int)((n + NT - 1) / NT < 2048 ? (n + NT - 1) / NT : 2048); }

}  // namespace

extern "C" {

int dvsof_count_image(const int64_t *x, const int64_t *y, int64_t n, int H, int W, uint32_t *out,
                      void *stream)
{
    if (!out || H < 1 || W < 1 || n < 0 || (n > 0 && (!x || !y))) return DVSOF_EINVAL;
    DVSOF_HIP_TRY(hipMemsetAsync(out, 0, sizeof(uint32_t) * (size_t)H * W, as_stream(stream)));
    if (n == 0) return DVSOF_OK;
    hipLaunchKernelGGL(count_image_kernel, dim3(grid_for(n)), dim3(NT), 0, as_stream(stream), x, y,
                       n, H, W, out);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_voxelize_fwd(const int64_t *x, const int64_t *y, const float *t, const int64_t *pol,
                       const int64_t *sample, int64_t n, const float *t0, const float *t1, int B,
                       int C, int H, int W, float *out, int32_t *bin0, int64_t *lin0, void *stream)
{
    if (!out || B < 1 || C < 1 || H < 1 || W < 1 || n < 0 || !t0 || !t1) return DVSOF_EINVAL;
    if (n > 0 && (!x || !y || !t || !pol || !sample)) return DVSOF_EINVAL;
    DVSOF_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * H * W, as_stream(stream)));
    if (n == 0) return DVSOF_OK;
    hipLaunchKernelGGL(voxelize_kernel, dim3(grid_for(n)), dim3(NT), 0, as_stream(stream), x, y, t,
                       pol, sample, n, t0, t1, B, C, H, W, out, bin0, lin0);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
