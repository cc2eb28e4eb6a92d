// Data augmentation on the device: horizontal flip -> nearest-neighbour LUT
// rotation -> crop, for frames and for event coordinates.
//
// Replaces (reference, CPU DataLoader workers): utils/dataset.py:753-769 (the
// order flip, rotate, crop), utils/data.py:155-220 (RandomRotation: rotated
// pixel o reads source pixel rint(R (o - c) + c), float64) with the native
// utils.transformation.map for the events, utils/data.py:24-42 (EventCrop) and
// :45-117 (image crops).  Events the rotation or the crop removes are not
// compacted away: they get x = y = -1, which the voxeliser ignores (no dynamic
// shapes, no host sync).  Several rotated pixels may read the same source
// pixel; an event there moves to the SMALLEST such pixel (atomicMin while the
// LUT is built) -- the native map's rule is unknown (source absent).
//
// HBM-bound, trivially parallel: one thread per output pixel / event.
#include "common.h"

namespace {

constexpr int NT = 256;

// source pixel of rotated pixel (oy, ox); the arithmetic order is numpy's
// elementwise c*x + (-s)*y + W/2 in float64 (no FMA: -ffp-contract=off)
__device__ __forceinline__ bool rot_src(double c, double s, int H, int W, int oy, int ox, int &sy,
                                        int &sx)
{
    const double x = (double)ox - (double)W / 2.0, y = (double)oy - (double)H / 2.0;
    const double fx = rint(c * x + (-s) * y + (double)W / 2.0);
    const double fy = rint(s * x + c * y + (double)H / 2.0);
    if (!(fx >= 0.0 && fx < (double)W && fy >= 0.0 && fy < (double)H)) return false;
    sx = (int)fx;
    sy = (int)fy;
    return true;
}

__global__ __launch_bounds__(NT) void aug_lut_kernel(const double *__restrict__ cs, int B, int H, int W,
                                                     int32_t *__restrict__ lut)
{
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= hw * B) return;
    const int b = (int)(i / hw), o = (int)(i - (long long)b * hw);
    int sy, sx;
    if (rot_src(cs[2 * b], cs[2 * b + 1], H, W, o / W, o % W, sy, sx))
        atomicMin(&lut[(long long)b * hw + (long long)sy * W + sx], o);
}

template <typename T>
__global__ __launch_bounds__(NT) void aug_frames_kernel(const T *__restrict__ src, int D, int H, int W,
                                                        const int32_t *__restrict__ frame_sample,
                                                        const uint8_t *__restrict__ flip,
                                                        const double *__restrict__ cs,
                                                        const int32_t *__restrict__ box, int h, int w,
                                                        float *__restrict__ dst)
{
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= (long long)D * h * w) return;
    const int x = (int)(i % w);
    const long long t = i / w;
    const int y = (int)(t % h), d = (int)(t / h);
    const int b = frame_sample[d];
    const int oy = y + box[4 * b], ox = x + box[4 * b + 1];
    float v = 0.f;
    int sy, sx;
    if (oy < H && ox < W && rot_src(cs[2 * b], cs[2 * b + 1], H, W, oy, ox, sy, sx)) {
        if (flip[b]) sx = W - 1 - sx;
        v = (float)src[((size_t)d * H + sy) * W + sx];
    }
    dst[i] = v;
}

__global__ __launch_bounds__(NT) void aug_events_kernel(const int64_t *__restrict__ x,
                                                        const int64_t *__restrict__ y,
                                                        const int64_t *__restrict__ sample, int64_t n,
                                                        const uint8_t *__restrict__ flip,
                                                        const int32_t *__restrict__ lut,
                                                        const int32_t *__restrict__ box, int B, int H,
                                                        int W, int64_t *__restrict__ xo,
                                                        int64_t *__restrict__ yo)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int64_t b = sample[i], xi = x[i], yi = y[i];
    int64_t nx = -1, ny = -1;
    if (b >= 0 && b < B && xi >= 0 && xi < W && yi >= 0 && yi < H) {
        const int64_t xf = flip[b] ? W - 1 - xi : xi;
        int64_t o = yi * W + xf;
        if (lut) {
            const int32_t v = lut[b * (int64_t)H * W + o];
            o = v < H * W ? v : -1;
        }
        if (o >= 0) {
            const int64_t oy = o / W, ox = o - oy * W;
            const int y0 = box[4 * b], x0 = box[4 * b + 1], bh = box[4 * b + 2], bw = box[4 * b + 3];
            if (ox >= x0 && ox < x0 + bw && oy >= y0 && oy < y0 + bh) {
                nx = ox - x0;
                ny = oy - y0;
            }
        }
    }
    xo[i] = nx;
    yo[i] = ny;
}

}  // namespace

extern "C" {

int dvsof_augment_lut(const double *cos_sin, int B, int H, int W, int32_t *lut, void *stream)
{
    if (!cos_sin || !lut || B < 1 || H < 1 || W < 1 || (long long)H * W > 0x7f000000LL)
        return DVSOF_EINVAL;
    hipStream_t st = as_stream(stream);
    const long long n = (long long)B * H * W;
    DVSOF_HIP_TRY((hipError_t)fill_u32(lut, 0x7f7f7f7fu, sizeof(int32_t) * (size_t)n, st));   // 0x7f7f7f7f: empty
    hipLaunchKernelGGL(aug_lut_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, cos_sin, B, H,
                       W, lut);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_augment_frames(const void *src, int src_is_u8, int D, int H, int W,
                         const int32_t *frame_sample, const uint8_t *flip, const double *cos_sin,
                         const int32_t *box, int h, int w, float *dst, void *stream)
{
    if (!src || !frame_sample || !flip || !cos_sin || !box || !dst || D < 0 || H < 1 || W < 1 ||
        h < 1 || w < 1)
        return DVSOF_EINVAL;
    if (D == 0) return DVSOF_OK;
    const long long n = (long long)D * h * w;
    const unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (src_is_u8)
        hipLaunchKernelGGL(aug_frames_kernel<uint8_t>, dim3(nb), dim3(NT), 0, as_stream(stream),
                           (const uint8_t *)src, D, H, W, frame_sample, flip, cos_sin, box, h, w, dst);
    else
        hipLaunchKernelGGL(aug_frames_kernel<float>, dim3(nb), dim3(NT), 0, as_stream(stream),
                           (const float *)src, D, H, W, frame_sample, flip, cos_sin, box, h, w, dst);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_augment_events(const int64_t *x, const int64_t *y, const int64_t *sample, int64_t n,
                         const uint8_t *flip, const int32_t *lut, const int32_t *box, int B, int H,
                         int W, int64_t *x_out, int64_t *y_out, void *stream)
{
    if (n < 0 || !flip || !box || B < 1 || H < 1 || W < 1) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    if (!x || !y || !sample || !x_out || !y_out) return DVSOF_EINVAL;
    hipLaunchKernelGGL(aug_events_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                       as_stream(stream), x, y, sample, n, flip, lut, box, B, H, W, x_out, y_out);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
