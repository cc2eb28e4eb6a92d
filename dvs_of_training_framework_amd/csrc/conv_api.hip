// C-ABI entry points of the predictor's convolution stack and the small
// HBM-bound kernels around it (weight flip-transpose, 1x1 flow head forward /
// backward fused with the activation backward, elementwise act backward).
#include "conv_common.h"


int gconv_launch(const GConvParams &P, int tile_hint, hipStream_t st);
int wgrad_launch(WGradParams P, float *dW, float *dbias, float *ws, size_t ws_floats,
                 const FlatWG *flat, int nflat, hipStream_t st);
size_t wgrad_flat_workspace_floats(const FlatWG *flat, int nflat);
size_t wgrad_workspace_floats(const WGradParams &P, bool with_bias);
int gconv_pick_tile(long long m, long long n);
int wgrad_splits(const WGradParams &P0, int *tile_out);
bool wino_eligible_shape(int nsrc, int layout_nhwc, int C, int N, int B, int H, int W, int ksize,
                         int stride, int pad, int upsample, int mfma);
int wino_components(int B, int H, int W, int mfma);
size_t wino_scratch_floats(int B, int H, int W, int C, int N, int mfma);
int wino_prepare(const float *weight, float *U, float *Ut, int N, int C, int B, int H, int W, int mfma,
                 hipStream_t st);
int wino_launch(const GConvParams &P, float *scratch, size_t scratch_floats, const WinoChain &ch, hipStream_t st);
bool wino_chain_ok(int B, int H, int W, int N, int mfma);
int wino_wgrad_tile(int B, int H, int W, int mfma);
size_t wino_wgrad_workspace_floats(int B, int H, int W, int C, int N, int mfma);
int wino_tile(int B, int H, int W, int mfma);
int wino_wgrad_launch(const GSrc &X, const float *V_in, const float *Z_in, const float *gout, float *dW, float *dbias, int B,
                      int H, int W, int C, int N, int mfma_bf16, float *ws, size_t ws_floats,
                      hipStream_t st);

// first.hip: the first encoder layer (planar voxel input, K = 9 C) as kernels of its own
bool first_layer_shape(int nsrc, int planar, int C, int Cout, int H, int W, int ksize, int stride,
                       int pad, int upsample);
int first_fwd_launch(const float *x, int B, int C, int H, int W, const float *w, const float *bias,
                     int act, float *y, float *z, unsigned short *y16, hipStream_t st);
size_t first_wgrad_workspace_floats(int B, int C, int H, int W);
int first_wgrad_launch(const float *x, int B, int C, int H, int W, const float *gout, float *dW,
                       float *dbias, float *ws, size_t ws_floats, hipStream_t st);

// set by the launches below / read by dvsof_conv2d_last_patch (profiling tools)
static thread_local int t_last_patch[3] = {0, 0, 0};
void conv_note_patch(int kind, int what) { t_last_patch[kind] = what; }

// fwd_patch.hip: forward of the finest decoder stage in the bf16-twins mode (patch in LDS,
// weights in registers)
bool fwd_patch_eligible(const GConvParams &P);
int fwd_patch_launch(const GConvParams &P, hipStream_t st);
// fwd_min.hip: the nine-product form of `nearest-up2 -> conv3x3` (exact f32)
bool min9_shape_ok(int mfma, int nsrc, const int *C, const int *nhwc, int Cout, int H, int W);
int min9_prepare_fwd(const float *w, float *wt, int Cout, int Ctot, hipStream_t st);
int fwd_min_launch(const GConvParams &P, hipStream_t st);
// dgrad_min.hip: its data gradient, nine products too (prepared form W'[9][Ctot][Cout])
bool min9_dgrad_shape_ok(const int *C, int Cout, int H);
int min9_prepare_dgrad(const float *w, float *wq, int Cout, int Ctot, hipStream_t st);
int dgrad_min_launch(const GConvParams &P, hipStream_t st);

namespace {

bool is_first_layer(const dvsof_conv_desc_t *d)
{
    return first_layer_shape(d->nsrc, d->src[0].layout == DVSOF_NCHW, d->src[0].C, d->Cout, d->H, d->W,
                             d->ksize, d->stride, d->pad, d->upsample) && !d->bias_cls;
}

GSrc make_src(const float *p, int C, int layout, int H, int W, const void *p16 = nullptr)
{
    GSrc s;
    s.p = p;
    s.p16 = layout == DVSOF_NHWC ? (const unsigned short *)p16 : nullptr;
    s.C = C;
    if (layout == DVSOF_NCHW) {
        s.sb = (long long)C * H * W;
        s.sy = W;
        s.sx = 1;
        s.sc = H * W;
        s.flat = 1;
    } else {
        s.sb = (long long)H * W * C;
        s.sy = W * C;
        s.sx = C;
        s.sc = 1;
        s.flat = ((C & 3) || C < BK) ? 1 : 0;
    }
    return s;
}

bool desc_ok(const dvsof_conv_desc_t *d, int &Ctot, int &Ho, int &Wo)
{
    if (!d || d->nsrc < 1 || d->nsrc > 3 || d->B < 1 || d->H < 1 || d->W < 1) return false;
    if (d->ksize != 1 && d->ksize != 3 && d->ksize != 5) return false;
    if (d->stride != 1 && d->stride != 2) return false;
    if (d->upsample < 0 || d->upsample > 2) return false;
    if (d->upsample && d->stride != 1) return false;
    // upsample = 2: zero insertion (transposed convolution).  One NHWC source of
    // whole 16-channel K slices, 3x3 taps, pad 1: the phased MFMA form below
    if (d->upsample == 2 &&
        !(d->ksize == 3 && d->pad == 1 && d->nsrc == 1 && d->src[0].layout == DVSOF_NHWC &&
          d->src[0].C % BK == 0 && d->mfma != 3))
        return false;
    if (d->Cout < 1 || d->pad < 0 || d->pad >= d->ksize) return false;
    Ctot = 0;
    for (int i = 0; i < d->nsrc; ++i) {
        if (!d->src[i].p || d->src[i].C < 1) return false;
        Ctot += d->src[i].C;
    }
    const int up = d->upsample ? 2 : 1;
    Ho = (d->H * up + 2 * d->pad - d->ksize) / d->stride + 1;
    Wo = (d->W * up + 2 * d->pad - d->ksize) / d->stride + 1;
    if (Ho < 1 || Wo < 1) return false;
    if ((long long)d->B * Ho * Wo > 0x7fffffffLL) return false;
    if ((long long)d->B * d->H * up * d->W * up > 0x7fffffffLL) return false;
    return true;
}

bool is_subpixel(const dvsof_conv_desc_t *d)
{   // up2 + 3x3/pad1/stride1 == four 2x2 phase convolutions on the low-res input
    return d->upsample == 1 && d->ksize == 3 && d->pad == 1 && d->stride == 1;
}

// ... and of those the ones whose FORWARD runs the nine-product minimal algorithm
// (fwd_min.hip): the prepared forward form is then Wt[9][Cout][Ctot] = G w G^T
bool is_min9(const dvsof_conv_desc_t *d)
{
    if (!is_subpixel(d)) return false;
    int C[3] = {0, 0, 0}, nhwc[3] = {0, 0, 0};
    for (int i = 0; i < d->nsrc && i < 3; ++i) {
        C[i] = d->src[i].C;
        nhwc[i] = d->src[i].layout == DVSOF_NHWC;
    }
    return min9_shape_ok(d->mfma, d->nsrc, C, nhwc, d->Cout, d->H, d->W);
}

bool is_min9_dgrad(const dvsof_conv_desc_t *d)
{
    if (!is_min9(d)) return false;
    const int C[2] = {d->src[0].C, d->src[1].C};
    return min9_dgrad_shape_ok(C, d->Cout, d->H);
}

// zero-insertion 2x + 3x3/pad 1 = transposed convolution with stride 2:
// y[Y][X] = sum_k W[ky][kx] xz[Y+ky-1][X+kx-1], xz[2i][2j] = x[i][j], else 0
// (torch: conv_transpose2d(x, W.flip(2,3).transpose(0,1), stride 2, padding 1,
// output_padding 1)).  Evaluated as four output-parity phases of (1+py)(1+px)
// taps on the low-resolution input -- the adjoint of the phased stride-2 layer.
bool is_transposed(const dvsof_conv_desc_t *d) { return d->upsample == 2; }

// wide 3x3 stride-1 layer evaluated as Winograd F(2x2,3x3) (winograd.hip)
bool is_wino(const dvsof_conv_desc_t *d)
{
    return d->nsrc == 1 &&
           wino_eligible_shape(d->nsrc, d->src[0].layout == DVSOF_NHWC, d->src[0].C, d->Cout, d->B,
                               d->H, d->W, d->ksize, d->stride, d->pad, d->upsample, d->mfma);
}

// ... and its weight gradient too: the tile count is the K dimension of the
// K-major kernel, which loads whole 16-element groups
bool is_wino_wgrad(const dvsof_conv_desc_t *d)
{
    static const bool off = getenv("DVSOF_NO_WINOGRAD_WGRAD") != nullptr;
    return !off && is_wino(d) && wino_wgrad_tile(d->B, d->H, d->W, d->mfma == 2 ? 2 : 0) != 0;
}

// Wf[ph][co][a][b][ci] = sum_{ky in S(py,a)} sum_{kx in S(px,b)} W[co][ky][kx][ci]
// S(0,0)={0} S(0,1)={1,2} S(1,0)={0,1} S(1,1)={2}; ph = 2*py + px.
// (every weight-form kernel below takes an optional bf16 destination: the twin
// of the form it writes, for compute_dtype 'bf16s' -- a separate conversion
// launch per form was 23 launches and 165 us of a 1.94 ms step at batch 8)
__global__ __launch_bounds__(256) void subpixel_fwd_weights_kernel(const float *__restrict__ w,
                                                                   float *__restrict__ wf, int Cout,
                                                                   int Ctot,
                                                                   unsigned short *__restrict__ wf16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Cout * Ctot) return;
    const int ci = (int)(i % Ctot), co = (int)(i / Ctot);
    float k[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t / 3][t % 3] = w[((size_t)co * 9 + t) * Ctot + ci];
    // row/column partial sums for (p, a): {0},{1,2},{0,1},{2}
    float r[4][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        r[0][x] = k[0][x];
        r[1][x] = k[1][x] + k[2][x];
        r[2][x] = k[0][x] + k[1][x];
        r[3][x] = k[2][x];
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float *rr = r[2 * py + a];
                    const int q = 2 * px + b;
                    const float v = q == 0 ? rr[0] : q == 1 ? rr[1] + rr[2] : q == 2 ? rr[0] + rr[1] : rr[2];
                    const size_t o = ((((size_t)(2 * py + px) * Cout + co) * 2 + a) * 2 + b) * Ctot + ci;
                    wf[o] = v;
                    if (wf16) wf16[o] = bf16_bits(v);
                }
}

// Wd[ci][ty][tx][co] = Wf[ph][co][a][b][ci], ty -> (py,a): 0->(1,1) 1->(0,1) 2->(1,0) 3->(0,0)
__global__ __launch_bounds__(256) void subpixel_dgrad_weights_kernel(const float *__restrict__ wf,
                                                                     float *__restrict__ wd,
                                                                     int Cout, int Ctot,
                                                                     unsigned short *__restrict__ wd16)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z, ty = z >> 2, tx = z & 3;
    const int py = (ty == 0 || ty == 2) ? 1 : 0, a = ty < 2 ? 1 : 0;
    const int px = (tx == 0 || tx == 2) ? 1 : 0, b = tx < 2 ? 1 : 0;
    const float *in = wf + ((size_t)(2 * py + px) * Cout * 4 + (a * 2 + b)) * Ctot;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + lx;
        tile[r][lx] = (co < Cout && ci < Ctot) ? in[(size_t)co * 4 * Ctot + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + lx;
        if (ci < Ctot && co < Cout) {
            const size_t o = ((size_t)ci * 16 + z) * Cout + co;
            wd[o] = tile[lx][r];
            if (wd16) wd16[o] = bf16_bits(tile[lx][r]);
        }
    }
}

// The same Wd straight from the RAW weights w[co][3][3][ci] (a caller that no longer holds
// the phase kernels, or whose forward form is another one: fwd_min.hip): per axis tap t of
// the 4x4 kernel sums the raw taps {2}, {1,2}, {0,1}, {0} -- the same additions in the same
// order as subpixel_fwd_weights_kernel, so both routes give the same bits.
__global__ __launch_bounds__(256) void subpixel_dgrad_weights_raw_kernel(const float *__restrict__ w,
                                                                         float *__restrict__ wd,
                                                                         int Cout, int Ctot,
                                                                         unsigned short *__restrict__ wd16)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z, ty = z >> 2, tx = z & 3;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + lx;
        float v = 0.f;
        if (co < Cout && ci < Ctot) {
            const float *k = w + (size_t)co * 9 * Ctot + ci;
            // (rows are summed FIRST in subpixel_fwd_weights_kernel, then columns)
            auto col = [&](int kx) -> float {
                const float *kc = k + (size_t)kx * Ctot;
                return ty == 0 ? kc[6 * Ctot] : ty == 1 ? kc[3 * Ctot] + kc[6 * Ctot]
                     : ty == 2 ? kc[0] + kc[3 * Ctot] : kc[0];
            };
            v = tx == 0 ? col(2) : tx == 1 ? col(1) + col(2) : tx == 2 ? col(0) + col(1) : col(0);
        }
        tile[r][lx] = v;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + lx;
        if (ci < Ctot && co < Cout) {
            const size_t o = ((size_t)ci * 16 + z) * Cout + co;
            wd[o] = tile[lx][r];
            if (wd16) wd16[o] = bf16_bits(tile[lx][r]);
        }
    }
}

// Stride-2 3x3/pad-1 data gradient as four input-parity phases of 2x2 taps:
// Wp[ph][ci][a][b][co] = W[co][ky(py,a)][kx(px,b)][ci], ky(0,0)=1, ky(0,1)=none,
// ky(1,0)=2, ky(1,1)=0 (unused taps are zero).  ph = 2*py + px.
// w16: bf16 twin of the RAW weights (each raw tap is read by exactly one z)
__global__ __launch_bounds__(256) void stride2_dgrad_weights_kernel(const float *__restrict__ w,
                                                                    float *__restrict__ wp,
                                                                    int Cout, int Ctot,
                                                                    unsigned short *__restrict__ wp16,
                                                                    unsigned short *__restrict__ w16)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z;            // ph*4 + a*2 + b
    const int ph = z >> 2, a = (z >> 1) & 1, b = z & 1, py = ph >> 1, px = ph & 1;
    const int ky = py ? (a ? 0 : 2) : (a ? -1 : 1), kx = px ? (b ? 0 : 2) : (b ? -1 : 1);
    const bool used = ky >= 0 && kx >= 0;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + lx;
        float v = 0.f;
        if (used && co < Cout && ci < Ctot) {
            const size_t o = ((size_t)co * 9 + ky * 3 + kx) * Ctot + ci;
            v = w[o];
            if (w16) w16[o] = bf16_bits(v);
        }
        tile[r][lx] = v;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + lx;
        if (ci < Ctot && co < Cout) {
            const size_t o = (((size_t)ph * Ctot + ci) * 4 + a * 2 + b) * Cout + co;
            wp[o] = tile[lx][r];
            if (wp16) wp16[o] = bf16_bits(tile[lx][r]);
        }
    }
}

// Transposed-convolution forward as four output-parity phases of 2x2 taps:
// Wt[ph][co][a][b][ci] = W[co][ky(py,a)][kx(px,b)][ci], ky(0,0)=1, ky(0,1)=none,
// ky(1,0)=0, ky(1,1)=2 (unused taps are zero).  ph = 2*py + px.
__global__ __launch_bounds__(256) void transposed_fwd_weights_kernel(const float *__restrict__ w,
                                                                     float *__restrict__ wt,
                                                                     int Cout, int Ctot,
                                                                     unsigned short *__restrict__ wt16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Cout * Ctot) return;
    const int ci = (int)(i % Ctot), co = (int)(i / Ctot);
#pragma unroll
    for (int z = 0; z < 16; ++z) {
        const int ph = z >> 2, a = (z >> 1) & 1, b = z & 1, py = ph >> 1, px = ph & 1;
        const int ky = py ? (a ? 2 : 0) : (a ? -1 : 1), kx = px ? (b ? 2 : 0) : (b ? -1 : 1);
        const float v = (ky >= 0 && kx >= 0) ? w[((size_t)co * 9 + ky * 3 + kx) * Ctot + ci] : 0.f;
        const size_t o = ((((size_t)ph * Cout + co) * 2 + a) * 2 + b) * Ctot + ci;
        wt[o] = v;
        if (wt16) wt16[o] = bf16_bits(v);
    }
}

// per-channel sum of an NHWC tensor, fixed order: partial sums per workgroup
// (64 pixels per pass and lane group), then one wave per channel
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float *__restrict__ g,
                                                                  long long npix, int C,
                                                                  float *__restrict__ part)
{
    // thread t owns channel c = t % C of pixel rows t / C, t / C + 256 / C, ... (C <= 256)
    const int per = 256 / C, c = threadIdx.x % C, r = threadIdx.x / C;
    float a = 0.f;
    if (r < per)
        for (long long p = (long long)blockIdx.x * per + r; p < npix; p += (long long)gridDim.x * per)
            a += g[p * C + c];
    __shared__ float sm[256];
    sm[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int j = 0; j < per; ++j) t += sm[j * C + threadIdx.x];
        part[(size_t)blockIdx.x * C + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float *__restrict__ part,
                                                                int nblocks, int C,
                                                                float *__restrict__ out)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double a = 0;
    for (int b = lane; b < nblocks; b += 64) a += (double)part[(size_t)b * C + c];
    a = wave_sum(a);
    if (lane == 0) out[c] = (float)a;
}

bool is_stride2_phased(const dvsof_conv_desc_t *d)
{
    return !d->upsample && d->stride == 2 && d->ksize == 3 && d->pad == 1 &&
           (d->H % 2 == 0) && (d->W % 2 == 0);
}

__global__ __launch_bounds__(256) void flip_transpose_kernel(const float *__restrict__ w,
                                                             float *__restrict__ wt, int Cout,
                                                             int taps, int Ctot,
                                                             unsigned short *__restrict__ wt16,
                                                             unsigned short *__restrict__ w16)
{
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Ctot) {
            const size_t o = ((size_t)co * taps + tap) * Ctot + ci;
            v = w[o];
            if (w16) w16[o] = bf16_bits(v);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Ctot && co < Cout) {
            const size_t o = ((size_t)ci * taps + (taps - 1 - tap)) * Cout + co;
            wt[o] = tile[tx][r];
            if (wt16) wt16[o] = bf16_bits(tile[tx][r]);
        }
    }
}

// ---- flow head ------------------------------------------------------------
// LPP = C/4 lanes share one pixel (one float4 of channels each).

template <int LPP>
__device__ __forceinline__ void head_fwd_body(const float *__restrict__ x, const float *__restrict__ w,
                                              const float *__restrict__ bias, float *__restrict__ flow,
                                              int B, int HW, int vblock, int nblocks)
{
    constexpr int C = LPP * 4, PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, sub = lane % LPP, pw = lane / LPP;
    const f32x4 w0 = *(const f32x4u *)(w + 4 * sub), w1 = *(const f32x4u *)(w + C + 4 * sub);
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
    const long long total = (long long)B * HW;
    const long long wave_id = (long long)vblock * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)nblocks * 4;
    for (long long base = wave_id * PPW; base < total; base += nwaves * PPW) {
        const long long pix = base + pw;
        float p0 = 0.f, p1 = 0.f;
        if (pix < total) {
            const f32x4 v = *(const f32x4u *)(x + pix * C + 4 * sub);
            p0 = v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
            p1 = v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
        }
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) {
            p0 += __shfl_xor(p0, off, 64);
            p1 += __shfl_xor(p1, off, 64);
        }
        if (sub == 0 && pix < total) {
            const long long b = pix / HW, r = pix - b * HW;
            flow[(b * 2) * HW + r] = p0 + b0;
            flow[(b * 2 + 1) * HW + r] = p1 + b1;
        }
    }
}

template <int LPP>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ w,
                                                       const float *__restrict__ bias,
                                                       float *__restrict__ flow, int B, int HW)
{
    head_fwd_body<LPP>(x, w, bias, flow, B, HW, blockIdx.x, gridDim.x);
}

// Up to 4 heads in ONE launch (the training forward with the flow member folded:
// nothing between the decoder stages reads a flow, so all of them are computed
// ahead of the loss; a launch of this size is mostly its ~4.5 us of dispatch)
constexpr int HEADS_MAX = 4;
struct HeadsFwd {
    const float *x[HEADS_MAX], *w[HEADS_MAX], *bias[HEADS_MAX];
    float *flow[HEADS_MAX];
    int HW[HEADS_MAX], C[HEADS_MAX], block_begin[HEADS_MAX + 1];
    int B, n;
};
__global__ __launch_bounds__(256) void heads_fwd_kernel(const HeadsFwd A)
{
    int h = 0;
#pragma unroll
    for (int i = 1; i < HEADS_MAX; ++i)
        if (i < A.n && (int)blockIdx.x >= A.block_begin[i]) h = i;
    const int vb = blockIdx.x - A.block_begin[h], nb = A.block_begin[h + 1] - A.block_begin[h];
    switch (A.C[h] / 4) {
    case 4: head_fwd_body<4>(A.x[h], A.w[h], A.bias[h], A.flow[h], A.B, A.HW[h], vb, nb); break;
    case 8: head_fwd_body<8>(A.x[h], A.w[h], A.bias[h], A.flow[h], A.B, A.HW[h], vb, nb); break;
    case 16: head_fwd_body<16>(A.x[h], A.w[h], A.bias[h], A.flow[h], A.B, A.HW[h], vb, nb); break;
    case 32: head_fwd_body<32>(A.x[h], A.w[h], A.bias[h], A.flow[h], A.B, A.HW[h], vb, nb); break;
    default: head_fwd_body<64>(A.x[h], A.w[h], A.bias[h], A.flow[h], A.B, A.HW[h], vb, nb); break;
    }
}

template <int LPP>
__global__ __launch_bounds__(256) void head_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ gflow,
    const float *gx_in, const float *__restrict__ actsrc, int act, float *gx,
    float *__restrict__ part, int B, int HW, unsigned short *__restrict__ gx16)
{
    constexpr int C = LPP * 4, PPW = 64 / LPP;
    __shared__ float red[4][2 * C + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPP, pw = lane / LPP;
    const f32x4 w0 = *(const f32x4u *)(w + 4 * sub), w1 = *(const f32x4u *)(w + C + 4 * sub);
    const long long total = (long long)B * HW;
    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long nwaves = (long long)gridDim.x * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;
    for (long long base = wave_id * PPW; base < total; base += nwaves * PPW) {
        const long long pix = base + pw;
        if (pix < total) {
            const long long b = pix / HW, r = pix - b * HW;
            const float g0 = gflow[(b * 2) * HW + r], g1 = gflow[(b * 2 + 1) * HW + r];
            const size_t o = (size_t)pix * C + 4 * sub;
            const f32x4 v = *(const f32x4u *)(x + o);
            a0 += g0 * v;
            a1 += g1 * v;
            if (sub == 0) {
                s0 += g0;
                s1 += g1;
            }
            if (!gx) continue;      // weight / bias gradient only (wave-uniform)
            f32x4 g = g0 * w0 + g1 * w1;
            if (gx_in) g += *(const f32x4u *)(gx_in + o);
            if (actsrc) {
                const f32x4 sv = *(const f32x4u *)(actsrc + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] *= act_bwd(sv[i], act);
            }
            *(f32x4u *)(gx + o) = g;
            if (gx16) {     // bf16 twin for the data gradient that reads it next
                typedef unsigned short u16x4 __attribute__((ext_vector_type(4), aligned(2)));
                const u16x4 h = {bf16_bits(g[0]), bf16_bits(g[1]), bf16_bits(g[2]), bf16_bits(g[3])};
                *(u16x4 *)(gx16 + o) = h;
            }
        }
    }
    // lanes with equal `sub` hold partial sums of the same channels
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] += __shfl_xor(a0[i], off, 64);
            a1[i] += __shfl_xor(a1[i], off, 64);
        }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (pw == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[wave][4 * sub + i] = a0[i];
            red[wave][C + 4 * sub + i] = a1[i];
        }
    }
    if (lane == 0) {
        red[wave][2 * C] = s0;
        red[wave][2 * C + 1] = s1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C + 2; i += 256)
        part[(size_t)blockIdx.x * (2 * C + 2) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// dw[2*C], dbias[2] from the per-workgroup partials, fixed order: one wave
// per output column, lanes stride over the workgroups, shuffle tree.
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const float *__restrict__ part,
                                                              int nblocks, int C, float *dw,
                                                              float *dbias)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= 2 * C + 2) return;
    double a = 0;
    for (int b = lane; b < nblocks; b += 64) a += (double)part[(size_t)b * (2 * C + 2) + i];
    a = wave_sum(a);
    if (lane == 0) {
        if (i < 2 * C) dw[i] = (float)a;
        else if (dbias) dbias[i - 2 * C] = (float)a;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *dy,
                                                      const float *__restrict__ actsrc, int act,
                                                      float *dz, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 g = *(const f32x4u *)(dy + i);
            const f32x4 s = *(const f32x4u *)(actsrc + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] *= act_bwd(s[j], act);
            *(f32x4u *)(dz + i) = g;
        } else {
            for (size_t j = i; j < n; ++j) dz[j] = dy[j] * act_bwd(actsrc[j], act);
        }
    }
}

// bf16 twins of prepared weights: 8 elements per thread and iteration
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *__restrict__ src,
                                                      unsigned short *__restrict__ dst, size_t n)
{
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8), aligned(4)));
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 7 < n) {
            const f32x4 a = *(const f32x4u *)(src + i), b = *(const f32x4u *)(src + i + 4);
            const u16x8 h = {bf16_bits(a[0]), bf16_bits(a[1]), bf16_bits(a[2]), bf16_bits(a[3]),
                             bf16_bits(b[0]), bf16_bits(b[1]), bf16_bits(b[2]), bf16_bits(b[3])};
            *(u16x8 *)(dst + i) = h;
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = bf16_bits(src[j]);
        }
    }
}

// several tensors in one launch (the raw-weight twins of a step): a workgroup
// converts 2048 elements of the tensor its index falls into
constexpr int BF16_MANY_MAX = 16;
struct Bf16Many {
    const float *src[BF16_MANY_MAX];
    unsigned short *dst[BF16_MANY_MAX];
    size_t n[BF16_MANY_MAX];
    unsigned block_begin[BF16_MANY_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void to_bf16_many_kernel(const Bf16Many J)
{
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8), aligned(4)));
    int t = 0;
#pragma unroll
    for (int i = 1; i < BF16_MANY_MAX; ++i)
        if (i < J.count && blockIdx.x >= J.block_begin[i]) t = i;
    const float *src = J.src[t];
    unsigned short *dst = J.dst[t];
    const size_t n = J.n[t];
    const size_t i = ((size_t)(blockIdx.x - J.block_begin[t]) * 256 + threadIdx.x) * 8;
    if (i + 7 < n) {
        const f32x4 a = *(const f32x4u *)(src + i), b = *(const f32x4u *)(src + i + 4);
        const u16x8 h = {bf16_bits(a[0]), bf16_bits(a[1]), bf16_bits(a[2]), bf16_bits(a[3]),
                         bf16_bits(b[0]), bf16_bits(b[1]), bf16_bits(b[2]), bf16_bits(b[3])};
        *(u16x8 *)(dst + i) = h;
    } else {
        for (size_t j = i; j < n; ++j) dst[j] = bf16_bits(src[j]);
    }
}

// workgroups of the head backward = partial rows of its weight-gradient reduce
int head_blocks(long long total, int lpp)
{
    static const int cap = getenv("DVSOF_HEAD_BLOCKS") ? atoi(getenv("DVSOF_HEAD_BLOCKS")) : 512;
    const long long per_block = 4LL * (64 / lpp);
    long long nb = (total + per_block - 1) / per_block;
    return (int)(nb < cap ? (nb < 1 ? 1 : nb) : cap);
}

void fill_wgrad(const dvsof_conv_desc_t *d, int Ctot, int Ho, int Wo, WGradParams &P)
{
    P.mfma_bf16 = d->mfma == 3 ? 1 : (d->mfma == 1 || d->mfma == 2) ? d->mfma : 0;   // twins: f32 tensors, rounded operands
    const int up = d->upsample ? 2 : 1;
    P.nsrc = d->nsrc;
    for (int i = 0; i < d->nsrc; ++i)
        P.src[i] = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W,
                            d->mfma == 3 ? d->src[i].p16 : nullptr);
    // bf16 twins (mode 3): gout and every vector member streamed from their bf16
    // copies when all of them exist and channel runs are whole 16-byte loads
    P.gout16 = d->mfma == 3 ? (const unsigned short *)d->gout16 : nullptr;
    P.twins = P.gout16 != nullptr && (d->Cout % 8) == 0;
    {
        static const bool off = getenv("DVSOF_WGRAD_NO_TWINS") != nullptr;
        if (off) P.twins = 0;
    }
    for (int i = 0; i < d->nsrc; ++i)
        if (!P.src[i].flat && (!P.src[i].p16 || (P.src[i].C % 8))) P.twins = 0;
    P.B = d->B;
    P.Hv = d->H * up;
    P.Wv = d->W * up;
    P.up = d->upsample ? UP_NEAREST : UP_NONE;
    P.Ho = Ho;
    P.Wo = Wo;
    P.stride = d->stride;
    P.pad = d->pad;
    P.ks = d->ksize;
    P.Cout = d->Cout;
    P.Cin_tot = Ctot;
    P.M = d->B * Ho * Wo;
    P.klen = 0;
    P.nph = 1;
    P.ph_pad = 0;
    P.S = 1;
    P.g_sb = (long long)Ho * Wo * d->Cout;
    P.g_sy = Wo * d->Cout;
    P.g_sx = d->Cout;
    P.g_py = P.g_px = 0;
    P.src_ph_stride = 0;
    if (is_subpixel(d)) {   // four 2x2 phase problems over the low-res pixels
        P.Hv = d->H;
        P.Wv = d->W;
        P.up = UP_NONE;
        P.Ho = d->H;
        P.Wo = d->W;
        P.ks = 2;
        P.M = d->B * d->H * d->W;
        P.nph = 4;
        P.ph_pad = 1;
        P.g_sy = 2 * Wo * d->Cout;
        P.g_sx = 2 * d->Cout;
        P.g_py = Wo * d->Cout;
        P.g_px = d->Cout;
    }
    P.gout = nullptr;
    P.dW = nullptr;
    P.dbias = nullptr;
}

int fill_flat(const dvsof_conv_desc_t *d, int Ctot, int Ho, int Wo, const float *gout, FlatWG *F)
{
    int n = 0, coff = 0;
    const int up = d->upsample ? 2 : 1;
    for (int i = 0; i < d->nsrc; ++i) {
        const GSrc g = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W);
        if (g.flat) {
            FlatWG &f = F[n++];
            f.S = g;
            f.gout = gout;
            f.B = d->B;
            f.Hv = d->H * up;
            f.Wv = d->W * up;
            f.up = d->upsample ? UP_NEAREST : UP_NONE;
            f.Ho = Ho;
            f.Wo = Wo;
            f.stride = d->stride;
            f.pad = d->pad;
            f.ks = d->ksize;
            f.Cout = d->Cout;
            f.M = d->B * Ho * Wo;
            f.ncol = d->ksize * d->ksize * g.C;
            f.coff = coff;
            f.Cin_tot = Ctot;
        }
        coff += d->src[i].C;
    }
    return n;
}

}  // namespace

extern "C" {

int dvsof_weight_flip_transpose(const float *w, float *wt, int Cout, int ksize, int Ctot,
                                void *stream);
static int flip_transpose16(const float *w, float *wt, int Cout, int ksize, int Ctot,
                            unsigned short *wt16, unsigned short *w16, void *stream);
size_t dvsof_conv2d_wgrad_workspace_bytes(const dvsof_conv_desc_t *d);
int dvsof_conv2d_wgrad(const dvsof_conv_desc_t *d, const float *gout, float *dweight, float *dbias,
                       void *ws, size_t ws_bytes, void *stream);

int dvsof_conv2d_fwd(const dvsof_conv_desc_t *d, const float *weight, const float *bias,
                     const float *residual, float *y, float *z, void *stream)
{
    int Ctot, Ho, Wo;
    t_last_patch[0] = 0;
    if (!desc_ok(d, Ctot, Ho, Wo) || !weight || !y) return DVSOF_EINVAL;
    if (is_first_layer(d) && !residual)     // exact f32 in every operand mode
        return first_fwd_launch(d->src[0].p, d->B, Ctot, d->H, d->W, weight, bias, d->act, y, z,
                                d->mfma == 3 ? (unsigned short *)d->y16 : nullptr, as_stream(stream));
    GConvParams P = {};
    P.nsrc = d->nsrc;
    for (int i = 0; i < d->nsrc; ++i)
        P.src[i] = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W, d->src[i].p16);
    P.ndst = 1;
    P.dst[0] = {y, residual, nullptr, nullptr, (long long)Ho * Wo * d->Cout, Wo * d->Cout, d->Cout, 1, d->Cout, 0, 0,
                d->mfma == 3 ? (unsigned short *)d->y16 : nullptr};
    P.W = weight;
    P.W16 = d->mfma == 3 ? (const unsigned short *)d->w16 : nullptr;
    P.bias = bias;
    P.bias_cls = d->bias_cls;
    P.out_H = Ho;
    P.out_W = Wo;
    P.zout = z;
    P.B = d->B;
    const int up = d->upsample ? 2 : 1;
    P.Hv = d->H * up;
    P.Wv = d->W * up;
    P.up = d->upsample ? UP_NEAREST : UP_NONE;
    P.Ho = Ho;
    P.Wo = Wo;
    P.stride = d->stride;
    P.pad = d->pad;
    P.ks = d->ksize;
    P.N = d->Cout;
    P.Cin_tot = Ctot;
    P.M = d->B * Ho * Wo;
    P.quad = 0;
    P.act = d->act;
    P.mfma_bf16 = (d->mfma >= 1 && d->mfma <= 3) ? d->mfma : 0;
    P.bwd_act = ACT_NONE;
    P.nph = 1;
    P.ph_pad = 0;
    P.w_phase_stride = 0;
    if (is_subpixel(d)) {   // `weight` is the prepared Wf[4][Cout][2][2][Ctot]
        P.up = UP_NONE;
        P.Hv = d->H;
        P.Wv = d->W;
        P.Ho = d->H;
        P.Wo = d->W;
        P.ks = 2;
        P.M = d->B * d->H * d->W;
        P.nph = 4;
        P.ph_pad = 1;
        P.w_phase_stride = (long long)d->Cout * 4 * Ctot;
        P.dst[0].sy = 2 * Wo * d->Cout;
        P.dst[0].sx = 2 * d->Cout;
        P.dst[0].ph_y = Wo * d->Cout;
        P.dst[0].ph_x = d->Cout;
    }
    if (is_transposed(d)) {   // `weight` is the prepared Wt[4][Cout][2][2][Ctot]
        P.up = UP_NONE;
        P.Hv = d->H;
        P.Wv = d->W;
        P.Ho = d->H;          // rows = low-resolution positions, four output phases
        P.Wo = d->W;
        P.ks = 2;
        P.pad = 0;
        P.M = d->B * d->H * d->W;
        P.nph = 4;
        P.ph_exact = 1;
        P.w_phase_stride = (long long)d->Cout * 4 * Ctot;
        P.dst[0].sy = 2 * Wo * d->Cout;
        P.dst[0].sx = 2 * d->Cout;
        P.dst[0].ph_y = Wo * d->Cout;
        P.dst[0].ph_x = d->Cout;
    }
    if (is_wino(d)) {  // `weight` is the prepared U[16][Cout][Ctot]
        if (d->winograd_next_gout) return DVSOF_EINVAL;     // a data gradient's option
        const WinoChain ch = {d->winograd_pre, d->winograd_next, nullptr};
        return wino_launch(P, (float *)d->scratch, d->scratch_bytes / sizeof(float), ch, as_stream(stream));
    }
    if (d->winograd_pre || d->winograd_next || d->winograd_next_gout) return DVSOF_EINVAL;
    if (is_min9(d)) {   // `weight` is the prepared Wt[9][Cout][Ctot]
        t_last_patch[0] = 2;
        return fwd_min_launch(P, as_stream(stream));
    }
    if (is_subpixel(d) && fwd_patch_eligible(P)) {  // finest decoder stage: fwd_patch.hip
        t_last_patch[0] = 1;
        return fwd_patch_launch(P, as_stream(stream));
    }
    return gconv_launch(P, 0, as_stream(stream));
}

static bool head_fold_ok(const dvsof_conv_desc_t *d)
{
    return is_min9_dgrad(d) || (is_subpixel(d) && d->src[0].layout == DVSOF_NHWC);
}

int dvsof_conv2d_dgrad_fuses_head(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    return d && desc_ok(d, Ctot, Ho, Wo) && head_fold_ok(d) ? 1 : 0;
}

int dvsof_conv2d_dgrad_head_rows(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    if (!d || !desc_ok(d, Ctot, Ho, Wo) || !is_min9_dgrad(d)) return 0;
    return d->B * (d->H / 8) * (d->W / 16);     // dgrad_min.hip's pixel blocks
}

int dvsof_flow_head_reduce(const float *part, int rows, int C, float *dw, float *dbias, void *stream)
{
    if (!part || !dw || rows < 1 || C < 1) return DVSOF_EINVAL;
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((2 * C + 2 + 3) / 4), dim3(256), 0, as_stream(stream), part,
                       rows, C, dw, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_conv2d_dgrad(const dvsof_conv_desc_t *d, const float *weight_t, const float *gout,
                       const dvsof_grad_dst_t *dst, int bwd_act, void *stream)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo) || !weight_t || !gout || !dst) return DVSOF_EINVAL;
    GConvParams P = {};
    P.nsrc = 1;
    P.src[0] = make_src(gout, d->Cout, DVSOF_NHWC, Ho, Wo, d->mfma == 3 ? d->gout16 : nullptr);
    P.ndst = d->nsrc;
    for (int i = 0; i < d->nsrc; ++i) {
        if (!dst[i].p) return DVSOF_EINVAL;
        const GSrc g = make_src(nullptr, d->src[i].C, d->src[i].layout, d->H, d->W);
        P.dst[i] = {dst[i].p, dst[i].addend, dst[i].addend2, dst[i].actsrc, g.sb, g.sy, g.sx, g.sc, g.C, 0, 0,
                    (d->mfma == 3 && d->src[i].layout == DVSOF_NHWC) ? (unsigned short *)dst[i].p16 : nullptr,
                    dst[i].head_w, dst[i].head_gflow, dst[i].head_x, dst[i].head_part};
        if ((dst[i].head_w != nullptr) != (dst[i].head_gflow != nullptr)) return DVSOF_EINVAL;
        if ((dst[i].head_x != nullptr) != (dst[i].head_part != nullptr)) return DVSOF_EINVAL;
        if (dst[i].head_part && (!dst[i].head_w || i != 0)) return DVSOF_EINVAL;
        // a head folds into the nine-product data gradient (dgrad_min.hip) and into the sub-pixel
        // layers' 4x4 stride-2 form on the general kernels (conv_epilogue), member 0 (NHWC) only
        if (dst[i].head_w && (i != 0 || !head_fold_ok(d))) return DVSOF_EINVAL;
        if (dst[i].head_part && !is_min9_dgrad(d)) return DVSOF_EINVAL;
    }
    P.W = weight_t;
    P.W16 = d->mfma == 3 ? (const unsigned short *)d->w16 : nullptr;
    P.bias = nullptr;
    P.zout = nullptr;
    P.B = d->B;
    P.ks = d->ksize;
    P.stride = 1;
    P.pad = d->ksize - 1 - d->pad;
    P.N = Ctot;
    P.Cin_tot = d->Cout;
    P.act = ACT_NONE;
    P.mfma_bf16 = (d->mfma >= 1 && d->mfma <= 3) ? d->mfma : 0;
    P.bwd_act = bwd_act;
    P.nph = 1;
    P.ph_pad = 0;
    P.w_phase_stride = 0;
    if (is_subpixel(d)) {
        // transpose of (up2 + 3x3) = 4x4 stride-2 convolution of gout with
        // the prepared Wd[Ctot][4][4][Cout]; rows = low-res input pixels
        P.up = UP_NONE;
        P.Hv = Ho;
        P.Wv = Wo;
        P.Ho = d->H;
        P.Wo = d->W;
        P.quad = 0;
        P.ks = 4;
        P.stride = 2;
        P.pad = 1;
    } else if (is_stride2_phased(d)) {
        // four input-parity phases, each a 2x2-tap stride-1 conv of gout
        // (weights prepared as Wp[4][Ctot][2][2][Cout]); rows = (H/2 x W/2)
        P.up = UP_NONE;
        P.Hv = Ho;
        P.Wv = Wo;
        P.Ho = d->H / 2;
        P.Wo = d->W / 2;
        P.quad = 0;
        P.ks = 2;
        P.stride = 1;
        P.pad = 0;
        P.nph = 4;
        P.ph_exact = 1;
        P.w_phase_stride = (long long)Ctot * 4 * d->Cout;
        for (int i = 0; i < d->nsrc; ++i) {
            P.dst[i].ph_y = P.dst[i].sy;
            P.dst[i].ph_x = P.dst[i].sx;
            P.dst[i].sy *= 2;
            P.dst[i].sx *= 2;
        }
    } else if (is_transposed(d)) {
        // adjoint of the zero insertion: a plain stride-2 3x3/pad-1 convolution
        // of gout (2H x 2W) with the flip-transposed weights
        P.up = UP_NONE;
        P.Hv = Ho;
        P.Wv = Wo;
        P.Ho = d->H;
        P.Wo = d->W;
        P.quad = 0;
        P.stride = 2;
        P.pad = 1;
    } else if (d->upsample) {  // rows = upsampled pixels, quad-summed to H x W
        P.up = UP_NONE;
        P.Hv = Ho;
        P.Wv = Wo;
        P.Ho = 2 * d->H;
        P.Wo = 2 * d->W;
        P.quad = 1;
    } else if (d->stride == 2) {  // zero-inserted gout
        P.up = UP_ZERO;
        P.Hv = 2 * Ho;
        P.Wv = 2 * Wo;
        P.Ho = d->H;
        P.Wo = d->W;
        P.quad = 0;
    } else {
        P.up = UP_NONE;
        P.Hv = Ho;
        P.Wv = Wo;
        P.Ho = d->H;
        P.Wo = d->W;
        P.quad = 0;
    }
    P.M = d->B * P.Ho * P.Wo;
    t_last_patch[1] = 0;
    if (is_min9_dgrad(d)) {  // weight_t is the prepared W'[9][Ctot][Cout]
        t_last_patch[1] = 2;
        return dgrad_min_launch(P, as_stream(stream));
    }
    if (is_wino(d)) {  // weight_t is the prepared U'[16][Ctot][Cout]
        const WinoChain ch = {d->winograd_pre, d->winograd_next, d->winograd_next_gout};
        return wino_launch(P, (float *)d->scratch, d->scratch_bytes / sizeof(float), ch, as_stream(stream));
    }
    if (d->winograd_pre || d->winograd_next || d->winograd_next_gout) return DVSOF_EINVAL;
    return gconv_launch(P, 0, as_stream(stream));
}

size_t dvsof_conv2d_scratch_bytes(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo) || !is_wino(d)) return 0;
    return wino_scratch_floats(d->B, d->H, d->W, Ctot, d->Cout, d->mfma == 2 ? 2 : 0) * sizeof(float);
}

// Transposed layer T: [B,H,W,Ctot] -> [B,2H,2W,Cout].  Its weight gradient is
// the weight gradient of the adjoint stride-2 layer S: [B,2H,2W,Cout] ->
// [B,H,W,Ctot] with the roles swapped (S's input = T's output gradient, S's
// output gradient = T's input), flip-transposed:
//   dW_T[co][ky][kx][ci] = dW_S[ci][2-ky][2-kx][co].
static dvsof_conv_desc_t adjoint_of_transposed(const dvsof_conv_desc_t *d, int Ctot, const float *gy)
{
    dvsof_conv_desc_t a = *d;
    a.nsrc = 1;
    a.src[0].p = gy;
    a.src[0].C = d->Cout;
    a.src[0].layout = DVSOF_NHWC;
    a.src[0].p16 = nullptr;
    a.H = 2 * d->H;
    a.W = 2 * d->W;
    a.upsample = 0;
    a.stride = 2;
    a.Cout = Ctot;
    a.scratch = nullptr;
    a.scratch_bytes = 0;
    a.winograd_input = nullptr;
    return a;
}
static int channel_sum_blocks(long long npix, int C)
{
    const long long per = 256 / C, want = (npix + per * 8 - 1) / (per * 8);
    return (int)(want < 1 ? 1 : want > 1024 ? 1024 : want);
}

size_t dvsof_conv2d_wgrad_workspace_bytes(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return 0;
    if (is_transposed(d)) {
        static const float dummy = 0.f;
        const dvsof_conv_desc_t a = adjoint_of_transposed(d, Ctot, &dummy);
        size_t n = dvsof_conv2d_wgrad_workspace_bytes(&a);
        n = (n + 255) & ~(size_t)255;
        n += (size_t)Ctot * 9 * d->Cout * sizeof(float);                       // dW of the adjoint
        if (d->Cout <= 256)
            n += (size_t)channel_sum_blocks((long long)d->B * Ho * Wo, d->Cout) * d->Cout * sizeof(float);
        return n + 256;
    }
    if (is_wino_wgrad(d)) return wino_wgrad_workspace_floats(d->B, d->H, d->W, Ctot, d->Cout, d->mfma == 2 ? 2 : 0) * sizeof(float) + 16;
    WGradParams P = {};     // (unused member slots are zeros, not stack residue: _audit.audit_exchange reads these words)
    fill_wgrad(d, Ctot, Ho, Wo, P);
    FlatWG F[3] = {};
    const int nflat = fill_flat(d, Ctot, Ho, Wo, nullptr, F);
    size_t n = wgrad_workspace_floats(P, true) + wgrad_flat_workspace_floats(F, nflat);
    if (is_first_layer(d)) {
        const size_t nf = first_wgrad_workspace_floats(d->B, Ctot, d->H, d->W);
        if (nf > n) n = nf;
    }
    return n * sizeof(float) + 16;
}

int dvsof_conv2d_wgrad(const dvsof_conv_desc_t *d, const float *gout, float *dweight, float *dbias,
                       void *ws, size_t ws_bytes, void *stream)
{
    int Ctot, Ho, Wo;
    t_last_patch[2] = 0;
    if (!desc_ok(d, Ctot, Ho, Wo) || !gout || !dweight) return DVSOF_EINVAL;
    if (is_transposed(d)) {
        if (dbias && d->Cout > 256) return DVSOF_EINVAL;
        const dvsof_conv_desc_t a = adjoint_of_transposed(d, Ctot, gout);
        size_t wsa = dvsof_conv2d_wgrad_workspace_bytes(&a);
        wsa = (wsa + 255) & ~(size_t)255;
        const size_t need = dvsof_conv2d_wgrad_workspace_bytes(d);
        if (!ws || ws_bytes < need) return DVSOF_ENOSPACE;
        float *dw_adj = (float *)((char *)ws + wsa);
        float *part = dw_adj + (size_t)Ctot * 9 * d->Cout;
        int rc = dvsof_conv2d_wgrad(&a, d->src[0].p, dw_adj, nullptr, ws, wsa, stream);
        if (rc) return rc;
        // [Ctot][tap][Cout] -> [Cout][8 - tap][Ctot]
        rc = flip_transpose16(dw_adj, dweight, Ctot, 3, d->Cout, nullptr, nullptr, stream);
        if (rc) return rc;
        if (dbias) {
            const long long npix = (long long)d->B * Ho * Wo;
            const int nb = channel_sum_blocks(npix, d->Cout);
            hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), gout,
                               npix, d->Cout, part);
            DVSOF_LAUNCH_CHECK();
            hipLaunchKernelGGL(channel_sum_final_kernel, dim3((d->Cout + 3) / 4), dim3(256), 0,
                               as_stream(stream), (const float *)part, nb, d->Cout, dbias);
            DVSOF_LAUNCH_CHECK();
        }
        return DVSOF_OK;
    }
    if (is_wino_wgrad(d))
    {
        // the forward's transformed input, when the caller kept it and both use the same form
        const int mf = d->mfma == 2 ? 2 : 0;
        const float *v_in = wino_tile(d->B, d->H, d->W, mf) == wino_wgrad_tile(d->B, d->H, d->W, mf)
                                ? d->winograd_input : nullptr;
        // ... and the gradient form of gout, when the data gradient that produced gout made it
        const float *z_in = wino_wgrad_tile(d->B, d->H, d->W, mf) == 4 ? d->winograd_gout : nullptr;
        return wino_wgrad_launch(make_src(d->src[0].p, d->src[0].C, d->src[0].layout, d->H, d->W), v_in, z_in,
                                 gout, dweight, dbias, d->B, d->H, d->W, Ctot, d->Cout, mf, (float *)ws,
                                 ws_bytes / sizeof(float), as_stream(stream));
    }
    if (is_first_layer(d) && !(d->flags & DVSOF_CONV_WGRAD_SKIP_FLAT))
        return first_wgrad_launch(d->src[0].p, d->B, Ctot, d->H, d->W, gout, dweight, dbias, (float *)ws,
                                  ws_bytes / sizeof(float), as_stream(stream));
    WGradParams P = {};     // (unused member slots are zeros, not stack residue: _audit.audit_exchange reads these words)
    fill_wgrad(d, Ctot, Ho, Wo, P);
    P.gout = gout;
    FlatWG F[3] = {};
    // DVSOF_CONV_WGRAD_SKIP_FLAT: the flat members' columns are the caller's
    // (dvsof_flow_fold_grads); the vector members' columns are written as usual
    const int nflat = (d->flags & DVSOF_CONV_WGRAD_SKIP_FLAT) ? 0 : fill_flat(d, Ctot, Ho, Wo, gout, F);
    if (wgrad_workspace_floats(P, dbias != nullptr) + wgrad_flat_workspace_floats(F, nflat) > 0 && !ws)
        return DVSOF_ENOSPACE;
    return wgrad_launch(P, dweight, dbias, (float *)ws, ws_bytes / sizeof(float), F, nflat,
                        as_stream(stream));
}

size_t dvsof_conv2d_fwd_weight_elems(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return 0;
    if (is_wino(d)) return (size_t)d->Cout * Ctot * wino_components(d->B, d->H, d->W, d->mfma == 2 ? 2 : 0);
    return (size_t)d->Cout * Ctot * ((is_subpixel(d) || is_transposed(d)) ? 16 : d->ksize * d->ksize);
}

size_t dvsof_conv2d_dgrad_weight_elems(const dvsof_conv_desc_t *d)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return 0;
    if (is_wino(d)) return (size_t)d->Cout * Ctot * wino_components(d->B, d->H, d->W, d->mfma == 2 ? 2 : 0);
    if (is_subpixel(d) || is_stride2_phased(d)) return (size_t)d->Cout * Ctot * 16;
    return (size_t)d->Cout * Ctot * d->ksize * d->ksize;
}

static int flip_transpose16(const float *w, float *wt, int Cout, int ksize, int Ctot,
                            unsigned short *wt16, unsigned short *w16, void *stream);
int dvsof_to_bf16(const float *src, void *dst, size_t n, void *stream);

// w_fwd16 / w_dgrad16 (optional): bf16 twins of the two forms, written by the
// kernels that make the forms.  For a layer whose forward form is the raw
// weight, w_fwd16 is the raw weight's twin (emitted by the data-gradient form
// kernel, which reads every raw element once; a conversion launch if no
// data-gradient form is asked for).
int dvsof_conv2d_prepare16(const dvsof_conv_desc_t *d, const float *weight, float *w_fwd,
                           float *w_dgrad, void *w_fwd16_, void *w_dgrad16_, void *stream)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return DVSOF_EINVAL;
    hipStream_t st = as_stream(stream);
    unsigned short *w_fwd16 = (unsigned short *)w_fwd16_, *w_dgrad16 = (unsigned short *)w_dgrad16_;
    if (w_dgrad16 && !w_dgrad) return DVSOF_EINVAL;
    if (is_subpixel(d)) {
        if (!w_fwd && !weight) return DVSOF_EINVAL;
        // weight == NULL: w_fwd already holds the phase kernels (made by an
        // earlier call); only the data-gradient form is derived from it
        if (!weight && !w_dgrad) return DVSOF_EINVAL;
        if (is_min9(d) || (weight && !w_fwd)) {
            // both forms from the RAW weights: forward Wt[9][Cout][Ctot] (fwd_min.hip) or the
            // phase kernels; the data gradient's 4x4 stride-2 form Wd
            if (!weight || (!w_fwd && !w_dgrad)) return DVSOF_EINVAL;
            if (w_fwd && is_min9(d)) {
                if (w_fwd16) return DVSOF_EINVAL;
                const int rc = min9_prepare_fwd(weight, w_fwd, d->Cout, Ctot, st);
                if (rc) return rc;
            } else if (w_fwd) {
                const size_t n = (size_t)d->Cout * Ctot;
                hipLaunchKernelGGL(subpixel_fwd_weights_kernel, dim3((unsigned)((n + 255) / 256)),
                                   dim3(256), 0, st, weight, w_fwd, d->Cout, Ctot, w_fwd16);
                DVSOF_LAUNCH_CHECK();
            }
            if (w_dgrad && is_min9_dgrad(d)) {
                if (w_dgrad16) return DVSOF_EINVAL;
                return min9_prepare_dgrad(weight, w_dgrad, d->Cout, Ctot, st);
            }
            if (w_dgrad) {
                dim3 grid((Ctot + 31) / 32, (d->Cout + 31) / 32, 16);
                hipLaunchKernelGGL(subpixel_dgrad_weights_raw_kernel, grid, dim3(256), 0, st, weight, w_dgrad,
                                   d->Cout, Ctot, w_dgrad16);
                DVSOF_LAUNCH_CHECK();
            }
            return DVSOF_OK;
        }
        if (weight) {
            const size_t n = (size_t)d->Cout * Ctot;
            hipLaunchKernelGGL(subpixel_fwd_weights_kernel, dim3((unsigned)((n + 255) / 256)),
                               dim3(256), 0, st, weight, w_fwd, d->Cout, Ctot, w_fwd16);
            DVSOF_LAUNCH_CHECK();
        } else if (w_fwd16) {
            const int rc = dvsof_to_bf16(w_fwd, w_fwd16, dvsof_conv2d_fwd_weight_elems(d), stream);
            if (rc) return rc;
        }
        if (w_dgrad) {
            dim3 grid((Ctot + 31) / 32, (d->Cout + 31) / 32, 16);
            hipLaunchKernelGGL(subpixel_dgrad_weights_kernel, grid, dim3(256), 0, st,
                               (const float *)w_fwd, w_dgrad, d->Cout, Ctot, w_dgrad16);
            DVSOF_LAUNCH_CHECK();
        }
        return DVSOF_OK;
    }
    if (is_transposed(d)) {   // phase kernels forward, flip-transpose backward
        if (!weight || (!w_fwd && !w_dgrad) || w_fwd == weight) return DVSOF_EINVAL;
        if (w_fwd) {
            const size_t n = (size_t)d->Cout * Ctot;
            hipLaunchKernelGGL(transposed_fwd_weights_kernel, dim3((unsigned)((n + 255) / 256)),
                               dim3(256), 0, st, weight, w_fwd, d->Cout, Ctot, w_fwd16);
            DVSOF_LAUNCH_CHECK();
        }
        if (w_dgrad) return flip_transpose16(weight, w_dgrad, d->Cout, 3, Ctot, w_dgrad16, nullptr, stream);
        return DVSOF_OK;
    }
    if (is_wino(d)) {   // either form (or both) from the raw weights
        if (!weight || (!w_fwd && !w_dgrad)) return DVSOF_EINVAL;
        if (w_fwd16 || w_dgrad16) return DVSOF_EINVAL;   // the bf16-twin mode runs these layers direct
        return wino_prepare(weight, w_fwd, w_dgrad, d->Cout, Ctot, d->B, d->H, d->W, d->mfma == 2 ? 2 : 0, st);
    }
    if (!weight) return DVSOF_EINVAL;
    // the forward form is the raw weight (or a copy of it)
    if (w_fwd && w_fwd != weight)
        DVSOF_HIP_TRY(hipMemcpyAsync(w_fwd, weight, dvsof_conv2d_fwd_weight_elems(d) * sizeof(float),
                                     hipMemcpyDeviceToDevice, st));
    if (is_stride2_phased(d) && w_dgrad) {
        dim3 grid((Ctot + 31) / 32, (d->Cout + 31) / 32, 16);
        hipLaunchKernelGGL(stride2_dgrad_weights_kernel, grid, dim3(256), 0, st, weight, w_dgrad,
                           d->Cout, Ctot, w_dgrad16, w_fwd16);
        DVSOF_LAUNCH_CHECK();
        return DVSOF_OK;
    }
    if (w_dgrad) return flip_transpose16(weight, w_dgrad, d->Cout, d->ksize, Ctot, w_dgrad16, w_fwd16, stream);
    if (w_fwd16) return dvsof_to_bf16(weight, w_fwd16, dvsof_conv2d_fwd_weight_elems(d), stream);
    return DVSOF_OK;
}

int dvsof_conv2d_prepare(const dvsof_conv_desc_t *d, const float *weight, float *w_fwd,
                         float *w_dgrad, void *stream)
{
    return dvsof_conv2d_prepare16(d, weight, w_fwd, w_dgrad, nullptr, nullptr, stream);
}

int dvsof_conv2d_tile_id(const dvsof_conv_desc_t *d, int kind);

// 0: direct implicit GEMM; 2 | 4: Winograd F(2x2,3x3) | F(4x4,3x3) (kind 0 fwd, 1 dgrad, 2 wgrad)
int dvsof_conv2d_winograd_tile(const dvsof_conv_desc_t *d, int kind)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo) || !is_wino(d)) return 0;
    const int mfma = d->mfma == 2 ? 2 : 0;
    if (kind == 2) return is_wino_wgrad(d) ? wino_wgrad_tile(d->B, d->H, d->W, mfma) : 0;
    return wino_tile(d->B, d->H, d->W, mfma);
}

// 1: this layer's forward (kind 0) / data gradient (kind 1) is a Winograd evaluation whose output
// transform can also write the consumer's forms (winograd_next / winograd_next_gout)
int dvsof_conv2d_winograd_chain(const dvsof_conv_desc_t *d, int kind)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo) || !is_wino(d) || (kind != 0 && kind != 1)) return 0;
    return wino_chain_ok(d->B, d->H, d->W, kind == 0 ? d->Cout : Ctot, d->mfma == 2 ? 2 : 0) ? 1 : 0;
}

// 2 when the LDS-DMA (v2) kernel serves this problem's vector members, else 1;
// 3: the first-layer kernels (first.hip); 0: flat members only on the VALU kernel
int dvsof_conv2d_kernel_generation(const dvsof_conv_desc_t *d, int kind)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return DVSOF_EINVAL;
    if (kind != 1 && is_first_layer(d)) return 3;
    if (kind == 2) {
        for (int i = 0; i < d->nsrc; ++i) {
            const GSrc g = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W);
            if (!g.flat) return (Wo % BK == 0 && (!d->upsample || is_subpixel(d))) ? 2 : 1;
        }
        return 0;   // flat members only: VALU kernel
    }
    if (kind == 1) return (d->Cout % BK == 0) ? 2 : 1;
    for (int i = 0; i < d->nsrc; ++i) {
        const GSrc g = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W);
        if (!g.flat) return (g.C % BK == 0) ? 2 : 1;
    }
    return 1;
}

int dvsof_conv2d_last_patch(int kind)
{
    return (kind >= 0 && kind <= 2) ? t_last_patch[kind] : 0;
}

int dvsof_conv2d_tile_id(const dvsof_conv_desc_t *d, int kind)
{
    int Ctot, Ho, Wo;
    if (!desc_ok(d, Ctot, Ho, Wo)) return DVSOF_EINVAL;
    if (kind == 0) return gconv_pick_tile((long long)d->B * Ho * Wo, d->Cout);
    if (kind == 1) {
        const int up = (d->upsample && !is_subpixel(d)) ? 2 : 1;
        int n = Ctot, trail = 0;   // narrow planar members are peeled off (gconv.hip)
        for (int i = d->nsrc - 1; i > 0; --i) {
            const GSrc g = make_src(d->src[i].p, d->src[i].C, d->src[i].layout, d->H, d->W);
            if (!g.flat || trail + g.C > 4) break;
            trail += g.C;
        }
        if (up == 1 && !(is_stride2_phased(d)) && n - trail >= 32) n -= trail;
        return gconv_pick_tile((long long)d->B * d->H * up * d->W * up, n);  // phases included
    }
    if (kind == 2) {
        WGradParams P = {};     // (unused member slots are zeros, not stack residue: _audit.audit_exchange reads these words)
        fill_wgrad(d, Ctot, Ho, Wo, P);
        int tile = 0;
        wgrad_splits(P, &tile);
        return tile;
    }
    return DVSOF_EINVAL;
}

static int flip_transpose16(const float *w, float *wt, int Cout, int ksize, int Ctot,
                            unsigned short *wt16, unsigned short *w16, void *stream)
{
    if (!w || !wt || Cout < 1 || ksize < 1 || Ctot < 1) return DVSOF_EINVAL;
    const int taps = ksize * ksize;
    dim3 grid((Ctot + 31) / 32, (Cout + 31) / 32, taps);
    hipLaunchKernelGGL(flip_transpose_kernel, grid, dim3(256), 0, as_stream(stream), w, wt, Cout,
                       taps, Ctot, wt16, w16);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_weight_flip_transpose(const float *w, float *wt, int Cout, int ksize, int Ctot,
                                void *stream)
{
    return flip_transpose16(w, wt, Cout, ksize, Ctot, nullptr, nullptr, stream);
}

#define HEAD_DISPATCH(KERNEL, nb, ...)                                                         \
    switch (C / 4) {                                                                           \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(nb), dim3(256), 0, st, __VA_ARGS__); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(nb), dim3(256), 0, st, __VA_ARGS__); break;   \
    case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(nb), dim3(256), 0, st, __VA_ARGS__); break; \
    case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(nb), dim3(256), 0, st, __VA_ARGS__); break; \
    case 64: hipLaunchKernelGGL((KERNEL<64>), dim3(nb), dim3(256), 0, st, __VA_ARGS__); break; \
    default: return DVSOF_EINVAL;                                                              \
    }

static bool head_c_ok(int C) { return C == 16 || C == 32 || C == 64 || C == 128 || C == 256; }

int dvsof_flow_head_fwd(const float *x, const float *w, const float *bias, float *flow, int B,
                        int H, int W, int C, void *stream)
{
    if (!x || !w || !flow || B < 1 || H < 1 || W < 1 || !head_c_ok(C)) return DVSOF_EINVAL;
    hipStream_t st = as_stream(stream);
    // no partial sums here, so the grid is free: ~4 pixels per lane group keeps enough
    // waves in flight (512 workgroups walked 32 dependent iterations: 22 us for 67 MB)
    static const int fwd_iters = getenv("DVSOF_HEAD_FWD_ITERS") ? atoi(getenv("DVSOF_HEAD_FWD_ITERS")) : 4;
    const long long per_block = 4LL * (64 / (C / 4)) * (fwd_iters > 0 ? fwd_iters : 1);
    long long nbl = ((long long)B * H * W + per_block - 1) / per_block;
    const int nb = (int)(nbl < 1 ? 1 : nbl > 65535 ? 65535 : nbl);
    HEAD_DISPATCH(head_fwd_kernel, nb, x, w, bias, flow, B, H * W);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_flow_heads_fwd(int n, const float *const *x, const float *const *w, const float *const *bias,
                         float *const *flow, int B, const int *H, const int *W, const int *C,
                         void *stream)
{
    if (n < 1 || n > HEADS_MAX || !x || !w || !flow || !H || !W || !C || B < 1) return DVSOF_EINVAL;
    static const int fwd_iters = getenv("DVSOF_HEAD_FWD_ITERS") ? atoi(getenv("DVSOF_HEAD_FWD_ITERS")) : 4;
    HeadsFwd A = {};
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !w[i] || !flow[i] || H[i] < 1 || W[i] < 1 || !head_c_ok(C[i])) return DVSOF_EINVAL;
        A.x[i] = x[i];
        A.w[i] = w[i];
        A.bias[i] = bias ? bias[i] : nullptr;
        A.flow[i] = flow[i];
        A.HW[i] = H[i] * W[i];
        A.C[i] = C[i];
        const long long per_block = 4LL * (64 / (C[i] / 4)) * (fwd_iters > 0 ? fwd_iters : 1);
        long long nbl = ((long long)B * H[i] * W[i] + per_block - 1) / per_block;
        nbl = nbl < 1 ? 1 : nbl > 65535 ? 65535 : nbl;
        A.block_begin[i] = (int)blocks;
        blocks += nbl;
    }
    A.block_begin[n] = (int)blocks;
    A.B = B;
    A.n = n;
    hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), A);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

size_t dvsof_flow_head_bwd_workspace_bytes(int B, int H, int W, int C)
{
    if (B < 1 || H < 1 || W < 1 || !head_c_ok(C)) return 0;
    return (size_t)head_blocks((long long)B * H * W, C / 4) * (2 * C + 2) * sizeof(float);
}

int dvsof_flow_head_bwd(const float *x, const float *w, const float *gflow, const float *gx_in,
                        const float *actsrc, int act, float *gx, float *dw, float *dbias, int B,
                        int H, int W, int C, void *ws, size_t ws_bytes, void *gx16, void *stream)
{
    // gx == NULL: the head's own weight / bias gradient only (its data part was folded into the
    // data gradient that produced gx_in's tensor: dvsof_grad_dst_t.head_w)
    if (!x || !w || !gflow || !dw || !ws || B < 1 || H < 1 || W < 1 || !head_c_ok(C))
        return DVSOF_EINVAL;
    if (ws_bytes < dvsof_flow_head_bwd_workspace_bytes(B, H, W, C)) return DVSOF_ENOSPACE;
    hipStream_t st = as_stream(stream);
    const int nb = head_blocks((long long)B * H * W, C / 4);
    float *part = (float *)ws;
    HEAD_DISPATCH(head_bwd_kernel, nb, x, w, gflow, gx_in, actsrc, act, gx, part, B, H * W,
                  (unsigned short *)gx16);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((2 * C + 2 + 3) / 4), dim3(256), 0, st,
                       (const float *)part, nb, C, dw, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_to_bf16(const float *src, void *dst, size_t n, void *stream)
{
    if (!src || !dst) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    size_t nb = (n + 2047) / 2048;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), src,
                       (unsigned short *)dst, n);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_to_bf16_many(const float *const *src, void *const *dst, const size_t *n, int count,
                       void *stream)
{
    if (count < 0 || count > BF16_MANY_MAX || (count && (!src || !dst || !n))) return DVSOF_EINVAL;
    if (count == 0) return DVSOF_OK;
    Bf16Many J = {};
    size_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!src[i] || !dst[i]) return DVSOF_EINVAL;
        J.src[i] = src[i];
        J.dst[i] = (unsigned short *)dst[i];
        J.n[i] = n[i];
        J.block_begin[i] = (unsigned)blocks;
        blocks += (n[i] + 2047) / 2048;
    }
    J.count = count;
    J.block_begin[count] = (unsigned)blocks;
    if (blocks == 0) return DVSOF_OK;
    if (blocks > 0x7fffffffu) return DVSOF_EINVAL;
    hipLaunchKernelGGL(to_bf16_many_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), J);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_act_bwd(const float *dy, const float *actsrc, int act, float *dz, size_t n, void *stream)
{
    if (!dy || !actsrc || !dz) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    size_t nb = (n + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), dy,
                       actsrc, act, dz, n);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
