// The 2-channel flow member of a decoder stage, folded into weight space for
// the backward pass.
//
// Decoder stage i convolves cat[x, skip, flow] (docs/MODEL_SPEC.md) where
// flow = head(x) = Wh x + bh is the previous stage's flow prediction -- a
// LINEAR function of the x member at the same pixel.  The forward keeps the
// flow member as it is.  The backward used to need, per stage, a pass over the
// whole output gradient for the 2 flow rows of the data gradient
// (gconv_flat_rows_*) and another for the 18 flow columns of the weight
// gradient (wgrad_flat_*): 0.15 ms of a 3.2 ms step at batch 8.  Both follow
// from quantities the matrix-core kernels produce anyway:
//
//   data gradient     d/dx through the flow member = Wh^T (Wflow^T g): the
//                     x columns of the layer's weights become
//                     Weff[co][t][ci] = Wx[co][t][ci] + sum_f Wflow[co][t][f] Wh[f][ci]
//                     (dvsof_flow_fold_weights) and the data gradient runs on
//                     cat[x, skip] only; the head's backward then sees the
//                     loss gradient of the flow alone.
//   weight gradient   with dWx[co][t][ci] = sum_p g[p][co] x[p@t][ci] (the x
//                     columns of the MFMA weight gradient) and
//                     G[co][t] = sum over the pixels p whose tap t is inside
//                     the frame of g[p][co]:
//                       dWflow[co][t][f] = sum_ci dWx[co][t][ci] Wh[f][ci] + bh[f] G[co][t]
//                       dWh[f][ci]      += sum_{co,t} Wflow[co][t][f] dWx[co][t][ci]
//                       dbh[f]          += sum_{co,t} Wflow[co][t][f] G[co][t]
//                     (dvsof_flow_fold_grads).  G = (bias gradient) - (sums of
//                     g over the border lines the tap leaves the frame on),
//                     so only the border pixels of g are read.
//
// Same mathematics as the reference's autograd through cat / interpolate /
// conv2d (utils/training.py:158); f32 rounding differs in the order of sums.
// Every reduction runs in a fixed order: bitwise reproducible.
#include "common.h"

namespace {

constexpr int TAPS = 9;

__global__ __launch_bounds__(256) void flow_fold_weights_kernel(const float *__restrict__ w,
                                                                int Cout, int Ctot, int cx_off, int Cx,
                                                                int cf_off, const float *__restrict__ wh,
                                                                float *__restrict__ weff)
{
    const int Ce = Ctot - 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Cout * TAPS * Ce) return;
    const int j = (int)(i % Ce);
    const size_t row = i / Ce;                       // co * TAPS + t
    const int c = j < cf_off ? j : j + 2;            // source column (the flow pair is skipped)
    const float *wr = w + row * Ctot;
    float v = wr[c];
    if (c >= cx_off && c < cx_off + Cx) {
        const int ci = c - cx_off;
        v += wr[cf_off] * wh[ci] + wr[cf_off + 1] * wh[Cx + ci];
    }
    weff[i] = v;
}

// bias_eff / bias_cls of the forward fold; thread per output channel
__global__ __launch_bounds__(256) void flow_fold_bias_kernel(const float *__restrict__ w, int Cout,
                                                             int Ctot, int cf_off,
                                                             const float *__restrict__ bh,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ bias_eff,
                                                             float *__restrict__ bias_cls)
{
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= Cout) return;
    float tap[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const float *wf = w + ((size_t)co * TAPS + t) * Ctot + cf_off;
        tap[t] = wf[0] * bh[0] + wf[1] * bh[1];
    }
    float all = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) all += tap[t];
    bias_eff[co] = (bias ? bias[co] : 0.f) + all;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int cy = c / 3, cx = c - 3 * cy;      // 1: first line (ky / kx = 0 outside), 2: last (2 outside)
        float out = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const bool outside = (cy == 1 && ky == 0) || (cy == 2 && ky == 2) || (cx == 1 && kx == 0) ||
                                 (cx == 2 && kx == 2);
            if (outside) out += tap[t];
        }
        bias_cls[(size_t)c * Cout + co] = -out;
    }
}

// Sums of g over the four border lines of every image: part[(line * B + b) * C + c],
// line 0: y = 0, 1: y = H - 1, 2: x = 0, 3: x = W - 1.  One 1024-thread
// workgroup per (line, image); thread = (pixel group r, channel c), 8 loads in
// flight per thread; fixed-order LDS combine.
constexpr int FOLD_NT = 1024;
__global__ __launch_bounds__(FOLD_NT) void flow_border_sums_kernel(const float *__restrict__ g, int B,
                                                                   int H, int W, int C,
                                                                   float *__restrict__ part)
{
    __shared__ float sm[FOLD_NT];
    const int line = blockIdx.x / B, b = blockIdx.x - line * B;
    const int len = line < 2 ? W : H;
    const float *base = g + (size_t)b * H * W * C;
    const size_t p0 = line == 1 ? (size_t)(H - 1) * W : line == 3 ? (size_t)(W - 1) : 0;
    const size_t step = line < 2 ? 1 : (size_t)W;        // pixels between consecutive line elements
    for (int c0 = 0; c0 < C; c0 += FOLD_NT) {
        const int cw = min(C - c0, FOLD_NT);             // channels of this pass
        const int R = FOLD_NT / cw;                      // pixel groups
        const int c = threadIdx.x % cw, r = threadIdx.x / cw;
        float a = 0.f;
        if (r < R) {
            int i = r;
            for (; i + 7 * R < len; i += 8 * R) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = base[(p0 + (size_t)(i + u * R) * step) * C + c0 + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; i < len; i += R) a += base[(p0 + (size_t)i * step) * C + c0 + c];
        }
        sm[threadIdx.x] = a;
        __syncthreads();
        if ((int)threadIdx.x < cw) {
            float t = 0.f;
            for (int k = 0; k < R; ++k) t += sm[k * cw + threadIdx.x];
            part[((size_t)line * B + b) * C + c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

struct FoldArgs {
    float *dW;               // [Cout][9][Ctot]: x columns read, flow columns written
    const float *w;          // [Cout][9][Ctot]
    const float *wh, *bh;    // [2][Cx], [2]
    const float *db;         // [Cout]: sum over all pixels of g
    const float *part;       // border line sums (flow_border_sums_kernel)
    const float *g;          // [B][H][W][Cout] (corner pixels are read)
    float *dwh, *dbh;        // [2][Cx], [2]: accumulated into
    int Cout, Ctot, cx_off, Cx, cf_off, B, H, W;
    int nb_flow, nb_wh;      // workgroups of roles 0 and 1 (role 2: one more)
};

// G[co][t]: sum of g[p][co] over the pixels whose tap t = (ky, kx) reads inside the frame
__device__ float fold_G(const FoldArgs &A, int co, int t)
{
    const int ky = t / 3, kx = t - 3 * ky;
    const int ly = ky == 0 ? 0 : ky == 2 ? 1 : -1;     // border line the tap leaves the frame on
    const int lx = kx == 0 ? 2 : kx == 2 ? 3 : -1;
    float v = A.db[co];
    for (int b = 0; b < A.B; ++b) {
        if (ly >= 0) v -= A.part[((size_t)ly * A.B + b) * A.Cout + co];
        if (lx >= 0) v -= A.part[((size_t)lx * A.B + b) * A.Cout + co];
        if (ly >= 0 && lx >= 0) {       // the corner pixel was subtracted twice
            const int y = ly == 0 ? 0 : A.H - 1, x = lx == 2 ? 0 : A.W - 1;
            v += A.g[(((size_t)b * A.H + y) * A.W + x) * A.Cout + co];
        }
    }
    return v;
}

// 1024 threads = 16 waves per workgroup: these are a few hundred short dot
// products over data that was just written -- latency, not bandwidth; the waves
// of a workgroup split the long axis and meet in LDS in a fixed order.
__global__ __launch_bounds__(FOLD_NT) void flow_fold_grads_kernel(const FoldArgs A)
{
    constexpr int NW = FOLD_NT / 64;
    __shared__ double sm[NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrow = A.Cout * TAPS;
    if ((int)blockIdx.x < A.nb_flow) {
        // role 0: dW flow columns; one wave per (co, t), lanes stride the x channels
        const int row = blockIdx.x * NW + wave;          // co * 9 + t
        if (row >= nrow) return;
        const float *dx = A.dW + (size_t)row * A.Ctot + A.cx_off;
        double s0 = 0, s1 = 0;
        for (int ci = lane; ci < A.Cx; ci += 64) {
            const float d = dx[ci];
            s0 += (double)d * A.wh[ci];
            s1 += (double)d * A.wh[A.Cx + ci];
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            const float G = fold_G(A, row / TAPS, row % TAPS);
            float *df = A.dW + (size_t)row * A.Ctot + A.cf_off;
            df[0] = (float)(s0 + (double)A.bh[0] * G);
            df[1] = (float)(s1 + (double)A.bh[1] * G);
        }
        return;
    }
    if ((int)blockIdx.x < A.nb_flow + A.nb_wh) {
        // role 1: dWh[f][ci] += sum_{co,t} Wflow[co][t][f] dWx[co][t][ci]; a workgroup
        // owns 64 (f, ci) outputs, its 16 waves split the rows, fixed-order combine
        const int i = (blockIdx.x - A.nb_flow) * 64 + lane;
        const bool ok = i < 2 * A.Cx;
        const int f = ok ? i / A.Cx : 0, ci = ok ? i - f * A.Cx : 0;
        const float *wf = A.w + A.cf_off + f, *dx = A.dW + A.cx_off + ci;
        double s = 0;
        int row = wave;
        for (; row + 7 * NW < nrow; row += 8 * NW) {
            float a[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = wf[(size_t)(row + u * NW) * A.Ctot];
                d[u] = dx[(size_t)(row + u * NW) * A.Ctot];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)a[u] * d[u];
        }
        for (; row < nrow; row += NW) s += (double)wf[(size_t)row * A.Ctot] * dx[(size_t)row * A.Ctot];
        sm[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && ok) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) t += sm[k][lane];
            A.dwh[i] += (float)t;
        }
        return;
    }
    // role 2: dbh[f] += sum_{co,t} Wflow[co][t][f] G[co][t]; thread per row, fixed-order combine
    double s0 = 0, s1 = 0;
    for (int row = threadIdx.x; row < nrow; row += FOLD_NT) {
        const float G = fold_G(A, row / TAPS, row % TAPS);
        s0 += (double)A.w[(size_t)row * A.Ctot + A.cf_off] * G;
        s1 += (double)A.w[(size_t)row * A.Ctot + A.cf_off + 1] * G;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    __shared__ double s2[2][NW];
    if (lane == 0) {
        s2[0][wave] = s0;
        s2[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0;
        for (int k = 0; k < NW; ++k) t += s2[threadIdx.x][k];
        A.dbh[threadIdx.x] += (float)t;
    }
}

}  // namespace

extern "C" {

int dvsof_flow_fold_weights(const float *w, int Cout, int Ctot, int cx_off, int Cx, int cf_off,
                            const float *wh, float *w_eff, void *stream)
{
    if (!w || !wh || !w_eff || Cout < 1 || Ctot < 3 || Cx < 1 || cx_off < 0 || cf_off < 0 ||
        cf_off + 2 > Ctot || cx_off + Cx > Ctot || (cx_off < cf_off + 2 && cf_off < cx_off + Cx))
        return DVSOF_EINVAL;
    const size_t n = (size_t)Cout * TAPS * (Ctot - 2);
    hipLaunchKernelGGL(flow_fold_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, Cout, Ctot, cx_off, Cx, cf_off, wh, w_eff);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_flow_fold_bias(const float *w, int Cout, int Ctot, int cf_off, const float *bh,
                         const float *bias, float *bias_eff, float *bias_cls, void *stream)
{
    if (!w || !bh || !bias_eff || !bias_cls || Cout < 1 || Ctot < 3 || cf_off < 0 || cf_off + 2 > Ctot)
        return DVSOF_EINVAL;
    hipLaunchKernelGGL(flow_fold_bias_kernel, dim3((Cout + 255) / 256), dim3(256), 0, as_stream(stream), w,
                       Cout, Ctot, cf_off, bh, bias, bias_eff, bias_cls);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

size_t dvsof_flow_fold_workspace_bytes(int B, int Cout) { return (size_t)4 * B * Cout * sizeof(float) + 64; }

int dvsof_flow_fold_grads(float *dW, const float *w, int Cout, int Ctot, int cx_off, int Cx, int cf_off,
                          const float *wh, const float *bh, const float *db_conv, const float *g, int B,
                          int H, int W, float *dwh, float *dbh, void *ws, size_t ws_bytes, void *stream)
{
    if (!dW || !w || !wh || !bh || !db_conv || !g || !dwh || !dbh || !ws || Cout < 1 || Ctot < 3 || Cx < 1 ||
        B < 1 || H < 1 || W < 1 || cf_off + 2 > Ctot || cx_off + Cx > Ctot)
        return DVSOF_EINVAL;
    if (ws_bytes < dvsof_flow_fold_workspace_bytes(B, Cout)) return DVSOF_ENOSPACE;
    hipStream_t st = as_stream(stream);
    float *part = (float *)ws;
    hipLaunchKernelGGL(flow_border_sums_kernel, dim3(4 * B), dim3(FOLD_NT), 0, st, g, B, H, W, Cout, part);
    DVSOF_LAUNCH_CHECK();
    FoldArgs A = {dW, w, wh, bh, db_conv, part, g, dwh, dbh, Cout, Ctot, cx_off, Cx, cf_off, B, H, W, 0, 0};
    A.nb_flow = (Cout * TAPS + FOLD_NT / 64 - 1) / (FOLD_NT / 64);
    A.nb_wh = (2 * Cx + 63) / 64;
    hipLaunchKernelGGL(flow_fold_grads_kernel, dim3(A.nb_flow + A.nb_wh + 1), dim3(FOLD_NT), 0, st, A);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
