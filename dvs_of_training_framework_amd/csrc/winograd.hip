// Winograd F(2x2, 3x3) for the wide 3x3 / stride-1 layers (EV-FlowNet's 512-channel
// residual blocks): 16 multiplies per 2x2 output tile and channel pair instead
// of 36, i.e. 2.25x fewer matrix-core FLOPs than the direct implicit GEMM.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray 2016, F(2x2,3x3))
//
// d = 4x4 input patch of one tile (stride 2 between tiles, zero padded),
// g = 3x3 kernel.  The element-wise product summed over input channels is 16
// independent GEMMs [tiles x Cin] x [Cin x Cout], one per Winograd component:
//
//   wino_input_kernel    V[16][T][C]  = B^T d B          (adds only)
//   gconv2_kernel        Mb[16][T][N] = V[g] * U[g]^T    (the LDS-DMA MFMA kernel run
//                                                         as a 1x1 conv with 16 "phases")
//   wino_output_kernel   y = epilogue(A^T Mb A)          (bias, residual / gradient
//                                                         addends, act' multiply, act)
//
// with U[16][N][C] = G g G^T made once per step by dvsof_conv2d_prepare
// (wino_weight_kernel), and the data-gradient form U'[16][C][N] derived from U
// by a transpose plus the component permutation (3,1,2,0) x (3,1,2,0): the
// 180-degree rotated kernel g' satisfies G g' G^T = P (G g G^T) P.
//
// Weight gradient: dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G, again 16
// GEMMs, now contracting over the tiles (the K-major LDS-DMA kernel of
// wgrad2.hip run as a 1x1-conv weight gradient with 16 phases):
//
//   wino_input_kernel    V[16][T][C]   = B^T d B
//   wino_gout_kernel     Z[16][T][N]   = A dY A^T
//   wgrad2_kernel        dU[16][S][N][C] = sum_t Z[g][t][n] V[g][t][c]   (S K-splits)
//   wino_dw_kernel       dW[n][3][3][c] = G^T (sum_S dU) G;  dbias = sum_S colsum(Z[5])
//
// (component 5 = (1,1) of A dY A^T is the plain 2x2 sum of dY, so the bias
// gradient is the column sum wgrad2 already takes of its A operand.)
//
// V and Mb live in a caller-provided scratch (dvsof_conv_desc_t.scratch); at
// the residual layers' size (T = 512 tiles, 512 channels) they are 16 MiB each
// and stay in the 256 MiB memory-side cache between the three launches.
// All transforms are exact up to f32 rounding of sums of at most four terms;
// results agree with the direct kernel to ~1e-6 relative (tests/test_gpu_conv.py).
#include "conv_common.h"

bool gconv2_eligible(const GConvParams &P, long long max_src_bytes, long long w_bytes);
int gconv2_launch(const GConvParams &P, int tile, hipStream_t st);
bool wgrad2_eligible(const WGradParams &P);
int wgrad2_launch(const WGradParams &P, int tile, int ntiles, hipStream_t st);

namespace {

__device__ __forceinline__ f32x4 ld4(const float *p) { return *(const f32x4 *)p; }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *(f32x4 *)p = v; }

// U[g][n][c] from w[n][3][3][c]; one thread per (n, channel quad)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w,
                                                          float *__restrict__ U, int N, int C)
{
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * c4n) return;
    const int n = (int)(idx / c4n), c = (int)(idx - (long long)n * c4n) * 4;
    f32x4 g[3][3], t[4][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = ld4(w + ((size_t)n * 9 + i * 3 + j) * C + c);
#pragma unroll
    for (int j = 0; j < 3; ++j) {   // t = G g
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    const size_t plane = (size_t)N * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // U = t G^T
        float *o = U + (size_t)(4 * i) * plane + (size_t)n * C + c;
        st4(o, t[i][0]);
        st4(o + plane, 0.5f * (t[i][0] + t[i][1] + t[i][2]));
        st4(o + 2 * plane, 0.5f * (t[i][0] - t[i][1] + t[i][2]));
        st4(o + 3 * plane, t[i][2]);
    }
}

// Ut[g'][c][n] = U[perm(g')][n][c], perm = (3,1,2,0) on both component indices
__global__ __launch_bounds__(256) void wino_weight_transpose_kernel(const float *__restrict__ U,
                                                                    float *__restrict__ Ut, int N,
                                                                    int C)
{
    __shared__ float tile[32][33];
    const int g = blockIdx.z, gy = g >> 2, gx = g & 3;
    const int py = gy == 0 ? 3 : gy == 3 ? 0 : gy, px = gx == 0 ? 3 : gx == 3 ? 0 : gx;
    const float *src = U + (size_t)(py * 4 + px) * N * C;
    float *dst = Ut + (size_t)g * N * C;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (n < N && c < C) ? src[(size_t)n * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, n = n0 + tx;
        if (c < C && n < N) dst[(size_t)c * N + n] = tile[tx][ty + 8 * i];
    }
}

struct WinoGeom {
    int B, H, W, Th, Tw, T;   // image, tiles per column / row, tiles in total
};

// V[g][t][c] = (B^T d B)[g]; one thread per (tile, channel quad), channel quad fastest
__global__ __launch_bounds__(256) void wino_input_kernel(const GSrc S, const WinoGeom G,
                                                         float *__restrict__ V)
{
    const int c4n = S.C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G.T * c4n) return;
    const int t = (int)(idx / c4n), c = (int)(idx - (long long)t * c4n) * 4;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const float *base = S.p + (size_t)b * S.sb + c;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 2 * tx - 1 + j;
            const bool ok = ((unsigned)y < (unsigned)G.H) & ((unsigned)x < (unsigned)G.W);
            d[i][j] = ok ? ld4(base + (size_t)y * S.sy + (size_t)x * S.sx) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 u[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // u = B^T d
        u[0][j] = d[0][j] - d[2][j];
        u[1][j] = d[1][j] + d[2][j];
        u[2][j] = d[2][j] - d[1][j];
        u[3][j] = d[1][j] - d[3][j];
    }
    const size_t plane = (size_t)G.T * S.C;
    float *o = V + (size_t)t * S.C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // V = u B
        st4(o + (size_t)(4 * i) * plane, u[i][0] - u[i][2]);
        st4(o + (size_t)(4 * i + 1) * plane, u[i][1] + u[i][2]);
        st4(o + (size_t)(4 * i + 2) * plane, u[i][2] - u[i][1]);
        st4(o + (size_t)(4 * i + 3) * plane, u[i][1] - u[i][3]);
    }
}

struct WinoOut {
    GDst D;
    const float *bias;
    float *zout;
    int act, bwd_act, N;
};

// y[b][2ty+dy][2tx+dx][n] = epilogue((A^T m A)[dy][dx]); one thread per (tile, channel quad)
__global__ __launch_bounds__(256) void wino_output_kernel(const float *__restrict__ Mb,
                                                          const WinoOut O, const WinoGeom G)
{
    const int n4n = O.N >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G.T * n4n) return;
    const int t = (int)(idx / n4n), n = (int)(idx - (long long)t * n4n) * 4;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const size_t plane = (size_t)G.T * O.N;
    const float *mp = Mb + (size_t)t * O.N + n;
    f32x4 m[4][4];
#pragma unroll
    for (int g = 0; g < 16; ++g) m[g >> 2][g & 3] = ld4(mp + (size_t)g * plane);
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // s = A^T m
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    f32x4 v[4];
    size_t o[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // y = s A
        v[2 * i] = s[i][0] + s[i][1] + s[i][2];
        v[2 * i + 1] = s[i][1] - s[i][2] - s[i][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            o[2 * i + j] = (size_t)b * O.D.sb + (size_t)(2 * ty + i) * O.D.sy + (size_t)(2 * tx + j) * O.D.sx + n;
    }
    if (O.bias) {
        const f32x4 bv = ld4(O.bias + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bv;
    }
    // all optional loads first, then the stores (stores count in vmcnt)
    f32x4 a1[4], a2[4], as[4];
    if (O.D.addend)
#pragma unroll
        for (int q = 0; q < 4; ++q) a1[q] = ld4(O.D.addend + o[q]);
    if (O.D.addend2)
#pragma unroll
        for (int q = 0; q < 4; ++q) a2[q] = ld4(O.D.addend2 + o[q]);
    if (O.D.actsrc)
#pragma unroll
        for (int q = 0; q < 4; ++q) as[q] = ld4(O.D.actsrc + o[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (O.D.addend) v[q] += a1[q];
        if (O.D.addend2) v[q] += a2[q];
        if (O.D.actsrc)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] *= act_bwd(as[q][e], O.bwd_act);
    }
    if (O.zout)
#pragma unroll
        for (int q = 0; q < 4; ++q) st4(O.zout + o[q], v[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_fwd(v[q][e], O.act);
        st4(O.D.p + o[q], y);
    }
}

// Z[g][t][n] = (A dY A^T)[g], dY = the tile's 2x2 output gradients (dense NHWC gout)
__global__ __launch_bounds__(256) void wino_gout_kernel(const float *__restrict__ gout,
                                                        const WinoGeom G, int N, float *__restrict__ Z)
{
    const int n4n = N >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)G.T * n4n) return;
    const int t = (int)(idx / n4n), n = (int)(idx - (long long)t * n4n) * 4;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const float *p = gout + (((size_t)b * G.H + 2 * ty) * G.W + 2 * tx) * N + n;
    const f32x4 d00 = ld4(p), d01 = ld4(p + N), d10 = ld4(p + (size_t)G.W * N),
                d11 = ld4(p + (size_t)G.W * N + N);
    // rows of A: (1,0) (1,1) (1,-1) (0,-1)
    f32x4 u[4][2];
    u[0][0] = d00;
    u[0][1] = d01;
    u[1][0] = d00 + d10;
    u[1][1] = d01 + d11;
    u[2][0] = d00 - d10;
    u[2][1] = d01 - d11;
    u[3][0] = -d10;
    u[3][1] = -d11;
    const size_t plane = (size_t)G.T * N;
    float *o = Z + (size_t)t * N + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st4(o + (size_t)(4 * i) * plane, u[i][0]);
        st4(o + (size_t)(4 * i + 1) * plane, u[i][0] + u[i][1]);
        st4(o + (size_t)(4 * i + 2) * plane, u[i][0] - u[i][1]);
        st4(o + (size_t)(4 * i + 3) * plane, -u[i][1]);
    }
}

// dW[n][3][3][c] = G^T (sum over the S slabs of dU[g][s][n][c]) G; the trailing
// workgroups add the S column-sum partials of component 5 into dbias.
__global__ __launch_bounds__(256) void wino_dw_kernel(const float *__restrict__ dU, int S, int N, int C,
                                                      float *__restrict__ dW, int nb_main,
                                                      const float *__restrict__ bias_part,
                                                      float *__restrict__ dbias)
{
    if ((int)blockIdx.x >= nb_main) {
        const int n = ((int)blockIdx.x - nb_main) * 256 + threadIdx.x;
        if (n < N) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += bias_part[(size_t)(5 * S + s) * N + n];
            dbias[n] = v;
        }
        return;
    }
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * c4n) return;
    const int n = (int)(idx / c4n), c = (int)(idx - (long long)n * c4n) * 4;
    const size_t plane = (size_t)N * C;
    f32x4 m[4][4];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float *p = dU + (size_t)g * S * plane + (size_t)n * C + c;
        f32x4 v = ld4(p);
        for (int s = 1; s < S; ++s) v += ld4(p + (size_t)s * plane);
        m[g >> 2][g & 3] = v;
    }
    // rows of G^T: (1,.5,.5,0) (0,.5,-.5,0) (0,.5,.5,1)
    f32x4 t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = m[0][j] + 0.5f * (m[1][j] + m[2][j]);
        t[1][j] = 0.5f * (m[1][j] - m[2][j]);
        t[2][j] = 0.5f * (m[1][j] + m[2][j]) + m[3][j];
    }
    float *o = dW + (size_t)n * 9 * C + c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        st4(o + (size_t)(3 * i) * C, t[i][0] + 0.5f * (t[i][1] + t[i][2]));
        st4(o + (size_t)(3 * i + 1) * C, 0.5f * (t[i][1] - t[i][2]));
        st4(o + (size_t)(3 * i + 2) * C, 0.5f * (t[i][1] + t[i][2]) + t[i][3]);
    }
}

}  // namespace

// A 3x3 / stride-1 / pad-1 problem over one dense NHWC source whose channel
// counts make the transforms' memory traffic (32 (C + N) bytes per output pixel)
// cheaper than the 5/9 of the matrix work they save: C N / (C + N) > ~100.
bool wino_eligible_shape(int nsrc, int layout_nhwc, int C, int N, int H, int W, int ksize, int stride,
                         int pad, int upsample, int mfma)
{
    static const bool off = getenv("DVSOF_NO_WINOGRAD") != nullptr;
    if (off) return false;
    if (nsrc != 1 || !layout_nhwc || upsample || ksize != 3 || stride != 1 || pad != 1) return false;
    if (mfma == 1) return false;   // bf16-rounded operands: the transforms amplify the rounding
    if ((C % 64) || (N % 64) || (H & 1) || (W & 1)) return false;
    return C >= 256 && N >= 256;
}

size_t wino_scratch_floats(int B, int H, int W, int C, int N)
{
    return (size_t)16 * B * (H / 2) * (W / 2) * ((size_t)C + N);
}

int wino_prepare(const float *weight, float *U, float *Ut, int N, int C, hipStream_t st)
{
    if (weight) {
        if (!U) return DVSOF_EINVAL;
        const long long n = (long long)N * (C / 4);
        hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           weight, U, N, C);
        DVSOF_LAUNCH_CHECK();
    }
    if (Ut) {
        if (!U) return DVSOF_EINVAL;
        hipLaunchKernelGGL(wino_weight_transpose_kernel, dim3((C + 31) / 32, (N + 31) / 32, 16),
                           dim3(256), 0, st, (const float *)U, Ut, N, C);
        DVSOF_LAUNCH_CHECK();
    }
    return DVSOF_OK;
}

// P: the direct problem (3x3, stride 1, pad 1, one NHWC source, one destination)
// with P.W = U[16][N][C].
int wino_launch(const GConvParams &P, float *scratch, size_t scratch_floats, hipStream_t st)
{
    const int C = P.Cin_tot, N = P.N;
    if (P.nsrc != 1 || P.ndst != 1 || P.src[0].flat || P.src[0].sc != 1 || P.dst[0].sc != 1 ||
        P.dst[0].C != N || P.src[0].C != C || P.Ho != P.Hv || P.Wo != P.Wv)
        return DVSOF_EINVAL;
    WinoGeom G = {P.B, P.Hv, P.Wv, P.Hv / 2, P.Wv / 2, P.B * (P.Hv / 2) * (P.Wv / 2)};
    if (!scratch || scratch_floats < wino_scratch_floats(P.B, P.Hv, P.Wv, C, N)) return DVSOF_ENOSPACE;
    float *V = scratch, *Mb = scratch + (size_t)16 * G.T * C;

    const long long nin = (long long)G.T * (C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, st,
                       P.src[0], G, V);
    DVSOF_LAUNCH_CHECK();

    GConvParams Q = {};
    Q.nsrc = 1;
    Q.src[0] = {V, (long long)G.T * C, G.T * C, C, 1, C, 0};
    Q.src_ph_stride = (long long)G.T * C;
    Q.ndst = 1;
    Q.dst[0] = {Mb, nullptr, nullptr, nullptr, (long long)G.T * N, G.T * N, N, 1, N, 0, 0};
    // phase g = 2 phy + phx -> plane g of Mb
    Q.dst[0].ph_y = 2 * G.T * N;
    Q.dst[0].ph_x = G.T * N;
    Q.W = P.W;
    Q.B = 1;
    Q.Hv = Q.Ho = 1;
    Q.Wv = Q.Wo = G.T;
    Q.up = UP_NONE;
    Q.stride = 1;
    Q.pad = 0;
    Q.ks = 1;
    Q.nph = 16;
    Q.ph_pad = 0;
    Q.w_phase_stride = (long long)N * C;
    Q.N = N;
    Q.Cin_tot = C;
    Q.M = G.T;
    Q.act = ACT_NONE;
    Q.bwd_act = ACT_NONE;
    Q.mfma_bf16 = P.mfma_bf16;
    if ((long long)16 * G.T * N * 4 >= 0x7fffffffLL || (long long)G.T * N >= 0x3fffffffLL ||
        !gconv2_eligible(Q, (long long)G.T * C * 4, (long long)16 * N * C * 4))
        return DVSOF_EINVAL;
    static const int tile = getenv("DVSOF_WINO_TILE") ? atoi(getenv("DVSOF_WINO_TILE")) : 3;   // tuning
    const int rc = gconv2_launch(Q, tile, st);
    if (rc) return rc;

    WinoOut O = {P.dst[0], P.bias, P.zout, P.act, P.bwd_act, N};
    const long long nout = (long long)G.T * (N / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st,
                       (const float *)Mb, O, G);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// K splits of the weight-gradient GEMMs (K = tiles): enough workgroups for two
// per CU with 64 x 128 tiles, at least 16 K steps each
static int wino_wgrad_splits(int T, int N, int C)
{
    static const int s_env = getenv("DVSOF_WINO_WGRAD_SPLITS") ? atoi(getenv("DVSOF_WINO_WGRAD_SPLITS")) : 0;
    int S = 1;
    if (s_env > 0) S = s_env;
    else {
        const long long tiles = (long long)16 * ((N + 63) / 64) * ((C + 127) / 128);
        while (tiles * S < 512 && T / (S * 2) >= 16 * BK) S *= 2;
    }
    while (S > 1 && (T + S - 1) / S < BK) --S;
    return S;
}

size_t wino_wgrad_workspace_floats(int B, int H, int W, int C, int N)
{
    const size_t T = (size_t)B * (H / 2) * (W / 2);
    const int S = wino_wgrad_splits((int)T, N, C);
    return 16 * T * ((size_t)C + N) + (size_t)16 * S * N * C + (size_t)16 * S * N;
}

int wino_wgrad_launch(const GSrc &X, const float *gout, float *dW, float *dbias, int B, int H, int W,
                      int C, int N, int mfma_bf16, float *ws, size_t ws_floats, hipStream_t st)
{
    if (X.flat || X.sc != 1 || X.C != C) return DVSOF_EINVAL;
    WinoGeom G = {B, H, W, H / 2, W / 2, B * (H / 2) * (W / 2)};
    if (!ws || ws_floats < wino_wgrad_workspace_floats(B, H, W, C, N)) return DVSOF_ENOSPACE;
    const int S = wino_wgrad_splits(G.T, N, C);
    float *V = ws, *Z = V + (size_t)16 * G.T * C, *dU = Z + (size_t)16 * G.T * N;
    float *bias_part = dU + (size_t)16 * S * N * C;

    const long long nin = (long long)G.T * (C / 4), ng = (long long)G.T * (N / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, st, X, G, V);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(wino_gout_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, gout, G, N, Z);
    DVSOF_LAUNCH_CHECK();

    WGradParams Q = {};
    Q.nsrc = 1;
    Q.src[0] = {V, (long long)G.T * C, G.T * C, C, 1, C, 0};
    Q.src_ph_stride = (long long)G.T * C;
    Q.gout = Z;
    Q.dW = dU;
    Q.dbias = dbias ? bias_part : nullptr;
    Q.B = 1;
    Q.Hv = Q.Ho = 1;
    Q.Wv = Q.Wo = G.T;
    Q.up = UP_NONE;
    Q.stride = 1;
    Q.pad = 0;
    Q.ks = 1;
    Q.Cout = N;
    Q.Cin_tot = C;
    Q.M = G.T;
    Q.S = S;
    Q.klen = (((G.T + S - 1) / S) + BK - 1) / BK * BK;
    Q.g_sb = (long long)G.T * N;
    Q.g_sy = G.T * N;
    Q.g_sx = N;
    Q.g_py = 2 * G.T * N;   // phase g = 2 phy + phx -> plane g of Z
    Q.g_px = G.T * N;
    Q.nph = 16;
    Q.ph_pad = 0;
    Q.mfma_bf16 = mfma_bf16;
    static const int tile = getenv("DVSOF_WINO_WGRAD_TILE") ? atoi(getenv("DVSOF_WINO_WGRAD_TILE")) : 4;
    const int bn = (tile == 2 || tile == 3) ? 64 : 128;
    const int nt = (C + bn - 1) / bn;
    Q.tile_begin[0] = 0;
    Q.tile_begin[1] = nt;
    if ((G.T % BK) || (long long)16 * G.T * (C > N ? C : N) * 4 >= 0x7fffffffLL || !wgrad2_eligible(Q))
        return DVSOF_EINVAL;
    const int rc = wgrad2_launch(Q, tile, nt, st);
    if (rc) return rc;

    const long long nw = (long long)N * (C / 4);
    const int nb_main = (int)((nw + 255) / 256), nb_bias = dbias ? (N + 255) / 256 : 0;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)(nb_main + nb_bias)), dim3(256), 0, st,
                       (const float *)dU, S, N, C, dW, nb_main, (const float *)bias_part, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
