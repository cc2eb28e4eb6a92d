// Winograd convolution for the wide 3x3 / stride-1 layers (EV-FlowNet's 512-channel
// residual blocks): F(4x4,3x3) when the image sides are multiples of 4 (36
// multiplies per 4x4 output tile and channel pair instead of 144: 4x fewer
// matrix-core FLOPs than the direct implicit GEMM), else F(2x2,3x3) (16 instead
// of 36: 2.25x fewer).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray 2016; points 0, +-1, +-2, inf)
//
// d = (m+2)x(m+2) input patch of one tile (stride m between tiles, zero padded),
// g = 3x3 kernel, m = 2 | 4.  The element-wise product summed over input channels
// is NG = (m+2)^2 independent GEMMs [tiles x Cin] x [Cin x Cout], one per component:
//
//   wino_input_kernel    V[NG][T][C]  = B^T d B
//   gconv2_kernel        Mb[NG][T][N] = V[g] * U[g]^T    (the LDS-DMA MFMA kernel run
//                                                         as a 1x1 conv with NG "phases")
//   wino_output_kernel   y = epilogue(A^T Mb A)          (bias, residual / gradient
//                                                         addends, act' multiply, act)
//
// with U[NG][N][C] = G g G^T made once per step by dvsof_conv2d_prepare
// (wino_weight_kernel) and the data-gradient form U'[NG][C][N] = G g' G^T of the
// 180-degree rotated, transposed kernel g'.
//
// Weight gradient: dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G, again NG
// GEMMs, now contracting over the tiles (the K-major LDS-DMA kernel of
// wgrad2.hip run as a 1x1-conv weight gradient with NG phases):
//
//   wino_input_kernel    V[NG][T][C]   = B^T d B
//   wino_gout_kernel     Z[NG][T][N]   = A dY A^T
//   wgrad2_kernel        dU[NG][S][N][C] = sum_t Z[g][t][n] V[g][t][c]   (S K-splits)
//   wino_dw_kernel       dW[n][3][3][c] = G^T (sum_S dU) G;  dbias = sum_S colsum(Z[(1,1)])
//
// (row 1 of A is all ones, so component (1,1) of A dY A^T is the plain sum of the
// tile's dY and the bias gradient is the column sum wgrad2 already takes of its A
// operand.)
//
// V, Mb, Z and dU live in caller-provided scratch (dvsof_conv_desc_t.scratch, the
// weight-gradient workspace); at the residual layers' size they are 9-38 MiB
// and stay in the 256 MiB memory-side cache between the launches.
// Accuracy in f32 (tests/test_gpu_conv.py, tools/winograd_error.py): F(2x2) ~1e-6,
// F(4x4) ~1e-5 of the output peak (direct kernel: ~3e-7); the parity bar is 1e-3.
#include "conv_common.h"

bool gconv2_eligible(const GConvParams &P, long long max_src_bytes, long long w_bytes);
int gconv2_launch(const GConvParams &P, int tile, hipStream_t st);
bool wgrad2_eligible(const WGradParams &P);
int wgrad2_launch(const WGradParams &P, int tile, int ntiles, hipStream_t st);

namespace {

__device__ __forceinline__ f32x4 ld4(const float *p) { return *(const f32x4 *)p; }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *(f32x4 *)p = v; }

// element type of the transform kernels: VW = 4 channels per thread (16-byte
// accesses) or 1 (4x the threads; the small problems are latency bound)
template <int VW>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T ld(const float *p) { return *(const f32x4 *)p; }
    static __device__ __forceinline__ void st(float *p, T v) { *(f32x4 *)p = v; }
    static __device__ __forceinline__ T zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ T actf(T v, int act)
    {
        T y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_fwd(v[e], act);
        return y;
    }
    static __device__ __forceinline__ T actb(T s, int act)
    {
        T y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_bwd(s[e], act);
        return y;
    }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T actf(T v, int act) { return act_fwd(v, act); }
    static __device__ __forceinline__ T actb(T s, int act) { return act_bwd(s, act); }
};

// ---- the 1-D transforms (F = output tile side m; NA = m + 2 points)
template <int F>
struct Wino;

template <>
struct Wino<2> {
    static constexpr int NA = 4;
    // B^T: data
    template <typename T>
    static __device__ __forceinline__ void bt(const T (&d)[4], T (&o)[4])
    {
        o[0] = d[0] - d[2];
        o[1] = d[1] + d[2];
        o[2] = d[2] - d[1];
        o[3] = d[1] - d[3];
    }
    // G: kernel taps -> points
    template <typename T>
    static __device__ __forceinline__ void g(const T (&w)[3], T (&o)[4])
    {
        o[0] = w[0];
        o[1] = 0.5f * (w[0] + w[1] + w[2]);
        o[2] = 0.5f * (w[0] - w[1] + w[2]);
        o[3] = w[2];
    }
    // A^T: points -> outputs
    template <typename T>
    static __device__ __forceinline__ void at(const T (&m)[4], T (&o)[2])
    {
        o[0] = m[0] + m[1] + m[2];
        o[1] = m[1] - m[2] - m[3];
    }
    // A: output gradients -> points
    template <typename T>
    static __device__ __forceinline__ void a(const T (&y)[2], T (&o)[4])
    {
        o[0] = y[0];
        o[1] = y[0] + y[1];
        o[2] = y[0] - y[1];
        o[3] = -y[1];
    }
    // G^T: points -> kernel-tap gradients
    template <typename T>
    static __device__ __forceinline__ void gt(const T (&m)[4], T (&o)[3])
    {
        o[0] = m[0] + 0.5f * (m[1] + m[2]);
        o[1] = 0.5f * (m[1] - m[2]);
        o[2] = 0.5f * (m[1] + m[2]) + m[3];
    }
};

template <>
struct Wino<4> {
    static constexpr int NA = 6;
    template <typename T>
    static __device__ __forceinline__ void bt(const T (&d)[6], T (&o)[6])
    {
        o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        o[1] = (d[3] + d[4]) - 4.f * (d[1] + d[2]);
        o[2] = 4.f * (d[1] - d[2]) + (d[4] - d[3]);
        o[3] = 2.f * (d[3] - d[1]) + (d[4] - d[2]);
        o[4] = 2.f * (d[1] - d[3]) + (d[4] - d[2]);
        o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
    template <typename T>
    static __device__ __forceinline__ void g(const T (&w)[3], T (&o)[6])
    {
        const T s = w[0] + w[2];
        const T q = (1.f / 24.f) * w[0] + (1.f / 6.f) * w[2], h = (1.f / 12.f) * w[1];
        o[0] = 0.25f * w[0];
        o[1] = (-1.f / 6.f) * (s + w[1]);
        o[2] = (-1.f / 6.f) * (s - w[1]);
        o[3] = q + h;
        o[4] = q - h;
        o[5] = w[2];
    }
    template <typename T>
    static __device__ __forceinline__ void at(const T (&m)[6], T (&o)[4])
    {
        const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        o[0] = m[0] + s12 + s34;
        o[1] = d12 + 2.f * d34;
        o[2] = s12 + 4.f * s34;
        o[3] = d12 + 8.f * d34 + m[5];
    }
    template <typename T>
    static __device__ __forceinline__ void a(const T (&y)[4], T (&o)[6])
    {
        const T e = y[0] + y[2], f = y[1] + y[3];
        const T e4 = y[0] + 4.f * y[2], f4 = 2.f * y[1] + 8.f * y[3];
        o[0] = y[0];
        o[1] = e + f;
        o[2] = e - f;
        o[3] = e4 + f4;
        o[4] = e4 - f4;
        o[5] = y[3];
    }
    template <typename T>
    static __device__ __forceinline__ void gt(const T (&m)[6], T (&o)[3])
    {
        const T s12 = m[1] + m[2], s34 = m[3] + m[4];
        o[0] = 0.25f * m[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        o[1] = (1.f / 6.f) * (m[2] - m[1]) + (1.f / 12.f) * (m[3] - m[4]);
        o[2] = (1.f / 6.f) * (s34 - s12) + m[5];
    }
};

// U[g][n][c] = (G w[n] G^T)[g] from w[n][3][3][c]; one thread per (n, channel quad).
// TRANSPOSED: the data-gradient form Ut[g][c][n] of the 180-degree rotated kernel;
// threads are then ordered n-fastest so that the (larger) write side is coalesced.
template <int F, bool TRANSPOSED>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w,
                                                          float *__restrict__ U, int N, int C)
{
    constexpr int NA = Wino<F>::NA;
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * c4n) return;
    int n, c;
    if (TRANSPOSED) {
        c = (int)(idx / N);
        n = (int)(idx - (long long)c * N);
        c *= 4;
    } else {
        n = (int)(idx / c4n);
        c = (int)(idx - (long long)n * c4n) * 4;
    }
    f32x4 g[3][3], t[3][NA];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tap = TRANSPOSED ? (2 - i) * 3 + (2 - j) : i * 3 + j;
            g[i][j] = ld4(w + ((size_t)n * 9 + tap) * C + c);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) Wino<F>::g(g[i], t[i]);   // along kx
    const size_t plane = (size_t)N * C;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const f32x4 col[3] = {t[0][j], t[1][j], t[2][j]};
        f32x4 u[NA];
        Wino<F>::g(col, u);                                // along ky
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float *o = U + (size_t)(i * NA + j) * plane;
            if (TRANSPOSED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[(size_t)(c + e) * N + n] = u[i][e];
            } else {
                st4(o + (size_t)n * C + c, u[i]);
            }
        }
    }
}

struct WinoGeom {
    int B, H, W, Th, Tw, T;   // image, tiles per column / row, tiles in total
};

// V[g][t][c] = (B^T d B)[g]; one thread per (tile, VW channels), channels fastest
template <int F, int NT, int VW>
__global__ __launch_bounds__(NT) void wino_input_kernel(const GSrc S, const WinoGeom G,
                                                        float *__restrict__ V)
{
    constexpr int NA = Wino<F>::NA;
    typedef typename Vec<VW>::T T;
    const int cn = S.C / VW;
    const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
    if (idx >= (long long)G.T * cn) return;
    const int t = (int)(idx / cn), c = (int)(idx - (long long)t * cn) * VW;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const float *base = S.p + (size_t)b * S.sb + c;
    T u[NA][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {   // column j of the patch: u[.][j] = B^T d[.][j]
        const int x = F * tx - 1 + j;
        T d[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int y = F * ty - 1 + i;
            const bool ok = ((unsigned)y < (unsigned)G.H) & ((unsigned)x < (unsigned)G.W);
            d[i] = ok ? Vec<VW>::ld(base + (size_t)y * S.sy + (size_t)x * S.sx) : Vec<VW>::zero();
        }
        T o[NA];
        Wino<F>::bt(d, o);
#pragma unroll
        for (int i = 0; i < NA; ++i) u[i][j] = o[i];
    }
    const size_t plane = (size_t)G.T * S.C;
    float *o = V + (size_t)t * S.C + c;
#pragma unroll
    for (int i = 0; i < NA; ++i) {   // V[i][.] = u[i][.] B
        T v[NA];
        Wino<F>::bt(u[i], v);
#pragma unroll
        for (int j = 0; j < NA; ++j) Vec<VW>::st(o + (size_t)(i * NA + j) * plane, v[j]);
    }
}

struct WinoOut {
    GDst D;
    const float *bias;
    float *zout;
    int act, bwd_act, N;
};

// y[b][F ty + i][F tx + j][n] = epilogue((A^T m A)[i][j]); one thread per (tile, VW channels)
template <int F, int NT, int VW>
__global__ __launch_bounds__(NT) void wino_output_kernel(const float *__restrict__ Mb, const WinoOut O,
                                                         const WinoGeom G)
{
    constexpr int NA = Wino<F>::NA;
    typedef typename Vec<VW>::T T;
    const int nn = O.N / VW;
    const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
    if (idx >= (long long)G.T * nn) return;
    const int t = (int)(idx / nn), n = (int)(idx - (long long)t * nn) * VW;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const size_t plane = (size_t)G.T * O.N;
    const float *mp = Mb + (size_t)t * O.N + n;
    T s[F][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {   // s[.][j] = A^T m[.][j]
        T m[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) m[i] = Vec<VW>::ld(mp + (size_t)(i * NA + j) * plane);
        T o[F];
        Wino<F>::at(m, o);
#pragma unroll
        for (int i = 0; i < F; ++i) s[i][j] = o[i];
    }
    T bv = Vec<VW>::zero();
    if (O.bias) bv = Vec<VW>::ld(O.bias + n);
    const size_t o0 = (size_t)b * O.D.sb + (size_t)(F * ty) * O.D.sy + (size_t)(F * tx) * O.D.sx + n;
#pragma unroll
    for (int i = 0; i < F; ++i) {   // one output row at a time: loads, then stores
        T v[F];
        Wino<F>::at(s[i], v);
        size_t o[F];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            o[j] = o0 + (size_t)i * O.D.sy + (size_t)j * O.D.sx;
            v[j] += bv;
        }
        T a1[F], a2[F], as[F];
        if (O.D.addend)
#pragma unroll
            for (int j = 0; j < F; ++j) a1[j] = Vec<VW>::ld(O.D.addend + o[j]);
        if (O.D.addend2)
#pragma unroll
            for (int j = 0; j < F; ++j) a2[j] = Vec<VW>::ld(O.D.addend2 + o[j]);
        if (O.D.actsrc)
#pragma unroll
            for (int j = 0; j < F; ++j) as[j] = Vec<VW>::ld(O.D.actsrc + o[j]);
#pragma unroll
        for (int j = 0; j < F; ++j) {
            if (O.D.addend) v[j] += a1[j];
            if (O.D.addend2) v[j] += a2[j];
            if (O.D.actsrc) v[j] *= Vec<VW>::actb(as[j], O.bwd_act);
        }
        if (O.zout)
#pragma unroll
            for (int j = 0; j < F; ++j) Vec<VW>::st(O.zout + o[j], v[j]);
#pragma unroll
        for (int j = 0; j < F; ++j) Vec<VW>::st(O.D.p + o[j], Vec<VW>::actf(v[j], O.act));
    }
}

// The output transform of a layer whose result feeds ANOTHER Winograd layer of the same frame
// (the residual chain: forward y -> next convolution; data gradient dz -> the data gradient
// and the weight gradient below), with the consumer's transforms made from the workgroup's own
// copy of the tile it just wrote -- one launch and one pass over y instead of three:
//     y  = epilogue(A^T Mb A)                      exactly wino_output_kernel's
//     Vn = B^T y B   [NG][T][N]                    the consumer's wino_input_kernel   (or null)
//     Zn = A y A^T   [NG][T][N]                    its weight gradient's wino_gout_kernel (or null)
// B^T y B needs a 1-pixel halo, i.e. the neighbouring tiles' outputs: a workgroup owns a WHOLE
// image x 16 channels (thread = (tile, channel); frames of <= 64 tiles), stages y in LDS with
// a zero border and transforms from there.  Bitwise the same y, Vn, Zn as the three kernels.
constexpr int WC_CG = 16;      // channels per workgroup
template <int F>
__global__ __launch_bounds__(1024) void wino_chain_kernel(const float *__restrict__ Mb, const WinoOut O,
                                                          const WinoGeom G, float *__restrict__ Vn,
                                                          float *__restrict__ Zn)
{
    constexpr int NA = Wino<F>::NA;
    extern __shared__ float ybuf[];     // [(H + 2)][(W + 2)][WC_CG]
    const int LWp = G.W + 2;
    const int tid = threadIdx.x, cl = tid % WC_CG, tl = tid / WC_CG;
    const int n = blockIdx.x * WC_CG + cl, b = blockIdx.y;
    const int tx = tl % G.Tw, ty = tl / G.Tw;
    const int t = b * (G.Th * G.Tw) + tl;
    // zero border of the staged frame
    for (int i = tid; i < 2 * (G.W + 2 + G.H) * WC_CG; i += blockDim.x) {
        const int c = i % WC_CG, q = i / WC_CG;
        int yy, xx;
        if (q < G.W + 2) { yy = 0; xx = q; }
        else if (q < 2 * (G.W + 2)) { yy = G.H + 1; xx = q - (G.W + 2); }
        else if (q < 2 * (G.W + 2) + G.H) { yy = q - 2 * (G.W + 2) + 1; xx = 0; }
        else { yy = q - 2 * (G.W + 2) - G.H + 1; xx = G.W + 1; }
        ybuf[(yy * LWp + xx) * WC_CG + c] = 0.f;
    }
    const size_t plane = (size_t)G.T * O.N;
    const float *mp = Mb + (size_t)t * O.N + n;
    float s[F][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {   // s[.][j] = A^T m[.][j]
        float m[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) m[i] = mp[(size_t)(i * NA + j) * plane];
        float o[F];
        Wino<F>::at(m, o);
#pragma unroll
        for (int i = 0; i < F; ++i) s[i][j] = o[i];
    }
    const float bv = O.bias ? O.bias[n] : 0.f;
    const size_t o0 = (size_t)b * O.D.sb + (size_t)(F * ty) * O.D.sy + (size_t)(F * tx) * O.D.sx + n;
    float yv[F][F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        float v[F];
        Wino<F>::at(s[i], v);
        size_t o[F];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            o[j] = o0 + (size_t)i * O.D.sy + (size_t)j * O.D.sx;
            v[j] += bv;
        }
        float a1[F], a2[F], as[F];
        if (O.D.addend)
#pragma unroll
            for (int j = 0; j < F; ++j) a1[j] = O.D.addend[o[j]];
        if (O.D.addend2)
#pragma unroll
            for (int j = 0; j < F; ++j) a2[j] = O.D.addend2[o[j]];
        if (O.D.actsrc)
#pragma unroll
            for (int j = 0; j < F; ++j) as[j] = O.D.actsrc[o[j]];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            if (O.D.addend) v[j] += a1[j];
            if (O.D.addend2) v[j] += a2[j];
            if (O.D.actsrc) v[j] *= act_bwd(as[j], O.bwd_act);
        }
        if (O.zout)
#pragma unroll
            for (int j = 0; j < F; ++j) O.zout[o[j]] = v[j];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float y = act_fwd(v[j], O.act);
            O.D.p[o[j]] = y;
            yv[i][j] = y;
            ybuf[((F * ty + i + 1) * LWp + F * tx + j + 1) * WC_CG + cl] = y;
        }
    }
    if (Zn) {   // A y A^T of the tile itself (wino_gout_kernel)
        float u[NA][F];
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float d[F] = {yv[0][j], yv[1][j], yv[2][j], yv[3][j]};
            float o[NA];
            Wino<F>::a(d, o);
#pragma unroll
            for (int i = 0; i < NA; ++i) u[i][j] = o[i];
        }
        float *zo = Zn + (size_t)t * O.N + n;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float v[NA];
            Wino<F>::a(u[i], v);
#pragma unroll
            for (int j = 0; j < NA; ++j) zo[(size_t)(i * NA + j) * plane] = v[j];
        }
    }
    if (!Vn) return;
    __syncthreads();
    float u[NA][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {   // column j of the patch (wino_input_kernel)
        float d[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) d[i] = ybuf[((F * ty + i) * LWp + F * tx + j) * WC_CG + cl];
        float o[NA];
        Wino<F>::bt(d, o);
#pragma unroll
        for (int i = 0; i < NA; ++i) u[i][j] = o[i];
    }
    float *vo = Vn + (size_t)t * O.N + n;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float v[NA];
        Wino<F>::bt(u[i], v);
#pragma unroll
        for (int j = 0; j < NA; ++j) vo[(size_t)(i * NA + j) * plane] = v[j];
    }
}

// Z[g][t][n] = (A dY A^T)[g], dY = the tile's F x F output gradients (dense NHWC gout)
template <int F, int NT, int VW>
__global__ __launch_bounds__(NT) void wino_gout_kernel(const float *__restrict__ gout, const WinoGeom G,
                                                       int N, float *__restrict__ Z)
{
    constexpr int NA = Wino<F>::NA;
    typedef typename Vec<VW>::T T;
    const int nn = N / VW;
    const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
    if (idx >= (long long)G.T * nn) return;
    const int t = (int)(idx / nn), n = (int)(idx - (long long)t * nn) * VW;
    const int tx = t % G.Tw, r = t / G.Tw, ty = r % G.Th, b = r / G.Th;
    const float *p = gout + (((size_t)b * G.H + F * ty) * G.W + F * tx) * N + n;
    T u[NA][F];
#pragma unroll
    for (int j = 0; j < F; ++j) {   // u[.][j] = A dY[.][j]
        T d[F];
#pragma unroll
        for (int i = 0; i < F; ++i) d[i] = Vec<VW>::ld(p + ((size_t)i * G.W + j) * N);
        T o[NA];
        Wino<F>::a(d, o);
#pragma unroll
        for (int i = 0; i < NA; ++i) u[i][j] = o[i];
    }
    const size_t plane = (size_t)G.T * N;
    float *o = Z + (size_t)t * N + n;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        T v[NA];
        Wino<F>::a(u[i], v);
#pragma unroll
        for (int j = 0; j < NA; ++j) Vec<VW>::st(o + (size_t)(i * NA + j) * plane, v[j]);
    }
}

// dW[n][3][3][c] = G^T (sum over the S slabs of dU[g][s][n][c]) G; the trailing
// workgroups add the S column-sum partials of component (1,1) into dbias.
template <int F>
__global__ __launch_bounds__(256) void wino_dw_kernel(const float *__restrict__ dU, int S, int N, int C,
                                                      float *__restrict__ dW, int nb_main,
                                                      const float *__restrict__ bias_part,
                                                      float *__restrict__ dbias)
{
    constexpr int NA = Wino<F>::NA;
    if ((int)blockIdx.x >= nb_main) {
        const int n = ((int)blockIdx.x - nb_main) * 256 + threadIdx.x;
        if (n < N) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += bias_part[(size_t)((NA + 1) * S + s) * N + n];
            dbias[n] = v;
        }
        return;
    }
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * c4n) return;
    const int n = (int)(idx / c4n), c = (int)(idx - (long long)n * c4n) * 4;
    const size_t plane = (size_t)N * C;
    f32x4 t[3][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {   // t[.][j] = G^T m[.][j]
        f32x4 m[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float *p = dU + (size_t)(i * NA + j) * S * plane + (size_t)n * C + c;
            f32x4 v = ld4(p);
            for (int s = 1; s < S; ++s) v += ld4(p + (size_t)s * plane);
            m[i] = v;
        }
        f32x4 o[3];
        Wino<F>::gt(m, o);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i][j] = o[i];
    }
    float *o = dW + (size_t)n * 9 * C + c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f32x4 v[3];
        Wino<F>::gt(t[i], v);
#pragma unroll
        for (int j = 0; j < 3; ++j) st4(o + (size_t)(3 * i + j) * C, v[j]);
    }
}

inline int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

inline unsigned nblocks(long long n, int nt) { return (unsigned)((n + nt - 1) / nt); }

}  // namespace

// Output tile side of the forward / data-gradient evaluation: 4 when the image
// allows it (DVSOF_WINO_F=2 forces the 2x2 form), 0 = direct kernel.
// The tile count is the row dimension of the component GEMMs and the
// transformed weights are 4x (1.8x) the raw ones: with few tiles the layer is
// bound by reading them.  Measured, batch 1 at 16x16 (16 | 64 tiles): whole
// forward pass 1.88 | 1.63 ms against 0.75 ms direct; batch 8 (128 tiles): 52 us
// per layer against 87 us direct.
// The bf16x3 operand mode (mfma = 2, ~2^-16 per product) stays with the 2x2
// form: the 4x4 transforms amplify the product error past its 1e-4 test bound.
int wino_tile(int B, int H, int W, int mfma)
{
    static const int f_env = env_int("DVSOF_WINO_F", 0);
    if ((H & 1) || (W & 1)) return 0;
    const long long t4 = (long long)B * (H / 4) * (W / 4), t2 = (long long)B * (H / 2) * (W / 2);
    if (f_env != 2 && mfma == 0 && (H % 4) == 0 && (W % 4) == 0 && t4 >= 64) return 4;
    return t2 >= 128 ? 2 : 0;
}

int wino_components(int B, int H, int W, int mfma)
{
    const int f = wino_tile(B, H, W, mfma);
    return (f + 2) * (f + 2);
}

// A 3x3 / stride-1 / pad-1 problem over one dense NHWC source whose channel
// counts make the transforms' memory traffic (~32 (C + N) bytes per output pixel
// for the 2x2 form) cheaper than the matrix work they save: C N / (C + N) > ~100.
bool wino_eligible_shape(int nsrc, int layout_nhwc, int C, int N, int B, int H, int W, int ksize,
                         int stride, int pad, int upsample, int mfma)
{
    static const bool off = getenv("DVSOF_NO_WINOGRAD") != nullptr;
    if (off) return false;
    if (nsrc != 1 || !layout_nhwc || upsample || ksize != 3 || stride != 1 || pad != 1) return false;
    if (mfma == 1 || mfma == 3) return false;   // bf16-rounded operands: the transforms amplify the rounding
    if ((C % 64) || (N % 64) || C < 256 || N < 256) return false;
    return wino_tile(B, H, W, mfma == 2 ? 2 : 0) != 0;
}

// Can the output transform of a Winograd evaluation at this frame also make the consumer's
// forms (wino_chain_kernel)?  F(4x4) form, whole image per workgroup: <= 64 tiles per image, a
// multiple of 4 (whole waves), the staged frame within 64 KB of LDS.  DVSOF_NO_WINO_CHAIN=1: off
bool wino_chain_ok(int B, int H, int W, int N, int mfma)
{
    static const bool off = getenv("DVSOF_NO_WINO_CHAIN") != nullptr;
    if (off || wino_tile(B, H, W, mfma) != 4 || (N % WC_CG)) return false;
    const int tiles = (H / 4) * (W / 4);
    return tiles <= 64 && (tiles % 4) == 0 && (size_t)(H + 2) * (W + 2) * WC_CG * 4 <= 64 * 1024;
}

size_t wino_scratch_floats(int B, int H, int W, int C, int N, int mfma)
{
    const int f = wino_tile(B, H, W, mfma);
    if (f == 0) return 0;
    return (size_t)(f + 2) * (f + 2) * B * (H / f) * (W / f) * ((size_t)C + N);
}

// U (forward form) and / or Ut (data-gradient form) from the raw weights
int wino_prepare(const float *weight, float *U, float *Ut, int N, int C, int B, int H, int W, int mfma,
                 hipStream_t st)
{
    if (!weight) return DVSOF_EINVAL;
    const int f = wino_tile(B, H, W, mfma);
    if (f == 0) return DVSOF_EINVAL;
    const unsigned nb = nblocks((long long)N * (C / 4), 256);
    if (U) {
        if (f == 4) hipLaunchKernelGGL((wino_weight_kernel<4, false>), dim3(nb), dim3(256), 0, st, weight, U, N, C);
        else hipLaunchKernelGGL((wino_weight_kernel<2, false>), dim3(nb), dim3(256), 0, st, weight, U, N, C);
        DVSOF_LAUNCH_CHECK();
    }
    if (Ut) {
        if (f == 4) hipLaunchKernelGGL((wino_weight_kernel<4, true>), dim3(nb), dim3(256), 0, st, weight, Ut, N, C);
        else hipLaunchKernelGGL((wino_weight_kernel<2, true>), dim3(nb), dim3(256), 0, st, weight, Ut, N, C);
        DVSOF_LAUNCH_CHECK();
    }
    return DVSOF_OK;
}

template <int F>
static int wino_launch_f(const GConvParams &P, float *scratch, const WinoChain &ch, hipStream_t st)
{
    constexpr int NA = Wino<F>::NA, NG = NA * NA;
    constexpr int NT = 256;
    const int C = P.Cin_tot, N = P.N;
    WinoGeom G = {P.B, P.Hv, P.Wv, P.Hv / F, P.Wv / F, P.B * (P.Hv / F) * (P.Wv / F)};
    // (ch.v_pre: the producer's output transform already made this call's transformed input)
    float *V = ch.v_pre ? const_cast<float *>(ch.v_pre) : scratch, *Mb = scratch + (size_t)NG * G.T * C;

    // few tiles: one channel per thread (4x the threads; these launches are latency bound)
    static const int vw_env = env_int("DVSOF_WINO_VW", 0);
    const bool scalar = vw_env ? vw_env == 1 : (long long)G.T * (C > N ? C : N) / 4 < 256 * 256;
    if (ch.v_pre) {
    } else if (scalar)
        hipLaunchKernelGGL((wino_input_kernel<F, NT, 1>), dim3(nblocks((long long)G.T * C, NT)), dim3(NT), 0,
                           st, P.src[0], G, V);
    else
        hipLaunchKernelGGL((wino_input_kernel<F, NT, 4>), dim3(nblocks((long long)G.T * (C / 4), NT)),
                           dim3(NT), 0, st, P.src[0], G, V);
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
    Q.nph = NG;
    Q.ph_pad = 0;
    Q.w_phase_stride = (long long)N * C;
    Q.N = N;
    Q.Cin_tot = C;
    Q.M = G.T;
    Q.act = ACT_NONE;
    Q.bwd_act = ACT_NONE;
    Q.mfma_bf16 = P.mfma_bf16;
    if ((long long)NG * G.T * N * 4 >= 0x7fffffffLL ||
        !gconv2_eligible(Q, (long long)G.T * C * 4, (long long)NG * N * C * 4))
        return DVSOF_EINVAL;
    static const int tile = env_int("DVSOF_WINO_TILE", 3);   // tuning
    const int rc = gconv2_launch(Q, tile, st);
    if (rc) return rc;

    WinoOut O = {P.dst[0], P.bias, P.zout, P.act, P.bwd_act, N};
    if (ch.v_next || ch.z_next) {
        if constexpr (F == 4) {
            if (!wino_chain_ok(P.B, P.Hv, P.Wv, N, P.mfma_bf16)) return DVSOF_EINVAL;
            const size_t lds = (size_t)(G.H + 2) * (G.W + 2) * WC_CG * sizeof(float);
            hipLaunchKernelGGL((wino_chain_kernel<4>), dim3(N / WC_CG, G.B), dim3(G.Th * G.Tw * WC_CG), lds, st,
                               (const float *)Mb, O, G, ch.v_next, ch.z_next);
            DVSOF_LAUNCH_CHECK();
            return DVSOF_OK;
        } else {
            return DVSOF_EINVAL;
        }
    }
    if (scalar)
        hipLaunchKernelGGL((wino_output_kernel<F, NT, 1>), dim3(nblocks((long long)G.T * N, NT)), dim3(NT), 0,
                           st, (const float *)Mb, O, G);
    else
        hipLaunchKernelGGL((wino_output_kernel<F, NT, 4>), dim3(nblocks((long long)G.T * (N / 4), NT)),
                           dim3(NT), 0, st, (const float *)Mb, O, G);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// P: the direct problem (3x3, stride 1, pad 1, one NHWC source, one destination)
// with P.W = U[NG][N][C].
int wino_launch(const GConvParams &P, float *scratch, size_t scratch_floats, const WinoChain &ch, hipStream_t st)
{
    const int C = P.Cin_tot, N = P.N;
    if (P.nsrc != 1 || P.ndst != 1 || P.src[0].flat || P.src[0].sc != 1 || P.dst[0].sc != 1 ||
        P.dst[0].C != N || P.src[0].C != C || P.Ho != P.Hv || P.Wo != P.Wv)
        return DVSOF_EINVAL;
    if (!scratch || scratch_floats < wino_scratch_floats(P.B, P.Hv, P.Wv, C, N, P.mfma_bf16))
        return DVSOF_ENOSPACE;
    const int f = wino_tile(P.B, P.Hv, P.Wv, P.mfma_bf16);
    if (f == 0) return DVSOF_EINVAL;
    return f == 4 ? wino_launch_f<4>(P, scratch, ch, st) : wino_launch_f<2>(P, scratch, ch, st);
}

// ---- weight gradient
// Output tile side: the tile count is the K dimension of the GEMMs, so the 4x4
// form (4x fewer tiles, 2.25x more components to write and fold) only pays
// with enough tiles: measured 57 vs 62 us at 128 tiles (batch 8, 16x16, 512
// channels).  DVSOF_WINO_WGRAD_F overrides.  0: not available.
int wino_wgrad_tile(int B, int H, int W, int mfma)
{
    static const int f_env = env_int("DVSOF_WINO_WGRAD_F", 0);
    static const int min_t4 = env_int("DVSOF_WINO_WGRAD_F4_MIN_TILES", 128);
    const bool ok2 = !(H & 1) && !(W & 1) && (B * (H / 2) * (W / 2)) % BK == 0;
    const bool ok4 = mfma == 0 && !(H & 3) && !(W & 3) && (B * (H / 4) * (W / 4)) % BK == 0;
    if (f_env == 2) return ok2 ? 2 : 0;
    if (f_env == 4) return ok4 ? 4 : ok2 ? 2 : 0;
    if (ok4 && B * (H / 4) * (W / 4) >= min_t4) return 4;
    return ok2 ? 2 : ok4 ? 4 : 0;
}

// K splits of the weight-gradient GEMMs (K = tiles): enough workgroups for two
// per CU with 64 x 128 tiles, at least 16 K steps each
static int wino_wgrad_splits(int NG, int T, int N, int C)
{
    static const int s_env = env_int("DVSOF_WINO_WGRAD_SPLITS", 0);
    int S = 1;
    if (s_env > 0) S = s_env;
    else {
        const long long tiles = (long long)NG * ((N + 63) / 64) * ((C + 127) / 128);
        while (tiles * S < 512 && T / (S * 2) >= 16 * BK) S *= 2;
    }
    while (S > 1 && (T + S - 1) / S < BK) --S;
    return S;
}

size_t wino_wgrad_workspace_floats(int B, int H, int W, int C, int N, int mfma)
{
    const int f = wino_wgrad_tile(B, H, W, mfma);
    if (f == 0) return 0;
    const int NG = (f + 2) * (f + 2);
    const size_t T = (size_t)B * (H / f) * (W / f);
    const int S = wino_wgrad_splits(NG, (int)T, N, C);
    return NG * T * ((size_t)C + N) + (size_t)NG * S * N * C + (size_t)NG * S * N;
}

template <int F>
static int wino_wgrad_f(const GSrc &X, const float *V_in, const float *Z_in, const float *gout, float *dW,
                        float *dbias, int B, int H, int W, int C, int N, int mfma_bf16, float *ws, hipStream_t st)
{
    constexpr int NA = Wino<F>::NA, NG = NA * NA;
    constexpr int NT = 256;
    WinoGeom G = {B, H, W, H / F, W / F, B * (H / F) * (W / F)};
    const int S = wino_wgrad_splits(NG, G.T, N, C);
    float *Vws = ws, *Zws = Vws + (size_t)NG * G.T * C, *dU = Zws + (size_t)NG * G.T * N;
    const float *Z = Z_in ? Z_in : Zws;   // Z_in: the producer of gout already transformed it (wino_chain_kernel)
    float *bias_part = dU + (size_t)NG * S * N * C;
    const float *V = V_in ? V_in : Vws;   // V_in: the forward pass already transformed this input

    static const int vw_env = env_int("DVSOF_WINO_VW", 0);
    const bool scalar = vw_env ? vw_env == 1 : (long long)G.T * (C > N ? C : N) / 4 < 256 * 256;
    if (scalar) {
        if (!V_in)
            hipLaunchKernelGGL((wino_input_kernel<F, NT, 1>), dim3(nblocks((long long)G.T * C, NT)), dim3(NT),
                               0, st, X, G, Vws);
        DVSOF_LAUNCH_CHECK();
        if (!Z_in)
            hipLaunchKernelGGL((wino_gout_kernel<F, NT, 1>), dim3(nblocks((long long)G.T * N, NT)), dim3(NT), 0,
                               st, gout, G, N, Zws);
    } else {
        if (!V_in)
            hipLaunchKernelGGL((wino_input_kernel<F, NT, 4>), dim3(nblocks((long long)G.T * (C / 4), NT)),
                               dim3(NT), 0, st, X, G, Vws);
        DVSOF_LAUNCH_CHECK();
        if (!Z_in)
            hipLaunchKernelGGL((wino_gout_kernel<F, NT, 4>), dim3(nblocks((long long)G.T * (N / 4), NT)),
                               dim3(NT), 0, st, gout, G, N, Zws);
    }
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
    Q.nph = NG;
    Q.ph_pad = 0;
    Q.mfma_bf16 = mfma_bf16;
    static const int tile = env_int("DVSOF_WINO_WGRAD_TILE", 3);
    const int bn = (tile == 2 || tile == 3) ? 64 : 128;
    const int nt = (C + bn - 1) / bn;
    Q.tile_begin[0] = 0;
    Q.tile_begin[1] = nt;
    if ((G.T % BK) || (long long)NG * G.T * (C > N ? C : N) * 4 >= 0x7fffffffLL || !wgrad2_eligible(Q))
        return DVSOF_EINVAL;
    const int rc = wgrad2_launch(Q, tile, nt, st);
    if (rc) return rc;

    const int nb_main = (int)nblocks((long long)N * (C / 4), 256), nb_bias = dbias ? (N + 255) / 256 : 0;
    hipLaunchKernelGGL((wino_dw_kernel<F>), dim3((unsigned)(nb_main + nb_bias)), dim3(256), 0, st,
                       (const float *)dU, S, N, C, dW, nb_main, (const float *)bias_part, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int wino_wgrad_launch(const GSrc &X, const float *V_in, const float *Z_in, const float *gout, float *dW,
                      float *dbias, int B, int H, int W, int C, int N, int mfma_bf16, float *ws, size_t ws_floats,
                      hipStream_t st)
{
    if (X.flat || X.sc != 1 || X.C != C) return DVSOF_EINVAL;
    const int f = wino_wgrad_tile(B, H, W, mfma_bf16);
    if (f == 0) return DVSOF_EINVAL;
    if (!ws || ws_floats < wino_wgrad_workspace_floats(B, H, W, C, N, mfma_bf16)) return DVSOF_ENOSPACE;
    return f == 4 ? wino_wgrad_f<4>(X, V_in, Z_in, gout, dW, dbias, B, H, W, C, N, mfma_bf16, ws, st)
                  : wino_wgrad_f<2>(X, V_in, Z_in, gout, dW, dbias, B, H, W, C, N, mfma_bf16, ws, st);
}
