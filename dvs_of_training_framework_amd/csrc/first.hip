// The first encoder layer as kernels of its own: 3x3 / stride 2 / pad 1 over
// the PLANAR voxel grid [B][C][H][W] (C = event bins, 3..16) into 64 NHWC
// channels, forward and weight gradient.  (EV_FlowNet predictor, reference
// call site utils/training.py:59-64; docs/MODEL_SPEC.md enc.0.)
//
// Why: K = 9 C is 27..144 and the input is planar, so the general LDS-DMA
// implicit-GEMM kernels do not apply (16-channel NHWC slices); the layer ran on
// the register-staged v1 kernel (matrix pipe 0.14 busy: 27 us forward in f32,
// 36-50 us with a bf16 twin) and its weight gradient on a kernel that gathers
// im2col values from global memory (39 + 4 us) -- against ~10 us of traffic
// each (10.5 MB of voxels in, 33.5 MB of activations out / gradients in at
// batch 8, 256 x 256 x 5).
//
// Both kernels: a workgroup owns 4 x 32 output pixels of one sample (a row per wave), stages
// the (9 x 65 x C) input patch ONCE in LDS (coalesced row reads of the planar
// grid, zero padding applied there) and feeds v_mfma_f32_32x32x2_f32 -- exact
// f32 in every operand mode of the stack -- with im2col values read from the
// patch through a k -> patch-offset table.
//   forward:  D[pixel][cout] += A[pixel][k] B[k][cout]; a register of the
//             32x32 accumulator is 32 consecutive channels of one pixel (two
//             128-byte segments per store instruction), bias / activation /
//             pre-activation copy / bf16 twin in the epilogue;
//   weight gradient: D[cout][col] += A[cout][pixel] B[pixel][col], A straight
//             from global memory (32 consecutive channels of a pixel per half
//             wave), B = im2col from the patch, one extra column of ones = the
//             bias gradient.  Waves split the tile's pixels, workgroups keep
//             their accumulators over several tiles, partial sums are added by
//             a second kernel in FIXED order: bitwise reproducible.
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr int FT_H = 4, FT_W = 32;                  // output pixels per workgroup: a row per wave
constexpr int RPW = FT_H / 4;                       // rows per wave
constexpr int F_PH = 2 * FT_H + 1, F_PW = 2 * FT_W + 2;   // patch rows / row pitch (65 used)
constexpr int F_N = 64;                             // output channels
constexpr int F_MAXC = 16;

struct FirstP {
    const float *x;            // [B][C][H][W]
    const float *w;            // [64][9][C]
    const float *bias;         // [64] or null
    float *y, *z;              // [B][Ho][Wo][64]; z: optional pre-activation copy
    unsigned short *y16;       // optional bf16 twin of y
    const float *gout;         // weight gradient: [B][Ho][Wo][64]
    float *part;               // weight gradient: [G][64][NCB*32] partial sums
    int B, C, H, W, Ho, Wo, K, KP, act;
    int tiles_x, tiles_y, ntiles;
    int dbg;                   // DVSOF_FIRST_DBG (probe build): 1 no stores, 2 no matrix loop, 4 no patch load, 8 no weight load
};

// LDS image shared by both kernels: patch[C][F_PH][F_PW] + one zero + one 1.0f
__device__ __forceinline__ int patch_floats(int C) { return C * F_PH * F_PW + 2; }

// All loads of a batch first, then the LDS stores: a rolled load -> store loop
// is one dependent memory round trip per element (22 per thread at C = 5: the
// first version of this file spent 25 of its 37 us there).
__device__ __forceinline__ void load_patch(const FirstP &P, float *patch, int b, int ty0, int tx0, int tid,
                                           int nthr = CONV_NT)
{
    const int n = P.C * F_PH * 65;
    const float *xb = P.x + (size_t)b * P.C * P.H * P.W;
    const int gy0 = 2 * ty0 - 1, gx0 = 2 * tx0 - 1;
    constexpr int PBATCH = 24;
    for (int e0 = 0; e0 < n; e0 += PBATCH * nthr) {
        float v[PBATCH];
        int dst[PBATCH];
#pragma unroll
        for (int j = 0; j < PBATCH; ++j) {
            const int e = e0 + j * nthr + tid;
            const int c = e / (F_PH * 65), r = e - c * (F_PH * 65);
            const int py = r / 65, px = r - py * 65;
            const int gy = gy0 + py, gx = gx0 + px;
            const bool ok = (e < n) & ((unsigned)gy < (unsigned)P.H) & ((unsigned)gx < (unsigned)P.W);
            dst[j] = e < n ? (c * F_PH + py) * F_PW + px : -1;
            v[j] = ok ? xb[((size_t)c * P.H + gy) * P.W + gx] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PBATCH; ++j)
            if (dst[j] >= 0) patch[dst[j]] = v[j];
    }
    if (tid == 0) {
        patch[P.C * F_PH * F_PW] = 0.f;
        patch[P.C * F_PH * F_PW + 1] = 1.f;
    }
}

// weight row order [tap][c] (the physical [Cout][kh][kw][Cin] layout): k -> patch offset
__device__ __forceinline__ int k_offset(const FirstP &P, int k)
{
    if (k >= P.K) return P.C * F_PH * F_PW;       // padding column: reads the zero
    const int tap = k / P.C, c = k - tap * P.C;
    const int ky = tap / 3, kx = tap - 3 * ky;
    return (c * F_PH + ky) * F_PW + kx;
}

__global__ __launch_bounds__(CONV_NT) void first_fwd_kernel(const FirstP P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ float smem[];
    float *patch = smem;
    float *wl = patch + patch_floats(P.C);         // [KP][64]
    int *koff = (int *)(wl + P.KP * F_N);          // [KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int b = bid / (P.tiles_x * P.tiles_y), t = bid - b * (P.tiles_x * P.tiles_y);
    const int ty0 = (t / P.tiles_x) * FT_H, tx0 = (t % P.tiles_x) * FT_W;

    // Two independent fills, each a round trip to memory: half the workgroup stages the
    // weights (wl[k][cout] = w[cout][k]), the other half the input patch -- one round trip
    // instead of two in a row (probe build, batch 8: each costs ~5 us of the launch)
    if (tid < CONV_NT / 2) {
        if (!(DVSOF_DBG(P) & 8)) {
            const int nw = F_N * P.KP, nthr = CONV_NT / 2;
            constexpr int WBATCH = 24;
            for (int i0 = 0; i0 < nw; i0 += WBATCH * nthr) {
                float v[WBATCH];
                int dst[WBATCH];
#pragma unroll
                for (int j = 0; j < WBATCH; ++j) {
                    const int i = i0 + j * nthr + tid;
                    const int cout = i / P.KP, k = i - cout * P.KP;
                    dst[j] = i < nw ? k * F_N + cout : -1;
                    v[j] = (i < nw && k < P.K) ? P.w[(size_t)cout * P.K + k] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < WBATCH; ++j)
                    if (dst[j] >= 0) wl[dst[j]] = v[j];
            }
        }
        for (int k = tid; k < P.KP; k += CONV_NT / 2) koff[k] = k_offset(P, k);
    } else if (!(DVSOF_DBG(P) & 4)) {
        load_patch(P, patch, b, ty0, tx0, tid - CONV_NT / 2, CONV_NT / 2);
    }
    __syncthreads();

    const int cx = lane & 31, half = lane >> 5;
    const int base0 = (2 * (RPW * wave)) * F_PW + 2 * cx;
    f32x16 acc[RPW][2];
#pragma unroll
    for (int a = 0; a < RPW; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll 4
    for (int k0 = 0; k0 < ((DVSOF_DBG(P) & 2) ? 2 : P.KP); k0 += 2) {
        const int k = k0 + half;
        const int off = koff[k];
        const float b0 = wl[k * F_N + cx], b1 = wl[k * F_N + 32 + cx];
#pragma unroll
        for (int a = 0; a < RPW; ++a) {
            const float av = patch[base0 + a * 2 * F_PW + off];
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[a][1], 0, 0, 0);
        }
    }
    // epilogue through LDS: a register of the accumulator is 32 channels of one pixel
    // (128-byte pieces; 64-byte ones for the bf16 twin), staged per wave as [pixel][64]
    // and stored as whole pixels -- 16 bytes per lane, 1 KiB contiguous per instruction
    __syncthreads();                        // every wave is done with the patch
    float *stage = smem + (size_t)wave * (RPW * 32 * F_N);
#pragma unroll
    for (int mb = 0; mb < RPW; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int cout = nb * 32 + cx;
            const float bv = P.bias ? P.bias[cout] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = mb * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                stage[px * F_N + cout] = acc[mb][nb][r] + bv;
            }
        }
    // (a wave reads back only what it wrote: no barrier, the LDS queue is in order)
    const int c4 = (lane & 15) * 4, pl = lane >> 4;
#pragma unroll 4
    for (int it = 0; it < RPW * 8; ++it) {
        const int px = 4 * it + pl, mb = px >> 5;
        const int oy = ty0 + RPW * wave + mb, ox = tx0 + (px & 31);
        if (oy >= P.Ho || ox >= P.Wo || ((DVSOF_DBG(P) & 1) && px > 0)) continue;
        const f32x4 v = *(const f32x4 *)(stage + px * F_N + c4);
        const size_t o = (((size_t)b * P.Ho + oy) * P.Wo + ox) * F_N + c4;
        if (P.z) *(f32x4 *)(P.z + o) = v;
        f32x4 yv;
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[e] = act_fwd(v[e], P.act);
        *(f32x4 *)(P.y + o) = yv;
        if (P.y16) {
            typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
            u16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = bf16_bits(yv[e]);
            *(u16x4 *)(P.y16 + o) = h;
        }
    }
#endif
}

// Columns of the weight-gradient GEMM: k = 0 .. K-1 (im2col), K = ones (bias
// gradient), the rest padding.  NCB = blocks of 32 columns.
template <int NCB>
__global__ __launch_bounds__(CONV_NT) void first_wgrad_kernel(const FirstP P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ float smem[];
    float *patch = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cx = lane & 31, half = lane >> 5;
    int coff[NCB];      // patch offset of this lane's column in every column block
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int col = 32 * j + cx;
        coff[j] = col == P.K ? P.C * F_PH * F_PW + 1 : k_offset(P, col);
    }
    f32x16 acc[2][NCB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    for (int t = blockIdx.x; t < P.ntiles; t += gridDim.x) {
        const int b = t / (P.tiles_x * P.tiles_y), tt = t - b * (P.tiles_x * P.tiles_y);
        const int ty0 = (tt / P.tiles_x) * FT_H, tx0 = (tt % P.tiles_x) * FT_W;
        __syncthreads();                    // the previous tile's readers are done
        load_patch(P, patch, b, ty0, tx0, tid);
        __syncthreads();
        // this wave's pixels: tile row(s) RPW wave ..; two pixels per matrix
        // instruction.  A[cout][pixel]: 32 consecutive channels of a pixel per half wave,
        // ALL 64 values of the lane loaded before the first matrix instruction
        float g[RPW][FT_W / 2][2];
#pragma unroll
        for (int mb = 0; mb < RPW; ++mb) {
            const int oy = ty0 + RPW * wave + mb;
            const float *grow = P.gout + (((size_t)b * P.Ho + (oy < P.Ho ? oy : 0)) * P.Wo) * F_N;
#pragma unroll
            for (int s = 0; s < FT_W / 2; ++s) {
                const int ox = tx0 + 2 * s + half;
                const bool ok = (oy < P.Ho) & (ox < P.Wo);
                g[mb][s][0] = ok ? grow[(size_t)ox * F_N + cx] : 0.f;
                g[mb][s][1] = ok ? grow[(size_t)ox * F_N + 32 + cx] : 0.f;
            }
        }
#pragma unroll
        for (int mb = 0; mb < RPW; ++mb) {
            const int ly = RPW * wave + mb;
#pragma unroll
            for (int s = 0; s < FT_W / 2; ++s) {
                const int pbase = (2 * ly) * F_PW + 2 * (2 * s + half);
#pragma unroll
                for (int j = 0; j < NCB; ++j) {
                    const float v = patch[coff[j] + (coff[j] >= P.C * F_PH * F_PW ? 0 : pbase)];
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[mb][s][0], v, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[mb][s][1], v, acc[1][j], 0, 0, 0);
                }
            }
        }
    }
    // waves add up through LDS in fixed order (wave 0 + 1 + 2 + 3), then one store per value
    __syncthreads();
    float *xch = smem;      // [3][2 * NCB * 16][64]
    constexpr int NV = 2 * NCB * 16;
    if (wave > 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xch[((size_t)(wave - 1) * NV + (a * NCB + j) * 16 + r) * 64 + lane] = acc[a][j][r];
    }
    __syncthreads();
    if (wave > 0) return;
    float *out = P.part + (size_t)blockIdx.x * F_N * (NCB * 32);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][j][r];
#pragma unroll
                for (int wv = 0; wv < 3; ++wv) v += xch[((size_t)wv * NV + (a * NCB + j) * 16 + r) * 64 + lane];
                // D[cout][col]: col = lane & 31 of block j, cout = 32 a + 8 (r / 4) + 4 half + r % 4
                const int cout = 32 * a + 8 * (r >> 2) + 4 * half + (r & 3);
                out[(size_t)cout * (NCB * 32) + 32 * j + cx] = v;
            }
#endif
}

// dW[cout][k] = sum over the G partials; column K of them = dbias.  A workgroup
// of 16 waves owns 64 outputs: wave v adds partials [v chunk, (v + 1) chunk) of
// its lane's output in order (four chains: loads in flight), then the 16 wave
// sums are added in wave order -- a fixed order, bitwise reproducible.  (One
// thread per output walking all G partials took 42 us for 8 MB.)
constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(RED_WAVES *kWave) void first_wgrad_reduce_kernel(
    const float *__restrict__ part, int G, int ncol, int K, float *__restrict__ dW,
    float *__restrict__ dbias)
{
    __shared__ float red[RED_WAVES][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int i = blockIdx.x * kWave + lane;
    const bool live = i < F_N * (K + 1);
    const int cout = live ? i / (K + 1) : 0, k = live ? i - cout * (K + 1) : 0;
    const int chunk = (G + RED_WAVES - 1) / RED_WAVES;
    const int g0 = wave * chunk, g1 = min(G, g0 + chunk);
    const float *p = part + (size_t)cout * ncol + k;
    const size_t gs = (size_t)F_N * ncol;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = g0;
    for (; g + 4 <= g1; g += 4) {
        s0 += p[(size_t)(g + 0) * gs];
        s1 += p[(size_t)(g + 1) * gs];
        s2 += p[(size_t)(g + 2) * gs];
        s3 += p[(size_t)(g + 3) * gs];
    }
    for (; g < g1; ++g) s0 += p[(size_t)g * gs];
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave != 0 || !live) return;
    float s = red[0][lane];
#pragma unroll
    for (int v = 1; v < RED_WAVES; ++v) s += red[v][lane];
    if (k < K) dW[(size_t)cout * K + k] = s;
    else if (dbias) dbias[cout] = s;
}

inline int ncb_of(int K) { return (K + 1 + 31) / 32; }

inline int wgrad_groups(int ntiles)
{
    static const int g = getenv("DVSOF_FIRST_WGRAD_GROUPS") ? atoi(getenv("DVSOF_FIRST_WGRAD_GROUPS")) : 512;
    return ntiles < g ? ntiles : g;
}

void fill(FirstP &P, const float *x, int B, int C, int H, int W)
{
    P.x = x;
    P.B = B;
    P.C = C;
    P.H = H;
    P.W = W;
    P.Ho = H / 2;
    P.Wo = W / 2;
    P.K = 9 * C;
    P.KP = (P.K + 1) & ~1;
    P.tiles_x = (P.Wo + FT_W - 1) / FT_W;
    P.tiles_y = (P.Ho + FT_H - 1) / FT_H;
    P.ntiles = B * P.tiles_x * P.tiles_y;
}

}  // namespace

// 3x3 / stride 2 / pad 1, one planar source of <= 16 channels, 64 output
// channels, even frame sides (DVSOF_NO_FIRST_KERNEL=1: the general kernels)
bool first_layer_shape(int nsrc, int planar, int C, int Cout, int H, int W, int ksize, int stride,
                       int pad, int upsample)
{
    static const bool off = getenv("DVSOF_NO_FIRST_KERNEL") != nullptr;
    return !off && nsrc == 1 && planar && C >= 1 && C <= F_MAXC && Cout == F_N && ksize == 3 &&
           stride == 2 && pad == 1 && !upsample && !(H & 1) && !(W & 1) && H >= 2 && W >= 2;
}

int first_fwd_launch(const float *x, int B, int C, int H, int W, const float *w, const float *bias,
                     int act, float *y, float *z, unsigned short *y16, hipStream_t st)
{
    FirstP P = {};
    fill(P, x, B, C, H, W);
    P.w = w;
    P.bias = bias;
    P.y = y;
    P.z = z;
    P.y16 = y16;
    P.act = act;
#ifdef DVSOF_PROBES
    P.dbg = getenv("DVSOF_FIRST_DBG") ? atoi(getenv("DVSOF_FIRST_DBG")) : 0;
#endif
    size_t lds = ((size_t)C * F_PH * F_PW + 2 + (size_t)P.KP * F_N + P.KP) * 4;
    if (lds < (size_t)4 * RPW * 32 * F_N * 4) lds = (size_t)4 * RPW * 32 * F_N * 4;     // the epilogue's staging area
    static size_t lds_set = 0;
    if (lds > 64 * 1024 && lds > lds_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)first_fwd_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        lds_set = 150 * 1024;
    }
    hipLaunchKernelGGL(first_fwd_kernel, dim3(P.ntiles), dim3(CONV_NT), lds, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

size_t first_wgrad_workspace_floats(int B, int C, int H, int W)
{
    FirstP P = {};
    fill(P, nullptr, B, C, H, W);
    return (size_t)wgrad_groups(P.ntiles) * F_N * ncb_of(P.K) * 32;
}

int first_wgrad_launch(const float *x, int B, int C, int H, int W, const float *gout, float *dW,
                       float *dbias, float *ws, size_t ws_floats, hipStream_t st)
{
    FirstP P = {};
    fill(P, x, B, C, H, W);
    if (!ws || ws_floats < first_wgrad_workspace_floats(B, C, H, W)) return DVSOF_ENOSPACE;
    P.gout = gout;
    P.part = ws;
    const int G = wgrad_groups(P.ntiles), ncb = ncb_of(P.K);
    // LDS: the patch, re-used for the cross-wave sums at the end
    size_t lds = ((size_t)C * F_PH * F_PW + 2) * 4;
    const size_t xch = (size_t)3 * 2 * ncb * 16 * 64 * 4;
    if (xch > lds) lds = xch;
#define FIRST_WG(NCB_)                                                                               \
    case NCB_: {                                                                                     \
        static bool set_ = false;                                                                    \
        if (lds > 64 * 1024 && !set_) {                                                              \
            DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)first_wgrad_kernel<NCB_>,                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
            set_ = true;                                                                             \
        }                                                                                            \
        hipLaunchKernelGGL(first_wgrad_kernel<NCB_>, dim3(G), dim3(CONV_NT), lds, st, P);            \
        break;                                                                                       \
    }
    switch (ncb) {
        FIRST_WG(1)
        FIRST_WG(2)
        FIRST_WG(3)
        FIRST_WG(4)
        FIRST_WG(5)
    default: return DVSOF_EINVAL;
    }
#undef FIRST_WG
    DVSOF_LAUNCH_CHECK();
    const int nout = F_N * (P.K + 1);
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((nout + kWave - 1) / kWave), dim3(RED_WAVES * kWave), 0, st,
                       (const float *)ws, G, ncb * 32, P.K, dW, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
