// Gather-convolution as an implicit GEMM on the f32 matrix cores.
//
//   out[m][n] = sum_k A[m][k] * W[n][k]
//   m = output pixel (b,oy,ox)      n = output channel
//   k = (tap, input channel) over a VIRTUAL input: the channel concatenation
//       of up to three tensors, optionally 2x nearest-upsampled or 2x
//       zero-inserted, never materialised in memory.
//
// One kernel serves (see conv.py for the wiring):
//   * forward of every EV_FlowNet predictor layer (strided encoder convs,
//     residual convs, decoder convs on upsampled concat[x, skip, flow]) with a
//     fused bias + residual + ReLU/Mish epilogue;
//   * data-gradient of all of them, as a convolution of the output gradient
//     with tap-flipped, transposed weights: stride-2 layers read a
//     zero-inserted virtual input, upsampled decoder layers use quad-major
//     rows so that the 2x2 sum back to the low-resolution tensor happens in
//     registers (the 4 quad members are the 4 consecutive accumulator
//     registers of one lane), and the epilogue scatters channel ranges to
//     up to three destinations with optional accumulate and act' multiply.
// Replaces ATen conv2d / conv_transpose / cat / upsample / activation and their
// autograd inside the (absent) EV_FlowNet predictor called at
// utils/training.py:59-64 and differentiated at utils/training.py:158.
//
// MFMA-bound (v_mfma_f32_32x32x2_f32: 64 cycles per 32x32x2 block per SIMD).
// Workgroup = 4 waves; wave tile = (TM*32) x (TN*32); K staged through LDS in
// 16-wide slices [row][k] (row stride 20 floats: conflict-free ds_read_b128),
// double buffered with register prefetch, one barrier per slice.  Lane half
// h = lane>>5 reads k = 8j + 4h .. +3 as one b128 and feeds 4 consecutive
// MFMAs; A and B use the same k permutation, so the sum is unchanged.
#include "conv_common.h"
#include <stdlib.h>

namespace {

struct KIter {
    int s, tap, c0, coff;
};

__device__ __forceinline__ void kiter_advance(const GConvParams &P, KIter &it, int taps)
{
    const int C = P.src[it.s].C;
    it.c0 += BK;
    bool next_src;
    if (P.src[it.s].flat) {
        next_src = it.c0 >= taps * C;
    } else {
        next_src = false;
        if (it.c0 >= C) {
            it.c0 = 0;
            it.tap += 1;
            next_src = it.tap >= taps;
        }
    }
    if (next_src) {
        it.coff += C;
        it.s += 1;
        it.tap = 0;
        it.c0 = 0;
    }
}

template <int WROWS, int WCOLS, int TM, int TN>
__global__ __launch_bounds__(CONV_NT) void gconv_kernel(const GConvParams P, const int nsteps)
{
    constexpr int BM = WROWS * TM * 32, BN = WCOLS * TN * 32;
    constexpr int RA = BM / 16 < 4 ? 4 : BM / 16;
    constexpr int RB = BN / 16 < 4 ? 4 : BN / 16;
    static_assert(WROWS * WCOLS == CONV_NT / kWave, "4 waves");

    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];
    __shared__ int rowB[BM], rowY[BM], rowX[BM];
    __shared__ long long rowO[3 * BM];   // output offsets (conv_epilogue)
    __shared__ int rowC[BM];             // border class of the output row (bias_cls)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int taps = P.ks * P.ks;
    const int ph = blockIdx.z, phy = ph >> 1, phx = ph & 1;
    const int pad_y = P.pad - phy * P.ph_pad, pad_x = P.pad - phx * P.ph_pad;
    const float *Wp = P.W + (size_t)ph * P.w_phase_stride;

    for (int r = tid; r < BM; r += CONV_NT) {
        const int m = m0 + r;
        int b = 0, y = -(1 << 20), x = -(1 << 20), oy = 0, ox = 0;
        if (m < P.M) {
            if (!P.quad) {
                ox = m % P.Wo;
                const int t = m / P.Wo;
                oy = t % P.Ho;
                b = t / P.Ho;
            } else {
                const int j = m & 3, q = m >> 2, wq = P.Wo >> 1, hq = P.Ho >> 1;
                const int t = q / wq;
                ox = 2 * (q - t * wq) + (j & 1);
                oy = 2 * (t % hq) + (j >> 1);
                b = t / hq;
            }
            y = oy * P.stride - pad_y;
            x = ox * P.stride - pad_x;
        }
        rowB[r] = b;
        rowY[r] = y;
        rowX[r] = x;
        conv_row_offsets(P, rowO, rowC, BM, r, m < P.M, b, oy, ox, phy, phx);
    }
    __syncthreads();

    float ra[RA], rb[RB];
    bool ldflat = false;

    auto load_tiles = [&](const KIter &it) {
        const GSrc &S = P.src[it.s];
        const size_t wrow = (size_t)taps * P.Cin_tot;
        ldflat = S.flat != 0;
        if (!ldflat) {
            const int ky = it.tap / P.ks, kx = it.tap - ky * P.ks;
            const int c = it.c0 + 4 * (tid & 3);
            const bool cok = c < S.C;
#pragma unroll
            for (int i = 0; i < (BM + 63) / 64; ++i) {
                const int r = (tid >> 2) + 64 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (BM >= 64 || r < BM) {
                    const int Y = rowY[r] + ky, X = rowX[r] + kx;
                    bool ok = cok & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
                    if (P.up == UP_ZERO) ok &= ((Y | X) & 1) == 0;
                    if (ok) {
                        const int ys = P.up ? Y >> 1 : Y, xs = P.up ? X >> 1 : X;
                        v = *(const f32x4u *)(S.p + (size_t)rowB[r] * S.sb + (size_t)ys * S.sy +
                                              (size_t)xs * S.sx + c);
                    }
                }
                ra[4 * i + 0] = v[0]; ra[4 * i + 1] = v[1];
                ra[4 * i + 2] = v[2]; ra[4 * i + 3] = v[3];
            }
#pragma unroll
            for (int i = 0; i < (BN + 63) / 64; ++i) {
                const int r = (tid >> 2) + 64 * i;
                const int n = n0 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((BN >= 64 || r < BN) && cok && n < P.N)
                    v = *(const f32x4u *)(Wp + (size_t)n * wrow + (size_t)it.tap * P.Cin_tot +
                                          it.coff + c);
                rb[4 * i + 0] = v[0]; rb[4 * i + 1] = v[1];
                rb[4 * i + 2] = v[2]; rb[4 * i + 3] = v[3];
            }
        } else {
            const int f = it.c0 + (tid & 15);
            const bool fok = f < taps * S.C;
            const int tap = fok ? f / S.C : 0, c = f - tap * S.C;
            const int ky = tap / P.ks, kx = tap - ky * P.ks;
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) {
                const int r = (tid >> 4) + 16 * i;
                const int Y = rowY[r] + ky, X = rowX[r] + kx;
                bool ok = fok & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
                if (P.up == UP_ZERO) ok &= ((Y | X) & 1) == 0;
                float v = 0.f;
                if (ok) {
                    const int ys = P.up ? Y >> 1 : Y, xs = P.up ? X >> 1 : X;
                    v = S.p[(size_t)rowB[r] * S.sb + (size_t)ys * S.sy + (size_t)xs * S.sx +
                            (size_t)c * S.sc];
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const int n = n0 + (tid >> 4) + 16 * i;
                rb[i] = (fok && n < P.N)
                            ? Wp[(size_t)n * wrow + (size_t)tap * P.Cin_tot + it.coff + c]
                            : 0.f;
            }
        }
    };

    auto store_tiles = [&](int buf) {
        if (!ldflat) {
            const int kq = 4 * (tid & 3);
#pragma unroll
            for (int i = 0; i < (BM + 63) / 64; ++i) {
                const int r = (tid >> 2) + 64 * i;
                if (BM >= 64 || r < BM)
                    *(f32x4 *)&As[buf][r][kq] =
                        f32x4{ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]};
            }
#pragma unroll
            for (int i = 0; i < (BN + 63) / 64; ++i) {
                const int r = (tid >> 2) + 64 * i;
                if (BN >= 64 || r < BN)
                    *(f32x4 *)&Bs[buf][r][kq] =
                        f32x4{rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]};
            }
        } else {
            const int k = tid & 15;
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) As[buf][(tid >> 4) + 16 * i][k] = ra[i];
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) Bs[buf][(tid >> 4) + 16 * i][k] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    KIter it = {0, 0, 0, 0};
    load_tiles(it);
    store_tiles(0);
    __syncthreads();

    const int lrow = lane & 31, lk = 4 * (lane >> 5);
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool has_next = step + 1 < nsteps;
        if (has_next) {
            kiter_advance(P, it, taps);
            load_tiles(it);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t] = *(const f32x4 *)&As[cur][(wr * TM + t) * 32 + lrow][8 * j + lk];
#pragma unroll
            for (int t = 0; t < TN; ++t)
                b[t] = *(const f32x4 *)&Bs[cur][(wc * TN + t) * 32 + lrow][8 * j + lk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i],
                                                                           acc[tm][tn], 0, 0, 0);
        }
        if (has_next) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue (conv_common.h)
    conv_epilogue<TM, TN>(P, acc, rowO, rowC, BM, n0, wr, wc, lane);
}

int count_steps(const GConvParams &P)
{
    const int taps = P.ks * P.ks;
    int n = 0;
    for (int s = 0; s < P.nsrc; ++s)
        n += P.src[s].flat ? (taps * P.src[s].C + BK - 1) / BK : taps * ((P.src[s].C + BK - 1) / BK);
    return n;
}

template <int WROWS, int WCOLS, int TM, int TN>
int launch(const GConvParams &P, hipStream_t st)
{
    constexpr int BM = WROWS * TM * 32, BN = WCOLS * TN * 32;
    dim3 grid((P.M + BM - 1) / BM, (P.N + BN - 1) / BN, P.nph);
    hipLaunchKernelGGL((gconv_kernel<WROWS, WCOLS, TM, TN>), grid, dim3(CONV_NT), 0, st, P,
                       count_steps(P));
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // namespace

// MFMA tile of the forward / data-gradient kernels: 1 = 128x128, 2 = 128x64,
// 3 = 64x64, 4 = 256x32, 5 = 128x32 (rows x output channels).
// Chosen by measurement, not by a cost model: with every tile forced in turn
// (DVSOF_GCONV_TILE, tools/tile_sweep.sh) over the EV-FlowNet layers at batch
// 1, 8 and 32, 64x64 is the fastest or within 1 % of it on every layer with
// more than 32 output channels -- the larger tiles reuse operands better but
// run fewer workgroups per CU, and the barrier-synchronised K loop wants the
// extra workgroups to cover its stalls -- and 128x32 wins for the 32-channel
// decoder stage (+7 % over 256x32; 64x64 would pad N to 64).  At batch 1 the
// 64x64 tile is 2.5-3.5x faster than 128x128 on the 16x16 and 32x32 stages.
int gconv_pick_tile(long long m, long long n)
{
    (void)m;
    return n <= 32 ? 5 : 3;
}

// ---- trailing "flat" output channel ranges (the 2-channel flow gradient of a
// decoder data-gradient) on the VALU: out[m][n] = sum_k A[m][k] * W[n][k] for
// the few rows n >= n_begin of W.  Keeps them out of the MFMA problem, whose N
// is then 128/256/512 instead of 130/258/514 (one fewer column tile, and
// 128-wide tiles with no padding).
// One lane per output pixel (64 consecutive pixels per workgroup); the four
// waves split the taps, so a lane walks the channels of ONE input pixel per tap
// (its cache line is fetched once and read 16 B at a time); the weight rows are
// wave-uniform (scalar loads); partial sums meet in LDS in a fixed order.
namespace {
constexpr int FLATN_MAX = 4;

template <int NW>   // waves per workgroup: 16 for small M (more taps in flight per pixel)
__global__ __launch_bounds__(64 * NW) void gconv_flat_rows_kernel(const GConvParams P, const int n_begin,
                                                                  const int nrows)
{
    __shared__ float part[NW][FLATN_MAX][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const GSrc &S = P.src[0];
    const int taps = P.ks * P.ks;
    const size_t wrow = (size_t)taps * P.Cin_tot;
    const long long m = (long long)blockIdx.x * 64 + lane;
    const bool pix = m < P.M;
    const int ox = (int)(m % P.Wo);
    const long long t = m / P.Wo;
    const int oy = (int)(t % P.Ho), b = (int)(t / P.Ho);
    float acc[FLATN_MAX] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = wave; tap < taps; tap += NW) {
        const int ky = tap / P.ks, kx = tap - ky * P.ks;
        const int Y = oy * P.stride - P.pad + ky, X = ox * P.stride - P.pad + kx;
        const bool ok = pix & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
        const float *ap = S.p + (ok ? (size_t)b * S.sb + (size_t)Y * S.sy + (size_t)X * S.sx : 0);
        const float *wp = P.W + (size_t)n_begin * wrow + (size_t)tap * P.Cin_tot;
#pragma unroll 4
        for (int c = 0; c < S.C; c += 4) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (ok) a = *(const f32x4u *)(ap + c);
#pragma unroll
            for (int r = 0; r < FLATN_MAX; ++r)
                if (r < nrows) {
                    const f32x4 w = *(const f32x4u *)(wp + (size_t)r * wrow + c);
                    acc[r] = fmaf(a[0], w[0], acc[r]);
                    acc[r] = fmaf(a[1], w[1], acc[r]);
                    acc[r] = fmaf(a[2], w[2], acc[r]);
                    acc[r] = fmaf(a[3], w[3], acc[r]);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < FLATN_MAX; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave != 0 || !pix) return;
    int d = 0, off = 0;
    for (int dd = 0; dd + 1 < P.ndst; ++dd)
        if (n_begin >= off + P.dst[dd].C && d == dd) {
            off += P.dst[dd].C;
            d = dd + 1;
        }
    for (int r = 0; r < nrows; ++r) {
        const int n = n_begin + r;
        while (n >= off + P.dst[d].C) {
            off += P.dst[d].C;
            ++d;
        }
        const GDst &D = P.dst[d];
        const size_t o = (size_t)b * D.sb + (size_t)oy * D.sy + (size_t)ox * D.sx +
                         (size_t)(n - off) * D.sc;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            v += (part[w][r][lane] + part[w + 1][r][lane]) + (part[w + 2][r][lane] + part[w + 3][r][lane]);
        if (D.addend) v += D.addend[o];
        if (D.addend2) v += D.addend2[o];
        if (D.actsrc) v *= act_bwd(D.actsrc[o], P.bwd_act);
        D.p[o] = v;
    }
}

// Large-M form: four lanes share one output pixel and read 64 contiguous bytes
// of an input pixel per instruction (16 cache lines per wave-load instead of
// 64); the weight rows sit in LDS; a wave owns 16 pixels and all taps, the four
// lanes of a pixel meet through two shuffles (fixed order).
constexpr int FLATQ_MAX_K = 4096;   // taps * C floats per weight row in LDS

__global__ __launch_bounds__(256) void gconv_flat_rows_quad_kernel(const GConvParams P, const int n_begin,
                                                                   const int nrows)
{
    extern __shared__ __attribute__((aligned(16))) float wlds[];   // [nrows][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GSrc &S = P.src[0];
    const int taps = P.ks * P.ks, K = taps * S.C;
    const size_t wrow = (size_t)taps * P.Cin_tot;
    for (int i = threadIdx.x * 4; i < nrows * K; i += 256 * 4) {
        const int r = i / K, k = i - r * K, tap = k / S.C, c = k - tap * S.C;
        *(f32x4 *)(wlds + i) =
            *(const f32x4u *)(P.W + (size_t)(n_begin + r) * wrow + (size_t)tap * P.Cin_tot + c);
    }
    __syncthreads();
    const int g = lane & 3;
    const long long m = (long long)blockIdx.x * 64 + wave * 16 + (lane >> 2);
    const bool pix = m < P.M;
    const int ox = (int)(m % P.Wo);
    const long long t = m / P.Wo;
    const int oy = (int)(t % P.Ho), b = (int)(t / P.Ho);
    float acc[FLATN_MAX] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / P.ks, kx = tap - ky * P.ks;
        const int Y = oy * P.stride - P.pad + ky, X = ox * P.stride - P.pad + kx;
        const bool ok = pix & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
        const float *ap = S.p + (ok ? (size_t)b * S.sb + (size_t)Y * S.sy + (size_t)X * S.sx : 0);
        const float *wp = wlds + tap * S.C;
#pragma unroll 2
        for (int c = 4 * g; c < S.C; c += 16) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (ok) a = *(const f32x4u *)(ap + c);
#pragma unroll
            for (int r = 0; r < FLATN_MAX; ++r)
                if (r < nrows) {
                    const f32x4 w = *(const f32x4 *)(wp + r * K + c);
                    acc[r] = fmaf(a[0], w[0], acc[r]);
                    acc[r] = fmaf(a[1], w[1], acc[r]);
                    acc[r] = fmaf(a[2], w[2], acc[r]);
                    acc[r] = fmaf(a[3], w[3], acc[r]);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < FLATN_MAX; ++r) {
        acc[r] += __shfl_xor(acc[r], 1);
        acc[r] += __shfl_xor(acc[r], 2);
    }
    if (g != 0 || !pix) return;
    int d = 0, off = 0;
    for (int dd = 0; dd + 1 < P.ndst; ++dd)
        if (n_begin >= off + P.dst[dd].C && d == dd) {
            off += P.dst[dd].C;
            d = dd + 1;
        }
    for (int r = 0; r < nrows; ++r) {
        const int n = n_begin + r;
        while (n >= off + P.dst[d].C) {
            off += P.dst[d].C;
            ++d;
        }
        const GDst &D = P.dst[d];
        const size_t o = (size_t)b * D.sb + (size_t)oy * D.sy + (size_t)ox * D.sx +
                         (size_t)(n - off) * D.sc;
        float v = acc[r];
        if (D.addend) v += D.addend[o];
        if (D.addend2) v += D.addend2[o];
        if (D.actsrc) v *= act_bwd(D.actsrc[o], P.bwd_act);
        D.p[o] = v;
    }
}
}  // namespace

bool gconv2_eligible(const GConvParams &P, long long max_src_bytes, long long w_bytes);
int gconv2_launch(const GConvParams &P, int tile, hipStream_t st);

// Internal entry (not part of the C ABI): picks the tile shape and launches.
int gconv_launch(const GConvParams &P, int tile_hint, hipStream_t st)
{
    if (P.M <= 0 || P.N <= 0 || P.nsrc < 1 || P.nsrc > 3 || P.ndst < 1 || P.ndst > 3)
        return DVSOF_EINVAL;
    if (P.quad && ((P.Ho | P.Wo) & 1)) return DVSOF_EINVAL;
    if (P.stride != 1 && P.stride != 2) return DVSOF_EINVAL;
    if (P.nph != 1 && P.nph != 4) return DVSOF_EINVAL;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && (P.src[s].sc != 1 || (P.src[s].C & 3))) return DVSOF_EINVAL;
    // data-gradient form with trailing narrow planar destinations: peel them off
    {
        static const bool no_split = getenv("DVSOF_GCONV_NO_SPLIT") != nullptr;
        int ntrail = 0, d = P.ndst;
        while (d > 1 && (P.dst[d - 1].sc != 1 || P.dst[d - 1].C < BK) && ntrail + P.dst[d - 1].C <= FLATN_MAX) {
            ntrail += P.dst[d - 1].C;
            --d;
        }
        if (!no_split && ntrail > 0 && P.nsrc == 1 && !P.src[0].flat && P.src[0].sc == 1 &&
            (P.src[0].C & 3) == 0 && P.up == UP_NONE && !P.quad && P.nph == 1 && !P.bias &&
            !P.zout && P.act == ACT_NONE && P.N - ntrail >= 32) {
            GConvParams Q = P;
            Q.N = P.N - ntrail;
            Q.ndst = d;
            const int rc = gconv_launch(Q, tile_hint, st);
            if (rc) return rc;
            const long long nb = ((long long)P.M + 63) / 64;
            const int Kq = P.ks * P.ks * P.src[0].C;
            if (nb >= 512 && Kq <= FLATQ_MAX_K && (P.src[0].C & 3) == 0 && ((Kq * ntrail) & 3) == 0)
                hipLaunchKernelGGL(gconv_flat_rows_quad_kernel, dim3((unsigned)nb), dim3(256),
                                   (size_t)ntrail * Kq * sizeof(float), st, P, Q.N, ntrail);
            else if (nb < 512)
                hipLaunchKernelGGL(gconv_flat_rows_kernel<16>, dim3((unsigned)nb), dim3(1024), 0, st, P,
                                   Q.N, ntrail);
            else
                hipLaunchKernelGGL(gconv_flat_rows_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, P,
                                   Q.N, ntrail);
            DVSOF_LAUNCH_CHECK();
            return DVSOF_OK;
        }
    }
    static const int tile_env = getenv("DVSOF_GCONV_TILE") ? atoi(getenv("DVSOF_GCONV_TILE")) : 0;   // tuning sweeps
    const int tile = tile_hint > 0 ? tile_hint : tile_env > 0 ? tile_env
                                                              : gconv_pick_tile((long long)P.M * P.nph, P.N);
    {   // v2 (LDS-DMA ring, VALU-free main loop) when the shape allows it
        static const bool force_v1 = getenv("DVSOF_GCONV_V1") != nullptr;
        long long src_bytes = 0;
        for (int s = 0; s < P.nsrc; ++s) {
            const long long b = (long long)P.B * P.src[s].sb * 4;
            src_bytes = b > src_bytes ? b : src_bytes;
        }
        const long long w_bytes = (long long)P.N * P.ks * P.ks * P.Cin_tot * 4 * P.nph;
        if (!force_v1 && gconv2_eligible(P, src_bytes, w_bytes)) return gconv2_launch(P, tile, st);
    }
    switch (tile) {
    case 1: return launch<2, 2, 2, 2>(P, st);  // 128 x 128
    case 2: return launch<2, 2, 2, 1>(P, st);  // 128 x 64
    case 3: return launch<2, 2, 1, 1>(P, st);  // 64 x 64
    case 4: return launch<4, 1, 2, 1>(P, st);  // 256 x 32
    case 5: return launch<4, 1, 1, 1>(P, st);  // 128 x 32
    default: return DVSOF_EINVAL;
    }
}
