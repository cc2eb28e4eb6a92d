// Multi-scale flow-warp photometric + Charbonnier smoothness + out-of-border
// loss, forward and analytic backward, all scales in one launch.
//
// Replaces (reference paths): utils/loss.py:20-21 (bilinear resize),
// :24-35 (Charbonnier), :58-74 (warp + photometric), :76-90 (smoothness),
// :92-119 (out-of-border), :121-171 (Loss.__call__), and the autograd
// backward that utils/training.py:158 runs through them.
//
// HBM-bound kernels.  Layout: flow [N,2,h,w] and frames [D,h,w] row-major;
// a 256-thread workgroup owns a 64x16 pixel tile of one sample at one scale,
// stages the flow tile + 1 px halo in LDS (the 3x3 smoothness stencil reads
// it 9x), gathers the 4 bilinear taps of the second frame straight from
// L2/HBM, reduces its sums with wave shuffles and writes ONE partial record;
// a single-workgroup finalize kernel adds the records in a fixed order
// (bitwise reproducible, no float atomics).
#include "common.h"

namespace {

constexpr int TW = 64, TH = 16, NT = 256;
constexpr int LW = TW + 2, LH = TH + 2;
constexpr int NPART = 8;  // photo, smooth x4, border sum, border count, pad

struct ScaleDev {
    const float *frames;
    const float *flow;
    float *grad;
    int h, w;
    int tiles_x, tiles_per_sample, block_begin;
    float half_w, half_h;        // (w-1)/2, (h-1)/2 (utils/loss.py:152-154)
    float k_smooth[3];           // 1/(4*count) for ->, v, diagonal crops
    float k_photo;               // 1/(N*h*w)
    double c_smooth[3];          // crop element counts (utils/loss.py:77-85)
};

struct Params {
    ScaleDev s[DVSOF_MAX_SCALES];
    int K, N;
    const int32_t *start, *stop;
    float *partials;
    int32_t *oob;               // [K*N]
    const float *seeds_dev;     // [3*K] or null
    float seeds_host[3];        // used when seeds_dev == null
};

__device__ __forceinline__ int find_scale(const Params &P, int bid)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < DVSOF_MAX_SCALES; ++i)
        if (i < P.K && bid >= P.s[i].block_begin) k = i;
    return k;
}

// utils/loss.py:150-156 in fp32, op for op (IEEE division).
__device__ __forceinline__ void warp_grid(const ScaleDev &S, int x, int y, float u,
                                          float v, float &gx, float &gy)
{
    gx = ((float)x + u) / S.half_w - 1.f;
    gy = ((float)y + v) / S.half_h - 1.f;
}
__device__ __forceinline__ bool out_of_border(float gx, float gy)
{  // utils/loss.py:92-94 (strict)
    return (gx < -1.f) | (gx > 1.f) | (gy < -1.f) | (gy > 1.f);
}

template <bool FWD, bool BWD>
__global__ __launch_bounds__(NT) void loss_main_kernel(const Params P)
{
    __shared__ float sF[2][LH][LW];
    __shared__ float red[NT / kWave][NPART];

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int k = find_scale(P, bid);
    const ScaleDev &S = P.s[k];
    const int local = bid - S.block_begin;
    const int n = local / S.tiles_per_sample;
    const int t = local - n * S.tiles_per_sample;
    const int ty0 = (t / S.tiles_x) * TH, tx0 = (t % S.tiles_x) * TW;
    const int h = S.h, w = S.w;
    const size_t hw = (size_t)h * w;
    const float *U = S.flow + (size_t)n * 2 * hw;

    for (int i = tid; i < LH * LW; i += NT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
        const bool in = (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
        const size_t o = (size_t)gy * w + gx;
        sF[0][ly][lx] = in ? U[o] : 0.f;
        sF[1][ly][lx] = in ? U[hw + o] : 0.f;
    }
    const float *I0 = S.frames + (size_t)P.start[n] * hw;
    const float *I1 = S.frames + (size_t)P.stop[n] * hw;
    float seed[3] = {0.f, 0.f, 0.f};
    float k_border = 0.f;
    if (BWD) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            seed[i] = P.seeds_dev ? P.seeds_dev[i * P.K + k] : P.seeds_host[i];
        const int cnt = P.oob[k * P.N + n];
        k_border = cnt > 0 ? seed[2] / (2.f * (float)cnt * (float)P.N) : 0.f;
    }
    __syncthreads();

    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tx = tid & (TW - 1), tr = tid >> 6;
    const int x = tx0 + tx;
    constexpr int NP = TH / 4;   // pixels per thread (rows tr, tr+4, ...)
    // phase A: addresses and ALL gathers of the thread's pixels first, so the
    // 5*NP loads are in flight together instead of NP dependent round trips
    bool valid[NP], oobv[NP];
    float uu[NP], vv[NP], axv[NP], ayv[NP], nwv[NP], nev[NP], swv[NP], sev[NP], prv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int ly = tr + 4 * j, y = ty0 + ly;
        valid[j] = (y < h) & (x < w);
        const float u = sF[0][ly + 1][tx + 1], v = sF[1][ly + 1][tx + 1];
        uu[j] = u;
        vv[j] = v;
        float gx, gy;
        warp_grid(S, x, y, u, v, gx, gy);
        oobv[j] = out_of_border(gx, gy);
        // grid_sample(bilinear, zeros, align_corners=True): utils/loss.py:70
        const float ix = (gx + 1.f) * S.half_w, iy = (gy + 1.f) * S.half_h;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        axv[j] = ix - fx0;
        ayv[j] = iy - fy0;
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
        const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x0 + 1 >= 0) & (x0 + 1 < w);
        const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y0 + 1 >= 0) & (y0 + 1 < h);
        const float *r0 = I1 + (ptrdiff_t)y0 * w + x0;
        nwv[j] = (valid[j] & vy0 & vx0) ? r0[0] : 0.f;
        nev[j] = (valid[j] & vy0 & vx1) ? r0[1] : 0.f;
        swv[j] = (valid[j] & vy1 & vx0) ? r0[w] : 0.f;
        sev[j] = (valid[j] & vy1 & vx1) ? r0[w + 1] : 0.f;
        prv[j] = valid[j] ? I0[(size_t)y * w + x] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int ly = tr + 4 * j, y = ty0 + ly;
        if (!valid[j]) continue;
        const float u = uu[j], v = vv[j];
        const bool oob = oobv[j];
        const float ax = axv[j], ay = ayv[j], cx = 1.f - ax, cy = 1.f - ay;
        const float nw = nwv[j], ne = nev[j], sw = swv[j], se = sev[j];
        const float warped = nw * cx * cy + ne * ax * cy + sw * cx * ay + se * ax * ay;
        const Charb ph = charbonnier(warped - prv[j]);

        float gu = 0.f, gv = 0.f;
        if (FWD) acc[0] += ph.val;
        if (BWD) {
            const float gp = seed[1] * S.k_photo * ph.der;
            gu = gp * ((ne - nw) * cy + (se - sw) * ay);
            gv = gp * ((sw - nw) * cx + (se - ne) * ax);
        }
        if (oob) {  // utils/loss.py:96-119
            const Charb bu = charbonnier(u), bv = charbonnier(v);
            if (FWD) {
                acc[5] += bu.val + bv.val;
                acc[6] += 1.f;
            }
            if (BWD) {
                gu += k_border * bu.der;
                gv += k_border * bv.der;
            }
        }
        // smoothness, utils/loss.py:76-90: pairs (->, v, diag \, diag /)
        const bool xr = x + 1 < w, xl = x >= 1, yd = y + 1 < h, yu = y >= 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float(*F)[LW] = sF[c];
            const int a = ly + 1, b = tx + 1;
            const float ctr = F[a][b];
            float g = 0.f;
            if (xr) {
                const Charb q = charbonnier(F[a][b + 1] - ctr);
                if (FWD) acc[1] += q.val;
                if (BWD) g -= S.k_smooth[0] * q.der;
            }
            if (yd) {
                const Charb q = charbonnier(F[a + 1][b] - ctr);
                if (FWD) acc[2] += q.val;
                if (BWD) g -= S.k_smooth[1] * q.der;
            }
            if (xr & yd) {
                const Charb q = charbonnier(F[a + 1][b + 1] - ctr);
                if (FWD) acc[3] += q.val;
                if (BWD) g -= S.k_smooth[2] * q.der;
                if (FWD) acc[4] += charb_val(F[a][b + 1] - F[a + 1][b]);
            }
            if (BWD) {
                if (xl) g += S.k_smooth[0] * charbonnier(ctr - F[a][b - 1]).der;
                if (yu) g += S.k_smooth[1] * charbonnier(ctr - F[a - 1][b]).der;
                if (xl & yu) g += S.k_smooth[2] * charbonnier(ctr - F[a - 1][b - 1]).der;
                if (xl & yd) g += S.k_smooth[2] * charbonnier(ctr - F[a + 1][b - 1]).der;
                if (yu & xr) g -= S.k_smooth[2] * charbonnier(F[a - 1][b + 1] - ctr).der;
                g *= seed[0];
                if (c == 0) gu += g; else gv += g;
            }
        }
        if (BWD) {
            float *G = S.grad + (size_t)n * 2 * hw + (size_t)y * w + x;
            G[0] = gu;
            G[hw] = gv;
        }
    }

    if (FWD) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float s = wave_sum(acc[i]);
            if ((tid & (kWave - 1)) == 0) red[tid >> 6][i] = s;
        }
        __syncthreads();
        if (tid < NPART) {
            float s = 0.f;
            if (tid < 7)
                for (int wv = 0; wv < NT / kWave; ++wv) s += red[wv][tid];
            P.partials[(size_t)bid * NPART + tid] = s;
        }
    }
}

// Per-sample out-of-border pixel counts (utils/loss.py:101) ahead of the
// fused forward+backward sweep.  Integer atomics: order-independent.
__global__ __launch_bounds__(NT) void loss_count_oob_kernel(const Params P)
{
    __shared__ int red[NT / kWave];
    const int tid = threadIdx.x, bid = blockIdx.x;
    const int k = find_scale(P, bid);
    const ScaleDev &S = P.s[k];
    const int local = bid - S.block_begin;
    const int n = local / S.tiles_per_sample;
    const int t = local - n * S.tiles_per_sample;
    const int ty0 = (t / S.tiles_x) * TH, tx0 = (t % S.tiles_x) * TW;
    const int h = S.h, w = S.w;
    const size_t hw = (size_t)h * w;
    const float *U = S.flow + (size_t)n * 2 * hw;
    const int x = tx0 + (tid & (TW - 1)), tr = tid >> 6;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < TH / 4; ++j) {
        const int y = ty0 + tr + 4 * j;
        if (y < h && x < w) {
            const size_t o = (size_t)y * w + x;
            float gx, gy;
            warp_grid(S, x, y, U[o], U[hw + o], gx, gy);
            cnt += out_of_border(gx, gy) ? 1 : 0;
        }
    }
    cnt = wave_sum(cnt);
    if ((tid & (kWave - 1)) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        const int s = red[0] + red[1] + red[2] + red[3];
        if (s) atomicAdd(&P.oob[k * P.N + n], s);
    }
}

// One workgroup adds the per-tile records in a fixed order (double) and applies
// the reference's normalisers.  terms[t*K + k], t = smooth/photo/border.
// Work items (scale k, sample n) and (scale k, sum i) are dealt to the 4 waves;
// each is a lane-strided sum + shuffle tree, so there are only two barriers.
__global__ __launch_bounds__(NT) void loss_finalize_kernel(const Params P, float *terms,
                                                           float *loss_out, float w0,
                                                           float w1, float w2,
                                                           float loss_scale, int write_oob)
{
    __shared__ double s_sum[DVSOF_MAX_SCALES][5];
    __shared__ double s_border[DVSOF_MAX_SCALES][64];   // per (scale, sample slot)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    constexpr int NW = NT / kWave;
    // (k, i): global sums of photo and the four smoothness directions
    for (int item = wave; item < P.K * 5; item += NW) {
        const int k = item / 5, i = item - 5 * k;
        const ScaleDev &S = P.s[k];
        const float *part = P.partials + (size_t)S.block_begin * NPART;
        const int nb = P.N * S.tiles_per_sample;
        double a = 0;
        for (int b = lane; b < nb; b += kWave) a += (double)part[(size_t)b * NPART + i];
        a = wave_sum(a);
        if (lane == 0) s_sum[k][i] = a;
    }
    // (k, n): border sum / count of one sample; partial border per wave slot
    for (int k = 0; k < P.K; ++k)
        if (tid < 64) s_border[k][tid] = 0;
    __syncthreads();
    // sample n belongs to wave n % NW (64 % NW == 0: every slot n & 63 has one
    // owner wave, which visits its samples in increasing n)
    for (int item = 0; item < P.K * ((P.N + NW - 1) / NW); ++item) {
        const int k = item % P.K, n = (item / P.K) * NW + wave;
        if (n >= P.N) continue;
        const ScaleDev &S = P.s[k];
        const float *part = P.partials + ((size_t)S.block_begin + (size_t)n * S.tiles_per_sample) * NPART;
        double bs = 0, c = 0;
        for (int b = lane; b < S.tiles_per_sample; b += kWave) {
            bs += (double)part[(size_t)b * NPART + 5];
            c += (double)part[(size_t)b * NPART + 6];
        }
        bs = wave_sum(bs);
        c = wave_sum(c);
        if (lane == 0) {
            // utils/loss.py:101,113 -- samples are visited in increasing n by
            // the same wave slot, so the accumulation order is fixed
            if (c > 0) s_border[k][n & 63] += bs / (2.0 * c * (double)P.N);
            if (write_oob) P.oob[k * P.N + n] = (int)c;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double total[3] = {0, 0, 0};
        for (int k = 0; k < P.K; ++k) {
            const ScaleDev &S = P.s[k];
            const double *a = s_sum[k];
            double border = 0;
            for (int n = 0; n < 64; ++n) border += s_border[k][n];
            // empty crops contribute 0 (utils/loss.py:29-30)
            const double sm = ((S.c_smooth[0] > 0 ? a[1] / S.c_smooth[0] : 0) +
                               (S.c_smooth[1] > 0 ? a[2] / S.c_smooth[1] : 0) +
                               (S.c_smooth[2] > 0 ? (a[3] + a[4]) / S.c_smooth[2] : 0)) / 4.0;
            const double ph = a[0] / ((double)P.N * S.h * S.w);
            terms[0 * P.K + k] = (float)sm;
            terms[1 * P.K + k] = (float)ph;
            terms[2 * P.K + k] = (float)border;
            total[0] += sm;
            total[1] += ph;
            total[2] += border;
        }
        if (loss_out)  // combined_loss, utils/training.py:23
            loss_out[0] = (float)((w0 * total[0] + w1 * total[1] + w2 * total[2]) /
                                  (double)P.K * (double)loss_scale);
    }
}

// F.interpolate(bilinear, align_corners=True), utils/loss.py:20-21.
__global__ __launch_bounds__(NT) void resize_bilinear_ac_kernel(const float *__restrict__ src,
                                                                float *__restrict__ dst,
                                                                int hin, int win, int hout,
                                                                int wout, float sh, float sw)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= wout || y >= hout) return;
    const float *s = src + (size_t)n * hin * win;
    const float fy = sh * (float)y, fx = sw * (float)x;
    const int y0 = min((int)fy, hin - 1), x0 = min((int)fx, win - 1);
    const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    dst[((size_t)n * hout + y) * wout + x] =
        hy * (hx * s[(size_t)y0 * win + x0] + lx * s[(size_t)y0 * win + x1]) +
        ly * (hx * s[(size_t)y1 * win + x0] + lx * s[(size_t)y1 * win + x1]);
}

int build_params(const dvsof_loss_scale_t *sc, int K, int N, Params &P, int &total_blocks)
{
    if (!sc || K < 1 || K > DVSOF_MAX_SCALES || N < 1) return DVSOF_EINVAL;
    P.K = K;
    P.N = N;
    int begin = 0;
    for (int k = 0; k < K; ++k) {
        const int h = sc[k].h, w = sc[k].w;
        if (h < 1 || w < 1 || !sc[k].frames || !sc[k].flow) return DVSOF_EINVAL;
        ScaleDev &S = P.s[k];
        S.frames = sc[k].frames;
        S.flow = sc[k].flow;
        S.grad = sc[k].grad_flow;
        S.h = h;
        S.w = w;
        S.tiles_x = (w + TW - 1) / TW;
        S.tiles_per_sample = S.tiles_x * ((h + TH - 1) / TH);
        S.block_begin = begin;
        begin += N * S.tiles_per_sample;
        S.half_w = (float)((w - 1) / 2.0);
        S.half_h = (float)((h - 1) / 2.0);
        S.c_smooth[0] = (double)N * 2 * h * (w - 1);
        S.c_smooth[1] = (double)N * 2 * (h - 1) * w;
        S.c_smooth[2] = (double)N * 2 * (h - 1) * (w - 1);
        for (int i = 0; i < 3; ++i)
            S.k_smooth[i] = S.c_smooth[i] > 0 ? (float)(1.0 / (4.0 * S.c_smooth[i])) : 0.f;
        S.k_photo = (float)(1.0 / ((double)N * h * w));
    }
    total_blocks = begin;
    return DVSOF_OK;
}

}  // namespace

extern "C" {

size_t dvsof_loss_workspace_bytes(const dvsof_loss_scale_t *sc, int K, int N)
{
    Params P;
    int nb = 0;
    if (build_params(sc, K, N, P, nb) != DVSOF_OK) return 0;
    return (size_t)nb * NPART * sizeof(float);
}

int dvsof_resize_bilinear_ac(const float *src, float *dst, int n, int hin, int win, int hout,
                             int wout, void *stream)
{
    if (!src || !dst || n < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    if (n > 65535) return DVSOF_EINVAL;
    const float sh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
    const float sw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
    dim3 grid((wout + 63) / 64, (hout + 3) / 4, n);
    hipLaunchKernelGGL(resize_bilinear_ac_kernel, grid, dim3(NT), 0, as_stream(stream), src, dst,
                       hin, win, hout, wout, sh, sw);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_loss_fwd(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                   const int32_t *stop, float *terms, int32_t *oob, void *ws, size_t ws_bytes,
                   void *stream)
{
    Params P;
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !terms || !oob || !ws) return DVSOF_EINVAL;
    if (ws_bytes < (size_t)nb * NPART * sizeof(float)) return DVSOF_ENOSPACE;
    P.start = start;
    P.stop = stop;
    P.partials = (float *)ws;
    P.oob = oob;
    P.seeds_dev = nullptr;
    hipLaunchKernelGGL((loss_main_kernel<true, false>), dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(NT), 0, as_stream(stream), P, terms,
                       (float *)nullptr, 0.f, 0.f, 0.f, 1.f, 1);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_loss_bwd(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                   const int32_t *stop, const float *seeds, const int32_t *oob, void *stream)
{
    Params P;
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !seeds || !oob) return DVSOF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!sc[k].grad_flow) return DVSOF_EINVAL;
    P.start = start;
    P.stop = stop;
    P.partials = nullptr;
    P.oob = const_cast<int32_t *>(oob);
    P.seeds_dev = seeds;
    hipLaunchKernelGGL((loss_main_kernel<false, true>), dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_loss_fused(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                     const int32_t *stop, const float *w, float loss_scale, float *terms,
                     float *loss_out, int32_t *oob, void *ws, size_t ws_bytes, void *stream)
{
    Params P;
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !w || !terms || !loss_out || !oob || !ws) return DVSOF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!sc[k].grad_flow) return DVSOF_EINVAL;
    if (ws_bytes < (size_t)nb * NPART * sizeof(float)) return DVSOF_ENOSPACE;
    P.start = start;
    P.stop = stop;
    P.partials = (float *)ws;
    P.oob = oob;
    P.seeds_dev = nullptr;
    for (int i = 0; i < 3; ++i) P.seeds_host[i] = w[i] / (float)K * loss_scale;
    DVSOF_HIP_TRY(hipMemsetAsync(oob, 0, sizeof(int32_t) * (size_t)K * N, as_stream(stream)));
    hipLaunchKernelGGL(loss_count_oob_kernel, dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL((loss_main_kernel<true, true>), dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(NT), 0, as_stream(stream), P, terms,
                       loss_out, w[0], w[1], w[2], loss_scale, 0);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"
