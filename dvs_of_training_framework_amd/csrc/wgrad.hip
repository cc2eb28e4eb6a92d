// Weight gradient of the gather-convolution as an implicit GEMM on the f32
// matrix cores:
//   dW[co][tap][ci] = sum_pix gout[pix][co] * Xvirt[pix @ tap][ci]
//   GEMM rows = co, columns = flattened (tap, ci) of one concat member,
//   K = output pixels, split over blockIdx.z into slabs that a separate pass
//   adds in a fixed order (bitwise reproducible, no float atomics).
// The bias gradient (column sums of gout) is accumulated on the VALU by the
// workgroups of the first column tile while they stream gout anyway.
// Replaces the ATen convolution_backward(weight, bias) that
// utils/training.py:158 triggers inside the (absent) EV_FlowNet predictor.
//
// Both operands are K-major in memory (pixel rows, channels contiguous), so
// LDS holds [k][row] slices written with ds_write_b128 and read one float per
// lane (consecutive lanes -> consecutive banks); at 64 cycles per MFMA the
// reads are far off the critical path.
#include "conv_common.h"
#include <stdio.h>
#include <stdlib.h>

size_t wgrad_flat_workspace_floats(const FlatWG *flat, int nflat);
bool wgrad2_eligible(const WGradParams &P);
int wgrad2_launch(const WGradParams &P, int tile, int ntiles, hipStream_t st);


namespace {

template <int WROWS, int WCOLS, int TM, int TN>
__global__ __launch_bounds__(CONV_NT) void wgrad_kernel(const WGradParams P)
{
    constexpr int BMc = WROWS * TM * 32, BN = WCOLS * TN * 32;
    constexpr int LDA = BMc + 4, LDB = BN + 4;
    constexpr int QA = BMc / 4, RPA = CONV_NT / QA, PA = (BK + RPA - 1) / RPA;
    constexpr int QB = BN / 4, RPB = CONV_NT / QB, PB = (BK + RPB - 1) / RPB;
    constexpr int RPF = CONV_NT / BN, PF = BK / RPF;  // flat: one column per thread
    constexpr int NBREG = PB * 4 > PF ? PB * 4 : PF;
    static_assert(WROWS * WCOLS == CONV_NT / kWave, "4 waves");

    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int taps = P.ks * P.ks;

    // which concat member / column range this workgroup owns
    int s = 0;
    for (int i = 1; i < P.nsrc; ++i)
        if ((int)blockIdx.x >= P.tile_begin[i]) s = i;
    const GSrc &S = P.src[s];
    int coff = 0;
    for (int i = 0; i < s; ++i) coff += P.src[i].C;
    const int f0 = ((int)blockIdx.x - P.tile_begin[s]) * BN;
    const int fmax = taps * S.C;
    const int co0 = blockIdx.y * BMc;
    const int ph = blockIdx.z / P.S, split = blockIdx.z - ph * P.S;
    const int phy = ph >> 1, phx = ph & 1;
    const int pad_y = P.pad - phy * P.ph_pad, pad_x = P.pad - phx * P.ph_pad;
    const size_t g_ph = (size_t)phy * P.g_py + (size_t)phx * P.g_px;
    const int kbeg = split * P.klen;
    const int kend = min(P.M, kbeg + P.klen);
    const int nsteps = (kend - kbeg + BK - 1) / BK;
    const bool flat = S.flat != 0;
    const bool do_bias = (blockIdx.x == 0) && (P.dbias != nullptr);

    // per-thread constants of the loaders
    const int qa = tid % QA, pra = tid / QA;
    const bool a_ok = co0 + 4 * qa < P.Cout;
    const int colB = flat ? tid % BN : 4 * (tid % QB);
    const int prb = flat ? tid / BN : tid / QB;
    const int fB = f0 + colB;
    const bool b_ok = fB < fmax;
    const int tapB = b_ok ? fB / S.C : 0, cB = fB - tapB * S.C;
    const int kyB = tapB / P.ks, kxB = tapB - kyB * P.ks;

    f32x4 rga[PA];
    float rgb[NBREG];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    auto gather = [&](int pix, bool &ok, size_t &o) {
        ok = b_ok && pix < kend;
        o = 0;
        if (!ok) return;
        const int ox = pix % P.Wo, t = pix / P.Wo;
        const int oy = t % P.Ho, b = t / P.Ho;
        const int Y = oy * P.stride - pad_y + kyB, X = ox * P.stride - pad_x + kxB;
        ok = ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
        if (P.up == UP_ZERO) ok &= ((Y | X) & 1) == 0;
        const int ys = P.up ? Y >> 1 : Y, xs = P.up ? X >> 1 : X;
        o = (size_t)b * S.sb + (size_t)ys * S.sy + (size_t)xs * S.sx + (size_t)cB * S.sc;
    };

    auto load_tiles = [&](int kbase) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = pra + RPA * i, pix = kbase + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < BK && a_ok && pix < kend) {
                const int ox = pix % P.Wo, t = pix / P.Wo;
                const int oy = t % P.Ho, b = t / P.Ho;
                v = *(const f32x4u *)(P.gout + (size_t)b * P.g_sb + (size_t)oy * P.g_sy +
                                      (size_t)ox * P.g_sx + g_ph + co0 + 4 * qa);
            }
            rga[i] = v;
        }
        if (!flat) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int r = prb + RPB * i;
                bool ok;
                size_t o;
                gather(kbase + r, ok, o);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < BK && ok) v = *(const f32x4u *)(S.p + o);
                rgb[4 * i] = v[0]; rgb[4 * i + 1] = v[1];
                rgb[4 * i + 2] = v[2]; rgb[4 * i + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                bool ok;
                size_t o;
                gather(kbase + prb + RPF * i, ok, o);
                rgb[i] = ok ? S.p[o] : 0.f;
            }
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = pra + RPA * i;
            if (r < BK) *(f32x4 *)&As[buf][r][4 * qa] = rga[i];
            if (do_bias) bsum += rga[i];
        }
        if (!flat) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int r = prb + RPB * i;
                if (r < BK)
                    *(f32x4 *)&Bs[buf][r][colB] =
                        f32x4{rgb[4 * i], rgb[4 * i + 1], rgb[4 * i + 2], rgb[4 * i + 3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < PF; ++i) Bs[buf][prb + RPF * i][colB] = rgb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lrow = lane & 31, lh = lane >> 5;
    if (nsteps > 0) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool has_next = step + 1 < nsteps;
        if (has_next) load_tiles(kbeg + (step + 1) * BK);
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[cur][2 * q + lh][(wr * TM + t) * 32 + lrow];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = Bs[cur][2 * q + lh][(wc * TN + t) * 32 + lrow];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] =
                        __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        if (has_next) store_tiles(cur ^ 1);
        __syncthreads();
    }

    const size_t wsize = (size_t)P.Cout * taps * P.Cin_tot;
    float *dW = P.dW + (size_t)blockIdx.z * wsize;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int f = f0 + (wc * TN + tn) * 32 + lrow;
        if (f >= fmax) continue;
        const int tap = f / S.C, c = f - tap * S.C;
        const size_t col = (size_t)tap * P.Cin_tot + coff + c;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + (wr * TM + tm) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (co < P.Cout) dW[(size_t)co * taps * P.Cin_tot + col] = acc[tm][tn][reg];
            }
    }

    if (do_bias) {  // column sums of gout for this K split (fixed order)
        float *scr = &As[0][0][0];  // RPA x BMc floats, well inside As
        __syncthreads();
        *(f32x4 *)&scr[pra * BMc + 4 * qa] = bsum;
        __syncthreads();
        if (tid < BMc && co0 + tid < P.Cout) {
            float t = 0.f;
            for (int r = 0; r < RPA; ++r) t += scr[r * BMc + tid];
            P.dbias[(size_t)blockIdx.z * P.Cout + co0 + tid] = t;
        }
    }
}

constexpr int COLSUM_MAX_BLOCKS = 512;

int colsum_blocks(long long rows)
{
    const long long nb = (rows + 127) / 128;
    return (int)(nb < 1 ? 1 : nb > COLSUM_MAX_BLOCKS ? COLSUM_MAX_BLOCKS : nb);
}

// part[block][c] = sum over this block's rows of g[row][c]   (g dense [rows][C]).
// C % 4 == 0: C/4 lanes per row load float4 (whole rows = contiguous bytes),
// 256/(C/4) rows in flight per pass, 4 passes unrolled.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g, long long rows,
                                                     int C, float *__restrict__ part)
{
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x;
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per;
    const long long r1 = r0 + per < rows ? r0 + per : rows;
    const int C4 = C >> 2;
    for (int c0 = 0; c0 < C4; c0 += 256) {
        const int G = (C4 - c0) < 256 ? (C4 - c0) : 256;   // float4 lanes per row
        const int RP = 256 / G;                             // rows per pass
        const int c = tid % G, rl = tid / G;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (rl < RP) {
            const float *base = g + (size_t)(c0 + c) * 4;
            long long r = r0 + rl;
            for (; r + 3 * RP < r1; r += 4 * RP) {
                a0 += *(const f32x4u *)(base + (size_t)r * C);
                a1 += *(const f32x4u *)(base + (size_t)(r + RP) * C);
                a2 += *(const f32x4u *)(base + (size_t)(r + 2 * RP) * C);
                a3 += *(const f32x4u *)(base + (size_t)(r + 3 * RP) * C);
            }
            for (; r < r1; r += RP) a0 += *(const f32x4u *)(base + (size_t)r * C);
        }
        red[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < G) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < RP; ++k) t += red[k * G + tid];
            *(f32x4u *)(part + (size_t)blockIdx.x * C + (size_t)(c0 + tid) * 4) = t;
        }
        __syncthreads();
    }
}

// scalar fallback for C % 4 != 0
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const float *__restrict__ g,
                                                            long long rows, int C,
                                                            float *__restrict__ part)
{
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * per;
    const long long r1 = r0 + per < rows ? r0 + per : rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (long long r = r0; r < r1; ++r) a += g[r * C + c];
        part[(size_t)blockIdx.x * C + c] = a;
    }
}

// Tail workgroups of the reduce kernels: dbias[c] = sum_z bias_part[z][c]
// (the per-slab column sums wgrad2 leaves behind), fixed order.
// The 256 threads of a tail workgroup split into channel lanes x slab lanes: a
// single thread per channel walked all nslab partials as one dependent chain
// (27 us for 128 slabs x 32 channels, longer than the reduce it rides on).
__device__ __forceinline__ void bias_tail(int blk, const float *__restrict__ bias_part, int nslab,
                                          int Cout, float *__restrict__ dbias)
{
    __shared__ float bred[256];
    const int c0 = blk * 256;
    const int G = Cout - c0 < 256 ? Cout - c0 : 256;     // channels of this workgroup
    int gl = 1;
    while (gl < G) gl <<= 1;                              // channel lanes (power of two <= 256)
    const int RP = 256 / gl;                              // slab lanes
    const int cl = threadIdx.x % gl, zl = threadIdx.x / gl;
    float a = 0.f;
    if (cl < G) {
        const float *p = bias_part + c0 + cl;
        int z = zl;
        for (; z + 3 * RP < nslab; z += 4 * RP) {         // four loads in flight
            const float v0 = p[(size_t)z * Cout], v1 = p[(size_t)(z + RP) * Cout];
            const float v2 = p[(size_t)(z + 2 * RP) * Cout], v3 = p[(size_t)(z + 3 * RP) * Cout];
            a += v0;
            a += v1;
            a += v2;
            a += v3;
        }
        for (; z < nslab; z += RP) a += p[(size_t)z * Cout];
    }
    bred[threadIdx.x] = a;
    __syncthreads();
    if (zl == 0 && cl < G) {
        for (int k = 1; k < RP; ++k) a += bred[k * gl + cl];   // fixed order
        dbias[c0 + cl] = a;
    }
}

// dbias[c] = sum_b part[b][c] over the colsum workgroups: one wave per channel,
// lane-strided partial sums + shuffle tree (fixed order).  (slab_reduce_kernel
// would walk the nb partials of a channel serially in one thread: 35 us for
// 512 x 64.)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ part, int nb,
                                                            int C, float *__restrict__ dbias)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float a = 0.f;
    for (int b = lane; b < nb; b += 64) a += part[(size_t)b * C + c];
    a = wave_sum(a);
    if (lane == 0) dbias[c] = a;
}

// out[i] = sum_z slab[z][i], fixed order; four slabs in flight per thread
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slab,
                                                          float *__restrict__ out, size_t n, int S,
                                                          int nb_main, const float *bias_part,
                                                          int nslab, int Cout, float *dbias)
{
    if ((int)blockIdx.x >= nb_main) {
        bias_tail(blockIdx.x - nb_main, bias_part, nslab, Cout, dbias);
        return;
    }
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 3 < S; z += 4) {
            const f32x4 v0 = *(const f32x4u *)(slab + (size_t)z * n + i);
            const f32x4 v1 = *(const f32x4u *)(slab + (size_t)(z + 1) * n + i);
            const f32x4 v2 = *(const f32x4u *)(slab + (size_t)(z + 2) * n + i);
            const f32x4 v3 = *(const f32x4u *)(slab + (size_t)(z + 3) * n + i);
            a = (((a + v0) + v1) + v2) + v3;
        }
        for (; z < S; ++z) a += *(const f32x4u *)(slab + (size_t)z * n + i);
        *(f32x4u *)(out + i) = a;
    } else {
        for (size_t j = i; j < n; ++j) {
            float a = 0.f;
            for (int z = 0; z < S; ++z) a += slab[(size_t)z * n + j];
            out[j] = a;
        }
    }
}

// Sub-pixel fold: the 3x3 weight gradient from the 4 phases' 2x2 gradients,
//   dW[co][ky][kx][ci] = sum over (py,a) with ky in S(py,a), (px,b) with kx in S(px,b),
//   S(0,0)={0} S(0,1)={1,2} S(1,0)={0,1} S(1,1)={2}   (and over the K splits).
// slab[(ph*S + s)][co][a][b][ci], ph = 2*py + px.
// ZG threads share one output quad and split the K-split index z between them
// (decoder layers with few output channels have tiny dW but many splits: one
// thread per quad would walk 4*S dependent loads); partial sums meet in LDS in
// the fixed order zg = 0..ZG-1.
template <int ZG>
__global__ __launch_bounds__(256) void subpixel_fold_kernel(const float *__restrict__ slab,
                                                            float *__restrict__ dW, int Cout,
                                                            int Ctot, int S, int nb_main,
                                                            const float *bias_part, int nslab,
                                                            float *dbias)
{
    constexpr int QPB = 256 / ZG;          // quads per workgroup
    __shared__ f32x4 red[ZG > 1 ? 256 : 1];
    if ((int)blockIdx.x >= nb_main) {
        bias_tail(blockIdx.x - nb_main, bias_part, nslab, Cout, dbias);
        return;
    }
    // one quad per (co, tap, 4 consecutive ci)
    const int Q = (Ctot + 3) >> 2;
    const int ql = threadIdx.x % QPB, zg = threadIdx.x / QPB;
    const size_t idx = (size_t)blockIdx.x * QPB + ql;
    const bool live = idx < (size_t)Cout * 9 * Q;
    const int ci = (int)(idx % Q) * 4;
    const int tap = (int)((idx / Q) % 9), co = (int)(idx / ((size_t)9 * Q));
    const int ky = tap / 3, kx = tap - 3 * ky;
    const size_t wsize = (size_t)Cout * 4 * Ctot;
    const int cnt = Ctot - ci;          // >= 1; static lane indices only (no scratch)
    const bool full = cnt >= 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // contributors of kernel index k along one axis: (phase bit, tap bit)
    //   k=0: (0,0),(1,0)   k=1: (0,1),(1,0)   k=2: (0,1),(1,1)
    const int py0 = 0, a0 = ky > 0, py1 = 1, a1 = ky == 2;
    const int px0 = 0, b0 = kx > 0, px1 = 1, b1 = kx == 2;
#define FOLD_ADD(PY, A, PX, B)                                                              \
    {                                                                                       \
        const float *sl = slab + (size_t)(2 * (PY) + (PX)) * S * wsize +                    \
                          ((size_t)co * 4 + (A) * 2 + (B)) * Ctot + ci;                     \
        if (full) {                                                                         \
            for (int z = zg; z < S; z += ZG) acc += *(const f32x4u *)(sl + (size_t)z * wsize); \
        } else {                                                                            \
            for (int z = zg; z < S; z += ZG) {                                              \
                const float *q = sl + (size_t)z * wsize;                                    \
                acc[0] += q[0];                                                             \
                if (cnt > 1) acc[1] += q[1];                                                \
                if (cnt > 2) acc[2] += q[2];                                                \
            }                                                                               \
        }                                                                                   \
    }
    if (live) {
        FOLD_ADD(py0, a0, px0, b0)
        FOLD_ADD(py0, a0, px1, b1)
        FOLD_ADD(py1, a1, px0, b0)
        FOLD_ADD(py1, a1, px1, b1)
    }
#undef FOLD_ADD
    if (ZG > 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (zg != 0) return;
        for (int k = 1; k < ZG; ++k) acc += red[k * QPB + ql];
    }
    if (!live) return;
    float *o = dW + ((size_t)co * 9 + tap) * Ctot + ci;
    if (full) {
        *(f32x4u *)o = acc;
    } else {
        o[0] = acc[0];
        if (cnt > 1) o[1] = acc[1];
        if (cnt > 2) o[2] = acc[2];
    }
}

// ---- weight gradient of "flat" concat members (2-channel flow, 5-bin voxel
// grid) on the VALU: dW[co][tap][c] = sum_pix gout[pix][co] * Xvirt[pix@tap][c]
// straight from the layer definition (nearest-upsampled virtual input), so it
// needs neither the sub-pixel phases nor the MFMA tiles (an MFMA column tile
// for 8..45 columns would run the whole K loop at <10 % utilisation).
// HBM-bound: reads gout once.  part[block][co][NCOL], then a fixed-order sum.
constexpr int FLAT_BLOCKS = 512;    // most partial sums of a flat member (workspace bound)
constexpr int FLAT_PIX = 64;   // pixels staged per round


template <int NCOL>
__global__ __launch_bounds__(256) void wgrad_flat_kernel(const FlatWG P, float *__restrict__ part)
{
    constexpr int C = NCOL / 9;              // 3x3 taps (flat_ncol_ok)
    constexpr int NC4 = (NCOL + 3) / 4 * 4;  // row stride: b128 reads
    __shared__ __attribute__((aligned(16))) float xs[FLAT_PIX][NC4];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int G = P.Cout < 256 ? P.Cout : 256;      // channel lanes
    const int RP = 256 / G;                         // pixel lanes
    const int cl = tid % G, pl = tid / G;
    const long long per = ((long long)P.M + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * per;
    const long long p1 = p0 + per < P.M ? p0 + per : P.M;
    // staging role: pixel sp of the round, columns sg, sg+4, ... (tap and
    // channel of a column are compile-time constants)
    const int sp = tid & (FLAT_PIX - 1), sg = tid >> 6;
    if (tid < FLAT_PIX)
        for (int i = NCOL; i < NC4; ++i) xs[tid][i] = 0.f;
    for (int cbase = 0; cbase < P.Cout; cbase += G) {
        float acc[NC4];
#pragma unroll
        for (int i = 0; i < NC4; ++i) acc[i] = 0.f;
        for (long long base = p0; base < p1; base += FLAT_PIX) {
            __syncthreads();
            {
                const long long pix = base + sp;
                const bool pok = pix < p1;
                const int ox = (int)(pix % P.Wo);
                const long long t = pix / P.Wo;
                const int oy = (int)(t % P.Ho), b = (int)(t / P.Ho);
                const int Y0 = oy * P.stride - P.pad, X0 = ox * P.stride - P.pad;
                const float *sb = P.S.p + (size_t)b * P.S.sb;
#pragma unroll
                for (int j = 0; j < (NCOL + 3) / 4; ++j) {
                    const int col = sg + 4 * j;     // sg < 4: col < NCOL checked below
                    float v = 0.f;
                    if (col < NCOL) {
                        // tap/c from col: small runtime div on constants C (folds per sg value)
                        const int tap = col / C, c = col - tap * C;
                        const int ky = tap / 3, kx = tap - 3 * ky;
                        const int Y = Y0 + ky, X = X0 + kx;
                        if (pok && (unsigned)Y < (unsigned)P.Hv && (unsigned)X < (unsigned)P.Wv) {
                            const int ys = P.up ? Y >> 1 : Y, xsrc = P.up ? X >> 1 : X;
                            v = sb[(size_t)ys * P.S.sy + (size_t)xsrc * P.S.sx + (size_t)c * P.S.sc];
                        }
                        xs[sp][col] = v;
                    }
                }
            }
            __syncthreads();
            if (pl < RP && cbase + cl < P.Cout) {
                const int np = (int)((p1 - base) < FLAT_PIX ? (p1 - base) : FLAT_PIX);
                // eight gout values in flight per thread (one dependent load per
                // pixel left the loop waiting on memory: 67 us for 33 MB)
                for (int p0b = pl; p0b < np; p0b += 8 * RP) {
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int p = p0b + u * RP;
                        g[u] = p < np ? P.gout[(size_t)(base + p) * P.Cout + cbase + cl] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int p = p0b + u * RP;
                        if (p >= np) break;
#pragma unroll
                        for (int i = 0; i < NC4; i += 4) {
                            const f32x4 x = *(const f32x4 *)&xs[p][i];
                            acc[i] = fmaf(g[u], x[0], acc[i]);
                            acc[i + 1] = fmaf(g[u], x[1], acc[i + 1]);
                            acc[i + 2] = fmaf(g[u], x[2], acc[i + 2]);
                            acc[i + 3] = fmaf(g[u], x[3], acc[i + 3]);
                        }
                    }
                }
            }
        }
        // sum the pixel lanes, one column at a time (static indices: a
        // runtime-indexed acc[] would live in scratch memory)
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
            __syncthreads();
            red[tid] = acc[i];
            __syncthreads();
            if (tid < G && cbase + tid < P.Cout) {
                float t = 0.f;
                for (int k = 0; k < RP; ++k) t += red[k * G + tid];
                part[((size_t)blockIdx.x * P.Cout + cbase + tid) * NCOL + i] = t;
            }
        }
    }
}

// The same gradient on the matrix cores, no LDS in the loop: for
// v_mfma_f32_32x32x2_f32 the A operand of lane (i = lane & 31, k = lane >> 5) is
// gout[pixel 2q + k][channel i] -- consecutive lanes read consecutive channels,
// a coalesced global load straight into the operand register -- and the B
// operand is the im2col value of column j = lane & 31 at pixel 2q + k, gathered
// from the small flat tensor (cache resident).  The kernel streams gout once;
// it is bound by that read (67 MB at the finest decoder stage), not by the
// 64-cycle MFMA per pixel pair.  2 x UN pixel pairs are in flight per wave, 8
// waves per workgroup (a one-wave-per-workgroup version with UN pairs in
// flight was latency bound: 49 us at the finest decoder stage).
// NCB = ceil(ncol / 32) column blocks, NRB = 32-channel row blocks per pass.
// with_bias: im2col column `ncol` is the constant 1, so output column ncol is the
// column sum of gout = the bias gradient (free: the 32-wide MFMA block has idle
// columns); partial rows then have ncol + 1 entries.
template <int NCB, int NRB>
__global__ __launch_bounds__(512) void wgrad_flat_mfma_kernel(const FlatWG P, float *__restrict__ part,
                                                              const int with_bias)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int UN = 8, NW = 8;
    __shared__ float red[NRB * NCB * 16 * 64];
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncol = P.ncol, C = P.S.C;
    const int nout = ncol + (with_bias ? 1 : 0);      // output columns / partial row length
    // pixel pairs of this wave (contiguous range)
    const long long npair = ((long long)P.M + 1) / 2;
    const long long per = (npair + (long long)gridDim.x * NW - 1) / ((long long)gridDim.x * NW);
    const long long q0 = ((long long)blockIdx.x * NW + wave) * per;
    const long long q1 = q0 + per < npair ? q0 + per : npair;
    // per-lane column constants
    int cky[NCB], ckx[NCB];
    long long ccoff[NCB];
    bool cok[NCB], cone[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int col = cb * 32 + li;
        cok[cb] = col < ncol;
        cone[cb] = with_bias && col == ncol;
        const int tap = cok[cb] ? col / C : 0, c = col - tap * C;
        cky[cb] = tap / P.ks;
        ckx[cb] = tap - cky[cb] * P.ks;
        ccoff[cb] = (long long)c * P.S.sc;
    }
    const int nrb = P.Cout / 32;
    for (int rb0 = 0; rb0 < nrb; rb0 += NRB) {
        f32x16 acc[NRB][NCB];
#pragma unroll
        for (int a = 0; a < NRB; ++a)
#pragma unroll
            for (int b = 0; b < NCB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        // this lane's pixel (2q + lk) as (b, oy, ox), advanced by 2 per step
        long long pix = 2 * q0 + lk, qn = q0;
        int ox = 0, oy = 0, bb = 0;
        if (q0 < q1) {
            ox = (int)(pix % P.Wo);
            const long long t = pix / P.Wo;
            oy = (int)(t % P.Ho);
            bb = (int)(t / P.Ho);
        }
        const float *gp = P.gout + (size_t)pix * P.Cout + rb0 * 32 + li;
        // Per-lane state of the output row the pixel is in: element offset of the source
        // row each column reads and its validity -- recomputed on row changes only, so a
        // step costs an add, a compare, a shift and one multiply-add per column.
        long long ybase[NCB];
        bool yok[NCB];
        auto new_row = [&]() {
            const int Y0 = oy * P.stride - P.pad;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int Y = Y0 + cky[cb];
                yok[cb] = cok[cb] & ((unsigned)Y < (unsigned)P.Hv);
                const int ys = P.up ? Y >> 1 : Y;
                ybase[cb] = (long long)bb * P.S.sb + (long long)ys * P.S.sy + ccoff[cb];
            }
        };
        new_row();
        const int xstep = 2 * P.stride, sx = P.S.sx;
        int X0 = ox * P.stride - P.pad;
        // UN pixel pairs: gout values and im2col values of this lane
        auto load_set = [&](float (&av)[UN][NRB], float (&bv)[UN][NCB]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool pok = (qn + u < q1) && (pix < P.M);
#pragma unroll
                for (int a = 0; a < NRB; ++a) av[u][a] = (pok && rb0 + a < nrb) ? gp[a * 32] : 0.f;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int X = X0 + ckx[cb];
                    const bool ok = pok & yok[cb] & ((unsigned)X < (unsigned)P.Wv);
                    const int xs = P.up ? X >> 1 : X;
                    bv[u][cb] = ok ? P.S.p[ybase[cb] + (long long)xs * sx]
                                   : (cone[cb] && pok) ? 1.f : 0.f;
                }
                pix += 2;
                gp += 2 * (size_t)P.Cout;
                ox += 2;
                X0 += xstep;
                if (ox >= P.Wo) {
                    ox -= P.Wo;
                    X0 = ox * P.stride - P.pad;
                    if (++oy == P.Ho) {
                        oy = 0;
                        ++bb;
                    }
                    new_row();
                }
            }
            qn += UN;
        };
        auto mfma_set = [&](const float (&av)[UN][NRB], const float (&bv)[UN][NCB]) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int a = 0; a < NRB; ++a)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[a][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][cb], acc[a][cb], 0, 0, 0);
        };
        // two register sets: the loads of the next UN pairs are in flight while
        // the MFMAs of the current ones run
        float a0[UN][NRB], b0[UN][NCB], a1[UN][NRB], b1[UN][NCB];
        if (q0 < q1) load_set(a0, b0);
        for (long long q = q0; q < q1; q += 2 * UN) {
            if (q + UN < q1) load_set(a1, b1);
            mfma_set(a0, b0);
            if (q + UN < q1) {
                if (q + 2 * UN < q1) load_set(a0, b0);
                mfma_set(a1, b1);
            }
        }
        // the 8 waves add up in LDS in wave order (fixed order: reproducible);
        // the last one writes the workgroup's partial sum.
        // acc[reg] <-> channel (reg & 3) + 8 (reg >> 2) + 4 lk of the row block, column li
        for (int w = 0; w < NW; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < NRB; ++a)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float *slot = &red[((a * NCB + cb) * 16 + r) * 64 + lane];
                            const float v = w == 0 ? acc[a][cb][r] : *slot + acc[a][cb][r];
                            if (w < NW - 1) {
                                *slot = v;
                            } else {
                                const int col = cb * 32 + li;
                                const int co = (rb0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                                if (rb0 + a < nrb && col < nout)
                                    part[((size_t)blockIdx.x * P.Cout + co) * nout + col] = v;
                            }
                        }
            }
            __syncthreads();
        }
    }
#endif
}

// dW[co][tap][coff + c] = sum_blocks part[blk][co][tap*C + c]; one wave per
// output, lanes stride over the workgroups, shuffle tree (fixed order)
// (nout = ncol + 1 with dbias: the last entry of a partial row is the bias gradient)
__global__ __launch_bounds__(256) void wgrad_flat_reduce_kernel(const float *__restrict__ part,
                                                                int nblocks, int Cout, int ncol,
                                                                int C, int coff, int row_stride,
                                                                int tap_stride, float *dW, int nout,
                                                                float *dbias)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= Cout * nout) return;
    const int co = i / nout, col = i - co * nout;
    float a = 0.f;
    for (int b = lane; b < nblocks; b += 64) a += part[(size_t)b * Cout * nout + i];
    a = wave_sum(a);
    if (lane == 0) {
        if (col == ncol) {
            dbias[co] = a;
        } else {
            const int tap = col / C, c = col - tap * C;
            dW[(size_t)co * row_stride + (size_t)tap * tap_stride + coff + c] = a;
        }
    }
}

bool flat_ncol_ok(int ncol)
{
    return ncol == 18 || ncol == 27 || ncol == 45 || ncol == 81 || ncol == 108;
}

// Does this flat member take the matrix-core kernel?
// measured (batch 8): MFMA 33 vs VALU 46 us at M = 524288 / 18 columns, 39 vs 52 us at
// M = 131072 / 45 columns; a tie at M = 131072 / 18 columns; VALU wins below
bool flat_uses_mfma(const FlatWG &F)
{
    static const bool no_mfma = getenv("DVSOF_WGRAD_FLAT_VALU") != nullptr;
    static const bool force_mfma = getenv("DVSOF_WGRAD_FLAT_MFMA") != nullptr;
    const int ncb = (F.ncol + 31) / 32;
    const bool big = F.M >= 262144 || (ncb == 2 && F.M >= 65536);
    return !no_mfma && (big || force_mfma) && (F.Cout % 32) == 0 && ncb <= 2 && F.Wo >= 2;
}

// dbias != NULL (matrix-core kernel only, ncol % 32 != 0): the bias gradient comes out as
// one more output column
int flat_launch(const FlatWG &F, float *part, float *dW, float *dbias, hipStream_t st)
{
    const long long nbl = ((long long)F.M + FLAT_PIX - 1) / FLAT_PIX;
    int nb = (int)(nbl < 512 ? (nbl < 1 ? 1 : nbl) : 512);   // VALU kernel: 64 pixels per round
    const int ncb = (F.ncol + 31) / 32, nrb = F.Cout / 32;
    const int wb = dbias ? 1 : 0;
    if (dbias && (!flat_uses_mfma(F) || (F.ncol % 32) == 0)) return DVSOF_EINVAL;
    if (flat_uses_mfma(F)) {
        // 8 waves per workgroup, >= PPW pixel pairs per wave, at most 512 partial sums
        static const int ppw = getenv("DVSOF_FLAT_PPW") ? atoi(getenv("DVSOF_FLAT_PPW")) : 32;
        const long long npair = ((long long)F.M + 1) / 2;
        long long nw = npair / (ppw * 8);
        nw = nw < 1 ? 1 : nw > 512 ? 512 : nw;
        nb = (int)nw;
        if (ncb == 1 && nrb >= 4)
            hipLaunchKernelGGL((wgrad_flat_mfma_kernel<1, 4>), dim3(nb), dim3(512), 0, st, F, part, wb);
        else if (ncb == 1 && nrb >= 2)
            hipLaunchKernelGGL((wgrad_flat_mfma_kernel<1, 2>), dim3(nb), dim3(512), 0, st, F, part, wb);
        else if (ncb == 1)
            hipLaunchKernelGGL((wgrad_flat_mfma_kernel<1, 1>), dim3(nb), dim3(512), 0, st, F, part, wb);
        else if (nrb >= 2)
            hipLaunchKernelGGL((wgrad_flat_mfma_kernel<2, 2>), dim3(nb), dim3(512), 0, st, F, part, wb);
        else
            hipLaunchKernelGGL((wgrad_flat_mfma_kernel<2, 1>), dim3(nb), dim3(512), 0, st, F, part, wb);
    } else
    switch (F.ncol) {
    case 18: hipLaunchKernelGGL(wgrad_flat_kernel<18>, dim3(nb), dim3(256), 0, st, F, part); break;
    case 27: hipLaunchKernelGGL(wgrad_flat_kernel<27>, dim3(nb), dim3(256), 0, st, F, part); break;
    case 45: hipLaunchKernelGGL(wgrad_flat_kernel<45>, dim3(nb), dim3(256), 0, st, F, part); break;
    case 81: hipLaunchKernelGGL(wgrad_flat_kernel<81>, dim3(nb), dim3(256), 0, st, F, part); break;
    case 108: hipLaunchKernelGGL(wgrad_flat_kernel<108>, dim3(nb), dim3(256), 0, st, F, part); break;
    default: return DVSOF_EINVAL;
    }
    DVSOF_LAUNCH_CHECK();
    const int taps = F.ks * F.ks;
    hipLaunchKernelGGL(wgrad_flat_reduce_kernel, dim3((F.Cout * (F.ncol + wb) + 3) / 4), dim3(256), 0,
                       st, (const float *)part, nb, F.Cout, F.ncol, F.S.C, F.coff,
                       taps * F.Cin_tot, F.Cin_tot, dW, F.ncol + wb, dbias);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// column tiles of the members selected by `want_flat` (others get none)
int enumerate_tiles(WGradParams &P, int bn, int want_flat /* -1: all */)
{
    const int taps = P.ks * P.ks;
    int t = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        P.tile_begin[s] = t;
        if (want_flat < 0 || (P.src[s].flat != 0) == (want_flat != 0))
            t += (taps * P.src[s].C + bn - 1) / bn;
    }
    P.tile_begin[P.nsrc] = t;
    return t;
}

template <int WROWS, int WCOLS, int TM, int TN>
int launch(WGradParams &P, int want_flat, hipStream_t st)
{
    constexpr int BMc = WROWS * TM * 32, BN = WCOLS * TN * 32;
    const int t = enumerate_tiles(P, BN, want_flat);
    if (t == 0) return DVSOF_OK;
    dim3 grid(t, (P.Cout + BMc - 1) / BMc, P.S * P.nph);
    hipLaunchKernelGGL((wgrad_kernel<WROWS, WCOLS, TM, TN>), grid, dim3(CONV_NT), 0, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

void tile_dims(int tile, int &bm, int &bn)
{
    switch (tile) {
    case 1: bm = 128; bn = 128; break;
    case 2: bm = 128; bn = 64; break;
    case 3: bm = 64; bn = 64; break;
    case 4: bm = 64; bn = 128; break;
    default: bm = 32; bn = 128; break;
    }
}

// (tile, K splits) by a small occupancy model.  A CU runs `slots` workgroups of a
// tile at once (LDS ring footprint); workgroups are dealt round-robin to 256
// CUs, so the launch ends when the fullest CU has worked through its k blocks:
//   g(k) = full rounds of `slots` blocks + the last partial round, where a lone
//   block on a CU only reaches ~80 % of the matrix rate.
// Cost = (K steps per block + fixed prologue/epilogue) * g * step time / tile
// efficiency + the slab write/read when there is more than one slab.
// (Residual layers, 144 tiles of 128x128: S=4 -> 576 blocks on 512 slots ran
// 123 us, S=3 -> 432 blocks 105 us.)
int pick_tile_and_splits(const WGradParams &P, int *S_out)
{
    const int taps = P.ks * P.ks;
    static const int cand[5] = {1, 4, 3, 5, 2};
    static const int slots_of[6] = {0, 2, 3, 5, 3, 3};          // by tile id
    // (64x64 is 2.6-3 % faster than 64x128 on the three wide decoder layers one at a time,
    // tools/wgrad_sweep.sh, but beside the data-gradient stream the step then alternates
    // between 2340 and 2470 samples/s from run to run; 64x128 gives a steady 2445)
    static double eff_of[6] = {0, 0.85, 0.80, 0.70, 0.80, 0.65};
    static bool eff_init = false;
    if (!eff_init) {   // tuning: DVSOF_WGRAD_EFF="e1,e2,e3,e4,e5"
        eff_init = true;
        if (const char *e = getenv("DVSOF_WGRAD_EFF"))
            sscanf(e, "%lf,%lf,%lf,%lf,%lf", &eff_of[1], &eff_of[2], &eff_of[3], &eff_of[4], &eff_of[5]);
    }
    int best = -1, bestS = 1;
    double best_cost = 1e300;
    const long long ksteps = (P.M + BK - 1) / BK;
    for (int i = 0; i < 5; ++i) {
        int bm, bn;
        tile_dims(cand[i], bm, bn);
        if (bm > 32 && bm / 2 >= P.Cout) continue;      // mostly padding rows
        long long tiles = 0;
        for (int s = 0; s < P.nsrc; ++s)
            if (!P.src[s].flat) tiles += (taps * P.src[s].C + bn - 1) / bn;
        if (tiles == 0)
            for (int s = 0; s < P.nsrc; ++s) tiles += (taps * P.src[s].C + bn - 1) / bn;
        const long long rows = (P.Cout + bm - 1) / bm;
        tiles *= rows * P.nph;
        const int slots = slots_of[cand[i]];
        const double step_us = 2.0 * bm * bn * BK / (157.3e12 / 256) * 1e6 / eff_of[cand[i]];
        int maxS = (int)((P.M + 511) / 512);              // >= 32 K steps per split
        if (maxS > 64) maxS = 64;
        if (maxS < 1) maxS = 1;
        for (int S = 1; S <= maxS; ++S) {
            const long long blocks = tiles * S;
            const long long kmax = (blocks + 255) / 256;  // blocks on the fullest CU
            const long long full = kmax / slots, r = kmax % slots;
            const double g = (double)full * slots + (r == 1 ? 1.25 : (double)r);
            const double steps = (double)((ksteps + S - 1) / S) + 4.0;
            const double slab = (S * P.nph > 1)
                                    ? 2.0 * S * P.nph * P.Cout * taps * (double)P.Cin_tot * 4.0 / 4e6
                                    : 0.0;                 // us at ~4 TB/s
            const double cost = steps * g * step_us + slab;
            if (cost < best_cost) {
                best_cost = cost;
                best = cand[i];
                bestS = S;
            }
        }
    }
    if (S_out) *S_out = bestS;
    return best;
}

}  // namespace

// wgrad_patch.hip: the decoder stages in the bf16-twins mode, input patch resident in LDS
bool wgrad_patch_shape_ok(const WGradParams &P);
bool wgrad_patch_eligible(const WGradParams &P);
int wgrad_patch_splits(const WGradParams &P);
int wgrad_patch_launch(const WGradParams &P, hipStream_t st);
void conv_note_patch(int kind, int what);     // conv_api.hip
bool wgrad_min_ok(const WGradParams &P);  // wgrad_min.hip

// Number of K splits used for this problem (deterministic in the shape).
int wgrad_splits(const WGradParams &P0, int *tile_out)
{
    int S = 1;
    int tile = pick_tile_and_splits(P0, &S);
    if (wgrad_patch_eligible(P0)) {     // its own split rule (workgroups are K splits there)
        if (tile_out) *tile_out = tile;
        return wgrad_patch_splits(P0);
    }
    // tuning sweeps (tools/wgrad_sweep.sh): force the tile and / or the K splits
    static const int tile_env = getenv("DVSOF_WGRAD_TILE") ? atoi(getenv("DVSOF_WGRAD_TILE")) : 0;
    static const int s_env = getenv("DVSOF_WGRAD_SPLITS") ? atoi(getenv("DVSOF_WGRAD_SPLITS")) : 0;
    if (tile_env >= 1 && tile_env <= 5) tile = tile_env;
    if (s_env >= 1) {
        const long long cap = (P0.M + BK - 1) / BK;
        S = s_env > 64 ? 64 : s_env;
        if (S > cap) S = (int)cap;
    }
    if (tile_out) *tile_out = tile;
    return S;
}

// Internal entry: ws holds the nph*S slabs when needed, results go to dW/dbias.
// In phase mode (nph = 4) dW is the folded [Cout][3][3][Cin_tot] gradient.
int wgrad_launch(WGradParams P, float *dW, float *dbias, float *ws, size_t ws_floats,
                 const FlatWG *flat, int nflat, hipStream_t st)
{
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && (P.src[s].sc != 1 || (P.src[s].C & 3))) return DVSOF_EINVAL;
    int tile;
    const int S = wgrad_splits(P, &tile);
    P.S = S;
    const int nslab = S * P.nph;
    const size_t wsize = (size_t)P.Cout * P.ks * P.ks * P.Cin_tot;
    const bool direct = nslab == 1;
    const size_t need = (direct ? 0 : (size_t)nslab * wsize) +
                        (dbias ? (size_t)COLSUM_MAX_BLOCKS * P.Cout : 0);
    if (need + wgrad_flat_workspace_floats(flat, nflat) > ws_floats) return DVSOF_ENOSPACE;
    P.klen = (((P.M + S - 1) / S) + BK - 1) / BK * BK;
    P.dW = direct ? dW : ws;
    P.dbias = nullptr;
    float *bias_part = ws + (direct ? 0 : (size_t)nslab * wsize);   // [nslab][Cout] or colsum partials
    static const bool force_v1 = getenv("DVSOF_WGRAD_V1") != nullptr;
    // flat members: dedicated VALU kernel when every one of them qualifies
    bool flat_valu = nflat > 0 && !force_v1;
    for (int i = 0; i < nflat; ++i) flat_valu = flat_valu && flat_ncol_ok(flat[i].ncol);
    bool any_vec = false;
    for (int s = 0; s < P.nsrc; ++s) any_vec = any_vec || !P.src[s].flat;
    int want_flat = -1;  // v1 MFMA tiles for everything ...
    int rc = DVSOF_OK;
    bool bias_in_kernel = false, patch_folded = false;
    if (!force_v1 && wgrad2_eligible(P)) {  // ... or v2 for the vector members
        int bm, bn;
        tile_dims(tile, bm, bn);
        const int nt = enumerate_tiles(P, bn, 0);
        if (nt > 0) {
            // v2 also leaves the bias gradient: per-slab column sums of gout
            if (dbias) {
                P.dbias = direct ? dbias : bias_part;
                bias_in_kernel = true;
            }
            // (flat members on the v1 tiles would write phase-form columns into the same slabs)
            if (!direct && wgrad_patch_eligible(P) && (flat_valu || nflat == 0)) {
                conv_note_patch(2, (!P.twins && P.mfma_bf16 == 0 && wgrad_min_ok(P)) ? 2 : 1);
                patch_folded = true;    // its slabs are [S][Cout][3][3][Cin_tot] already
                rc = wgrad_patch_launch(P, st);
            } else {
                rc = wgrad2_launch(P, tile, nt, st);
            }
            P.dbias = nullptr;
        }
        if (rc) return rc;
        want_flat = 1;
    } else if (flat_valu) {
        want_flat = 0;   // v1 for the vector members only
    }
    const bool run_v1 = !(want_flat == 1 && (flat_valu || nflat == 0)) && (want_flat != 0 || any_vec);
    if (run_v1) {
        switch (tile) {
        case 1: rc = launch<2, 2, 2, 2>(P, want_flat, st); break;
        case 2: rc = launch<2, 2, 2, 1>(P, want_flat, st); break;
        case 3: rc = launch<2, 2, 1, 1>(P, want_flat, st); break;
        case 4: rc = launch<2, 2, 1, 2>(P, want_flat, st); break;
        default: rc = launch<1, 4, 1, 1>(P, want_flat, st); break;
        }
        if (rc) return rc;
    }
    // flat-only layer on the matrix-core flat kernel: the bias gradient is one more output
    // column of that kernel (no pass over gout of its own)
    const bool bias_by_flat = dbias && !bias_in_kernel && flat_valu && nflat >= 1 &&
                              flat_uses_mfma(flat[0]) && (flat[0].ncol % 32) != 0;
    if (dbias && !bias_in_kernel && !bias_by_flat) {  // column sums of gout (all phases cover gout exactly once)
        const long long rows = (long long)P.B * (P.g_sb / P.Cout);
        const int nb = colsum_blocks(rows);
        if (P.Cout & 3)
            hipLaunchKernelGGL(colsum_scalar_kernel, dim3(nb), dim3(256), 0, st, P.gout, rows, P.Cout,
                               bias_part);
        else
            hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, st, P.gout, rows, P.Cout,
                               bias_part);
        DVSOF_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((P.Cout + 3) / 4)), dim3(256), 0, st,
                           (const float *)bias_part, nb, P.Cout, dbias);
        DVSOF_LAUNCH_CHECK();
    }
    // the bias partials of the slabs ride along with the slab reduce / fold
    const bool bias_tail_needed = bias_in_kernel && !direct;
    const int nb_bias = bias_tail_needed ? (P.Cout + 255) / 256 : 0;
    if (!direct && (any_vec || !flat_valu)) {
        if (patch_folded) {
            const size_t w9 = (size_t)P.Cout * 9 * P.Cin_tot;
            const int nbm = (int)((w9 + 1023) / 1024);
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(nbm + nb_bias)), dim3(256), 0, st,
                               (const float *)ws, dW, w9, S, nbm, (const float *)bias_part, nslab,
                               P.Cout, dbias);
        } else if (P.nph == 4) {
            const size_t n = (size_t)P.Cout * 9 * ((P.Cin_tot + 3) / 4);
            const int zg = S <= 2 ? 1 : S <= 8 ? 4 : 16;
            const int nbm = (int)((n + 256 / zg - 1) / (256 / zg));
            auto kern = zg == 1 ? subpixel_fold_kernel<1> : zg == 4 ? subpixel_fold_kernel<4>
                                                                    : subpixel_fold_kernel<16>;
            hipLaunchKernelGGL(kern, dim3((unsigned)(nbm + nb_bias)), dim3(256), 0, st,
                               (const float *)ws, dW, P.Cout, P.Cin_tot, S, nbm,
                               (const float *)bias_part, nslab, dbias);
        } else {
            const int nbm = (int)((wsize + 1023) / 1024);
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(nbm + nb_bias)), dim3(256), 0, st,
                               (const float *)ws, dW, wsize, S, nbm, (const float *)bias_part, nslab,
                               P.Cout, dbias);
        }
        DVSOF_LAUNCH_CHECK();
    } else if (bias_tail_needed) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)nb_bias), dim3(256), 0, st,
                           (const float *)nullptr, (float *)nullptr, (size_t)0, 0, 0,
                           (const float *)bias_part, nslab, P.Cout, dbias);
        DVSOF_LAUNCH_CHECK();
    }
    if (flat_valu) {   // overwrites the flat members' columns of dW
        float *part = bias_part + (dbias ? (size_t)COLSUM_MAX_BLOCKS * P.Cout : 0);
        for (int i = 0; i < nflat; ++i) {
            rc = flat_launch(flat[i], part, dW, (i == 0 && bias_by_flat) ? dbias : nullptr, st);
            if (rc) return rc;
            part += (size_t)FLAT_BLOCKS * flat[i].Cout * (flat[i].ncol + 1);
        }
    }
    return DVSOF_OK;
}

size_t wgrad_flat_workspace_floats(const FlatWG *flat, int nflat)
{
    size_t n = 0;
    // (+ 1: room for the bias column of the matrix-core kernel's partial rows)
    for (int i = 0; i < nflat; ++i) n += (size_t)FLAT_BLOCKS * flat[i].Cout * (flat[i].ncol + 1);
    return n;
}

size_t wgrad_workspace_floats(const WGradParams &P, bool with_bias)
{
    int S = wgrad_splits(P, nullptr);
    // the twins may not be bound yet when the workspace is sized: room for either kernel
    if (wgrad_patch_shape_ok(P) && wgrad_patch_splits(P) > S) S = wgrad_patch_splits(P);
    const int nslab = S * P.nph;
    return (nslab <= 1 ? 0 : (size_t)nslab * P.Cout * P.ks * P.ks * P.Cin_tot) +
           (with_bias ? (size_t)COLSUM_MAX_BLOCKS * P.Cout : 0);
}
