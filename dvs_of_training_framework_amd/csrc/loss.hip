// Multi-scale flow-warp photometric + Charbonnier smoothness + out-of-border
// loss, forward and analytic backward, all scales in one launch.
//
// Replaces (reference paths): utils/loss.py:20-21 (bilinear resize),
// :24-35 (Charbonnier), :58-74 (warp + photometric), :76-90 (smoothness),
// :92-119 (out-of-border), :121-171 (Loss.__call__), and the autograd
// backward that utils/training.py:158 runs through them.
//
// HBM-bound kernels, three launches per evaluation (A, B and the one-wave C):
//   A  loss_pyramid_kernel: the cascaded frame pyramid; its tail workgroups
//      count out-of-border pixels per tile (plain stores, no atomics, nothing
//      to zero) and workgroup 0 clears the group accumulators of launch B;
//   B  loss_main_kernel: all scales, forward sums and flow gradients in one
//      sweep;
//   C  loss_reduce_kernel: one workgroup (a wave per scale) combines the per-group sums.
// Layout: flow [N,2,h,w] and frames [D,h,w] row-major.  A 256-thread
// workgroup owns a 64x16 pixel tile of one sample at one scale: wave v owns
// rows 4v..4v+3, lane l column l, so a thread holds a 4-pixel column strip.
// The flow tile + 1 px halo is staged in LDS; the 4 bilinear taps of the
// second frame are gathered straight from L2/HBM.
// Smoothness: every neighbour pair is evaluated ONCE, at its anchor pixel
// (the 4 pairs ->, v, \, / of utils/loss.py:76-90), value and derivative from
// one log2 + one exp2; the derivative reaches the pair's other pixel through
// registers (same column), a lane shuffle (column to the left) or a 12-entry
// LDS row per wave (left tile edge): 19 evaluations per channel and 4-pixel
// strip instead of 36.
// Reduction: a workgroup adds its 7 sums into its (scale, sample) group's
// 64-bit FIXED-POINT accumulators (2^-20; integer atomics, no return: the
// wave does not wait for them) -- integer addition commutes, so the group
// totals are bitwise reproducible whatever the arrival order.  A one-wave
// launch (C) then combines the groups, applies the reference's normalisers
// and writes the terms: the kernel boundary orders it behind the atomics, so
// there is no in-kernel publish / observe protocol at all.
#include "common.h"
#include <stdlib.h>

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
    float rcp_half_w, rcp_half_h;  // correctly rounded reciprocals (0 when the side is 1)
    float k_smooth[3];           // 1/(4*count) for ->, v, diagonal crops
    float k_photo;               // 1/(N*h*w)
    double c_smooth[3];          // crop element counts (utils/loss.py:77-85)
};

struct Params {
    ScaleDev s[DVSOF_MAX_SCALES];
    int K, N;
    const int32_t *start, *stop;
    int32_t *oob;               // [K*N] out-of-border pixels per (scale, sample)
    int32_t *oob_tile;          // [nb] the same per tile (launch A) or null: use oob
    unsigned long long *gacc;   // [K*N][NGROUP] group sums, 2^-20 fixed point (two's complement)
    const float *seeds_dev;     // [3*K] or null
    float seeds_host[3];        // used when seeds_dev == null
    float *terms, *loss_out;    // outputs of loss_reduce_kernel
    float wts[3], loss_scale;
    int dbg;                    // DVSOF_LOSS_DBG timing probes (results wrong by construction)
};

__device__ __forceinline__ int find_scale(const Params &P, int bid)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < DVSOF_MAX_SCALES; ++i)
        if (i < P.K && bid >= P.s[i].block_begin) k = i;
    return k;
}

// a / b correctly rounded from r = RN(1/b): q0 = RN(a r), e = a - b q0 (exact,
// one fma), q = RN(q0 + e r).  Markstein's theorem: correctly rounded whenever
// r is the correctly rounded reciprocal and b's significand is not all ones,
// barring over/underflow -- b = (size-1)/2 of an image side here, |a| < 2^24.
// The compiler's IEEE expansion of `/` is ~10 instructions (div_scale x2, rcp,
// 4 fma, div_fmas, div_fixup); this is 3, with the same bits.
__device__ __forceinline__ float div_exact(float a, float b, float r)
{
    const float q0 = a * r;
    const float e = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e, r, q0);
}

// utils/loss.py:150-156 in fp32, op for op (IEEE division).
__device__ __forceinline__ void warp_grid(const ScaleDev &S, int x, int y, float u,
                                          float v, float &gx, float &gy)
{
    gx = div_exact((float)x + u, S.half_w, S.rcp_half_w) - 1.f;
    gy = div_exact((float)y + v, S.half_h, S.rcp_half_h) - 1.f;
}
__device__ __forceinline__ bool out_of_border(float gx, float gy)
{  // utils/loss.py:92-94 (strict)
    return (gx < -1.f) | (gx > 1.f) | (gy < -1.f) | (gy > 1.f);
}

constexpr int NGROUP = 8;   // accumulators per group: 7 sums + pad
constexpr int NW = NT / kWave;
constexpr float FIX_SCALE = 1048576.f;              // 2^20
constexpr double FIX_INV = 1.0 / 1048576.0;

// A workgroup's sum into its group's accumulator: agent-scope integer RMW, no
// return (the wave does not wait for it).  The reader is the next launch.
__device__ __forceinline__ void accumulate(unsigned long long *p, float v)
{
    // v >= 0 here (sums of rho values, counts); two's complement keeps the door open
    const long long q = (long long)__builtin_rintf(v * FIX_SCALE);
    __hip_atomic_fetch_add(p, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double group_value(const unsigned long long *p)
{
    return (double)(long long)*p * FIX_INV;
}

// Every group is complete.  One workgroup combines them, wave v the scales v,
// v + 4: a group's 8 accumulators are 64 contiguous bytes, so lane l reads
// element l % 8 of sample 8 j + l / 8 -- 8 samples per load instruction, the
// loads of 64 samples in flight together (ONE memory round trip per wave up to
// batch 64; round 2's single wave walked the scales and 32-sample blocks in
// turn: 9-14 us for 40 KB, a third of the whole path at batch 8) -- lanes of
// equal l % 8 add their samples in j order and the eight lane groups meet by
// three shuffle steps.  Elements 0..4: sums over the samples of photo + the
// four smoothness directions (sums of exact multiples of 2^-20 in double:
// exact, whatever the order); element 5: border term
// = sum_n bs_n / (2 c_n N) with c_n = element 6 of the same sample
// (utils/loss.py:101,113), in the fixed order above.
__device__ __forceinline__ void final_terms(const Params &P, double (*s_tot)[3])
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, e = lane & 7, g = lane >> 3;
    constexpr int UB = 8;       // sample blocks of 8 per pass: UB loads in flight per lane
    for (int kk = wave; kk < P.K; kk += NW) {
        double acc = 0;
        for (int n0 = 0; n0 < P.N; n0 += 8 * UB) {
            double x[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int nn = min(n0 + 8 * u + g, P.N - 1);
                x[u] = group_value(P.gacc + ((size_t)kk * P.N + nn) * NGROUP + e);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int nn = n0 + 8 * u + g;
                const double c = __shfl(x[u], (lane & ~7) | 6, kWave);    // the sample's count
                if (nn >= P.N) continue;
                if (e < 5) acc += x[u];
                else if (e == 5 && c > 0) acc += x[u] / (2.0 * c * (double)P.N);
                else if (e == 6) P.oob[kk * P.N + nn] = (int)c;    // kept for dvsof_loss_bwd
            }
        }
        acc += __shfl_xor(acc, 8, kWave);
        acc += __shfl_xor(acc, 16, kWave);
        acc += __shfl_xor(acc, 32, kWave);
        double a6[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) a6[j] = __shfl(acc, j, kWave);
        if (lane == 0) {
            const ScaleDev &S = P.s[kk];
            // empty crops contribute 0 (utils/loss.py:29-30)
            const double sm = ((S.c_smooth[0] > 0 ? a6[1] / S.c_smooth[0] : 0) +
                               (S.c_smooth[1] > 0 ? a6[2] / S.c_smooth[1] : 0) +
                               (S.c_smooth[2] > 0 ? (a6[3] + a6[4]) / S.c_smooth[2] : 0)) / 4.0;
            const double ph = a6[0] / ((double)P.N * S.h * S.w);
            const double border = a6[5];
            P.terms[0 * P.K + kk] = (float)sm;
            P.terms[1 * P.K + kk] = (float)ph;
            P.terms[2 * P.K + kk] = (float)border;
            s_tot[kk][0] = sm;
            s_tot[kk][1] = ph;
            s_tot[kk][2] = border;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && P.loss_out) {   // combined_loss, utils/training.py:23
        double total[3] = {0, 0, 0};
        for (int kk = 0; kk < P.K; ++kk)
            for (int i = 0; i < 3; ++i) total[i] += s_tot[kk][i];
        P.loss_out[0] = (float)((P.wts[0] * total[0] + P.wts[1] * total[1] + P.wts[2] * total[2]) /
                                (double)P.K * (double)P.loss_scale);
    }
}

#ifndef DVSOF_LOSS_NO_FENCE
#define LOSS_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define LOSS_FENCE()
#endif
#ifdef DVSOF_LOSS_WPE     // experiment (tools/variant.sh): registers capped for that many waves per SIMD
#define LOSS_MAIN_ATTR __attribute__((amdgpu_waves_per_eu(DVSOF_LOSS_WPE, DVSOF_LOSS_WPE)))
#else
#define LOSS_MAIN_ATTR
#endif
template <bool FWD, bool BWD>
__global__ __launch_bounds__(NT) LOSS_MAIN_ATTR void loss_main_kernel(const Params P)
{
    // flow tile + 1-pixel halo, (u, v) interleaved: a pair is one ds_read_b64 into an aligned
    // register pair (the packed-f32 operand as it stands; two planes cost ~180 v_mov per wave
    // to assemble pairs).  Padded to FILL x NT entries: the fill writes unconditionally.
    constexpr int FILL = (LH * LW + NT - 1) / NT;
    __shared__ f32x2 sF[FILL * NT];
    __shared__ float red[NW][NPART];
    __shared__ int s_cnt[NW];

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int bid = blockIdx.x;
    if (DVSOF_DBG(P) & 128) return;
    const int k = find_scale(P, bid);
    const ScaleDev &S = P.s[k];
    const int local = bid - S.block_begin;
    const int n = local / S.tiles_per_sample;
    const int t = local - n * S.tiles_per_sample;
    const int ty0 = (t / S.tiles_x) * TH, tx0 = (t % S.tiles_x) * TW;
    const int h = S.h, w = S.w;
    const size_t hw = (size_t)h * w;
    const float *U = S.flow + (size_t)n * 2 * hw;

    // flow tile + halo -> LDS.  Fixed trip count, loads first: the 2*FILL loads
    // of a thread are in flight together (a rolled loop makes FILL dependent
    // load -> wait -> ds_write round trips, and a workgroup's life is a chain of
    // such round trips: at batch 8 every workgroup is resident at once and the
    // kernel takes exactly as long as one workgroup does)
    // Out-of-frame entries read as zero through the buffer range check (an offset past the
    // sample's two planes), like the frame taps below: no exec-masked loads, no branches.
    {
        const __amdgpu_buffer_rsrc_t rf =
            __builtin_amdgcn_make_buffer_rsrc((void *)U, 0, (int)(2 * hw * 4), 0x00020000);
        constexpr unsigned OOB = 0xffffffffu;
        float fu[FILL], fv[FILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int i = tid + it * NT;
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
            const bool in = (i < LH * LW) & ((unsigned)gy < (unsigned)h) & ((unsigned)gx < (unsigned)w);
            const unsigned o = (unsigned)((gy * w + gx) * 4);
            fu[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, in ? o : OOB, 0, 0));
            fv[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, in ? o + (unsigned)(hw * 4) : OOB, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < FILL; ++it) sF[tid + it * NT] = f32x2{fu[it], fv[it]};
    }
    auto F2 = [&](int ly, int lx) -> f32x2 { return sF[ly * LW + lx]; };
    const float *I0 = S.frames + (size_t)P.start[n] * hw;
    const float *I1 = S.frames + (size_t)P.stop[n] * hw;
    float seed[3] = {0.f, 0.f, 0.f};
    float k_border = 0.f;
    if (BWD) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            seed[i] = P.seeds_dev ? P.seeds_dev[i * P.K + k] : P.seeds_host[i];
        int cnt;
        if (P.oob_tile) {   // per-tile counts of launch A: integer sum, any order
            int c = 0;
            const int32_t *ot = P.oob_tile + S.block_begin + n * S.tiles_per_sample;
            for (int b = tid; b < S.tiles_per_sample; b += NT) c += ot[b];
            c = wave_sum(c);
            if (lane == 0) s_cnt[wave] = c;
            __syncthreads();
            cnt = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        } else {
            cnt = P.oob[k * P.N + n];
        }
        k_border = cnt > 0 ? seed[2] / (2.f * (float)cnt * (float)P.N) : 0.f;
    }
    __syncthreads();

    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int x = tx0 + lane;
    const int ly0 = wave * 4;            // tile-local row of the strip's first pixel
    constexpr int NP = 4;                // pixels per thread: rows ly0 .. ly0 + 3
    // phase A: addresses and ALL gathers of the thread's pixels first, so the
    // 5*NP loads are in flight together instead of NP dependent round trips
    // Taps outside the frame read as zero (grid_sample's zero padding) through
    // the buffer range check: an invalid tap gets an offset past the frame's
    // byte count, so there are no clamps, no validity factors to keep in
    // registers, no exec-mask branches -- and all 20 loads are in flight.
    bool valid[NP], oobv[NP];
    float axv[NP], ayv[NP], nwv[NP], nev[NP], swv[NP], sev[NP], prv[NP];
    {
        const __amdgpu_buffer_rsrc_t r1 =
            __builtin_amdgcn_make_buffer_rsrc((void *)I1, 0, (int)(hw * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r0 =
            __builtin_amdgcn_make_buffer_rsrc((void *)I0, 0, (int)(hw * 4), 0x00020000);
        constexpr unsigned OOB = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int ly = ly0 + j, y = ty0 + ly;
            valid[j] = (y < h) & (x < w);
            const f32x2 uv = F2(ly + 1, lane + 1);
            const float u = uv.x, v = uv.y;
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
            const bool vx0 = (unsigned)x0 < (unsigned)w, vx1 = (unsigned)(x0 + 1) < (unsigned)w;
            const bool vy0 = valid[j] & ((unsigned)y0 < (unsigned)h), vy1 = valid[j] & ((unsigned)(y0 + 1) < (unsigned)h);
            const unsigned o00 = (unsigned)((y0 * w + x0) * 4), o10 = o00 + (unsigned)(w * 4);
            nwv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (vy0 & vx0) ? o00 : OOB, 0, 0));
            nev[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (vy0 & vx1) ? o00 + 4 : OOB, 0, 0));
            swv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (vy1 & vx0) ? o10 : OOB, 0, 0));
            sev[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (vy1 & vx1) ? o10 + 4 : OOB, 0, 0));
            prv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                r0, valid[j] ? (unsigned)((y * w + x) * 4) : OOB, 0, 0));
            if (DVSOF_DBG(P) & 4) nwv[j] = nev[j] = swv[j] = sev[j] = prv[j] = u;     // probe: keeps the loads dead
        }
    }

    // left-edge column (x = tx0 - 1) derivatives the strip's lane 0 needs:
    // lane 12 c + 4 kind + r (< 24) of every wave evaluates one -- kind 0:
    // d0(r, xe), 1: d2(r-1, xe), 2: d3(r, xe) -- and lane 0 picks them up with
    // v_readlane (no LDS, no barrier).
    float edge = 0.f;
    if (BWD) {
        const int l = lane < 24 ? lane : 0;
        const int c = l / 12, q = l - 12 * c, kind = q >> 2, r = q & 3;
        const int db = kind == 1 ? -1 : (kind == 2 ? 1 : 0);
        const int ya = ty0 + ly0 + r, yb = ya + db;
        const bool ok = (tx0 >= 1) & (ya >= 0) & (ya < h) & (yb >= 0) & (yb < h);
        const f32x2 fa2 = F2(ly0 + r + 1, 1), fb2 = F2(ly0 + r + 1 + db, 0);
        const float fa = c ? fa2.y : fa2.x, fb = c ? fb2.y : fb2.x;
        edge = ok ? charbonnier(fa - fb).der : 0.f;
    }
    const int edge_bits = __builtin_bit_cast(int, edge);
    auto edge_at = [&](int src) -> float {      // uniform value of lane `src`
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(edge_bits, src));
    };

    float gu[NP], gv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) gu[j] = gv[j] = 0.f;
    // Scheduling fences between the phases (gathers | photometric | smoothness rows | tail).
    // As ONE scheduling region the compiler interleaves all of it for instruction-level
    // parallelism: 152 registers, 3 waves per SIMD.  The probe build, whose runtime probe
    // tests happen to cut the region in the same places, needs 80 (6 waves per SIMD) and was
    // 10 % faster at batch 64: this kernel wants occupancy, not a longer in-order window.
    LOSS_FENCE();
    // photometric + out-of-border, utils/loss.py:58-74, 96-119
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (!valid[j] || (DVSOF_DBG(P) & 2)) continue;
        const float ax = axv[j], ay = ayv[j], cx = 1.f - ax, cy = 1.f - ay;
        const float nw = nwv[j], ne = nev[j], sw = swv[j], se = sev[j];
        const float warped = nw * cx * cy + ne * ax * cy + sw * cx * ay + se * ax * ay;
        const Charb ph = charbonnier(warped - prv[j]);
        if (FWD) acc[0] += ph.val;
        if (BWD) {
            const float gp = seed[1] * S.k_photo * ph.der;
            gu[j] = gp * ((ne - nw) * cy + (se - sw) * ay);
            gv[j] = gp * ((sw - nw) * cx + (se - ne) * ax);
        }
        if (oobv[j]) {
            const f32x2 buv = F2(ly0 + j + 1, lane + 1);
            const Charb bu = charbonnier(buv.x), bv = charbonnier(buv.y);
            if (FWD) {
                acc[5] += bu.val + bv.val;
                acc[6] += 1.f;
            }
            if (BWD) {
                gu[j] += k_border * bu.der;
                gv[j] += k_border * bv.der;
            }
        }
    }
    LOSS_FENCE();
    // smoothness, utils/loss.py:76-90.  Anchor (r, x), r = -1..3 relative to the
    // strip: d0 = rho'(F[r][x+1] - F[r][x]), d1 = rho'(F[r+1][x] - F[r][x]),
    // d2 = rho'(F[r+1][x+1] - F[r][x]), d3 = rho'(F[r][x+1] - F[r+1][x]).
    // Pixel (r, x) is the SECOND operand of its own d0, d1, d2, the FIRST of
    // d0(r, x-1), d1(r-1, x), d2(r-1, x-1), d3(r, x-1), the SECOND of d3(r-1, x).
    // Branch-free: the LDS halo is zero-filled outside the frame, so every pair
    // can be evaluated and then multiplied by its 0/1 validity.  The two flow
    // channels of a pair go through the arithmetic together (packed f32).
    // (`h > 0` is always true: a wave-uniform branch the compiler cannot fold makes this
    // section a scheduling region of its own.  Without it the smoothness rows are interleaved
    // with the gathers' waits and the photometric part for instruction-level parallelism:
    // 154 registers, 3 waves per SIMD; with it 77 and 6 -- the sweep is bound by latency and
    // issue slots across waves, not by one wave's in-order window.  sched_barrier fences alone
    // did not move the allocation.)
    if (h > 0 && !(DVSOF_DBG(P) & 1)) {
        const float mx = x < w ? 1.f : 0.f, mxr = x + 1 < w ? 1.f : 0.f;
        float mrow[6];                     // rows -1..4 of the strip inside the frame
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = ty0 + ly0 + r - 1;
            mrow[r] = ((y >= 0) & (y < h)) ? 1.f : 0.f;
        }
        // Rolling window over the anchor rows -1..3: with the anchors of row
        // r and r - 1 in hand, pixel row r is complete -- only two rows of
        // derivatives live in registers (all five: 70 VGPRs more, occupancy 3).
        f32x2 s1 = {0.f, 0.f}, s2 = s1, s3 = s1, s4 = s1;
        f32x2 p0 = F2(ly0, lane + 1);     // row -1: columns x, x + 1
        f32x2 p1 = F2(ly0, lane + 2);
        f32x2 d1p = s1, d2p = s1, d3p = s1;                           // anchor row r - 1
        const float k0 = S.k_smooth[0] * seed[0], k1 = S.k_smooth[1] * seed[0],
                    k2 = S.k_smooth[2] * seed[0];
#pragma unroll
        for (int r = 0; r < 5; ++r) {          // anchor strip row r - 1
            const f32x2 n0 = F2(ly0 + r + 1, lane + 1);
            const f32x2 n1 = F2(ly0 + r + 1, lane + 2);
            const bool own = r >= 1;           // anchor row of this strip: sums count
            const float m0 = mrow[r] * mxr, mv = mrow[r] * mrow[r + 1];
            const float m1 = mv * mx, m2 = mv * mxr;
            f32x2 d0c = {0.f, 0.f};
            if (own) {
                const Charb2 q = charbonnier2(p1 - p0);
                if (FWD) s1 += q.val * m0;
                d0c = q.der * m0;
            }
            const Charb2 q1 = charbonnier2(n0 - p0);
            const Charb2 q2 = charbonnier2(n1 - p0);
            const Charb2 q3 = charbonnier2(p1 - n0);
            if (FWD && own) {
                s2 += q1.val * m1;
                s3 += q2.val * m2;
                s4 += q3.val * m2;
            }
            const f32x2 d1c = q1.der * m1, d2c = q2.der * m2, d3c = q3.der * m2;
            if (BWD && own) {
                const int j = r - 1;
                // from the column to the left: lane - 1 by a DPP wave shift (wave_shr:1; __shfl_up
                // is a ds_bpermute through the LDS crossbar); lane 0 keeps the shift's `old`
                // operand = the edge column's value (no select, no branch)
                auto from_left = [&](float v, int src) -> float {
                    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        __builtin_bit_cast(int, edge_at(src)), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
                };
                const f32x2 in0 = {from_left(d0c.x, j), from_left(d0c.y, 12 + j)};
                const f32x2 in2 = {from_left(d2p.x, 4 + j), from_left(d2p.y, 16 + j)};
                const f32x2 in3 = {from_left(d3c.x, 8 + j), from_left(d3c.y, 20 + j)};
                const f32x2 g = (in0 - d0c) * k0 + (d1p - d1c) * k1 + ((in2 - d2c) + (in3 - d3p)) * k2;
                gu[j] += g.x;
                gv[j] += g.y;
            }
            d1p = d1c;
            d2p = d2c;
            d3p = d3c;
            p0 = n0;
            p1 = n1;
            LOSS_FENCE();
        }
        if (FWD) {
            acc[1] = s1.x + s1.y;
            acc[2] = s2.x + s2.y;
            acc[3] = s3.x + s3.y;
            acc[4] = s4.x + s4.y;
        }
    }
    LOSS_FENCE();
    // ---- tail.  Waves 1..3 store their gradients and RETIRE; wave 0 alone
    // adds the workgroup's sums to its group's accumulators and, on small
    // grids, bumps the arrival counter, stores its own gradients while that
    // atomic is in flight, and combines everything if it was the last.
    auto store_grads = [&]() {
        if (!BWD || (DVSOF_DBG(P) & 8)) return;
        // pixels outside the frame get an offset past the sample's two planes: the buffer
        // range check drops the store (no exec-masked branches around 8 stores)
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(S.grad + (size_t)n * 2 * hw), 0, (int)(2 * hw * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const unsigned o = valid[j] ? (unsigned)(((ty0 + ly0 + j) * w + x) * 4) : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gu[j]), rg, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gv[j]), rg,
                                                  valid[j] ? o + (unsigned)(hw * 4) : 0xffffffffu, 0, 0);
        }
    };
    if (!FWD || (DVSOF_DBG(P) & 16)) {
        store_grads();
        return;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float v = (DVSOF_DBG(P) & 32) ? acc[i] : wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    if (DVSOF_DBG(P) & 64) {
        store_grads();
        return;
    }
    __syncthreads();
    if (wave != 0) {
        store_grads();
        return;
    }
    // 7 integer atomics (no return) and the wave is done: the combination of
    // the groups is loss_reduce_kernel's, ordered by the kernel boundary
    const int grp = k * P.N + n;
    if (lane < 7)
        accumulate(P.gacc + (size_t)grp * NGROUP + lane,
                   (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    store_grads();
}

// The final combination: one wave, a launch of its own.  (Folded into the
// sweep -- last workgroup to arrive combines -- it saved ~1 us at batch 8 and
// cost 30 us at batch 64: an arrival needs a store acknowledgement plus an
// atomic round trip per workgroup, a CU cannot start the next workgroup while
// its SIMD-0 slots are held by waves waiting for it, and the inlined
// combination took the sweep from 96 to 160 VGPRs, 5 -> 3 waves per SIMD.)
__global__ __launch_bounds__(NT) void loss_reduce_kernel(const Params P)
{
    __shared__ double s_tot[DVSOF_MAX_SCALES][3];
    final_terms(P, s_tot);
}

// Per-tile out-of-border pixel counts (utils/loss.py:101) ahead of the fused
// forward+backward sweep: plain stores, one int per tile.  Workgroup 0 also
// clears the arrival counters of the main kernel that follows in the stream.
__device__ __forceinline__ void count_oob_block(const Params &P, int bid, int *red)
{
    const int tid = threadIdx.x;
    if (bid == 0 && P.gacc)      // group accumulators of the sweep that follows
        for (int i = tid; i < P.K * P.N * NGROUP; i += NT) P.gacc[i] = 0ull;
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
    // all eight loads first (clamped addresses), then the arithmetic
    float uu[TH / 4], vv[TH / 4];
    const int xc = min(x, w - 1);
#pragma unroll
    for (int j = 0; j < TH / 4; ++j) {
        const int o = min(ty0 + tr + 4 * j, h - 1) * w + xc;
        uu[j] = U[o];
        vv[j] = U[hw + o];
    }
#pragma unroll
    for (int j = 0; j < TH / 4; ++j) {
        const int y = ty0 + tr + 4 * j;
        float gx, gy;
        warp_grid(S, x, y, uu[j], vv[j], gx, gy);
        cnt += ((y < h) & (x < w) & out_of_border(gx, gy)) ? 1 : 0;
    }
    cnt = wave_sum(cnt);
    if ((tid & (kWave - 1)) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) P.oob_tile[bid] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(NT) void loss_count_oob_kernel(const Params P)
{
    __shared__ int red[NT / kWave];
    count_oob_block(P, blockIdx.x, red);
}

// ---------------------------------------------------------------------------
// The whole cascaded frame pyramid (utils/loss.py:207-210: level k resamples
// level k-1, level 0 resamples the input frames) in ONE launch.  A workgroup
// owns a 16x64 tile of the finest level of one frame and recomputes, level by
// level in LDS, the (small) regions of the coarser levels that tile depends
// on -- the arithmetic per pixel is exactly resize_bilinear_ac_kernel's, so the
// values are bitwise those of K dependent launches.  Coarser-level pixels are
// written by the first tile (in y, then x) whose region contains them.
// Needs non-decreasing level sizes: then every pixel of level k-1 is a tap of
// some pixel of level k, so the regions of all tiles cover each level.
// Trailing workgroups of the same launch count out-of-border pixels.
// ---------------------------------------------------------------------------
constexpr int PYR_MAXR = 1536;   // floats per staged region

struct PyrParams {
    const float *src;
    float *lev[DVSOF_MAX_SCALES];
    int h[DVSOF_MAX_SCALES], w[DVSOF_MAX_SCALES];
    float sh[DVSOF_MAX_SCALES], sw[DVSOF_MAX_SCALES];   // input step per output pixel
    int K, D, H, W;
    int tiles_x, tiles_per_frame, nblocks;
    int big;        // 1: 32 x 128 tiles of the finest level (many frames), else 16 x 64
};

__device__ __forceinline__ int pyr_lo(float s, int o) { return (int)(s * (float)o); }
__device__ __forceinline__ int pyr_hi(float s, int o, int nin)
{
    const int y0 = min((int)(s * (float)o), nin - 1);
    return y0 + (y0 < nin - 1 ? 1 : 0);
}

// PTH x PTW: the tile of the finest level a workgroup owns.  16 x 64 at the benchmark batch
// (1 024 workgroups); 32 x 128 when there are many frames: a workgroup's life is a chain of K
// dependent phases behind barriers, and at batch 64 the 8 192 small workgroups were bound by
// that chain (30-35 us for 45 MB written) -- four times the pixels per chain, and the finest
// level leaves as 16-byte stores (4 consecutive pixels per thread; the arithmetic per pixel is
// unchanged, so the values are still bitwise those of K dependent launches).
template <int PTH, int PTW>
__global__ __launch_bounds__(NT) void loss_pyramid_kernel(const PyrParams Q, const Params P,
                                                          const int do_count)
{
    __shared__ float buf[2][PYR_MAXR];
    __shared__ int red[NT / kWave];
    if ((int)blockIdx.x >= Q.nblocks) {
        if (do_count && !(DVSOF_DBG(P) & 256)) count_oob_block(P, blockIdx.x - Q.nblocks, red);
        return;
    }
    if (DVSOF_DBG(P) & 512) return;
    const int tid = threadIdx.x, K = Q.K;
    const int d = blockIdx.x / Q.tiles_per_frame;
    const int t = blockIdx.x - d * Q.tiles_per_frame;
    const int ty = t / Q.tiles_x, tx = t - ty * Q.tiles_x;
    // regions [y0,y1] x [x0,x1] per level, and the previous tile's region end.
    // All loops are unrolled over DVSOF_MAX_SCALES with static indices (runtime
    // indexing would put these arrays in scratch memory).
    int ry0[DVSOF_MAX_SCALES], ry1[DVSOF_MAX_SCALES], rx0[DVSOF_MAX_SCALES], rx1[DVSOF_MAX_SCALES];
    int pey[DVSOF_MAX_SCALES], pex[DVSOF_MAX_SCALES];   // -1: no previous tile
#pragma unroll
    for (int k = DVSOF_MAX_SCALES - 1; k >= 0; --k) {
        ry0[k] = ry1[k] = rx0[k] = rx1[k] = 0;
        pey[k] = pex[k] = -1;
        if (k == K - 1) {
            ry0[k] = ty * PTH;
            ry1[k] = min(ry0[k] + PTH - 1, Q.h[k] - 1);
            rx0[k] = tx * PTW;
            rx1[k] = min(rx0[k] + PTW - 1, Q.w[k] - 1);
            pey[k] = ty * PTH - 1;
            pex[k] = tx * PTW - 1;
        } else if (k < K - 1) {
            ry0[k] = pyr_lo(Q.sh[k + 1], ry0[k + 1]);
            ry1[k] = pyr_hi(Q.sh[k + 1], ry1[k + 1], Q.h[k]);
            rx0[k] = pyr_lo(Q.sw[k + 1], rx0[k + 1]);
            rx1[k] = pyr_hi(Q.sw[k + 1], rx1[k + 1], Q.w[k]);
            pey[k] = pey[k + 1] < 0 ? -1 : pyr_hi(Q.sh[k + 1], pey[k + 1], Q.h[k]);
            pex[k] = pex[k + 1] < 0 ? -1 : pyr_hi(Q.sw[k + 1], pex[k + 1], Q.w[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < DVSOF_MAX_SCALES; ++k) {
        if (k >= K) break;
        const int km = k > 0 ? k - 1 : 0;
        const int hin = k ? Q.h[km] : Q.H, win = k ? Q.w[km] : Q.W;
        const int rh = ry1[k] - ry0[k] + 1, rw = rx1[k] - rx0[k] + 1;
        const float *in = k ? buf[km & 1] : Q.src + (size_t)d * Q.H * Q.W;
        const int iy0 = k ? ry0[km] : 0, ix0 = k ? rx0[km] : 0;
        const int ipitch = k ? rx1[km] - rx0[km] + 1 : Q.W;
        float *cur = buf[k & 1];
        float *out = Q.lev[k] + (size_t)d * Q.h[k] * Q.w[k];
        const float sh = Q.sh[k], sw = Q.sw[k];
        if (k == K - 1 && (rw & 3) == 0 && (Q.w[k] & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            // the finest level: 4 consecutive pixels per thread, one 16-byte store (its region
            // starts at a multiple of PTW, every pixel of it is this tile's to write)
            const int rw4 = rw >> 2;
            for (int i = tid; i < rh * rw4; i += NT) {
                const int ly = i / rw4, lx = (i - ly * rw4) * 4;
                const int y = ry0[k] + ly, x = rx0[k] + lx;
                const float fy = sh * (float)y;
                const int y0 = min((int)fy, hin - 1);
                const int y1 = y0 + (y0 < hin - 1 ? 1 : 0);
                const float wy = fy - (float)y0, hy = 1.f - wy;
                const float *r0 = in + (size_t)(y0 - iy0) * ipitch - ix0;
                const float *r1 = in + (size_t)(y1 - iy0) * ipitch - ix0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float fx = sw * (float)(x + e);
                    const int x0 = min((int)fx, win - 1);
                    const int x1 = x0 + (x0 < win - 1 ? 1 : 0);
                    const float wx = fx - (float)x0, hx = 1.f - wx;
                    v[e] = hy * (hx * r0[x0] + wx * r0[x1]) + wy * (hx * r1[x0] + wx * r1[x1]);
                }
                *(float4 *)(out + (size_t)y * Q.w[k] + x) = make_float4(v[0], v[1], v[2], v[3]);
            }
            break;
        }
        for (int i = tid; i < rh * rw; i += NT) {
            const int ly = i / rw, lx = i - ly * rw;
            const int y = ry0[k] + ly, x = rx0[k] + lx;
            const float fy = sh * (float)y, fx = sw * (float)x;
            const int y0 = min((int)fy, hin - 1), x0 = min((int)fx, win - 1);
            const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
            const float wy = fy - (float)y0, wx = fx - (float)x0, hy = 1.f - wy, hx = 1.f - wx;
            const float *r0 = in + (size_t)(y0 - iy0) * ipitch - ix0;
            const float *r1 = in + (size_t)(y1 - iy0) * ipitch - ix0;
            const float v = hy * (hx * r0[x0] + wx * r0[x1]) + wy * (hx * r1[x0] + wx * r1[x1]);
            if (k + 1 < K) cur[i] = v;
            if (y > pey[k] && x > pex[k]) out[(size_t)y * Q.w[k] + x] = v;
        }
        __syncthreads();
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
        // RN(1/b) via double: the double quotient is within 2^-53 relative of
        // 1/b and b = m/2 with m < 2^24 an integer, so no double-rounding tie
        S.rcp_half_w = w > 1 ? (float)(1.0 / (double)S.half_w) : 0.f;
        S.rcp_half_h = h > 1 ? (float)(1.0 / (double)S.half_h) : 0.f;
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

// workspace: [gacc K*N*NGROUP u64][oob_tile nb i32]
size_t ws_layout(int nb, int K, int N, size_t &tile_off)
{
    size_t o = (size_t)K * N * NGROUP * sizeof(unsigned long long);
    tile_off = o;
    o += (size_t)nb * sizeof(int32_t);
    return ((o + 15) & ~(size_t)15) + 16;
}

void bind_ws(Params &P, void *ws, int nb, float *terms, float *loss_out, const float *w,
             float loss_scale)
{
    size_t t;
    ws_layout(nb, P.K, P.N, t);
    P.gacc = (unsigned long long *)ws;
    P.oob_tile = (int32_t *)((char *)ws + t);
    P.terms = terms;
    P.loss_out = loss_out;
    for (int i = 0; i < 3; ++i) P.wts[i] = w ? w[i] : 0.f;
    P.loss_scale = loss_scale;
#ifdef DVSOF_PROBES
    static const int dbg = getenv("DVSOF_LOSS_DBG") ? atoi(getenv("DVSOF_LOSS_DBG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    P.dbg = dbg;
}

// Pyramid plan: fused single launch when the level sizes are non-decreasing
// and every staged region fits PYR_MAXR; else one resize launch per level.
bool pyramid_plan(int D, int H, int W, float *const *levels, const int *hs, const int *ws, int K,
                  const float *images, PyrParams &Q)
{
    if (K < 1 || K > DVSOF_MAX_SCALES) return false;
    Q.src = images;
    Q.K = K;
    Q.D = D;
    Q.H = H;
    Q.W = W;
    for (int k = 0; k < K; ++k) {
        const int hin = k ? hs[k - 1] : H, win = k ? ws[k - 1] : W;
        Q.lev[k] = levels[k];
        Q.h[k] = hs[k];
        Q.w[k] = ws[k];
        Q.sh[k] = hs[k] > 1 ? (float)(hin - 1) / (float)(hs[k] - 1) : 0.f;
        Q.sw[k] = ws[k] > 1 ? (float)(win - 1) / (float)(ws[k] - 1) : 0.f;
        if (k && (hs[k] < hs[k - 1] || ws[k] < ws[k - 1])) return false;
    }
    // conservative bound of the region sizes of a th x tw tile of the finest level
    auto fits = [&](int th, int tw) {
        double rh = th, rw = tw;
        for (int k = K - 1; k > 0; --k) {
            rh = rh * Q.sh[k] + 3;
            rw = rw * Q.sw[k] + 3;
            if (rh > hs[k - 1]) rh = hs[k - 1];
            if (rw > ws[k - 1]) rw = ws[k - 1];
            if (rh * rw > PYR_MAXR) return false;
        }
        return true;
    };
    if (!fits(TH, TW)) return false;
    // big tiles once the small ones would be more than 8 workgroups per CU
    // (DVSOF_PYR_BIG=0|1 forces the choice)
    static const int force = getenv("DVSOF_PYR_BIG") ? atoi(getenv("DVSOF_PYR_BIG")) : -1;
    const long long small = (long long)D * ((ws[K - 1] + TW - 1) / TW) * ((hs[K - 1] + TH - 1) / TH);
    Q.big = (force == 1 || (force < 0 && small >= 2048)) && fits(2 * TH, 2 * TW) ? 1 : 0;
    const int th = Q.big ? 2 * TH : TH, tw = Q.big ? 2 * TW : TW;
    Q.tiles_x = (ws[K - 1] + tw - 1) / tw;
    Q.tiles_per_frame = Q.tiles_x * ((hs[K - 1] + th - 1) / th);
    const long long nb = (long long)D * Q.tiles_per_frame;
    if (nb > 0x3fffffff) return false;
    Q.nblocks = (int)nb;
    return true;
}

int resize_launch(const float *src, float *dst, int n, int hin, int win, int hout, int wout,
                  hipStream_t st)
{
    const float sh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
    const float sw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
    dim3 grid((wout + 63) / 64, (hout + 3) / 4, n);
    hipLaunchKernelGGL(resize_bilinear_ac_kernel, grid, dim3(NT), 0, st, src, dst, hin, win, hout,
                       wout, sh, sw);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// pyramid (+ optional out-of-border count of `P`, nb_count workgroups) -> launches
int pyramid_launch(const float *images, int D, int H, int W, float *const *levels, const int *hs,
                   const int *ws, int K, const Params *P, int nb_count, hipStream_t st)
{
    if (!images || !levels || !hs || !ws || D < 0 || H < 1 || W < 1 || K < 1 ||
        K > DVSOF_MAX_SCALES)
        return DVSOF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!levels[k] || hs[k] < 1 || ws[k] < 1) return DVSOF_EINVAL;
    static const bool no_fuse = getenv("DVSOF_LOSS_NO_FUSED_PYRAMID") != nullptr;
    PyrParams Q = {};
    if (D > 0 && !no_fuse && pyramid_plan(D, H, W, levels, hs, ws, K, images, Q)) {
        Params dummy = {};
        if (Q.big)
            hipLaunchKernelGGL((loss_pyramid_kernel<2 * TH, 2 * TW>), dim3(Q.nblocks + (P ? nb_count : 0)),
                               dim3(NT), 0, st, Q, P ? *P : dummy, P ? 1 : 0);
        else
            hipLaunchKernelGGL((loss_pyramid_kernel<TH, TW>), dim3(Q.nblocks + (P ? nb_count : 0)),
                               dim3(NT), 0, st, Q, P ? *P : dummy, P ? 1 : 0);
        DVSOF_LAUNCH_CHECK();
        return DVSOF_OK;
    }
    if (D > 65535) return DVSOF_EINVAL;
    for (int k = 0; k < K && D > 0; ++k) {
        const int rc = resize_launch(k ? levels[k - 1] : images, levels[k], D, k ? hs[k - 1] : H,
                                     k ? ws[k - 1] : W, hs[k], ws[k], st);
        if (rc) return rc;
    }
    if (P) {
        hipLaunchKernelGGL(loss_count_oob_kernel, dim3(nb_count), dim3(NT), 0, st, *P);
        DVSOF_LAUNCH_CHECK();
    }
    return DVSOF_OK;
}

}  // namespace

extern "C" {

size_t dvsof_loss_workspace_bytes(const dvsof_loss_scale_t *sc, int K, int N)
{
    Params P = {};
    int nb = 0;
    if (build_params(sc, K, N, P, nb) != DVSOF_OK) return 0;
    size_t t;
    return ws_layout(nb, K, N, t);
}

int dvsof_resize_bilinear_ac(const float *src, float *dst, int n, int hin, int win, int hout,
                             int wout, void *stream)
{
    if (!src || !dst || n < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    if (n > 65535) return DVSOF_EINVAL;
    return resize_launch(src, dst, n, hin, win, hout, wout, as_stream(stream));
}

int dvsof_loss_pyramid(const float *images, int D, int H, int W, float *const *levels,
                       const int *hs, const int *ws, int K, void *stream)
{
    return pyramid_launch(images, D, H, W, levels, hs, ws, K, nullptr, 0, as_stream(stream));
}

int dvsof_loss_fwd(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                   const int32_t *stop, float *terms, int32_t *oob, void *ws, size_t ws_bytes,
                   void *stream)
{
    Params P = {};
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !terms || !oob || !ws) return DVSOF_EINVAL;
    size_t t;
    if (ws_bytes < ws_layout(nb, K, N, t)) return DVSOF_ENOSPACE;
    P.start = start;
    P.stop = stop;
    bind_ws(P, ws, nb, terms, nullptr, nullptr, 1.f);
    P.oob = oob;
    P.seeds_dev = nullptr;
    // launch A (per-tile counts are not needed by a forward-only sweep, but the
    // same kernel clears the arrival counters), then the sweep with the folded
    // reduction, which also writes oob[k*N + n] for dvsof_loss_bwd
    hipLaunchKernelGGL(loss_count_oob_kernel, dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL((loss_main_kernel<true, false>), dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_loss_bwd(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                   const int32_t *stop, const float *seeds, const int32_t *oob, void *stream)
{
    Params P = {};
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !seeds || !oob) return DVSOF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!sc[k].grad_flow) return DVSOF_EINVAL;
    P.start = start;
    P.stop = stop;
    P.gacc = nullptr;
    P.oob_tile = nullptr;       // totals of the forward call
    P.terms = P.loss_out = nullptr;
    P.oob = const_cast<int32_t *>(oob);
    P.seeds_dev = seeds;
    hipLaunchKernelGGL((loss_main_kernel<false, true>), dim3(nb), dim3(NT), 0, as_stream(stream), P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

namespace {
// images != NULL: the frame pyramid is built into sc[k].frames first (same
// launch as the out-of-border count)
int fused_impl(const float *images, int D, int H, int W, const dvsof_loss_scale_t *sc, int K, int N,
               const int32_t *start, const int32_t *stop, const float *w, float loss_scale,
               float *terms, float *loss_out, int32_t *oob, void *ws, size_t ws_bytes,
               hipStream_t st)
{
    Params P = {};
    int nb = 0;
    const int rc = build_params(sc, K, N, P, nb);
    if (rc) return rc;
    if (!start || !stop || !w || !terms || !loss_out || !oob || !ws) return DVSOF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!sc[k].grad_flow) return DVSOF_EINVAL;
    size_t t;
    if (ws_bytes < ws_layout(nb, K, N, t)) return DVSOF_ENOSPACE;
    P.start = start;
    P.stop = stop;
    bind_ws(P, ws, nb, terms, loss_out, w, loss_scale);
    P.oob = oob;
    P.seeds_dev = nullptr;
    for (int i = 0; i < 3; ++i) P.seeds_host[i] = w[i] / (float)K * loss_scale;
    if (images) {
        float *levels[DVSOF_MAX_SCALES];
        int hs[DVSOF_MAX_SCALES], wss[DVSOF_MAX_SCALES];
        for (int k = 0; k < K; ++k) {
            levels[k] = const_cast<float *>(sc[k].frames);
            hs[k] = sc[k].h;
            wss[k] = sc[k].w;
        }
        const int prc = pyramid_launch(images, D, H, W, levels, hs, wss, K, &P, nb, st);
        if (prc) return prc;
    } else {
        hipLaunchKernelGGL(loss_count_oob_kernel, dim3(nb), dim3(NT), 0, st, P);
        DVSOF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((loss_main_kernel<true, true>), dim3(nb), dim3(NT), 0, st, P);
    DVSOF_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(NT), 0, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
}  // namespace

int dvsof_loss_fused(const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                     const int32_t *stop, const float *w, float loss_scale, float *terms,
                     float *loss_out, int32_t *oob, void *ws, size_t ws_bytes, void *stream)
{
    return fused_impl(nullptr, 0, 0, 0, sc, K, N, start, stop, w, loss_scale, terms, loss_out, oob, ws,
                      ws_bytes, as_stream(stream));
}

int dvsof_loss_fused_pyramid(const float *images, int D, int H, int W,
                             const dvsof_loss_scale_t *sc, int K, int N, const int32_t *start,
                             const int32_t *stop, const float *w, float loss_scale, float *terms,
                             float *loss_out, int32_t *oob, void *ws, size_t ws_bytes, void *stream)
{
    if (!images || D < 1 || H < 1 || W < 1) return DVSOF_EINVAL;
    return fused_impl(images, D, H, W, sc, K, N, start, stop, w, loss_scale, terms, loss_out, oob, ws,
                      ws_bytes, as_stream(stream));
}

}  // extern "C"
