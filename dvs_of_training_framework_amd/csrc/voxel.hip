// Event stream -> per-pixel count image and polarity-signed voxel grid.
//
// Replaces (reference paths): get_count_image utils/data.py:120-136 and the
// EV_FlowNet quantization layer as called at utils/training.py:59-64 /
// scripts/quantize_preprocessed.py:87-91 (source absent upstream; arithmetic
// per docs/VOXEL_SPEC.md, restated on the CPU in oracle/dvsof_oracle.c).
//
// HBM-bound integer/byte work.  v1: one thread per event, coalesced reads of
// the five event columns, scatter with memory-side atomics into the grid that
// was zero-filled on the same stream (by a kernel: fill_u32, common.h).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void count_image_kernel(const int64_t *__restrict__ x,
                                                         const int64_t *__restrict__ y,
                                                         int64_t n, int H, int W,
                                                         uint32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t xi = x[i], yi = y[i];
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) atomicAdd(&out[yi * W + xi], 1u);
    }
}

__global__ __launch_bounds__(NT) void voxelize_kernel(
    const int64_t *__restrict__ x, const int64_t *__restrict__ y, const float *__restrict__ t,
    const int64_t *__restrict__ pol, const int64_t *__restrict__ sample, int64_t n,
    const float *__restrict__ t0, const float *__restrict__ t1, int B, int C, int H, int W,
    float *__restrict__ out, int32_t *__restrict__ bin0, int64_t *__restrict__ lin0)
{
    const int64_t stride = (int64_t)gridDim.x * NT;
    const size_t plane = (size_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int64_t b = sample[i], xi = x[i], yi = y[i];
        int c0 = -1;
        int64_t lin = -1;
        if (b >= 0 && b < B && xi >= 0 && xi < W && yi >= 0 && yi < H) {
            const float ts = t[i], lo = t0[b], hi = t1[b];
            if (ts >= lo && ts <= hi) {
                const float dt = hi - lo;
                const float tn = dt > 0.f ? ((ts - lo) / dt) * (float)(C - 1) : 0.f;
                c0 = min((int)floorf(tn), C - 1);
                const float f = tn - (float)c0;
                // polarity is a SIGN (VOXEL_SPEC): 0 contributes nothing, |p| > 1 counts once
                const int64_t pv = pol[i];
                const float p = pv > 0 ? 1.f : (pv < 0 ? -1.f : 0.f);
                lin = (int64_t)((((size_t)b * C + c0) * H + (size_t)yi) * W + (size_t)xi);
                if (pv != 0) {
                    atomicAdd(&out[lin], p * (1.f - f));
                    if (c0 + 1 < C) atomicAdd(&out[lin + plane], p * f);
                }
            }
        }
        if (bin0) bin0[i] = c0;
        if (lin0) lin0[i] = lin;
    }
}

int grid_for(int64_t n) { return (int)((n + NT - 1) / NT < 2048 ? (n + NT - 1) / NT : 2048); }

}  // namespace

extern "C" {

int dvsof_count_image(const int64_t *x, const int64_t *y, int64_t n, int H, int W, uint32_t *out,
                      void *stream)
{
    if (!out || H < 1 || W < 1 || n < 0 || (n > 0 && (!x || !y))) return DVSOF_EINVAL;
    DVSOF_HIP_TRY((hipError_t)fill_u32(out, 0u, sizeof(uint32_t) * (size_t)H * W, as_stream(stream)));
    if (n == 0) return DVSOF_OK;
    hipLaunchKernelGGL(count_image_kernel, dim3(grid_for(n)), dim3(NT), 0, as_stream(stream), x, y,
                       n, H, W, out);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_voxelize_fwd(const int64_t *x, const int64_t *y, const float *t, const int64_t *pol,
                       const int64_t *sample, int64_t n, const float *t0, const float *t1, int B,
                       int C, int H, int W, float *out, int32_t *bin0, int64_t *lin0, void *stream)
{
    if (!out || B < 1 || C < 1 || H < 1 || W < 1 || n < 0 || !t0 || !t1) return DVSOF_EINVAL;
    if (n > 0 && (!x || !y || !t || !pol || !sample)) return DVSOF_EINVAL;
    DVSOF_HIP_TRY((hipError_t)fill_u32(out, 0u, sizeof(float) * (size_t)B * C * H * W, as_stream(stream)));
    if (n == 0) return DVSOF_OK;
    hipLaunchKernelGGL(voxelize_kernel, dim3(grid_for(n)), dim3(NT), 0, as_stream(stream), x, y, t,
                       pol, sample, n, t0, t1, B, C, H, W, out, bin0, lin0);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // extern "C"

// ===========================================================================
// v2: LDS-staged voxel tiles, TWO launches, no memset, no zero-fill pass.
//   pass 1 (vox_bucket_kernel): coalesced read of the event columns (int64
//     wire columns or the 9 B/event compact columns); every workgroup ranks
//     its events per tile in an LDS histogram (cleared and scanned only over
//     the tile range its events touch), reserves a range in each touched
//     tile's bucket with ONE global atomic per (workgroup, tile) and writes
//     8-byte records {local pixel | lower bin | sign, fraction (exact f32)}.
//     Buckets have a fixed capacity (2x the mean + slack); events that do not
//     fit go to an overflow list (none for well-spread data).
//   pass 2 (vox_tile_kernel): one workgroup per (sample, tile of 1024 or 512
//     pixels): zero the [C][tile] accumulators in LDS, add the bucket's
//     records, then the overflow records that belong to this tile (the list
//     is empty unless the events pile up in a few tiles), store the tile with
//     coalesced rows.  No global float atomics at all -- and no LDS float
//     atomics either: ds_add_f32 measured ~150 cycles per wave instruction on
//     gfx950 (33 of the 59 us of this pass at batch 64), integer LDS atomics
//     are free next to the loads.  The accumulators are 64-bit FIXED POINT
//     (2^-32): an event adds sign * (2^32 - F) and sign * F, F = trunc(f * 2^32)
//     exactly, so a voxel is the exact sum of its weights to 2^-32 per event,
//     the two halves of an event add up to exactly +-1, and the result does
//     not depend on the order of additions: the tiled path is bitwise
//     reproducible.
//   The control words (bucket cursors, overflow count, finished-tile count)
//   are SELF-CLEANING: every tile workgroup zeroes its cursor after reading
//   it, the last one to finish zeroes the two counters.  A workspace whose
//   control region was zero before a call is zero again after it; the caller
//   says so with DVSOF_VOX_WS_CLEAN and no memset is enqueued.
// Integer parts (bin0 / lin0) are computed exactly as in v1 (bit-exact vs the
// oracle); float sums differ from v1 only in accumulation order.
// ===========================================================================
namespace {

// A tile is 2^lp pixels (lp = 10, or 9 when many bins would not leave room for
// several workgroups' 8-byte accumulators in LDS), 2^lx wide and 2^(lp-lx) high;
// lx is chosen per call (v2_plan): as wide as the frame allows, so that a
// workgroup stores long contiguous runs of every output row it owns.
constexpr int VPX_MAX = 1024;
// events per thread in pass 1: 4 up to ~2 M events, 8 above, 16 from ~3 M (measured, whole
// path, EPT 4 / 8 / 16: 16.8 / 19.2 / 24.7 us at 0.5 M events, 79.0 / 76.4 / 71.3 us at 4.2 M,
// 105.9 / 82.5 / 72.6 us at 4 M events on 512 x 512 x 12)
constexpr int64_t EPT8_FROM = 1 << 21;
// 16 from ~4 M events: half the bucket-cursor atomics, twice as long record runs per tile
constexpr int64_t EPT16_FROM = 3 << 20;
constexpr int V2_MAX_TILES = 8192;    // LDS histogram + base of the bucket pass: 8 bytes per tile (64 KiB)

struct VoxV2 {
    // wire format (int64 columns) ...
    const int64_t *x, *y, *pol, *sample;
    // ... or the reference's encoded columns (utils/dataset.py:286-289):
    // int16 x, int16 y, bool polarity; the sample of event i is found in
    // ev_off[B+1] (first event of every sample)
    const int16_t *x16, *y16;
    const uint8_t *p8;
    const int64_t *ev_off;
    int enc;
    const float *t, *t0, *t1;
    int64_t n;
    int B, C, H, W, TX, TY, ntile, cap, lx, lp;   // lx = log2(tile width), lp = log2(tile pixels)
    int32_t *cursor;      // [ntile] events reserved per tile (may exceed cap)
    int32_t *ovf_count;   // [1] overflow records
    int32_t *ovf_tiles;   // [1] tiles whose bucket overflowed
    int32_t *done;        // [1] of those, finished
    uint2 *records;       // [ntile][cap]
    int4 *ovf;            // [n]  {tile, key, bits(frac), 0}
    int64_t ovf_cap;
    float *out;
    int32_t *bin0;
    int64_t *lin0;
};

template <int EPT>
__global__ __launch_bounds__(NT) void vox_bucket_kernel(const VoxV2 P)
{
    extern __shared__ int sh[];          // hist[ntile], base[ntile]
    __shared__ int s_lo, s_hi;
    int *hist = sh, *base = sh + P.ntile;
    if (threadIdx.x == 0) {
        s_lo = P.ntile;
        s_hi = -1;
    }
    __syncthreads();
    const int64_t e0 = ((int64_t)blockIdx.x * NT) * EPT + threadIdx.x;
    int tile[EPT], rank[EPT];
    unsigned key[EPT];
    float frac[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int64_t i = e0 + (int64_t)k * NT;   // coalesced across the workgroup
        tile[k] = -1;
        if (i < P.n) {
            int64_t b, xi, yi;
            bool neg, zero = false;     // polarity is a sign; 0 contributes nothing
            if (P.enc) {
                xi = P.x16[i];
                yi = P.y16[i];
                neg = P.p8[i] == 0;
                int lo = 0, hi = P.B;          // last sample with ev_off <= i
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (P.ev_off[mid] <= i) lo = mid; else hi = mid;
                }
                b = lo;
            } else {
                b = P.sample[i];
                xi = P.x[i];
                yi = P.y[i];
                const int64_t pv = P.pol[i];
                neg = pv < 0;
                zero = pv == 0;
            }
            int c0 = -1;
            int64_t l = -1;
            if (b >= 0 && b < P.B && xi >= 0 && xi < P.W && yi >= 0 && yi < P.H) {
                const float ts = P.t[i], lo = P.t0[b], hi = P.t1[b];
                if (ts >= lo && ts <= hi) {
                    const float dt = hi - lo;
                    const float tn = dt > 0.f ? ((ts - lo) / dt) * (float)(P.C - 1) : 0.f;
                    c0 = min((int)floorf(tn), P.C - 1);
                    frac[k] = tn - (float)c0;
                    l = (int64_t)((((size_t)b * P.C + c0) * P.H + (size_t)yi) * P.W + (size_t)xi);
                    const int ty = (int)yi >> (P.lp - P.lx), tx = (int)xi >> P.lx;
                    if (!zero) tile[k] = ((int)b * P.TY + ty) * P.TX + tx;
                    key[k] = (unsigned)((((int)yi - (ty << (P.lp - P.lx))) << P.lx) + ((int)xi - (tx << P.lx))) |
                             ((unsigned)c0 << 10) | (neg ? 0x80000000u : 0u);
                }
            }
            if (P.bin0) P.bin0[i] = c0;
            if (P.lin0) P.lin0[i] = l;
        }
    }
    // tile range of this workgroup's events (events arrive grouped by sample,
    // so this is one or two samples' tiles out of ntile)
    {
        int lo = P.ntile, hi = -1;
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (tile[k] >= 0) {
                lo = min(lo, tile[k]);
                hi = max(hi, tile[k]);
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off, kWave));
            hi = max(hi, __shfl_xor(hi, off, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0 && hi >= 0) {
            atomicMin(&s_lo, lo);
            atomicMax(&s_hi, hi);
        }
    }
    __syncthreads();
    const int t_lo = s_lo, t_hi = s_hi;
    for (int i = t_lo + (int)threadIdx.x; i <= t_hi; i += NT) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (tile[k] >= 0) rank[k] = atomicAdd(&hist[tile[k]], 1);
    __syncthreads();
    for (int i = t_lo + (int)threadIdx.x; i <= t_hi; i += NT) {
        const int c = hist[i];
        base[i] = c ? atomicAdd(&P.cursor[i], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        if (tile[k] < 0) continue;
        const int pos = base[tile[k]] + rank[k];
        if (pos < P.cap) {
            P.records[(size_t)tile[k] * P.cap + pos] = make_uint2(key[k], __float_as_uint(frac[k]));
        } else {
            if (pos == P.cap) atomicAdd(P.ovf_tiles, 1);     // exactly one event per full bucket
            const int o = atomicAdd(P.ovf_count, 1);
            if (o < P.ovf_cap) P.ovf[o] = make_int4(tile[k], (int)key[k], __float_as_int(frac[k]), 0);
        }
    }
}

// one event into the tile's 2^-32 fixed-point accumulators (integer LDS atomics)
__device__ __forceinline__ void tile_add(unsigned long long *tl, unsigned key, float f, int C, int lp)
{
    const int pix = key & 0x3ff, c0 = (key >> 10) & 0x3ff;
    const unsigned long long F = (unsigned long long)(unsigned)(f * 4294967296.f);   // f in [0,1): exact
    const unsigned long long w1 = F, w0 = 4294967296ull - F;
    const bool neg = key & 0x80000000u;
    // two's complement: adding (0 - w) subtracts
    atomicAdd(&tl[(c0 << lp) + pix], neg ? 0ull - w0 : w0);
    if (c0 + 1 < C) atomicAdd(&tl[((c0 + 1) << lp) + pix], neg ? 0ull - w1 : w1);
}

__global__ __launch_bounds__(NT) void vox_tile_kernel(const VoxV2 P)
{
    extern __shared__ unsigned long long tl[];   // [C][2^lp] fixed-point accumulators
    const int tile = blockIdx.x;
    const int tx = tile % P.TX, ty = (tile / P.TX) % P.TY, b = tile / (P.TX * P.TY);
    const int nel = P.C << P.lp;
    // One memory round trip instead of two: the first 4*NT records are fetched
    // SPECULATIVELY (the bucket's address does not depend on its fill count)
    // together with the count, while the LDS tile is being zeroed.
    const uint2 *rec = P.records + (size_t)tile * P.cap;
    const int reserved = P.cursor[tile];
    uint2 r0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = u * NT + (int)threadIdx.x;
        r0[u] = i < P.cap ? rec[i] : make_uint2(0u, 0u);
    }
    for (int i = threadIdx.x * 2; i < nel; i += NT * 2) *(ulonglong2 *)(tl + i) = make_ulonglong2(0ull, 0ull);
    const int cnt = min(reserved, P.cap);
    __syncthreads();
    if (threadIdx.x == 0) P.cursor[tile] = 0;       // self-cleaning control words
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u * NT + (int)threadIdx.x < cnt) tile_add(tl, r0[u].x, __uint_as_float(r0[u].y), P.C, P.lp);
    // the rest, 4 loads in flight per thread (a rolled loop is one dependent
    // memory round trip per iteration)
    for (int i0 = 4 * NT; i0 < cnt; i0 += 4 * NT) {
        uint2 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT + (int)threadIdx.x;
            r[u] = i < cnt ? rec[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * NT + (int)threadIdx.x < cnt) tile_add(tl, r[u].x, __uint_as_float(r[u].y), P.C, P.lp);
    }
    // events that did not fit this tile's bucket (skewed inputs only): they
    // are on the overflow list among those of the other full buckets
    const bool spilled = reserved > P.cap;
    if (spilled) {
        const int64_t novf = min((int64_t)*P.ovf_count, P.ovf_cap);
        for (int64_t i = threadIdx.x; i < novf; i += NT) {
            const int4 r = P.ovf[i];
            if (r.x == tile) tile_add(tl, (unsigned)r.y, __int_as_float(r.z), P.C, P.lp);
        }
    }
    __syncthreads();
    // 16 bytes per lane along a tile row: full 128-byte lines, long runs for wide tiles
    const int y0 = ty << (P.lp - P.lx), x0 = tx << P.lx;
    const bool vec = (P.W & 3) == 0 && ((uintptr_t)P.out & 15) == 0;
    for (int i = threadIdx.x * 4; i < nel; i += NT * 4) {
        const int c = i >> P.lp, r = i - (c << P.lp), ly = r >> P.lx, lx = r - (ly << P.lx);
        const int y = y0 + ly, x = x0 + lx;
        if (y >= P.H || x >= P.W) continue;
        float *o = P.out + (((size_t)b * P.C + c) * P.H + y) * P.W + x;
        float4 v;
        // signed 2^-32 fixed point -> the correctly rounded float of the exact sum
        v.x = (float)((double)(long long)tl[i] * 2.3283064365386963e-10);
        v.y = (float)((double)(long long)tl[i + 1] * 2.3283064365386963e-10);
        v.z = (float)((double)(long long)tl[i + 2] * 2.3283064365386963e-10);
        v.w = (float)((double)(long long)tl[i + 3] * 2.3283064365386963e-10);
        if (vec) {          // W % 4 == 0 and x % 4 == 0: the quad is inside the row
            *(float4 *)o = v;
        } else {
            o[0] = v.x;
            if (x + 1 < P.W) o[1] = v.y;
            if (x + 2 < P.W) o[2] = v.z;
            if (x + 3 < P.W) o[3] = v.w;
        }
    }
    // the last SPILLED tile to finish (every one of them has read the list by
    // then) clears the overflow words; well-spread inputs never get here
    if (spilled && threadIdx.x == 0 && atomicAdd(P.done, 1) == *P.ovf_tiles - 1) {
        *P.ovf_count = 0;
        *P.ovf_tiles = 0;
        *P.done = 0;
    }
}

bool v2_plan(int64_t n, int B, int C, int H, int W, VoxV2 &P)
{
    // tile: 1024 pixels, or 512 when C 8-byte accumulators per pixel would leave
    // fewer than three workgroups per CU (160 KiB LDS)
    const int lp = (size_t)C * 1024 * 8 > 52 * 1024 ? 9 : 10;
    P.lp = lp;
    // tile width 2^lx, 64 <= 2^lx <= 2^lp: the widest one whose column padding
    // (TX * 2^lx - W) stays within an eighth of the frame, else the one with
    // the least padding (640 -> 128, 346 -> 128, 256 -> 256, 512 -> 512)
    static const int lx_env = getenv("DVSOF_VOX_TILE_LOG2X") ? atoi(getenv("DVSOF_VOX_TILE_LOG2X")) : 0;
    int lx = 6, best_pad = 1 << 30;
    for (int c = 6; c <= lp; ++c) {
        const int wd = 1 << c, padded = (W + wd - 1) / wd * wd;
        if (padded * 8 <= W * 9) {
            lx = c;             // within 12.5 %: wider is better
            best_pad = 0;
        } else if (best_pad && padded - W < best_pad) {
            best_pad = padded - W;
            lx = c;
        }
    }
    if (lx_env >= 2 && lx_env <= lp) lx = lx_env;
    P.lx = lx;
    const int vtx = 1 << lx, vty = 1 << (lp - lx);
    P.TX = (W + vtx - 1) / vtx;
    P.TY = (H + vty - 1) / vty;
    const int64_t nt = (int64_t)B * P.TX * P.TY;
    if (nt > V2_MAX_TILES || ((size_t)C << lp) * 8 > 150 * 1024 || C > 1023) return false;
    P.ntile = (int)nt;
    int64_t cap = 2 * (n / nt) + 256;
    cap = (cap + 63) / 64 * 64;
    if (cap > (1 << 24)) return false;
    P.cap = (int)cap;
    P.ovf_cap = n;
    return true;
}

size_t v2_control_bytes(const VoxV2 &P) { return (((size_t)P.ntile + 3) * 4 + 255) / 256 * 256; }

size_t v2_bytes(const VoxV2 &P, int64_t n)
{
    return v2_control_bytes(P) + (size_t)P.ntile * P.cap * 8 + (size_t)n * 16 + 256;
}

// control words first (that is the region DVSOF_VOX_WS_CLEAN speaks about)
void v2_bind(VoxV2 &P, void *workspace)
{
    unsigned char *w = (unsigned char *)workspace;
    P.cursor = (int32_t *)w;
    P.ovf_count = P.cursor + P.ntile;
    P.ovf_tiles = P.ovf_count + 1;
    P.done = P.ovf_count + 2;
    w += v2_control_bytes(P);
    P.records = (uint2 *)w;
    w += (size_t)P.ntile * P.cap * 8;
    P.ovf = (int4 *)(((uintptr_t)w + 15) & ~(uintptr_t)15);
}

int v2_launch(const VoxV2 &P, int flags, hipStream_t st)
{
    if (!(flags & DVSOF_VOX_WS_CLEAN))
        DVSOF_HIP_TRY((hipError_t)fill_u32(P.cursor, 0u, v2_control_bytes(P), st));
    const int64_t n = P.n;
    static const int ept_env = getenv("DVSOF_VOX_EPT") ? atoi(getenv("DVSOF_VOX_EPT")) : 0;
    if (ept_env == 16 || (ept_env == 0 && n >= EPT16_FROM))
        hipLaunchKernelGGL(vox_bucket_kernel<16>, dim3((unsigned)((n + NT * 16 - 1) / (NT * 16))), dim3(NT),
                           (size_t)P.ntile * 8, st, P);
    else if (ept_env == 8 || (ept_env == 0 && n >= EPT8_FROM))
        hipLaunchKernelGGL(vox_bucket_kernel<8>, dim3((unsigned)((n + NT * 8 - 1) / (NT * 8))), dim3(NT),
                           (size_t)P.ntile * 8, st, P);
    else
        hipLaunchKernelGGL(vox_bucket_kernel<4>, dim3((unsigned)((n + NT * 4 - 1) / (NT * 4))), dim3(NT),
                           (size_t)P.ntile * 8, st, P);
    DVSOF_LAUNCH_CHECK();
    const size_t tile_lds = ((size_t)P.C << P.lp) * 8;
    static bool attr_set = false;
    if (tile_lds > 64 * 1024 && !attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)vox_tile_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(vox_tile_kernel, dim3(P.ntile), dim3(NT), tile_lds, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

}  // namespace

namespace {
// encoded columns -> wire columns (fallback path of dvsof_voxelize_encoded)
__global__ __launch_bounds__(NT) void expand_encoded_kernel(
    const int16_t *__restrict__ x16, const int16_t *__restrict__ y16,
    const uint8_t *__restrict__ p8, const int64_t *__restrict__ ev_off, int B, int64_t n,
    int64_t *__restrict__ x, int64_t *__restrict__ y, int64_t *__restrict__ pol,
    int64_t *__restrict__ sample)
{
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        x[i] = x16[i];
        y[i] = y16[i];
        pol[i] = p8[i] ? 1 : -1;
        int lo = 0, hi = B;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ev_off[mid] <= i) lo = mid; else hi = mid;
        }
        sample[i] = lo;
    }
}
}  // namespace

extern "C" {

size_t dvsof_voxelize_workspace_bytes(int64_t n_events, int B, int C, int H, int W)
{
    VoxV2 P = {};
    // small inputs take the v1 kernel; the encoded entry then expands the
    // columns into the workspace (4 int64 columns)
    if (n_events < 4096 || !v2_plan(n_events, B, C, H, W, P)) return (size_t)n_events * 32 + 64;
    return v2_bytes(P, n_events);
}

size_t dvsof_voxelize_control_bytes(int64_t n_events, int B, int C, int H, int W)
{
    VoxV2 P = {};
    if (n_events < 4096 || !v2_plan(n_events, B, C, H, W, P)) return 0;
    return v2_control_bytes(P);
}

int dvsof_voxelize_tiled(const int64_t *x, const int64_t *y, const float *t, const int64_t *pol,
                         const int64_t *sample, int64_t n, const float *t0, const float *t1, int B,
                         int C, int H, int W, float *out, int32_t *bin0, int64_t *lin0,
                         void *workspace, size_t workspace_bytes, int flags, void *stream)
{
    VoxV2 P = {};
    if (n < 4096 || !workspace || !v2_plan(n, B, C, H, W, P) || workspace_bytes < v2_bytes(P, n))
        return dvsof_voxelize_fwd(x, y, t, pol, sample, n, t0, t1, B, C, H, W, out, bin0, lin0, stream);
    if (!out || !t0 || !t1 || !x || !y || !t || !pol || !sample) return DVSOF_EINVAL;
    v2_bind(P, workspace);
    P.enc = 0;
    P.x16 = P.y16 = nullptr;
    P.p8 = nullptr;
    P.ev_off = nullptr;
    P.x = x; P.y = y; P.pol = pol; P.sample = sample; P.t = t; P.t0 = t0; P.t1 = t1;
    P.n = n; P.B = B; P.C = C; P.H = H; P.W = W;
    P.out = out; P.bin0 = bin0; P.lin0 = lin0;
    return v2_launch(P, flags, as_stream(stream));
}

int dvsof_voxelize_encoded(const int16_t *x, const int16_t *y, const float *t,
                           const uint8_t *polarity, const int64_t *sample_event_offsets, int64_t n,
                           const float *t0, const float *t1, int B, int C, int H, int W, float *out,
                           int32_t *bin0, int64_t *lin0, void *workspace, size_t workspace_bytes,
                           int flags, void *stream)
{
    if (!out || B < 1 || C < 1 || H < 1 || W < 1 || n < 0 || !t0 || !t1) return DVSOF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        DVSOF_HIP_TRY((hipError_t)fill_u32(out, 0u, sizeof(float) * (size_t)B * C * H * W, st));
        return DVSOF_OK;
    }
    if (!x || !y || !t || !polarity || !sample_event_offsets || !workspace) return DVSOF_EINVAL;
    VoxV2 P = {};
    if (n < 4096 || !v2_plan(n, B, C, H, W, P) || workspace_bytes < v2_bytes(P, n)) {
        if (workspace_bytes < (size_t)n * 32) return DVSOF_ENOSPACE;
        int64_t *wx = (int64_t *)workspace, *wy = wx + n, *wp = wy + n, *wsmp = wp + n;
        hipLaunchKernelGGL(expand_encoded_kernel, dim3(grid_for(n)), dim3(NT), 0, st, x, y, polarity,
                           sample_event_offsets, B, n, wx, wy, wp, wsmp);
        DVSOF_LAUNCH_CHECK();
        return dvsof_voxelize_fwd(wx, wy, t, wp, wsmp, n, t0, t1, B, C, H, W, out, bin0, lin0, stream);
    }
    v2_bind(P, workspace);
    P.enc = 1;
    P.x16 = x; P.y16 = y; P.p8 = polarity; P.ev_off = sample_event_offsets;
    P.x = P.y = P.pol = P.sample = nullptr;
    P.t = t; P.t0 = t0; P.t1 = t1;
    P.n = n; P.B = B; P.C = C; P.H = H; P.W = W;
    P.out = out; P.bin0 = bin0; P.lin0 = lin0;
    return v2_launch(P, flags, st);
}

}  // extern "C"
