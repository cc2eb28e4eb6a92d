// gconv v3: gather-convolution with a PATCH-RESIDENT A operand.
//
// v2 fetches the A slice of every (tap, 16-channel chunk) from L2 again: a
// 3x3 layer reads each input element 9 times per column tile, and for small-N
// tiles the LDS-DMA rate (not the MFMA rate) sets the pace.  Here a workgroup
// owns a TH x TW rectangle of output positions of one image; for each 16-
// channel chunk it stages the input PATCH ((TH-1)*stride+ks) x ((TW-1)*stride+ks)
// pixels x 16 channels (64 B per pixel) ONCE and runs all ks*ks taps from LDS:
// a tap is a wave-uniform byte offset (ky*PW + kx)*64 added to the lane's
// patch address.  The next chunk's patch streams in during the first two tap
// steps of the current chunk (double buffer); weights stream per (chunk, tap)
// through a 4-stage ring as in v2.  Everything else (LDS-DMA with SGPR chunk
// offsets, range-check zero fill, counted vmcnt, raw barrier, VALU-free MFMA
// loop, epilogue semantics) follows gconv2.hip.
//
// Takes: no up-sampling / quad rows (the sub-pixel and phased forms need
// neither), vector members with C % 16 == 0, ks*ks >= 4.  Flat members run
// first through the register path.  LDS bank conflicts on the patch reads
// (64-B pixel stride: 4-way at stride 1) are affordable: an f32 MFMA takes 64
// cycles.
#include "conv_common.h"

namespace {
constexpr int G3_NS = 4;   // weight ring stages
constexpr unsigned G3_OOB = 0x80000000u;
}  // namespace

template <int WROWS, int WCOLS, int TM, int TN, int TH, int TW, int LAS, int BF16>
__global__ __launch_bounds__(CONV_NT) void gconv3_kernel(const GConvParams P, const int nflat,
                                                         const int nchunks, const int npp,
                                                         const int patch_kb, const int tiles_x,
                                                         const int tiles_y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = WROWS * TM * 32, BN = WCOLS * TN * 32;
    static_assert(BM == TH * TW, "rows = spatial tile");
    constexpr int PB0 = (BN + 15) / 16;
    constexpr int PB = (PB0 + 3) / 4 * 4;          // weight pieces per stage
    constexpr int LB = PB / 4;                     // ... per wave
    constexpr int BSTAGE = PB * 1024;
    constexpr int LA = 2 * LAS;                    // patch pieces per wave
    static_assert(WROWS * WCOLS == CONV_NT / kWave, "4 waves");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // [ weight ring | patch 0 | patch 1 | rowB rowY rowX ]
    // npp real pieces + 4 dummy KiB (one per wave) that absorb padding loads
    const int patch_bytes = patch_kb * 1024;
    unsigned char *patch0 = smem + G3_NS * BSTAGE;
    int *rowB = (int *)(patch0 + 2 * patch_bytes), *rowY = rowB + BM, *rowX = rowY + BM;
    long long *rowO = (long long *)(rowX + BM);   // [3][BM] output offsets (conv_epilogue)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int taps = P.ks * P.ks;
    const int ph = blockIdx.z, phy = ph >> 1, phx = ph & 1;
    const int pad_y = P.pad - phy * P.ph_pad, pad_x = P.pad - phx * P.ph_pad;
    const float *Wp = P.W + (size_t)ph * P.w_phase_stride;
    const size_t wrow = (size_t)taps * P.Cin_tot;
    const int n0 = blockIdx.y * BN;

    // spatial tile of this workgroup
    const int tix = blockIdx.x % tiles_x;
    const int tiy = (blockIdx.x / tiles_x) % tiles_y;
    const int img = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = tiy * TH, ox0 = tix * TW;
    const int Y0 = oy0 * P.stride - pad_y, X0 = ox0 * P.stride - pad_x;
    const int PW = (TW - 1) * P.stride + P.ks;

    for (int r = tid; r < BM; r += CONV_NT) {
        const int ry = r / TW, rx = r - ry * TW;
        const bool ok = (oy0 + ry < P.Ho) & (ox0 + rx < P.Wo);
        rowB[r] = img;
        rowY[r] = ok ? (oy0 + ry) * P.stride - pad_y : -(1 << 20);
        rowX[r] = ok ? (ox0 + rx) * P.stride - pad_x : -(1 << 20);
        conv_row_offsets(P, rowO, BM, r, ok, img, oy0 + ry, ox0 + rx, phy, phx);
    }
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lrow = lane & 31, lh = lane >> 5;
    // weight fragments: swizzled [n][16] rows as in v2
    int b_off[TN][2];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int R = (wc * TN + t) * 32 + lrow;
            b_off[t][j] = R * 64 + (((2 * j + lh) ^ ((R >> 2) & 3)) << 4);
        }
    // patch fragments: pixel (ry*stride, rx*stride) of the patch + k-quad lh
    int a_pix[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = (wr * TM + t) * 32 + lrow;
        const int ry = r / TW, rx = r - ry * TW;
        a_pix[t] = (ry * P.stride * PW + rx * P.stride) * 64 + lh * 16;
    }

    // ------------------------------------------------------------------
    // flat concat members first (register path, v1 style) through stage 0
    // of the weight ring and patch buffer 0 used as a plain [row][16] image
    // ------------------------------------------------------------------
    if (nflat > 0) {
        int s = 0, coff = 0, f0 = 0, done = 0;
        while (!P.src[s].flat) {
            coff += P.src[s].C;
            ++s;
        }
        while (done < nflat) {
            const GSrc &S = P.src[s];
            const int f = f0 + (tid & 15);
            const bool fok = f < taps * S.C;
            const int tap = fok ? f / S.C : 0, c = f - tap * S.C;
            const int ky = tap / P.ks, kx = tap - ky * P.ks;
            const int kk = tid & 15;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) {
                const int r = (tid >> 4) + 16 * i;
                const int Y = rowY[r] + ky, X = rowX[r] + kx;
                const bool ok = fok & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
                float v = 0.f;
                if (ok)
                    v = S.p[(size_t)rowB[r] * S.sb + (size_t)Y * S.sy + (size_t)X * S.sx +
                            (size_t)c * S.sc];
                *(float *)(patch0 + r * 64 + kk * 4) = v;      // linear [row][16]
            }
#pragma unroll
            for (int i = 0; i < BN / 16; ++i) {
                const int r = (tid >> 4) + 16 * i, n = n0 + r;
                const float v = (fok && n < P.N)
                                    ? Wp[(size_t)n * wrow + (size_t)tap * P.Cin_tot + coff + c] : 0.f;
                *(float *)(smem + r * 64 + ((((kk >> 2) ^ ((r >> 2) & 3))) << 4) + (kk & 3) * 4) = v;
            }
            __syncthreads();
            {
                f32x4 a[2][TM], b[2][TN];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int t = 0; t < TM; ++t)
                        a[j][t] = *(const f32x4 *)(patch0 + ((wr * TM + t) * 32 + lrow) * 64 + (2 * j + lh) * 16);
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[j][t] = *(const f32x4 *)(smem + b_off[t][j]);
                }
                mfma_k16<BF16, TM, TN>(acc, a, b);
            }
            ++done;
            f0 += BK;
            if (f0 >= taps * S.C) {
                f0 = 0;
                coff += S.C;
                ++s;
                while (done < nflat && !P.src[s].flat) {
                    coff += P.src[s].C;
                    ++s;
                }
            }
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------
    // vector members
    // ------------------------------------------------------------------
    if (nchunks > 0) {
        // weight load slots: piece p = wave + 4*i, rows 16p + (lane>>2)
        unsigned w_voff[LB];
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int p = wave + 4 * i;
            const int r = 16 * p + (lane >> 2), n = n0 + r;
            const unsigned kq4 = (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) << 4);
            w_voff[i] = (r < BN && n < P.N) ? (unsigned)(n * wrow * 4) + kq4 : G3_OOB;
        }
        // patch load slots: piece p = wave + 4*i (i < LA), pixel 16p + (lane>>2),
        // k-quad lane&3 (linear image: quad q of pixel i at i*64 + q*16)
        int p_lin[LA];            // Y*Wv + X of the input pixel, or -1
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int p = wave + 4 * i;
            const int pi = 16 * p + (lane >> 2);
            const int py = pi / PW, px = pi - py * PW;
            const int Y = Y0 + py, X = X0 + px;
            const bool ok = (p < npp) & ((unsigned)Y < (unsigned)P.Hv) & ((unsigned)X < (unsigned)P.Wv);
            p_lin[i] = ok ? Y * P.Wv + X : -1;
        }
        const unsigned kq16 = (unsigned)((lane & 3) << 4);
        const __amdgpu_buffer_rsrc_t wres =
            __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, 0x7fffffff, 0x00020000);

        // chunk iterator for PATCH loads (runs one chunk ahead of the taps)
        int ld_s = 0;
        while (P.src[ld_s].flat) ++ld_s;
        int ld_c0 = 0, ld_C = P.src[ld_s].C, ld_chunk = 0;
        unsigned p_voff[LA];
        __amdgpu_buffer_rsrc_t ares;
        auto bind_source = [&]() {   // per-lane offsets for member ld_s
            const GSrc &S = P.src[ld_s];
            ares = __builtin_amdgcn_make_buffer_rsrc((void *)(S.p + (size_t)img * S.sb), 0, 0x7fffffff,
                                                     0x00020000);
#pragma unroll
            for (int i = 0; i < LA; ++i)
                p_voff[i] = p_lin[i] >= 0 ? (unsigned)(p_lin[i] * S.sx * 4) + kq16 : G3_OOB;
        };
        bind_source();
        auto issue_patch = [&](int half) {   // pieces [half*LAS, half*LAS + LAS) of chunk ld_chunk
            unsigned char *dst0 = patch0 + (ld_chunk & 1) * patch_bytes;
            const int soff = __builtin_amdgcn_readfirstlane(ld_c0 * 4);
#pragma unroll
            for (int i = 0; i < LAS; ++i) {
                const int k = half * LAS + i;
                const int p = wave + 4 * k;
                // padding loads (p >= npp) land in this wave's dummy KiB
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(
                        dst0 + (p < npp ? p : patch_kb - 4 + wave) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, dst, 16, p < npp ? p_voff[k] : G3_OOB,
                                                         soff, 0, 0);
            }
        };
        auto next_patch_chunk = [&]() {
            ++ld_chunk;
            ld_c0 += BK;
            if (ld_c0 >= ld_C && ld_chunk < nchunks) {
                ld_c0 = 0;
                ++ld_s;
                while (P.src[ld_s].flat) ++ld_s;
                ld_C = P.src[ld_s].C;
                bind_source();
            }
        };

        // weight iterator (runs G3_NS-1 steps ahead): member, chunk, tap
        int w_s = ld_s, w_coff = 0;
        for (int i = 0; i < w_s; ++i) w_coff += P.src[i].C;
        int w_c0 = 0, w_tap = 0, w_C = P.src[w_s].C;
        auto issue_w = [&](int stage_idx) {
            const int soff =
                __builtin_amdgcn_readfirstlane((w_tap * P.Cin_tot + w_coff + w_c0) * 4);
            unsigned char *st = smem + stage_idx * BSTAGE;
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                const int p = wave + 4 * i;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(st + p * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, dst, 16, w_voff[i], soff, 0, 0);
            }
            if (++w_tap == taps) {
                w_tap = 0;
                w_c0 += BK;
                if (w_c0 >= w_C) {
                    w_c0 = 0;
                    w_coff += w_C;
                    ++w_s;
                    while (w_s < P.nsrc && P.src[w_s].flat) {
                        w_coff += P.src[w_s].C;
                        ++w_s;
                    }
                    if (w_s < P.nsrc) w_C = P.src[w_s].C;
                }
            }
        };

        const int nsteps = nchunks * taps;
        // prologue: whole first patch + first G3_NS-1 weight slices, then drain
        issue_patch(0);
        issue_patch(1);
        next_patch_chunk();
#pragma unroll
        for (int u = 0; u < G3_NS - 1; ++u)
            if (u < nsteps) issue_w(u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        const bool getenv_dbg_generic = (P.dbg & 256) != 0;   // probe: force the generic loop
        // compute iterator
        int c_tap = 0, c_ky = 0, c_kx = 0, c_chunk = 0;
        int hadA1 = 0, hadA2 = 0;   // patch loads issued 1 / 2 steps ago

#define G3_STEP(U)                                                                               \
    {                                                                                            \
        const int s = s0 + (U);                                                                  \
        if (s < nsteps) {                                                                        \
            if (s > 0) {                                                                         \
                if (s + G3_NS - 1 > nsteps) {                                                    \
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
                } else {                                                                         \
                    const int nA = hadA1 + hadA2;                                                \
                    if (nA == 0)                                                                 \
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB) : "memory");            \
                    else if (nA == 1)                                                            \
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB + LAS) : "memory");      \
                    else                                                                         \
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB + 2 * LAS) : "memory");  \
                }                                                                                \
                __builtin_amdgcn_s_barrier();                                                    \
            }                                                                                    \
            hadA2 = hadA1;                                                                       \
            hadA1 = 0;                                                                           \
            if (s + G3_NS - 1 < nsteps) issue_w(((U) + G3_NS - 1) % G3_NS);                      \
            if (c_tap < 2 && ld_chunk < nchunks) {                                               \
                issue_patch(c_tap);                                                              \
                hadA1 = 1;                                                                       \
                if (c_tap == 1) next_patch_chunk();                                              \
            }                                                                                    \
            {                                                                                    \
                const unsigned char *pb = patch0 + (c_chunk & 1) * patch_bytes +                 \
                                          (c_ky * PW + c_kx) * 64;                               \
                f32x4 a[2][TM], b[2][TN];                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                    \
                {                                                                                \
                    _Pragma("unroll") for (int t = 0; t < TM; ++t) a[j][t] =                     \
                        *(const f32x4 *)(pb + a_pix[t] + 32 * j);                                \
                    _Pragma("unroll") for (int t = 0; t < TN; ++t) b[j][t] =                     \
                        *(const f32x4 *)(smem + (U) * BSTAGE + b_off[t][j]);                     \
                }                                                                                \
                mfma_k16<BF16, TM, TN>(acc, a, b);                                               \
            }                                                                                    \
            if (++c_kx == P.ks) {                                                                \
                c_kx = 0;                                                                        \
                ++c_ky;                                                                          \
            }                                                                                    \
            if (++c_tap == taps) {                                                               \
                c_tap = 0;                                                                       \
                c_ky = 0;                                                                        \
                ++c_chunk;                                                                       \
            }                                                                                    \
        }                                                                                        \
    }
        static_assert(G3_NS == 4, "ring written out for 4 stages");
        // Fast path for the sub-pixel forward (2x2 taps, stride 1): one channel
        // chunk = exactly the four unrolled steps, so the tap of a step, its patch
        // offset and the vmcnt it needs are compile-time constants (the generic
        // step spends ~100 scalar instructions and ~30 branches per 16 MFMAs on
        // that bookkeeping: PMC matrix-pipe utilisation 0.50).
        //   loads in flight behind weight stage s at step U of a chunk that
        //   prefetches the next patch: U0: 2 LB; U1: + half 0; U2: + both halves;
        //   U3: + half 1 (half 0 of three steps ago is not counted: conservative)
#define G3_FAST(U)                                                                               \
    {                                                                                            \
        const int s = s0 + (U);                                                                  \
        if (s > 0) {                                                                             \
            if (s + G3_NS - 1 > nsteps) {                                                        \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
            } else if ((U) == 0 || !np) {                                                        \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB) : "memory");                    \
            } else if ((U) == 2) {                                                               \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB + 2 * LAS) : "memory");          \
            } else {                                                                             \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LB + LAS) : "memory");              \
            }                                                                                    \
            __builtin_amdgcn_s_barrier();                                                        \
        }                                                                                        \
        if (s + G3_NS - 1 < nsteps) issue_w(((U) + G3_NS - 1) % G3_NS);                          \
        if ((U) < 2 && np) {                                                                     \
            issue_patch(U);                                                                      \
            if ((U) == 1) next_patch_chunk();                                                    \
        }                                                                                        \
        {                                                                                        \
            const unsigned char *pb = pbase + (((U) >> 1) * (TW + 1) + ((U) & 1)) * 64;          \
            f32x4 a[2][TM], b[2][TN];                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
            {                                                                                    \
                _Pragma("unroll") for (int t = 0; t < TM; ++t) a[j][t] =                         \
                    *(const f32x4 *)(pb + a_pix[t] + 32 * j);                                    \
                _Pragma("unroll") for (int t = 0; t < TN; ++t) b[j][t] =                         \
                    *(const f32x4 *)(smem + (U) * BSTAGE + b_off[t][j]);                         \
            }                                                                                    \
            mfma_k16<BF16, TM, TN>(acc, a, b);                                                   \
        }                                                                                        \
    }
        if (taps == 4 && P.ks == 2 && P.stride == 1 && !getenv_dbg_generic) {
            for (int s0 = 0; s0 < nsteps; s0 += G3_NS) {
                const bool np = ld_chunk < nchunks;      // this chunk prefetches the next patch
                const unsigned char *pbase = patch0 + ((s0 >> 2) & 1) * patch_bytes;
                G3_FAST(0)
                G3_FAST(1)
                G3_FAST(2)
                G3_FAST(3)
            }
        } else {
            for (int s0 = 0; s0 < nsteps; s0 += G3_NS) {
                G3_STEP(0)
                G3_STEP(1)
                G3_STEP(2)
                G3_STEP(3)
            }
        }
#undef G3_FAST
#undef G3_STEP
    }

    // ---- epilogue (conv_common.h; no quad rows here)
    conv_epilogue<TM, TN>(P, acc, rowO, BM, n0, wr, wc, lane);
#endif
}

namespace {

struct G3Plan {
    int th, tw, npp, patch_kb, las, nflat, nchunks;
    size_t lds;
};

bool g3_plan(const GConvParams &P, int tile, G3Plan &pl)
{
    static const int thw[6][2] = {{0, 0}, {8, 16}, {8, 16}, {4, 16}, {16, 16}, {0, 0}};
    static const int bmn[6][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {256, 32}, {0, 0}};
    if (tile < 1 || tile > 4) return false;
    pl.th = thw[tile][0];
    pl.tw = thw[tile][1];
    const int taps = P.ks * P.ks;
    const int phh = (pl.th - 1) * P.stride + P.ks, pww = (pl.tw - 1) * P.stride + P.ks;
    pl.npp = (phh * pww + 15) / 16;
    const int la = (pl.npp + 3) / 4;               // patch pieces per wave
    pl.las = (la + 1) / 2;
    if (pl.las > 3 && pl.las <= 5) pl.las = 5;
    if (pl.las > 5) return false;
    pl.nflat = pl.nchunks = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) pl.nflat += (taps * P.src[s].C + BK - 1) / BK;
        else pl.nchunks += P.src[s].C / BK;
    }
    const int pb = ((bmn[tile][1] + 15) / 16 + 3) / 4 * 4;
    // buffer: the patch (or the flat path's [BM][16] image) + 4 dummy KiB
    pl.patch_kb = (pl.npp > bmn[tile][0] / 16 ? pl.npp : bmn[tile][0] / 16) + 4;
    pl.lds = (size_t)G3_NS * pb * 1024 + 2 * (size_t)pl.patch_kb * 1024 + 3 * bmn[tile][0] * (sizeof(int) + sizeof(long long));
    return pl.lds <= 160 * 1024;
}

template <int WROWS, int WCOLS, int TM, int TN, int TH, int TW, int LAS, int BF16>
int launch3x(const GConvParams &P, const G3Plan &pl, hipStream_t st)
{
    constexpr int BN = WCOLS * TN * 32;
    static size_t attr_lds = 0;
    if (pl.lds > attr_lds) {
        DVSOF_HIP_TRY(hipFuncSetAttribute(
            (const void *)gconv3_kernel<WROWS, WCOLS, TM, TN, TH, TW, LAS, BF16>,
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
        attr_lds = pl.lds;
    }
    const int tiles_x = (P.Wo + TW - 1) / TW, tiles_y = (P.Ho + TH - 1) / TH;
    dim3 grid(tiles_x * tiles_y * P.B, (P.N + BN - 1) / BN, P.nph);
    hipLaunchKernelGGL((gconv3_kernel<WROWS, WCOLS, TM, TN, TH, TW, LAS, BF16>), grid, dim3(CONV_NT), pl.lds,
                       st, P, pl.nflat, pl.nchunks, pl.npp, pl.patch_kb, tiles_x, tiles_y);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

template <int WROWS, int WCOLS, int TM, int TN, int TH, int TW, int LAS>
int launch3(const GConvParams &P, const G3Plan &pl, hipStream_t st)
{
    if (P.mfma_bf16 == 2) return launch3x<WROWS, WCOLS, TM, TN, TH, TW, LAS, 2>(P, pl, st);
    if (P.mfma_bf16 == 1) return launch3x<WROWS, WCOLS, TM, TN, TH, TW, LAS, 1>(P, pl, st);
    return launch3x<WROWS, WCOLS, TM, TN, TH, TW, LAS, 0>(P, pl, st);
}

template <int WROWS, int WCOLS, int TM, int TN, int TH, int TW>
int launch3_las(const GConvParams &P, const G3Plan &pl, hipStream_t st)
{
    switch (pl.las) {
    case 1: return launch3<WROWS, WCOLS, TM, TN, TH, TW, 1>(P, pl, st);
    case 2: return launch3<WROWS, WCOLS, TM, TN, TH, TW, 2>(P, pl, st);
    case 3: return launch3<WROWS, WCOLS, TM, TN, TH, TW, 3>(P, pl, st);
    case 5: return launch3<WROWS, WCOLS, TM, TN, TH, TW, 5>(P, pl, st);
    default: return DVSOF_EINVAL;
    }
}

}  // namespace

bool gconv3_eligible(const GConvParams &P, int tile, long long max_src_bytes, long long w_bytes)
{
    // measured (profiles/round1): v3 only beats v2 on the 256x32 tile, where
    // the A traffic per MFMA is highest; elsewhere its LDS footprint costs more
    if (tile != 4) return false;
    if (P.up != UP_NONE || P.quad || P.ks * P.ks < 4) return false;
    bool any_vec = false;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) continue;
        any_vec = true;
        if (P.src[s].sc != 1 || (P.src[s].C % BK) || P.src[s].sx != P.src[s].C ||
            P.src[s].sy != P.Wv * P.src[s].C)
            return false;
    }
    if (!any_vec) return false;
    if (max_src_bytes >= 0x7fffffffLL || w_bytes >= 0x7fffffffLL) return false;
    G3Plan pl;
    return g3_plan(P, tile, pl);
}

int gconv3_launch(const GConvParams &P, int tile, hipStream_t st)
{
    G3Plan pl;
    if (!g3_plan(P, tile, pl)) return DVSOF_EINVAL;
    switch (tile) {
    case 1: return launch3_las<2, 2, 2, 2, 8, 16>(P, pl, st);   // 128 x 128
    case 2: return launch3_las<2, 2, 2, 1, 8, 16>(P, pl, st);   // 128 x 64
    case 3: return launch3_las<2, 2, 1, 1, 4, 16>(P, pl, st);   // 64 x 64
    case 4: return launch3_las<4, 1, 2, 1, 16, 16>(P, pl, st);  // 256 x 32
    default: return DVSOF_EINVAL;
    }
}
