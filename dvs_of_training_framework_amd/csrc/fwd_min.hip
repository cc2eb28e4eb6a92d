// Forward of a 2x up-sampling decoder stage with NINE products per
// low-resolution pixel instead of sixteen (exact f32, v_mfma_f32_32x32x2_f32).
//
// The minimal bilinear algorithm of csrc/wgrad_min.hip for
// `nearest-up2 -> conv3x3(pad 1)` (EV_FlowNet decoder; reference call site
// utils/training.py:59-64 through the absent EV_FlowNet.net), forward direction:
//     Wt = G w G^T,  G = [[1,0,0],[1,1,1],[0,0,1]]              (weight form, once per step)
//     Xt[p][q] = D_p(rows) D_q(columns) x,  D = (x[-1] - x[0], x[0], x[+1] - x[0])
//     M[p][q]  = Wt[p][q] . Xt[p][q]                             (nine GEMMs over the channels)
//     y[2i+a][2j+b] = M[a][b] + M[a][b+1] + M[a+1][b] + M[a+1][b+1]
// x is zero outside the frame (the convolution's zero padding of the
// up-sampled image).  9/16 of the sub-pixel form's matrix FLOPs, 1/4 of the
// layer's as specified.
//
// Kernel (all four decoder stages): a workgroup of 8 waves owns 32 output
// channels x a block of NR rows x 16 columns of LOW-resolution pixels
// (NR = 4 | 8, chosen so that >= 256 workgroups remain) and walks the
// input channels in chunks of 32: per chunk ONE (NR+2) x 18-pixel patch and the
// nine 32 x 32 weight tiles go L2 -> LDS by LDS-DMA (double buffered).  A wave
// is (pixel tile of 2 rows x 16 pixels, K slice): it holds the nine 32 x 32
// accumulators M[p][q] of its pixels and takes every KS-th group of 4 channels
// (KS = 16 / NR slices: with few pixels per workgroup the waves split K; they
// meet once, in the epilogue).  Per group of 8 channels a wave reads 9 weight
// and 9 patch fragments (ds_read_b128: the lane halves take channels 0-3 / 4-7,
// four K steps each), makes the nine Xt quads with 12 x 2 packed subtractions
// and issues 36 matrix instructions.
// LDS images without swizzles: rows (patch slots / weight rows) are dealt to
// eight arrays by row mod 8, each array padded by 16 bytes -- eight lanes of
// a fragment read then cover the 32 banks once (16 bytes each, no conflict),
// and the chunk index stays an immediate offset.
// Epilogue: nine accumulators -> four phase tiles in registers, K slices and
// the transposition to pixel-major meet in LDS, 8 lanes store one output
// pixel's 128 bytes: bias, border-class bias (flow fold), pre-activation copy,
// activation -- the epilogue of fwd_patch_f32_kernel.
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned FM_OOB = 0x80000000u;
constexpr int FM_NT = 512;
constexpr int FM_APA = 5;                        // pieces per weight residue array (36 rows of 128 B + 4 unused)
constexpr int FM_AP = 8 * FM_APA;                // weight pieces per stage
constexpr int FM_AARR = FM_APA * 1024 + 16;      // bytes of one weight residue array + pad

// LDS image of a stage.  Rows of 128 bytes (32 channels of a weight row / a patch pixel slot)
// are dealt to EIGHT arrays by row mod 8; array m holds rows m, m + 8, ... at 128 B and starts
// at m x (array bytes + 16): consecutive rows then sit 16 bytes apart modulo 256, and the 32
// lanes of a ds_read_b64 fragment read (32 consecutive weight rows / 2 x 16 consecutive patch
// slots) meet 2 per bank pair -- without an XOR swizzle, so that the group of 4 channels is an
// immediate offset.  An LDS-DMA piece (1 KiB, lane-linear) is 8 rows of one array.
template <int NR>
struct FMGeom {
    static constexpr int PT = NR / 2;            // pixel tiles (2 rows x 16 pixels) per block
    static constexpr int KS = 8 / PT;            // K slices per pixel tile
    static constexpr int SLOTS = (NR + 2) * 18;
    static constexpr int BROWS = (SLOTS + 7) / 8;        // rows per patch residue array
    static constexpr int BPA = (BROWS + 7) / 8;          // pieces per patch residue array
    static constexpr int BARR = BPA * 1024 + 16;
    static constexpr int BOFF = 8 * FM_AARR;
    static constexpr int NPIECE = FM_AP + 8 * BPA;       // DMA pieces per stage (a multiple of 8)
    static constexpr int LPW = NPIECE / 8;
    static constexpr int STAGE = ((8 * FM_AARR + 8 * BARR + 1023) / 1024) * 1024;
    static constexpr int XCH = 8 * 4 * 4096;     // epilogue exchange
    static constexpr int LDS = 2 * STAGE > XCH ? 2 * STAGE : XCH;
};

// Wt[3 p + q][co][ci] = sum_{k,l} G[p][k] G[q][l] w[co][k][l][ci]
__global__ __launch_bounds__(256) void min9_fwd_weights_kernel(const float *__restrict__ w, float *__restrict__ wt,
                                                               int Cout, int Ctot)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Cout * Ctot) return;
    const int ci = (int)(i % Ctot), co = (int)(i / Ctot);
    float k[3][3], r[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t / 3][t % 3] = w[((size_t)co * 9 + t) * Ctot + ci];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        r[0][l] = k[0][l];
        r[1][l] = (k[0][l] + k[1][l]) + k[2][l];
        r[2][l] = k[2][l];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float v[3] = {r[p][0], (r[p][0] + r[p][1]) + r[p][2], r[p][2]};
#pragma unroll
        for (int q = 0; q < 3; ++q) wt[((size_t)(3 * p + q) * Cout + co) * Ctot + ci] = v[q];
    }
}

template <int NR, bool ZOUT>
__global__ __launch_bounds__(FM_NT) void fwd_min_f32_kernel(const GConvParams P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef FMGeom<NR> G;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = wave % G::PT, ks = wave / G::PT;
    const int lrow = lane & 31, lh = lane >> 5;
    const int H = P.Hv, W = P.Wv;
    const int nbx = W / 16, nby = H / NR;

    // block of pixels (XCD-aware: an XCD owns a contiguous range of blocks; the output-channel
    // tiles of one block read the same patch through one L2) and output-channel tile
    unsigned wg = blockIdx.x;
    {
        const unsigned total = gridDim.x, x = wg & 7u, q = total >> 3, r = total & 7u;
        wg = x * q + min(x, r) + (wg >> 3);
    }
    const int nco = P.N / 32;
    const int cot = (int)(wg % (unsigned)nco), blk = (int)(wg / (unsigned)nco);
    const int co0 = 32 * cot;
    const int bx = blk % nbx, by = (blk / nbx) % nby, b = blk / (nbx * nby);
    const int oy = NR * by, ox = 16 * bx;

    // ---- DMA roles: piece p = wave + 8 i (p < FM_AP: weight rows, else patch rows); the
    // destination of a piece is wave-uniform, lane l lands at + 16 l
    const GSrc &S0 = P.src[0], &S1 = P.src[1];
    // (FM_AP = 8 x 5: rounds i < 5 are weight pieces for every wave, the rest patch pieces)
    unsigned off_a[G::LPW], off_b[G::LPW - FM_APA];   // weights | member 0 of the patch; member 1
#pragma unroll
    for (int i = 0; i < G::LPW; ++i) {
        const int p = wave + 8 * i;
        off_a[i] = FM_OOB;
        if (i >= FM_APA) off_b[i - FM_APA] = FM_OOB;
        if (i < FM_APA) {
            const int m = p / FM_APA, pp = p - FM_APA * m;
            const int j = 8 * pp + (lane >> 3);         // row within array m
            if (j < 36) {
                const int R = 8 * j + m;                // (component, output channel) row
                const int c = R >> 5, r = R & 31;
                off_a[i] = (unsigned)((((size_t)c * P.N + co0 + r) * P.Cin_tot + 4 * (lane & 7)) * 4);
            }
        } else {
            const int pb = p - FM_AP;
            const int m = pb / G::BPA, pp = pb - G::BPA * m;
            const int n = 8 * (8 * pp + (lane >> 3)) + m;       // patch slot
            if (n < G::SLOTS) {
                const int r = n / 18, c = n - 18 * r;
                // (the resources' bases are shifted by (-1, -1): offsets stay non-negative)
                if (((unsigned)(oy + r - 1) < (unsigned)H) & ((unsigned)(ox + c - 1) < (unsigned)W)) {
                    off_a[i] = (unsigned)((r * S0.sy + c * S0.sx + 4 * (lane & 7)) * 4);
                    off_b[i - FM_APA] = (unsigned)((r * S1.sy + c * S1.sx + 4 * (lane & 7)) * 4);
                }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void *)P.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t res0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S0.p - ((long long)S0.sy + S0.sx)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t res1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S1.p - ((long long)S1.sy + S1.sx)), 0, 0x7fffffff, 0x00020000);
    const int base0 = __builtin_amdgcn_readfirstlane(
        (int)(((long long)b * S0.sb + (long long)oy * S0.sy + (long long)ox * S0.sx) * 4));
    const int base1 = __builtin_amdgcn_readfirstlane(
        (int)(((long long)b * S1.sb + (long long)oy * S1.sy + (long long)ox * S1.sx) * 4));
    const int nch0 = S0.C / 32, nchunks = P.Cin_tot / 32;

    auto issue = [&](int stage_idx, int ch) {
        unsigned char *st = smem + stage_idx * G::STAGE;
        const bool m1 = ch >= nch0;
        const int so_w = ch * 128;
        const int so_x = m1 ? base1 + (ch - nch0) * 128 : base0 + ch * 128;
#pragma unroll
        for (int i = 0; i < G::LPW; ++i) {
            const int p = wave + 8 * i;
            if (i < FM_APA) {
                const int m = p / FM_APA, pp = p - FM_APA * m;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(st + m * FM_AARR + pp * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, dst, 16, off_a[i], so_w, 0, 0);
            } else {
                const int pb = p - FM_AP;
                const int m = pb / G::BPA, pp = pb - G::BPA * m;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(st + G::BOFF + m * G::BARR + pp * 1024);
                if (m1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(res1, dst, 16, off_b[i - FM_APA], so_x, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(res0, dst, 16, off_a[i], so_x, 0, 0);
            }
        }
    };

    // ---- fragment addresses (relative to a stage)
    // A: row R = 32 c + lrow -> array lrow & 7, row 4 c + (lrow >> 3)
#ifndef FM_B64
    constexpr int FRAG = 16;    // bytes of a row a lane reads per group: 4 channels (ds_read_b128), see compute
#else
    constexpr int FRAG = 8;     // (-DFM_B64: 2 channels per read, groups of 4 channels -- the form before)
#endif
    const int abase = (lrow & 7) * FM_AARR + (lrow >> 3) * 128 + FRAG * lh;
    // B: slot n = (2 pt + (lrow >> 4) + r) 18 + (lrow & 15) + c -> array n & 7, row n >> 3
    int bbase[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n = (2 * pt + (lrow >> 4) + r) * 18 + (lrow & 15) + c;
            bbase[3 * r + c] = G::BOFF + (n & 7) * G::BARR + (n >> 3) * 128 + FRAG * lh;
        }

    f32x16 acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

#ifndef FM_PROBE
#define FM_PROBE 0      // timing probes (variant builds with -DFM_B64, tools/variant.sh; results wrong): 1 no DMA
#endif                  // behind the second chunk, 2 one weight and one patch fragment read per group instead of 18,
                        // 4 no barrier.  Per stage, us (8-byte fragment reads): 104 102 108 120 | 1: 101 99 106 117 |
                        // 2: 83 82 87 97 | 4: 91 92 98 110 | 7: 73 75 78 86 (= the matrix pipe's own time): the
                        // fragment reads cost 20 %, the chunk barrier 12 %, the DMA 3 %
    auto compute = [&](int u) {
        const unsigned char *st = smem + u * G::STAGE;
#ifdef FM_STAGGER       // experiment: the second wave of every SIMD starts its chunk late (x 64 cycles): 9 ->
        if (wave >= 4) __builtin_amdgcn_s_sleep(FM_STAGGER);     // 105 / 101 / 106 / 118 against 106 / 103 / 109 / 122 us
#endif
#ifndef FM_B64
        // Groups of EIGHT channels: a lane reads 16 bytes of its row (ds_read_b128: the low half of
        // the wave channels 0-3 of the group, the high half 4-7 -- any assignment of K to the two
        // lane halves will do as long as both operands use the same) and feeds FOUR matrix
        // instructions per component with them.  8 lanes x 16 bytes = one pass over the 32 banks
        // without a conflict in this layout, where the 8-byte reads met two per bank pair: half
        // the LDS time and half the read instructions per matrix instruction (the fragment reads
        // cost 20 % of a stage, FM_PROBE).  The weight fragments are fetched one component ahead.
#pragma unroll
        for (int gi = 0; gi < 4 / G::KS; ++gi) {
            const int kg = ks + G::KS * gi;             // group of 8 channels (32 bytes of a row)
            f32x4 x[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) x[c] = *(const f32x4 *)(st + bbase[c] + 32 * kg);
            f32x4 an = *(const f32x4 *)(st + abase + 32 * kg);
            // rows, then columns: Xt[p][q]
            f32x4 rw[3][3], xt[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rw[0][c] = x[c] - x[3 + c];
                rw[1][c] = x[3 + c];
                rw[2][c] = x[6 + c] - x[3 + c];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                xt[3 * p] = rw[p][0] - rw[p][1];
                xt[3 * p + 1] = rw[p][1];
                xt[3 * p + 2] = rw[p][2] - rw[p][1];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const f32x4 ac = an;
                if (c + 1 < 9) an = *(const f32x4 *)(st + abase + (c + 1) * 512 + 32 * kg);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[j], xt[c][j], acc[c], 0, 0, 0);
            }
        }
#else
#pragma unroll
        for (int gi = 0; gi < 8 / G::KS; ++gi) {
            const int kg = ks + G::KS * gi;             // group of 4 channels (16 bytes of a row)
            f32x2 a[9], x[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = *(const f32x2 *)(st + abase + ((FM_PROBE & 2) ? 0 : c) * 512 + 16 * kg);
#pragma unroll
            for (int c = 0; c < 9; ++c) x[c] = *(const f32x2 *)(st + bbase[(FM_PROBE & 2) ? 0 : c] + 16 * kg);
            // rows, then columns: Xt[p][q]
            f32x2 rw[3][3], xt[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rw[0][c] = x[c] - x[3 + c];
                rw[1][c] = x[3 + c];
                rw[2][c] = x[6 + c] - x[3 + c];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                xt[3 * p] = rw[p][0] - rw[p][1];
                xt[3 * p + 1] = rw[p][1];
                xt[3 * p + 2] = rw[p][2] - rw[p][1];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][0], xt[c][0], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][1], xt[c][1], acc[c], 0, 0, 0);
            }
        }
#endif
    };

    issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(FM_PROBE & 4) || ch == 0) __builtin_amdgcn_s_barrier();
        if (ch + 1 < nchunks && (!(FM_PROBE & 1) || ch == 0)) issue((ch + 1) & 1, ch + 1);
        compute(ch & 1);
    }

    // (Measured and not kept: NO barrier in the K loop -- per stage an arrival and a release counter
    // in LDS (a wave signals that its DMA pieces landed / that it has read its last fragment, spins
    // -- bounded -- until all eight have; the next chunk's DMA issued between a wave's two groups):
    // correct, and 113 / 95 / 101 / 113 us per stage against 93 / 93 / 98 / 110 -- the polls cost
    // more than the barrier's bubble.)
    // (Measured and not kept: the twelve subtractions pinned to v_pk_add_f32 by inline asm -- the
    // compiler splits most of them into scalar pairs, 86 v_add_f32 per unrolled chunk -- with the
    // two wait states a matrix instruction needs behind a vector write of its operand, which a
    // hand-written instruction hides from the hazard recogniser (without them: wrong sums):
    // 109 / 101 / 107 / 118 us per stage against 105 / 102 / 109 / 121, 2.518 against 2.526 ms per
    // step -- the K loop is not bound by its vector instructions.)
    // (... and again with the 16-byte reads, which left the registers for it -- the next group's nine
    // patch fragments fetched ahead of this group's 36 matrix instructions, 256 registers: 98 / 98 /
    // 102 / 114 us per stage against 97 / 95 / 103 / 112.)
    // (Measured and not kept, batch 8: the next group's 18 fragment reads issued ahead of this
    // group's matrix instructions -- register double buffering, with and without
    // sched_group_barrier ordering: 105-136 / 108-164 us per stage against 101-119; the two
    // waves of a SIMD already cover each other's read -> subtract phases.)
    // ---- epilogue.  Phase tiles (a, b) = sums of four components; [pixel 32][co 32] rows of
    // 128 bytes per (wave, phase), 16-byte chunks XOR-swizzled by pixel & 7
    // (the bias is loaded ahead of the barriers: behind them its latency is every workgroup's own)
    const int ecq = lane & 7;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias) bias4 = *(const f32x4 *)(P.bias + co0 + 4 * ecq);
    __builtin_amdgcn_s_barrier();       // every wave is done with the stages
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int fa = f >> 1, fb = f & 1;
        const f32x16 t = (acc[3 * fa + fb] + acc[3 * fa + fb + 1]) + (acc[3 * fa + 3 + fb] + acc[3 * fa + 4 + fb]);
        unsigned char *xt_ = smem + (wave * 4 + f) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
            *(f32x4 *)(xt_ + lrow * 128 + (((2 * g + lh) ^ (lrow & 7)) << 4)) = v;
        }
    }
    __builtin_amdgcn_s_barrier();
    const GDst &D = P.dst[0];
    // this wave finishes phases f = ks, ks + KS, ... of its pixel tile
#pragma unroll
    for (int fi = 0; fi < 4 / G::KS; ++fi) {
        const int f = ks + G::KS * fi;
        const int fa = f >> 1, fb = f & 1;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = 8 * it + (lane >> 3);
            const int xo = px * 128 + ((ecq ^ (px & 7)) << 4);
            f32x4 v = *(const f32x4 *)(smem + (pt * 4 + f) * 4096 + xo);
#pragma unroll
            for (int k = 1; k < G::KS; ++k) v += *(const f32x4 *)(smem + ((pt + G::PT * k) * 4 + f) * 4096 + xo);
            const int yy = oy + 2 * pt + (px >> 4), xx = ox + (px & 15);
            const long long o = (long long)b * D.sb + (long long)yy * D.sy + (long long)xx * D.sx +
                                (long long)fa * D.ph_y + (long long)fb * D.ph_x + co0 + 4 * ecq;
            v += bias4;
            if (P.bias_cls) {
                const int Y = 2 * yy + fa, X = 2 * xx + fb;
                const int cls = 3 * (Y == 0 ? 1 : Y == P.out_H - 1 ? 2 : 0) + (X == 0 ? 1 : X == P.out_W - 1 ? 2 : 0);
                if (cls) v += *(const f32x4 *)(P.bias_cls + cls * P.N + co0 + 4 * ecq);
            }
            if (ZOUT) *(f32x4 *)(P.zout + o) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], P.act);
            *(f32x4 *)(D.p + o) = v;
        }
    }
#endif
}

}  // namespace

// Shape test shared by the weight preparation and the launch (both see the same descriptor):
// exact f32, two NHWC vector members with multiples of 32 channels, 32 | Cout, 16 | W, 4 | H
// (DVSOF_NO_FWD_MIN=1: the sixteen-product kernels)
bool min9_shape_ok(int mfma, int nsrc, const int *C, const int *nhwc, int Cout, int H, int W)
{
    static const bool off = getenv("DVSOF_NO_FWD_MIN") != nullptr;
    if (off || mfma != 0 || nsrc != 2) return false;
    for (int s = 0; s < 2; ++s)
        if (!nhwc[s] || (C[s] & 31)) return false;
    return (Cout & 31) == 0 && (W % 16) == 0 && (H % 4) == 0 && H >= 4;
}

int min9_prepare_fwd(const float *w, float *wt, int Cout, int Ctot, hipStream_t st)
{
    const size_t n = (size_t)Cout * Ctot;
    hipLaunchKernelGGL(min9_fwd_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wt, Cout, Ctot);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

template <int NR, bool ZOUT>
static int fm_launch(const GConvParams &P, int grid, hipStream_t st)
{
    typedef FMGeom<NR> G;
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)fwd_min_f32_kernel<NR, ZOUT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((fwd_min_f32_kernel<NR, ZOUT>), dim3(grid), dim3(FM_NT), G::LDS, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// P as dvsof_conv2d_fwd fills it for a sub-pixel layer (P.W = Wt[9][Cout][Cin_tot] here)
int fwd_min_launch(const GConvParams &P, hipStream_t st)
{
    // 16-byte loads and stores everywhere
    for (int s = 0; s < 2; ++s) {
        const GSrc &S = P.src[s];
        if (!S.p || (reinterpret_cast<uintptr_t>(S.p) & 15) || ((S.sb | S.sy | S.sx) & 3) || S.sc != 1) return DVSOF_EINVAL;
    }
    const GDst &D = P.dst[0];
    if (D.addend || D.addend2 || D.actsrc || D.sc != 1 || ((D.sb | D.sy | D.sx | D.ph_y | D.ph_x) & 3)) return DVSOF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(D.p) & 15) || (reinterpret_cast<uintptr_t>(P.zout) & 15) ||
        (reinterpret_cast<uintptr_t>(P.W) & 15) || (reinterpret_cast<uintptr_t>(P.bias) & 15) ||
        (reinterpret_cast<uintptr_t>(P.bias_cls) & 15))
        return DVSOF_EINVAL;
    // rows per block: the largest of 16 | 8 | 4 that divides H and leaves >= 256 workgroups
    // (else the smallest that divides)
    static const int force = getenv("DVSOF_FWD_MIN_NR") ? atoi(getenv("DVSOF_FWD_MIN_NR")) : 0;
    const long long per_row = (long long)P.B * (P.Wv / 16) * (P.N / 32);
    int nr = 4;
    if (P.Hv % 8 == 0 && per_row * (P.Hv / 8) >= 256) nr = 8;
    if ((force == 4 || force == 8) && P.Hv % force == 0) nr = force;
    const int grid = (int)(per_row * (P.Hv / nr));
    const bool z = P.zout != nullptr;
    if (nr == 8) return z ? fm_launch<8, true>(P, grid, st) : fm_launch<8, false>(P, grid, st);
    return z ? fm_launch<4, true>(P, grid, st) : fm_launch<4, false>(P, grid, st);
}
