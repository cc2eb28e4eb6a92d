// Weight gradient of a 2x up-sampling decoder stage (sub-pixel form) from the
// bf16 twins with the INPUT PATCH RESIDENT in LDS.
//
//   dWf[(a,b)][co][(p,q)][ci] = sum over low-res pixels (i,j) of
//        gout[2i+a][2j+b][co] * x[i-1+a+p][j-1+b+q][ci]        a,b,p,q in {0,1}
//
// (the four 2x2 phase kernels of `up2 -> conv3x3`; subpixel_fold_kernel folds
// them back to the 3x3 gradient; EV_FlowNet decoder, reference call site
// utils/training.py:158 through the absent EV_FlowNet.net).
//
// wgrad2_twins_kernel runs these as 4 phase GEMMs whose column tiles are
// (tap, channel) ranges: every (phase, tap) view of the input is streamed
// through LDS on its own and every column tile re-reads the gradient -- the
// finest stage moves 671 MB through LDS-DMA for 67 MB of operands and is bound
// by exactly that (120-130 us; profiles/round3).  The 16 views are the SAME
// pixels shifted by -1/0/+1: here a workgroup owns 32 output channels x 32
// input channels x ALL 16 (phase, tap) combinations, stages per 16-pixel
// group ONE 3 x 18-pixel input patch and the four phase planes of the
// gradient, and the views are byte offsets into the patch:
//
//   wave w = phase (a,b); its A operand is the phase plane G_ab[16 px][32 co]
//   (shared by its 4 taps), its B operands X[(a+p)][(b+q) .. +15][32 ci];
//   v_mfma_f32_32x32x16_bf16, fragments by ds_read_b64_tr_b16 exactly as in
//   wgrad2_twins_kernel (64-byte pixel rows: no swizzle needed).
//
// Operands go L2 -> LDS by LDS-DMA (1 KiB pieces = 16 pixel slots of 64 B, the
// lane's global offset picks the pixel: a gather, so the patch layout is ours),
// 4-stage ring, 2 groups per stage.  Partial sums per (phase, K split) slab in
// the layout wgrad2 writes (the same fold / bias tail consumes them); K splits
// are workgroups.  Same products as the twins kernel, f32 summation order
// differs.
#include "conv_common.h"
#include <stdlib.h>

namespace {

__attribute__((unused)) constexpr unsigned WP_OOB = 0x80000000u;
constexpr int WP_XSLOT = 4 * 18;    // patch pixels of a block: rows i-1 .. i+2, columns j0-1 .. j0+16

// CT input channels per workgroup (32 | 64): a pixel slot of the patch is 2 CT bytes.
template <int CT>
struct WPGeom {
    static constexpr int PXB = 2 * CT;                  // bytes per patch pixel slot
    static constexpr int LPS = PXB / 16;                // lanes (16-byte chunks) per slot
    static constexpr int SPP = 64 / LPS;                // slots per 1 KiB piece
    static constexpr int XP = (WP_XSLOT + SPP - 1) / SPP;   // pieces of the patch
    static constexpr int GP = 8;                        // gradient planes: 4 phases x 32 px x 64 B
    static constexpr int PIECES = (XP + GP + 3) / 4 * 4;    // every wave issues the same number of loads
    static constexpr int LPW = PIECES / 4;
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int NS = CT == 32 ? 4 : 3;         // ring stages (64 | 60 KiB: two workgroups per CU)
    static constexpr int NB = CT / 32;                  // 32-column blocks per tap
};

// 128-byte pixel rows: the four pixel rows of a transposed fragment read are one pitch
// apart and rows q, q + 2 would share banks -- the 64-byte half h of slot n holds channel
// half h ^ ((n >> 1) & 1) (the lane FETCHES the permuted half; LDS-DMA destinations are
// lane-linear and cannot be padded); 64-byte rows need nothing
template <int CT>
__device__ __forceinline__ int wp_swz(int n)
{
    return CT == 64 ? (n >> 1) & 1 : 0;
}

// Epilogue of both kernels: the 16 (phase, tap) tiles of the workgroup FOLDED to the 9 taps of
// the 3x3 gradient before they leave it (subpixel_fold_kernel's rule: per axis kernel row
// k = 0 <- (phase 0, tap 0) + (phase 1, tap 0); 1 <- (0, 1) + (1, 0); 2 <- (0, 1) + (1, 1)).
// The phases are waves, so the tiles meet in LDS (64 KiB: the ring, no longer in use), in the
// accumulator layout [tile][register][lane] -- a lane adds ITS element of four tiles, in a
// fixed order.  Slabs are [split][Cout][3][3][Cin_tot]: 9/16 of the phase-form bytes, written
// once and read once by the plain slab reduce.
template <int NB>
__device__ __forceinline__ void wp_fold_store(const WGradParams &P, f32x16 (&acc)[4][NB], unsigned char *smem,
                                              int wave, int lane, int split, int co0, int col0)
{
    float *xs = (float *)smem;
    const int lrow = lane & 31, lh = lane >> 5;
    const size_t row9 = (size_t)9 * P.Cin_tot;
    float *dW = P.dW + (size_t)split * P.Cout * row9;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __builtin_amdgcn_s_barrier();       // last stage read / previous pass folded
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) xs[((wave * 4 + t) * 16 + reg) * 64 + lane] = acc[t][nb][reg];
        __builtin_amdgcn_s_barrier();
        for (int k = wave; k < 9; k += 4) {
            const int ky = k / 3, kx = k - 3 * ky;
            // contributors (phase bit, tap bit) per axis: (0, k > 0) and (1, k == 2)
            const int ty0 = ky > 0, ty1 = ky == 2, tx0 = kx > 0, tx1 = kx == 2;
            const float *s00 = xs + ((0 * 4 + 2 * ty0 + tx0) * 16) * 64 + lane;     // phase (0,0)
            const float *s01 = xs + ((1 * 4 + 2 * ty0 + tx1) * 16) * 64 + lane;     // phase (0,1)
            const float *s10 = xs + ((2 * 4 + 2 * ty1 + tx0) * 16) * 64 + lane;     // phase (1,0)
            const float *s11 = xs + ((3 * 4 + 2 * ty1 + tx1) * 16) * 64 + lane;     // phase (1,1)
            const size_t col = (size_t)k * P.Cin_tot + col0 + 32 * nb + lrow;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                dW[(size_t)co * row9 + col] = ((s00[reg * 64] + s01[reg * 64]) + s10[reg * 64]) + s11[reg * 64];
            }
        }
    }
}

template <int CT>
__global__ __launch_bounds__(CONV_NT) void wgrad_patch_twins_kernel(const WGradParams P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef WPGeom<CT> G;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pa = wave >> 1, pb = wave & 1;            // this wave's output phase

    int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
    if (P.xcd) {
        // linear id % 8 = XCD: every XCD takes a contiguous range of (split, tile) pairs, tiles
        // fastest -- the tiles of a split read the SAME gradient planes and the two 64-byte
        // halves of the same 128-byte input lines, and now meet in one L2 (any grid size)
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned x = L & 7u, q = total >> 3, r = total & 7u;
        const unsigned t = x * q + min(x, r) + (L >> 3);
        bx = (int)(t % gx);
        by = (int)((t / gx) % gy);
        split = (int)(t / (gx * gy));
    }
    int s = 0;
    for (int i = 1; i < P.nsrc; ++i)
        if (bx >= P.tile_begin[i]) s = i;
    const GSrc &S = P.src[s];
    int coff = 0;
    for (int i = 0; i < s; ++i) coff += P.src[i].C;
    const int c0 = (bx - P.tile_begin[s]) * CT;          // channel tile of the member
    const int co0 = by * 32;
    const int H = P.Hv, W = P.Wv;                         // low-resolution frame
    // K = pixels in blocks of 2 rows x 16 columns; this split's range of blocks
    const int nbx = W / 16, nby = H / 2;
    const int nblocks = P.B * nby * nbx, bps = (nblocks + P.S - 1) / P.S;
    const int blk0 = split * bps;
    const int nsteps = max(0, min(nblocks, blk0 + bps) - blk0);

    // load slots: piece p = wave + 4 i; a lane fetches one 16-byte chunk of one pixel slot
    unsigned v_off[G::LPW];
    int v_dy[G::LPW], v_dx[G::LPW];
    bool v_on[G::LPW];
#pragma unroll
    for (int i = 0; i < G::LPW; ++i) {
        const int p = wave + 4 * i;
        v_off[i] = WP_OOB;
        v_dy[i] = v_dx[i] = 0;
        v_on[i] = false;
        if (p < G::XP) {
            const int n = G::SPP * p + lane / G::LPS, q = lane % G::LPS;
            if (n < WP_XSLOT) {
                const int r = n / 18, c = n - 18 * r;
                v_on[i] = true;
                v_dy[i] = r - 1;
                v_dx[i] = c - 1;
                const int qq = q ^ (4 * wp_swz<CT>(n));
                // the resource's base is shifted by (-1, -1): offsets stay non-negative
                v_off[i] = (unsigned)((r * S.sy + c * S.sx + c0 + 8 * qq) * 2);
            }
        } else if (p < G::XP + G::GP) {
            const int n = 16 * (p - G::XP) + (lane >> 2), q = lane & 3;
            const int ph = n >> 5, rr = (n >> 4) & 1, c = n & 15;
            v_on[i] = true;
            v_off[i] = (unsigned)(((ph >> 1) * P.g_py + (ph & 1) * P.g_px + rr * P.g_sy + c * P.g_sx + co0 +
                                   8 * q) * 2);
        }
    }
    const __amdgpu_buffer_rsrc_t gres =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.gout16, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S.p16 - ((long long)S.sy + S.sx)), 0, 0x7fffffff, 0x00020000);

    int k_bx = blk0 % nbx, k_by = (blk0 / nbx) % nby, k_b = blk0 / (nbx * nby);
    auto issue = [&](int stage_idx) {
        const int oy = 2 * k_by, ox = 16 * k_bx;
        const int a_so = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * P.g_sb + (long long)oy * P.g_sy + (long long)ox * P.g_sx) * 2));
        const int b_so = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S.sb + (long long)oy * S.sy + (long long)ox * S.sx) * 2));
        if (++k_bx == nbx) {
            k_bx = 0;
            if (++k_by == nby) {
                k_by = 0;
                ++k_b;
            }
        }
        unsigned char *st = smem + stage_idx * G::STAGE;
#pragma unroll
        for (int i = 0; i < G::LPW; ++i) {
            const int p = wave + 4 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            if (p < G::XP) {
                const bool ok = v_on[i] & ((unsigned)(oy + v_dy[i]) < (unsigned)H) &
                                ((unsigned)(ox + v_dx[i]) < (unsigned)W);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sres, dst, 16, ok ? v_off[i] : WP_OOB, b_so, 0, 0);
            } else {    // gradient planes (always inside the frame) and the padding pieces
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, v_on[i] ? v_off[i] : WP_OOB, a_so, 0, 0);
            }
        }
    };

    f32x16 acc[4][G::NB];       // tap t = 2 p + q of this wave's phase, column block: D[co 32][ci 32]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nb][r] = 0.f;
    const int lrow = lane & 31, lh = lane >> 5;
    const bool do_bias = P.dbias != nullptr && bx == 0;
    float bsum = 0.f;
    // transposed fragment reads (wgrad2_twins_kernel): group tg = lane >> 4 covers channels
    // 16 (tg & 1) .. and pixels 8 (tg >> 1) + 4 rd ..; lane 4 tq + tp of the group addresses
    // pixel row tq, channels 4 tp .. 4 tp + 3
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int kpix = 8 * (tg >> 1) + tq;            // this lane's pixel within a 16-pixel K step (lo read)
    const int cbyte = 32 * (tg & 1) + 8 * tp;       // its bytes within a 32-channel block

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * G::STAGE;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {            // the block's two pixel rows: one K step each
            const unsigned char *ga = st + G::XP * 1024 + ((wave * 2 + rr) * 16 + kpix) * 64 + cbyte;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)ga);
            const s16x4 ahi =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ga + 4 * 64));
            const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
            if (do_bias) {
                const s16x8 v = __builtin_bit_cast(s16x8, fa);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += __builtin_bit_cast(float, (unsigned)(unsigned short)v[e] << 16);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // patch row rr + a + p, first patch column b + q
                const int n_lo = (rr + pa + (t >> 1)) * 18 + pb + (t & 1) + kpix, n_hi = n_lo + 4;
#pragma unroll
                for (int nb = 0; nb < G::NB; ++nb) {
                    const unsigned char *xl = st + n_lo * G::PXB + 64 * (nb ^ wp_swz<CT>(n_lo)) + cbyte;
                    const unsigned char *xh = st + n_hi * G::PXB + 64 * (nb ^ wp_swz<CT>(n_hi)) + cbyte;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)xl);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)xh);
                    const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t][nb], 0, 0, 0);
                }
            }
        }
    };

#pragma unroll
    for (int u = 0; u < G::NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int s0 = 0; s0 < nsteps; s0 += G::NS) {
#pragma unroll
        for (int u = 0; u < G::NS; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                if (st + G::NS - 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NS - 2) * G::LPW) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (st + G::NS - 1 < nsteps) issue((u + G::NS - 1) % G::NS);
                compute(u);
            }
        }
    }

    const int slab = wave * P.S + split;            // [phase][split], as wgrad2's blockIdx.z
    if (do_bias) {   // lanes l and l ^ 32 hold the two k halves of the same channel
        const float v = bsum + __shfl_xor(bsum, 32);
        if (lh == 0) P.dbias[(size_t)slab * P.Cout + co0 + lrow] = v;
    }
    wp_fold_store<G::NB>(P, acc, smem, wave, lane, split, co0, coff + c0);
#endif
}


// ---------------------------------------------------------------------------
// The same organisation in exact f32 (v_mfma_f32_32x32x2_f32, the benchmark's
// operand mode): wgrad2_kernel's column tiles reach 59-68 % of the f32 matrix
// peak on these stages (32 x 128 and 64 x 128 tiles re-read the gradient per
// column tile and pay a fragment read per MFMA).  Here the operands of a
// 2-row x 16-pixel block are staged once for all 16 (phase, tap) combinations
// and read as plain ds_read_b32 (lane = channel: conflict-free without
// transposition), 64 or 128 MFMAs per wave and stage.
//   stage = X patch 72 slots x 4 CT bytes + 4 phase planes x 32 px x 128 B,
//   double-buffered (56 | 72 KiB: two workgroups per CU)
// ---------------------------------------------------------------------------
template <int CT>
struct WPGeomF {
    static constexpr int PXB = 4 * CT;                  // bytes per patch pixel slot
    static constexpr int LPS = PXB / 16;                // lanes (16-byte chunks) per slot
    static constexpr int SPP = 64 / LPS;                // slots per 1 KiB piece
    static constexpr int XP = WP_XSLOT / SPP;           // pieces of the patch (9 | 18)
    static constexpr int GP = 16;                       // gradient planes: 4 phases x 32 px x 128 B
    static constexpr int PIECES = (XP + GP + 3) / 4 * 4;
    static constexpr int LPW = PIECES / 4;
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int NS = 2;
    static constexpr int NB = CT / 32;
};

template <int CT>
__global__ __launch_bounds__(CONV_NT) void wgrad_patch_f32_kernel(const WGradParams P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef WPGeomF<CT> G;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pa = wave >> 1, pb = wave & 1;            // this wave's output phase

    int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
    if (P.xcd) {    // as in wgrad_patch_twins_kernel
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned x = L & 7u, q = total >> 3, r = total & 7u;
        const unsigned t = x * q + min(x, r) + (L >> 3);
        bx = (int)(t % gx);
        by = (int)((t / gx) % gy);
        split = (int)(t / (gx * gy));
    }
    int s = 0;
    for (int i = 1; i < P.nsrc; ++i)
        if (bx >= P.tile_begin[i]) s = i;
    const GSrc &S = P.src[s];
    int coff = 0;
    for (int i = 0; i < s; ++i) coff += P.src[i].C;
    const int c0 = (bx - P.tile_begin[s]) * CT;
    const int co0 = by * 32;
    const int H = P.Hv, W = P.Wv;
    const int nbx = W / 16, nby = H / 2;
    const int nblocks = P.B * nby * nbx, bps = (nblocks + P.S - 1) / P.S;
    const int blk0 = split * bps;
    const int nsteps = max(0, min(nblocks, blk0 + bps) - blk0);

    unsigned v_off[G::LPW];
    int v_dy[G::LPW], v_dx[G::LPW];
    bool v_on[G::LPW];
#pragma unroll
    for (int i = 0; i < G::LPW; ++i) {
        const int p = wave + 4 * i;
        v_off[i] = WP_OOB;
        v_dy[i] = v_dx[i] = 0;
        v_on[i] = false;
        if (p < G::XP) {
            const int n = G::SPP * p + lane / G::LPS, q = lane % G::LPS;
            const int r = n / 18, c = n - 18 * r;
            v_on[i] = true;
            v_dy[i] = r - 1;
            v_dx[i] = c - 1;
            // 256-byte slots: the two pixels of a K step would meet in the same banks --
            // odd slots hold their 128-byte halves swapped
            const int qq = CT == 64 ? q ^ (8 * (n & 1)) : q;
            v_off[i] = (unsigned)((r * S.sy + c * S.sx + c0 + 4 * qq) * 4);
        } else if (p < G::XP + G::GP) {
            const int n = 8 * (p - G::XP) + (lane >> 3), q = lane & 7;
            const int ph = n >> 5, rr = (n >> 4) & 1, c = n & 15;
            v_on[i] = true;
            v_off[i] = (unsigned)(((ph >> 1) * P.g_py + (ph & 1) * P.g_px + rr * P.g_sy + c * P.g_sx + co0 +
                                   4 * q) * 4);
        }
    }
    const __amdgpu_buffer_rsrc_t gres =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.gout, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S.p - ((long long)S.sy + S.sx)), 0, 0x7fffffff, 0x00020000);

    int k_bx = blk0 % nbx, k_by = (blk0 / nbx) % nby, k_b = blk0 / (nbx * nby);
    auto issue = [&](int stage_idx) {
        const int oy = 2 * k_by, ox = 16 * k_bx;
        const int a_so = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * P.g_sb + (long long)oy * P.g_sy + (long long)ox * P.g_sx) * 4));
        const int b_so = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S.sb + (long long)oy * S.sy + (long long)ox * S.sx) * 4));
        if (++k_bx == nbx) {
            k_bx = 0;
            if (++k_by == nby) {
                k_by = 0;
                ++k_b;
            }
        }
        unsigned char *st = smem + stage_idx * G::STAGE;
#pragma unroll
        for (int i = 0; i < G::LPW; ++i) {
            const int p = wave + 4 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            if (p < G::XP) {
                const bool ok = ((unsigned)(oy + v_dy[i]) < (unsigned)H) & ((unsigned)(ox + v_dx[i]) < (unsigned)W);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sres, dst, 16, ok ? v_off[i] : WP_OOB, b_so, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, v_on[i] ? v_off[i] : WP_OOB, a_so, 0, 0);
            }
        }
    };

    f32x16 acc[4][G::NB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nb][r] = 0.f;
    const int lrow = lane & 31, lh = lane >> 5;
    const bool do_bias = P.dbias != nullptr && bx == 0;
    float bsum = 0.f;

    // fragment addresses relative to a stage: A = G_ab[pixel 2 kk + lh][co lrow], B = X[slot of
    // that pixel under tap t][ci 32 nb + lrow].  The parity of a slot (which half of a 256-byte
    // slot holds which channels) is a per-lane constant: 18 and 2 kk are even.
    const int par = (pb + lh) & 1;
    int xaddr[4][G::NB];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
            xaddr[t][nb] = ((pa + (t >> 1)) * 18 + pb + (t & 1) + lh) * G::PXB +
                           (CT == 64 ? 128 * (nb ^ par ^ (t & 1)) : 0) + 4 * lrow;
    const int gaddr = G::XP * 1024 + (wave * 32 + lh) * 128 + 4 * lrow;
    constexpr int KU = 2;               // K steps (of 2 pixels) per register batch
    constexpr int NBATCH = 2 * 8 / KU;  // batches per stage: (row rr, kk0)

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * G::STAGE;
        float av[2][KU], bv[2][KU][4][G::NB];
        auto fetch = [&](int buf, int batch) {
            const int rr = batch / (8 / KU), kk0 = (batch % (8 / KU)) * KU;
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                av[buf][j] = *(const float *)(st + gaddr + (rr * 16 + 2 * (kk0 + j)) * 128);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nb = 0; nb < G::NB; ++nb)
                        bv[buf][j][t][nb] =
                            *(const float *)(st + xaddr[t][nb] + (rr * 18 + 2 * (kk0 + j)) * G::PXB);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int batch = 0; batch < NBATCH; ++batch) {
            const int buf = batch & 1;
            if (batch + 1 < NBATCH) fetch(buf ^ 1, batch + 1);   // in flight under this batch's MFMAs
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                if (do_bias) bsum += av[buf][j];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nb = 0; nb < G::NB; ++nb)
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][j], bv[buf][j][t][nb],
                                                                          acc[t][nb], 0, 0, 0);
            }
        }
    };

    if (nsteps > 0) issue(0);
    for (int s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (st + 1 < nsteps) issue(u ^ 1);
                compute(u);
            }
        }
    }

    const int slab = wave * P.S + split;
    if (do_bias) {
        const float v = bsum + __shfl_xor(bsum, 32);
        if (lh == 0) P.dbias[(size_t)slab * P.Cout + co0 + lrow] = v;
    }
    wp_fold_store<G::NB>(P, acc, smem, wave, lane, split, co0, coff + c0);
#endif
}

}  // namespace

// Decoder stages in the bf16-twins mode: four sub-pixel phases of 2x2 taps over vector members
// whose channel counts are multiples of 32, 32 | Cout, 16 | width (DVSOF_NO_WGRAD_PATCH=1: the
// column-tile kernel)
// (the shape alone: what the workspace is sized for, whether or not the twins are bound yet)
bool wgrad_patch_shape_ok(const WGradParams &P)
{
    static const bool off = getenv("DVSOF_NO_WGRAD_PATCH") != nullptr;
    if (off || P.nph != 4 || P.ks != 2 || P.stride != 1 || P.up != UP_NONE) return false;
    if (P.ph_pad != 1 || P.pad != 1 || P.src_ph_stride != 0) return false;
    if ((P.Cout & 31) || (P.Wo % 16) || (P.Ho & 1) || P.Ho != P.Hv || P.Wo != P.Wv) return false;
    // flat members (the 2-channel flow of a decoder stage) are not this kernel's: their
    // columns belong to the caller (dvsof_flow_fold_grads) or to the flat-member kernels
    int nvec = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) continue;
        if (P.src[s].sc != 1 || (P.src[s].C & 31)) return false;
        ++nvec;
    }
    return nvec >= 1;
}

// exact-f32 operand mode: wgrad_patch_f32_kernel (DVSOF_NO_WGRAD_PATCH_F32=1: wgrad2_kernel)
static bool wp_f32(const WGradParams &P)
{
    static const bool off = getenv("DVSOF_NO_WGRAD_PATCH_F32") != nullptr;
    return !off && !P.twins && P.mfma_bf16 == 0;
}

bool wgrad_patch_eligible(const WGradParams &P)
{
    if (!wgrad_patch_shape_ok(P)) return false;
    if (wp_f32(P)) {
        if (!P.gout || (reinterpret_cast<uintptr_t>(P.gout) & 15)) return false;
        if ((P.g_sb | P.g_sy | P.g_sx | P.g_py | P.g_px) & 3) return false;
        for (int s = 0; s < P.nsrc; ++s)
            if (!P.src[s].flat && (!P.src[s].p || (reinterpret_cast<uintptr_t>(P.src[s].p) & 15) ||
                                   ((P.src[s].sb | P.src[s].sy | P.src[s].sx) & 3)))
                return false;
        return true;
    }
    if (!P.twins || !P.gout16) return false;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && !P.src[s].p16) return false;
    return true;
}

// Bound on the K splits: >= 2 stages per split; a slab is a whole phase-form gradient -- at
// most ~32 MB of partial sums per layer, and no more than DVSOF_WGRAD_PATCH_MAXS (64) slabs
// per phase (the fold reads them all)
static long long wp_max_splits(const WGradParams &P)
{
    const long long blocks = (long long)P.B * (P.Hv / 2) * (P.Wv / 16);
    long long maxS = blocks / 2 > 0 ? blocks / 2 : 1;
    const long long slab_bytes = 4LL * P.Cout * 4 * P.Cin_tot * 4;
    long long capS = (32LL << 20) / (slab_bytes > 0 ? slab_bytes : 1);
    static const int max_env = getenv("DVSOF_WGRAD_PATCH_MAXS") ? atoi(getenv("DVSOF_WGRAD_PATCH_MAXS")) : 64;
    if (capS > max_env) capS = max_env;
    if (capS < 1) capS = 1;
    return maxS < capS ? maxS : capS;
}

// 64 input channels per workgroup halve the gradient planes' re-reads
// (DVSOF_WGRAD_PATCH_CT = 32 | 64 forces one where every vector member allows it)
static int wp_channel_tile(const WGradParams &P)
{
    static const int force = getenv("DVSOF_WGRAD_PATCH_CT") ? atoi(getenv("DVSOF_WGRAD_PATCH_CT")) : 0;
    long long ct = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        if (P.src[s].flat) continue;
        if (P.src[s].C & 63) return 32;
        ct += P.src[s].C / 64;
    }
    if (force == 32 || force == 64) return force;
    // exact f32: matrix-bound once the planes are read half as often -- as long as one
    // workgroup per CU remains
    if (wp_f32(P)) return (P.Cout / 32) * ct * wp_max_splits(P) >= 256 ? 64 : 32;
    // bf16 twins, measured (batch 8, the four decoder stages): 32 wins everywhere -- the slab
    // bound on the K splits leaves the 64-channel form with 224-256 workgroups
    return 32;
}

// wgrad_min.hip: the nine-product form of the same gradient (exact f32)
bool wgrad_min_ok(const WGradParams &P);
int wgrad_min_splits(const WGradParams &P);
int wgrad_min_launch(WGradParams &P, hipStream_t st);

// K splits for this kernel: enough workgroups for two per CU
int wgrad_patch_splits(const WGradParams &P)
{
    if (wp_f32(P) && wgrad_min_ok(P)) return wgrad_min_splits(P);
    const int CT = wp_channel_tile(P);
    long long tiles = (long long)(P.Cout / 32);
    long long ct = 0;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat) ct += P.src[s].C / CT;
    tiles *= ct;
    static const int target16 = getenv("DVSOF_WGRAD_PATCH_WGS") ? atoi(getenv("DVSOF_WGRAD_PATCH_WGS")) : 512;
    static const int target32 = getenv("DVSOF_WGRAD_PATCH_F32_WGS") ? atoi(getenv("DVSOF_WGRAD_PATCH_F32_WGS")) : 512;
    const int target = wp_f32(P) ? target32 : target16;
    long long S = (target + tiles - 1) / tiles;
    const long long maxS = wp_max_splits(P);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return (int)S;
}

template <int CT>
static int wp_launch(WGradParams &P, hipStream_t st)
{
    int nt = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        P.tile_begin[s] = nt;
        if (!P.src[s].flat) nt += P.src[s].C / CT;
    }
    P.tile_begin[P.nsrc] = nt;
    constexpr size_t LDS0 = (size_t)WPGeom<CT>::NS * WPGeom<CT>::STAGE;
    constexpr size_t LDS = LDS0 < 65536 ? 65536 : LDS0;     // the fold of the epilogue: 16 tiles of 4 KiB
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_patch_twins_kernel<CT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid(nt, P.Cout / 32, P.S);
    static const bool xcd_off = getenv("DVSOF_WGRAD_XCD") && atoi(getenv("DVSOF_WGRAD_XCD")) == 0;
    P.xcd = xcd_off ? 0 : 1;
    hipLaunchKernelGGL(wgrad_patch_twins_kernel<CT>, grid, dim3(CONV_NT), LDS, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

template <int CT>
static int wp_launch_f32(WGradParams &P, hipStream_t st)
{
    int nt = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        P.tile_begin[s] = nt;
        if (!P.src[s].flat) nt += P.src[s].C / CT;
    }
    P.tile_begin[P.nsrc] = nt;
    constexpr size_t LDS0 = (size_t)WPGeomF<CT>::NS * WPGeomF<CT>::STAGE;
    constexpr size_t LDS = LDS0 < 65536 ? 65536 : LDS0;     // the fold of the epilogue: 16 tiles of 4 KiB
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_patch_f32_kernel<CT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid(nt, P.Cout / 32, P.S);
    static const bool xcd_off = getenv("DVSOF_WGRAD_XCD") && atoi(getenv("DVSOF_WGRAD_XCD")) == 0;
    P.xcd = xcd_off ? 0 : 1;
    hipLaunchKernelGGL(wgrad_patch_f32_kernel<CT>, grid, dim3(CONV_NT), LDS, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int wgrad_patch_launch(const WGradParams &P0, hipStream_t st)
{
    WGradParams P = P0;
    if (wp_f32(P) && wgrad_min_ok(P)) return wgrad_min_launch(P, st);
    const int ct = wp_channel_tile(P);
    if (wp_f32(P)) return ct == 64 ? wp_launch_f32<64>(P, st) : wp_launch_f32<32>(P, st);
    return ct == 64 ? wp_launch<64>(P, st) : wp_launch<32>(P, st);
}
