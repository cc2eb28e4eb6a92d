// Weight gradient of a 2x up-sampling decoder stage with NINE products per
// low-resolution pixel instead of sixteen (exact f32, v_mfma_f32_32x32x2_f32).
//
// `nearest-up2 -> conv3x3(pad 1)` (the EV_FlowNet decoder; reference call site
// utils/training.py:158 through the absent EV_FlowNet.net) is, per axis,
//     y[2i]   = w0 x[i-1] + (w1 + w2) x[i]
//     y[2i+1] = (w0 + w1) x[i] + w2 x[i+1]
// -- the sub-pixel form (csrc/gconv2.hip, wgrad_patch.hip): four products for
// two outputs.  The four effective weights span only three dimensions
// ((w0 + w1) = w0 + (w1 + w2) - w2), and three products suffice:
//     m0 = w0 (x[i-1] - x[i]),  m1 = (w0 + w1 + w2) x[i],  m2 = w2 (x[i+1] - x[i])
//     y[2i] = m0 + m1,   y[2i+1] = m1 + m2
// (a Toom-Cook / Winograd style minimal algorithm for this particular pair of
// 2-tap filters; in 2-D 3 x 3 = 9 products for the 2 x 2 outputs of a
// low-resolution pixel).  With G = [[1,0,0],[1,1,1],[0,0,1]] the transformed
// weights are Wt = G w G^T (the same nine numbers per (co, ci) pair, well
// conditioned: sums of at most nine weights), the transformed input
// Xt[p][q] = D_p(rows) D_q(columns) x with D = (x[-1] - x[0], x[0], x[+1] - x[0]),
// and the weight gradient is
//     dWt[p][q] = sum over pixels of GM[p][q]^T Xt[p][q],   dw = G^T dWt G
//     GM[p][q]  = sum of the gradient's phase planes a in A(p), b in A(q),
//                 A(0) = {0}, A(1) = {0, 1}, A(2) = {1}
// nine GEMMs [Cout x pixels] x [pixels x Cin] instead of sixteen.  Checked
// against autograd through interpolate + conv2d in tests/test_gpu_conv.py.
//
// Kernel: the staging of wgrad_patch_f32_kernel (ONE 4 x 18-pixel input patch
// and the four phase planes of the gradient per block of 2 rows x 16
// low-resolution pixels, L2 -> LDS by LDS-DMA, lane = channel fragment reads)
// with a different split of the work.  A workgroup owns 32 output channels x
// 64 input channels x all nine components; its EIGHT waves are (column block
// nb of 32 input channels) x (quarter kq of the block's 16 K steps): every
// wave holds the nine 32 x 32 accumulators of its column block and sums its
// share of the pixels -- K splits inside the workgroup are free for a weight
// gradient, the waves meet only once, after the last block.  Per K step a wave
// reads 4 + 9 fragments and makes the nine operand pairs with 5 + 12 adds (the
// f32 matrix instruction runs on the vector ALUs: ~17 x 4 of 9 x 64 cycles).
// Epilogue: the accumulators meet in LDS one component row p at a time
// (8 waves x 3 tiles = 96 KiB), a wave adds the four K quarters of its share,
// applies G^T . G in registers and writes the 3 x 3 gradient slab
// [split][Cout][3][3][Cin_tot] that the plain slab reduce consumes -- the
// layout wgrad_patch's fold writes.
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned WM_OOB = 0x80000000u;
constexpr int WM_NT = 512;              // 8 waves
constexpr int WM_CT = 64;               // input channels per workgroup
constexpr int WM_PXB = 4 * WM_CT;       // bytes per patch pixel slot
constexpr int WM_XP = 18;               // patch pieces: 72 slots x 256 B
constexpr int WM_GP = 16;               // gradient planes: 4 phases x 32 px x 128 B
constexpr int WM_LPW = 5;               // loads per wave and stage (40 pieces, 34 real)
constexpr int WM_STAGE = 8 * WM_LPW * 1024;
constexpr int WM_NS = 3;
constexpr int WM_LDS = WM_NS * WM_STAGE;    // 120 KiB; the epilogue uses 96 KiB of it

__global__ __launch_bounds__(WM_NT) void wgrad_min_f32_kernel(const WGradParams P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wave >> 2, kq = wave & 3;            // column block; quarter of a block's K steps
    const int rr = kq >> 1, kbase = 4 * (kq & 1);       // pixel row of the block, first K step

    int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
    if (P.xcd) {    // as in wgrad_patch_f32_kernel: an XCD owns a contiguous range of (split, tile) pairs
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
    const int c0 = (bx - P.tile_begin[s]) * WM_CT;
    const int co0 = by * 32;
    const int H = P.Hv, W = P.Wv;
    const int nbx = W / 16, nby = H / 2;
    const int nblocks = P.B * nby * nbx, bps = (nblocks + P.S - 1) / P.S;
    const int blk0 = split * bps;
    const int nsteps = max(0, min(nblocks, blk0 + bps) - blk0);

    // load slots: piece p = wave + 8 i; a lane fetches one 16-byte chunk of one pixel slot
    unsigned v_off[WM_LPW];
    int v_dy[WM_LPW], v_dx[WM_LPW];
#pragma unroll
    for (int i = 0; i < WM_LPW; ++i) {
        const int p = wave + 8 * i;
        v_off[i] = WM_OOB;
        v_dy[i] = v_dx[i] = 0;
        if (p < WM_XP) {
            const int n = 4 * p + (lane >> 4), q = lane & 15;
            const int r = n / 18, c = n - 18 * r;
            v_dy[i] = r - 1;
            v_dx[i] = c - 1;
            // odd slots hold their 128-byte halves swapped (wgrad_patch_f32_kernel's layout)
            const int qq = q ^ (8 * (n & 1));
            v_off[i] = (unsigned)((r * S.sy + c * S.sx + c0 + 4 * qq) * 4);
        } else if (p < WM_XP + WM_GP) {
            const int n = 8 * (p - WM_XP) + (lane >> 3), q = lane & 7;
            const int ph = n >> 5, r = (n >> 4) & 1, c = n & 15;
            v_off[i] = (unsigned)(((ph >> 1) * P.g_py + (ph & 1) * P.g_px + r * P.g_sy + c * P.g_sx + co0 +
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
        unsigned char *st = smem + stage_idx * WM_STAGE;
#pragma unroll
        for (int i = 0; i < WM_LPW; ++i) {
            const int p = wave + 8 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            if (p < WM_XP) {
                const bool ok = ((unsigned)(oy + v_dy[i]) < (unsigned)H) & ((unsigned)(ox + v_dx[i]) < (unsigned)W);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sres, dst, 16, ok ? v_off[i] : WM_OOB, b_so, 0, 0);
            } else {    // gradient planes (always inside the frame) and the padding pieces
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, v_off[i], a_so, 0, 0);
            }
        }
    };

    f32x16 acc[3][3];       // component (p, q): D[co 32][ci 32] of this wave's column block
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
    const int lrow = lane & 31, lh = lane >> 5;
    const bool do_bias = P.dbias != nullptr && bx == 0 && nb == 0;
    float bsum = 0.f;

    // fragment addresses relative to a stage.  A: G_ab[pixel 2 kk + lh of row rr][co lrow];
    // B: X[slot (rr + r) 18 + 2 kk + lh + c][ci 32 nb + lrow], r, c = 0..2 -- the 3 x 3
    // neighbourhood of that pixel; the half of a 256-byte slot that holds the column block
    // depends on the slot's parity = (lh + c) & 1 (18 and 2 kk are even)
    const int gaddr = WM_XP * 1024 + (rr * 16 + 2 * kbase + lh) * 128 + 4 * lrow;
    const int xrow = (rr * 18 + 2 * kbase + lh) * WM_PXB + 4 * lrow;
    const int xeven = xrow + 128 * (nb ^ (lh & 1)), xodd = xrow + 128 * (nb ^ ((lh + 1) & 1));

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * WM_STAGE;
        float g[2][4], x[2][9];
#ifndef WM_PROBE
#define WM_PROBE 0      // timing probes (variant builds; results wrong): 1 no DMA behind the ring's first fill,
#endif                  // 2 one gradient and one input fragment read per K step instead of 13, 4 no stage barrier.
                        // Per stage with its slab reduce, us: 117 112 107 107 | 1: 110 101 97 97 | 2: 103 95 93 94 |
                        // 4: 110 103 99 99 | 7: 92 83 83 83: reads 13 %, DMA 8 %, barrier 7 % -- no single cost
        auto fetch = [&](int buf, int j) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
                g[buf][ph] = *(const float *)(st + gaddr + (((WM_PROBE & 2) ? 0 : ph) * 32 + 2 * j) * 128);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    x[buf][3 * r + c] = *(const float *)(st + (((WM_PROBE & 2) ? 0 : (c & 1)) ? xodd : xeven) +
                                                         (((WM_PROBE & 2) ? 0 : r * 18 + c) + 2 * j) * WM_PXB);
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int buf = j & 1;
            if (j + 1 < 4) fetch(buf ^ 1, j + 1);       // in flight under this step's MFMAs
            // gradient side: phase planes (a, b) -> GM[p][q]
            const float g00 = g[buf][0], g01 = g[buf][1], g10 = g[buf][2], g11 = g[buf][3];
            float gm[3][3];
            gm[0][0] = g00;
            gm[0][2] = g01;
            gm[2][0] = g10;
            gm[2][2] = g11;
            gm[0][1] = g00 + g01;
            gm[2][1] = g10 + g11;
            gm[1][0] = g00 + g10;
            gm[1][2] = g01 + g11;
            gm[1][1] = gm[0][1] + gm[2][1];
            if (do_bias) bsum += gm[1][1];
            // input side: rows first, then columns
            float rw[3][3], xt[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rw[0][c] = x[buf][c] - x[buf][3 + c];
                rw[1][c] = x[buf][3 + c];
                rw[2][c] = x[buf][6 + c] - x[buf][3 + c];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                xt[p][0] = rw[p][0] - rw[p][1];
                xt[p][1] = rw[p][1];
                xt[p][2] = rw[p][2] - rw[p][1];
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(gm[p][q], xt[p][q], acc[p][q], 0, 0, 0);
        }
    };

    // ring of 3: two stages in flight
#pragma unroll
    for (int u = 0; u < WM_NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int s0 = 0; s0 < nsteps; s0 += WM_NS) {
#pragma unroll
        for (int u = 0; u < WM_NS; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                if (st + WM_NS - 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WM_NS - 2) * WM_LPW) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!(WM_PROBE & 4) || st == 0) __builtin_amdgcn_s_barrier();
                if (st + WM_NS - 1 < nsteps && (!(WM_PROBE & 1) || st == 0)) issue((u + WM_NS - 1) % WM_NS);
                compute(u);
            }
        }
    }

    if (do_bias) {   // lanes l and l ^ 32 hold the two k halves of the same channel; rows [kq][split]
        const float v = bsum + __shfl_xor(bsum, 32);
        if (lh == 0) P.dbias[(size_t)(kq * P.S + split) * P.Cout + co0 + lrow] = v;
    }

    // ---- epilogue: K quarters meet, dw = G^T dWt G, slab [split][Cout][3][3][Cin_tot]
    float *xs = (float *)smem;      // [wave 8][q 3][reg 16][lane 64]
    const int onb = wave >> 2, orq = wave & 3;      // this wave's share: column block, registers 4 orq ..
    float out[3][3][4];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[ky][kx][e] = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        __builtin_amdgcn_s_barrier();       // last stage read / previous pass read
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) xs[((wave * 3 + q) * 16 + reg) * 64 + lane] = acc[p][q][reg];
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int reg = 4 * orq + e;
            float v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float *src = xs + (((onb * 4) * 3 + q) * 16 + reg) * 64 + lane;
                v[q] = ((src[0] + src[3 * 16 * 64]) + src[2 * 3 * 16 * 64]) + src[3 * 3 * 16 * 64];
            }
            const float t[3] = {v[0] + v[1], v[1], v[1] + v[2]};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (p <= 1) out[0][kx][e] += t[kx];     // kernel row 0 <- components 0, 1
                if (p == 1) out[1][kx][e] += t[kx];     // 1 <- 1
                if (p >= 1) out[2][kx][e] += t[kx];     // 2 <- 1, 2
            }
        }
    }
    const size_t row9 = (size_t)9 * P.Cin_tot;
    float *dW = P.dW + (size_t)split * P.Cout * row9;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int reg = 4 * orq + e;
        const int co = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                dW[(size_t)co * row9 + (size_t)(3 * ky + kx) * P.Cin_tot + coff + c0 + 32 * onb + lrow] =
                    out[ky][kx][e];
    }
#endif
}

}  // namespace

// wgrad_patch.hip routes the exact-f32 decoder stages here when every vector member has a
// multiple of 64 channels (DVSOF_NO_WGRAD_MIN=1: the sixteen-product patch kernel)
bool wgrad_min_ok(const WGradParams &P)
{
    static const bool off = getenv("DVSOF_NO_WGRAD_MIN") != nullptr;
    if (off || P.twins || P.mfma_bf16 != 0) return false;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && (P.src[s].C & 63)) return false;
    return true;
}

// K splits: one workgroup (8 waves) per CU; >= 2 blocks per split; <= 128 slabs
int wgrad_min_splits(const WGradParams &P)
{
    long long tiles = 0;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat) tiles += P.src[s].C / WM_CT;
    tiles *= P.Cout / 32;
    static const int target = getenv("DVSOF_WGRAD_MIN_WGS") ? atoi(getenv("DVSOF_WGRAD_MIN_WGS")) : 256;
    long long S = (target + tiles - 1) / (tiles > 0 ? tiles : 1);
    const long long blocks = (long long)P.B * (P.Hv / 2) * (P.Wv / 16);
    if (S > blocks / 2) S = blocks / 2;
    if (S > 128) S = 128;
    if (S < 1) S = 1;
    return (int)S;
}

int wgrad_min_launch(WGradParams &P, hipStream_t st)
{
    int nt = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        P.tile_begin[s] = nt;
        if (!P.src[s].flat) nt += P.src[s].C / WM_CT;
    }
    P.tile_begin[P.nsrc] = nt;
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_min_f32_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, WM_LDS));
        attr_set = true;
    }
    dim3 grid(nt, P.Cout / 32, P.S);
    static const bool xcd_off = getenv("DVSOF_WGRAD_XCD") && atoi(getenv("DVSOF_WGRAD_XCD")) == 0;
    P.xcd = xcd_off ? 0 : 1;
    hipLaunchKernelGGL(wgrad_min_f32_kernel, grid, dim3(WM_NT), WM_LDS, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
