// Data gradient of a 2x up-sampling decoder stage with NINE products per
// low-resolution pixel instead of sixteen (exact f32, v_mfma_f32_32x32x2_f32).
//
// The adjoint of `nearest-up2 -> conv3x3(pad 1)` (EV_FlowNet decoder; reference
// call site utils/training.py:158 through the absent EV_FlowNet.net).  Per axis
// the forward is y[2i] = a x[i-1] + b x[i], y[2i+1] = c x[i] + d x[i+1] with
// a = w0, b = w1 + w2, c = w0 + w1 = a + b - d, d = w2 (csrc/wgrad_min.hip), so
//     gx[i] = b g[2i] + c g[2i+1] + a g[2i+2] + d g[2i-1]
//           = b (g[2i] + g[2i+1]) + a (g[2i+1] + g[2i+2]) + d (g[2i-1] - g[2i+1])
// three products instead of four, all of them gathered AT the output pixel (the
// sub-pixel form's 4x4 stride-2 kernel has sixteen taps in 2-D, this has nine):
//     gx[i][j] = sum_{p,q} W'[p][q]^T S[p][q][i][j]
//     S[p][q]  = R_p(rows) C_q(columns) of the 4 x 4 fine-resolution window
//                g[2i-1 .. 2i+2][2j-1 .. 2j+2], R = C = (g1 + g2, g2 + g3, g0 - g2)
//     W'       = G' w G'^T,  G' = [[0,1,1],[1,0,0],[0,0,1]]
// g is zero outside the frame.  All nine components accumulate into ONE
// accumulator per output tile (a K concatenation): no output transform.
//
// Kernel: a workgroup of 8 waves owns 64 input channels x a block of 8 rows x
// 16 columns of low-resolution pixels and walks the output channels (K) in
// chunks of 16: per chunk the 18 x 34 fine-resolution patch of the gradient
// and the nine 64 x 16 weight tiles go L2 -> LDS by LDS-DMA (double buffered).
// A wave is (pixel tile of 2 rows x 16 pixels, K half): per group of 4 channels
// it reads the 16 window fragments (ds_read_b64: two K steps each), makes the
// nine S pairs with 21 packed additions -- shared by its TWO 32-channel output
// tiles -- reads 18 weight fragments and issues 36 matrix instructions.  LDS
// images as in fwd_min.hip (rows dealt to arrays by residue, arrays padded by
// 16 bytes: no swizzle, the channel group is an immediate offset); the fine
// patch is cut into even and odd columns because a fragment read walks it with
// stride 2.  Epilogue: the two K halves meet in LDS (a wave hands its partner the
// tile it does not finish), each wave turns its tile to pixel-major through 4 KB
// of its own; 8 lanes store one pixel's 128 bytes into the member (x or skip) the
// channel tile belongs to: + addend(s), + a folded flow head's term, x act'(actsrc).
// Finest stage (32 output channels = two chunks): the weights stay in LDS and a
// workgroup walks the pixel blocks of its channel tile (IPW = 0 below).
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned DM_OOB = 0x80000000u;
constexpr int DM_NT = 512;
constexpr int DM_NR = 8;                         // low-resolution rows per block (x 16 columns)
constexpr int DM_AARR = 9 * 1024 + 16;           // weight array: 144 rows of 64 B (9 components x 16)
constexpr int DM_AP = 4 * 9;                     // weight pieces
constexpr int DM_PR = 17;                        // column pairs per fine row (34 columns)
constexpr int DM_NPAIR = (2 * DM_NR + 2) * DM_PR;    // 306
constexpr int DM_BPA = 5;                        // pieces per patch array (77 rows of 64 B)
constexpr int DM_BARR = DM_BPA * 1024 + 16;
constexpr int DM_BOFF = 4 * DM_AARR;
constexpr int DM_BP = 8 * DM_BPA;                // 2 planes x 4 arrays
constexpr int DM_NPIECE = DM_AP + DM_BP;         // 76
constexpr int DM_LPW = (DM_NPIECE + 7) / 8;      // 10 rounds; the last one is partial
constexpr int DM_STAGE = ((DM_BOFF + 8 * DM_BARR + 1023) / 1024) * 1024;
constexpr int DM_LDS = 2 * DM_STAGE;

// W'[3 p + q][ci][co] = sum_{k,l} G'[p][k] G'[q][l] w[co][k][l][ci]: rows of G' pick the raw taps
// {1,2}, {0}, {2}
__global__ __launch_bounds__(256) void min9_dgrad_weights_kernel(const float *__restrict__ w, float *__restrict__ wq,
                                                                 int Cout, int Ctot)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z, p = z / 3, q = z - 3 * p;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + lx;
        float v = 0.f;
        if (co < Cout && ci < Ctot) {
            const float *k = w + (size_t)co * 9 * Ctot + ci;
            auto col = [&](int kx) -> float {   // rows first
                const float *kc = k + (size_t)kx * Ctot;
                return p == 0 ? kc[3 * Ctot] + kc[6 * Ctot] : p == 1 ? kc[0] : kc[6 * Ctot];
            };
            v = q == 0 ? col(1) + col(2) : q == 1 ? col(0) : col(2);
        }
        tile[r][lx] = v;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + lx;
        if (ci < Ctot && co < Cout) wq[((size_t)z * Ctot + ci) * Cout + co] = tile[lx][r];
    }
}

template <int IPW>
__global__ __launch_bounds__(DM_NT) void dgrad_min_f32_kernel(const GConvParams P, const int total)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = wave & 3, ks = wave >> 2;
    const int lrow = lane & 31, lh = lane >> 5;
    const int H = P.Ho, W = P.Wo;                   // low-resolution frame (the outputs)
    const int FH = 2 * H, FW = 2 * W;               // fine frame (the gradient)
    const int Cout = P.Cin_tot, Ctot = P.N;         // K = output channels of the layer
    const int nbx = W / 16, nby = H / DM_NR;
    const int nct = Ctot / 64;
    const int nchunks = Cout / 16;

    // A workgroup walks IPW consecutive ITEMS = (block of pixels, 64-channel tile), channel
    // tile fastest (XCD-aware: an XCD owns a contiguous range of workgroups; the channel tiles of
    // a block read the same patch through one L2).  The LDS ring runs on across the items: the
    // first chunk of item n + 1 is in flight while item n's last chunk is multiplied and its
    // epilogue runs -- with 2-4 chunks of K per item (the fine stages: 32 | 64 output channels)
    // a workgroup per item spent a third of its life waiting for its first chunk with nothing
    // else resident on the CU (151 | 121 us against 107 for the coarsest stage's 16 chunks).
    // The item loop is unrolled (IPW = 1 | 2): as a loop, or unrolled four times, the compiler
    // keeps every item's address arithmetic alive across the others (256 registers + 50-111
    // spilled against 185 | 230).
    unsigned wg = blockIdx.x;
    {
        const unsigned tot = gridDim.x, x = wg & 7u, q = tot >> 3, r = tot & 7u;
        wg = x * q + min(x, r) + (wg >> 3);
    }
    // IPW = 0, RESIDENT weights (two chunks of K only: 32 output channels, the finest stage): a
    // workgroup keeps ONE channel tile and walks pixel blocks r0, r0 + rows, ...; the weights of
    // chunk 0 / 1 stay in stage 0 / 1 for all of them and only the patch is fetched per item --
    // half of an item's LDS-DMA bytes (74 of 152 KB), which at 2.6 us of matrix work per item is
    // what an item waits for.  The epilogue's exchange then lives in stage 1's PATCH area.
    constexpr bool RES = IPW == 0;
    const int rows = RES ? (int)gridDim.x / nct : 1;
    const int item0 = RES ? ((int)wg / nct) * nct + (int)wg % nct : (int)wg * IPW;
    const int istep = RES ? rows * nct : 1;
    const int item1 = RES ? total : item0 + IPW;       // (the launch makes IPW divide `total`)

    const GSrc &GS = P.src[0];
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void *)P.W, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(GS.p - ((long long)GS.sy + GS.sx)), 0, 0x7fffffff, 0x00020000);

    // ---- DMA roles: piece p = wave + 8 i; lane l lands at piece base + 16 l = row l >> 2, quarter l & 3
    // weight pieces: offsets relative to the item's channel tile (wbase); patch pieces: relative to
    // the item's block (gbase), with the frame test of that block
    unsigned off[DM_LPW];
    int ci0 = 0, b = 0, oy = 0, ox = 0, wbase = 0, gbase = 0, blk_ = 0;      // of the item being ISSUED
    auto setup = [&](int item) {
        const int ct64 = item % nct, blk = item / nct;
        blk_ = blk;
        ci0 = 64 * ct64;
        const int bx = blk % nbx, by = (blk / nbx) % nby;
        b = blk / (nbx * nby);
        oy = DM_NR * by;
        ox = 16 * bx;
        wbase = __builtin_amdgcn_readfirstlane((int)((size_t)ci0 * Cout * 4));
        gbase = __builtin_amdgcn_readfirstlane(
            (int)(((long long)b * GS.sb + (long long)(2 * oy) * GS.sy + (long long)(2 * ox) * GS.sx) * 4));
#pragma unroll
        for (int i = 0; i < DM_LPW; ++i) {
            const int p = wave + 8 * i;
            if (p < DM_AP) {
                const int m = p / 9, pp = p - 9 * m;
                const int j = 16 * pp + (lane >> 2);        // row within array m: (component, ci >> 2)
                const int c = j >> 4, r = 4 * (j & 15) + m;
                off[i] = (unsigned)((((size_t)c * Ctot + r) * Cout + 4 * (lane & 3)) * 4);
            } else if (p < DM_NPIECE) {
                off[i] = DM_OOB;
                const int pb = p - DM_AP;
                const int arr = pb / DM_BPA, pp = pb - DM_BPA * arr;
                const int pr = 4 * (16 * pp + (lane >> 2)) + (arr & 3);     // column pair
                if (pr < DM_NPAIR) {
                    const int n = 2 * pr + (arr >> 2);      // fine slot: row n / 34, column n % 34
                    const int fr = n / 34, fc = n - 34 * fr;
                    // (the resource's base is shifted by (-1, -1): offsets stay non-negative)
                    if (((unsigned)(2 * oy + fr - 1) < (unsigned)FH) & ((unsigned)(2 * ox + fc - 1) < (unsigned)FW))
                        off[i] = (unsigned)((fr * GS.sy + fc * GS.sx + 4 * (lane & 3)) * 4);
                }
            } else {
                off[i] = DM_OOB;
            }
        }
    };

    bool with_w = true;     // (resident mode: false once both stages hold their weights)
    auto issue = [&](int stage_idx, int ch) {
        unsigned char *st = smem + stage_idx * DM_STAGE;
        // (scalar offsets pinned to SGPRs here: as loop-carried values the compiler takes them
        // for divergent and wraps every LDS-DMA in a waterfall loop)
        const int so_w = __builtin_amdgcn_readfirstlane(wbase + ch * 64);
        const int so_g = __builtin_amdgcn_readfirstlane(gbase + ch * 64);
#pragma unroll
        for (int i = 0; i < DM_LPW; ++i) {
            const int p = wave + 8 * i;
            if (p < DM_AP) {
                if (RES && !with_w) continue;
                const int m = p / 9, pp = p - 9 * m;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(st + m * DM_AARR + pp * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, dst, 16, off[i], so_w, 0, 0);
            } else if (p < DM_NPIECE) {
                const int pb = p - DM_AP;
                const int arr = pb / DM_BPA, pp = pb - DM_BPA * arr;
                __attribute__((address_space(3))) void *dst =
                    (__attribute__((address_space(3))) void *)(st + DM_BOFF + arr * DM_BARR + pp * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, off[i], so_g, 0, 0);
            }
        }
    };

    // ---- fragment addresses (relative to a stage)
    // A: weight row (component c, ci r = 32 ct + lrow) -> array r & 3, row 16 c + (r >> 2)
    const int abase = (lrow & 3) * DM_AARR + (lrow >> 2) * 64 + 16 * lh;    // (16-byte fragment reads, see compute)
    // B: window element (u, v) of pixel (li, lj) of this wave's tile: fine slot
    // n = (2 (2 pt + li) + u) 34 + 2 lj + v -> plane v & 1, pair (n >> 1) = (...) 17 + lj + (v >> 1)
    int bb[4][2];
    {
        const int li = lrow >> 4, lj = lrow & 15;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pr = (2 * (2 * pt + li) + u) * DM_PR + lj + h;
                bb[u][h] = DM_BOFF + (pr & 3) * DM_BARR + (pr >> 2) * 64 + 16 * lh;
            }
    }

    f32x16 acc[2];

    // A wave takes ONE group of eight output channels of the chunk (its K half): a lane reads 16
    // bytes of a row (ds_read_b128; the lane halves take channels 0-3 / 4-7 of the group: any
    // assignment of K to the halves will do as long as both operands use the same) and feeds four
    // matrix instructions per component and tile.  Eight lanes x 16 bytes cover the 32 banks once;
    // the 8-byte reads of the first version met two per bank pair (csrc/fwd_min.hip: the fragment
    // reads cost a fifth of a stage).
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    auto compute = [&](int u_) {
        const unsigned char *st = smem + u_ * DM_STAGE;
        const int ko = 32 * ks;
        f32x4_ g[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                g[u][v] = *(const f32x4_ *)(st + bb[u][v >> 1] + (v & 1) * (4 * DM_BARR) + ko);
        // columns, then rows
        f32x4_ cc[4][3], s[9];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cc[u][0] = g[u][1] + g[u][2];
            cc[u][1] = g[u][2] + g[u][3];
            cc[u][2] = g[u][0] - g[u][2];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            s[q] = cc[1][q] + cc[2][q];
            s[3 + q] = cc[2][q] + cc[3][q];
            s[6 + q] = cc[0][q] - cc[2][q];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const f32x4_ a = *(const f32x4_ *)(st + abase + (16 * c + 8 * t) * 64 + ko);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], s[c][j], acc[t], 0, 0, 0);
            }
    };

    if (item0 >= item1) return;
    // (more than one item per workgroup only with an EVEN number of chunks: every item then
    // starts in stage 0 and the stage of a chunk is its parity -- a compile-time constant of the
    // twice-unrolled chunk loop, i.e. immediate offsets in every fragment read)
    setup(item0);
    issue(0, 0);
#pragma unroll
    for (int item = item0; item < item1; item += istep) {
        const int e_ci0 = ci0, e_b = b, e_oy = oy, e_ox = ox, e_blk = blk_;      // this item's (setup moves on below)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ch + 1 < nchunks) issue((ch + 1) & 1, ch + 1);
            else if (item + istep < item1) {    // the next item's first chunk
                with_w = false;
                setup(item + istep);
                issue((ch + 1) & 1, 0);
            }
            compute(ch & 1);
        }

        // ---- epilogue: [wave][tile][pixel 32][ci 32] rows of 128 bytes, chunks XOR-swizzled by pixel & 7,
        // in the stage that was multiplied last (the other one is receiving the next item's chunk)
        // (resident weights: in stage 1's patch area -- 41 KB, the exchange takes 32 + 2)
        unsigned char *xch = RES ? smem + DM_STAGE + DM_BOFF : smem + ((nchunks - 1) & 1) * DM_STAGE;
        // this wave finishes channel tile t = ks of its pixel tile: which member, where in it
        const int cit = e_ci0 + 32 * ks;
        const int c_first = P.dst[0].C;
        const int sel = cit >= c_first ? 1 : 0;
        const GDst &D = P.dst[sel];
        const int cm = cit - (sel ? c_first : 0);
        // (the epilogue's lane-derived addresses are made from a laundered copy of the lane id:
        // as loop invariants of the item loop they would be hoisted above it and stay in registers
        // through the K loop -- 256 registers + 50 spilled against 188)
        int le = lane;
        asm volatile("" : "+v"(le));
        const int ecq = le & 7, elrow = le & 31, elh = le >> 5;
        // The epilogue's global operands first: their loads are in flight across the exchange of
        // the accumulators through LDS (a barrier pins memory operations: issued behind it, every
        // workgroup of the finest stage waited for them with nothing else to do: 158 against 127
        // us when the head's term joined the epilogue).
        // A flow head folded into this member's gradient (dvsof_grad_dst_t.head_w): the head's two
        // weight rows for this lane's 4 channels, the flow's gradient at the lane's pixels.
        f32x4 hw0 = {0.f, 0.f, 0.f, 0.f}, hw1 = hw0;
        if (D.head_w) {
            hw0 = *(const f32x4 *)(D.head_w + cm + 4 * ecq);
            hw1 = *(const f32x4 *)(D.head_w + D.C + cm + 4 * ecq);
        }
        long long eo[4];
        f32x4 ead[4], ead2[4], eas[4], ehx[4];
        float eg0[4], eg1[4];
        const bool hx_own = D.head_part && D.head_x != D.actsrc;   // (ReLU: the head's input IS actsrc)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = 8 * it + (le >> 3);
            const int yy = e_oy + 2 * pt + (px >> 4), xx = e_ox + (px & 15);
            eo[it] = (long long)e_b * D.sb + (long long)yy * D.sy + (long long)xx * D.sx + cm + 4 * ecq;
            if (D.addend) ead[it] = *(const f32x4 *)(D.addend + eo[it]);
            if (D.addend2) ead2[it] = *(const f32x4 *)(D.addend2 + eo[it]);
            if (D.actsrc) eas[it] = *(const f32x4 *)(D.actsrc + eo[it]);
            if (hx_own) ehx[it] = *(const f32x4 *)(D.head_x + eo[it]);
            if (D.head_w) {
                const long long hwp = (long long)H * W, r = (long long)yy * W + xx;
                eg0[it] = D.head_g[((long long)e_b * 2) * hwp + r];
                eg1[it] = D.head_g[((long long)e_b * 2 + 1) * hwp + r];
            }
        }
        __builtin_amdgcn_s_barrier();       // every wave is done with the last chunk's stage
        // (1) the K halves meet: a wave hands the tile it does NOT finish (t = 1 - ks) to its partner
        // (pt, 1 - ks) in the matrix instruction's own register layout, 4 KB per wave ...
        f32x16 mine = ks ? acc[1] : acc[0];
        {
            const f32x16 other = ks ? acc[0] : acc[1];
            unsigned char *r1 = xch + wave * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4 *)(r1 + (g * 64 + le) * 16) = f32x4{other[4 * g], other[4 * g + 1], other[4 * g + 2], other[4 * g + 3]};
        }
        __builtin_amdgcn_s_barrier();
        {
            const unsigned char *p1 = xch + (pt + 4 * (1 - ks)) * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o_ = *(const f32x4 *)(p1 + (g * 64 + le) * 16);
                // (the K half 0 wave's value first, as the two-tile exchange summed them)
#pragma unroll
                for (int e = 0; e < 4; ++e) mine[4 * g + e] = ks ? o_[e] + mine[4 * g + e] : mine[4 * g + e] + o_[e];
            }
        }
        __builtin_amdgcn_s_barrier();       // the partners have read: the regions are the waves' own again
        // (2) ... and the wave turns its tile to pixel-major through its own 4 KB (rows of 128 bytes,
        // chunks XOR-swizzled by pixel & 7): 8 lanes then hold one pixel's 32 channels
        unsigned char *r2 = xch + wave * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {mine[4 * g], mine[4 * g + 1], mine[4 * g + 2], mine[4 * g + 3]};
            *(f32x4 *)(r2 + elrow * 128 + (((2 * g + elh) ^ (elrow & 7)) << 4)) = v;
        }
        f32x4 ha0 = {0.f, 0.f, 0.f, 0.f}, ha1 = ha0;
        float hs0 = 0.f, hs1 = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = 8 * it + (le >> 3);
            const int xo = px * 128 + ((ecq ^ (px & 7)) << 4);
            f32x4 v = *(const f32x4 *)(r2 + xo);
            if (D.addend) v += ead[it];
            if (D.addend2) v += ead2[it];
            if (D.head_w) v += eg0[it] * hw0 + eg1[it] * hw1;     // + W_h^T g_flow (dvsof_flow_head_bwd's data part)
            if (D.head_part) {  // the head's own weight / bias gradient: sum_px g_flow (x) x
                const f32x4 xv = hx_own ? ehx[it] : eas[it];
                ha0 += eg0[it] * xv;
                ha1 += eg1[it] * xv;
                hs0 += eg0[it];
                hs1 += eg1[it];
            }
            if (D.actsrc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd(eas[it][e], P.bwd_act);
            }
            *(f32x4 *)(D.p + eo[it]) = v;
        }
        if (D.head_part) {      // (member 0 only, both K halves of the item: uniform over the workgroup)
            // lanes of equal `ecq` hold the same 4 channels at 8 different pixels each
#pragma unroll
            for (int o_ = 8; o_ < 64; o_ <<= 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ha0[e] += __shfl_xor(ha0[e], o_, 64);
                    ha1[e] += __shfl_xor(ha1[e], o_, 64);
                }
                hs0 += __shfl_xor(hs0, o_, 64);
                hs1 += __shfl_xor(hs1, o_, 64);
            }
            // [wave][2][32] + [wave][2] behind the exchange tiles (32 KiB)
            float *red = (float *)(xch + 32768);
            if (le < 8) {
                *(f32x4 *)(red + (wave * 2 + 0) * 32 + 4 * ecq) = ha0;
                *(f32x4 *)(red + (wave * 2 + 1) * 32 + 4 * ecq) = ha1;
                if (le == 0) {
                    red[512 + wave * 2] = hs0;
                    red[512 + wave * 2 + 1] = hs1;
                }
            }
            __builtin_amdgcn_s_barrier();
            const int Cm = D.C;
            float *prow = D.head_part + (size_t)e_blk * (2 * Cm + 2);
            if (tid < 128) {    // (K half, flow channel, channel): the four pixel tiles in order
                const int ks_ = tid >> 6, k_ = (tid >> 5) & 1, c_ = tid & 31;
                float t_ = 0.f;
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_) t_ += red[((p_ + 4 * ks_) * 2 + k_) * 32 + c_];
                prow[k_ * Cm + e_ci0 + 32 * ks_ + c_] = t_;
            } else if (tid < 130 && e_ci0 == 0) {
                const int k_ = tid - 128;
                prow[2 * Cm + k_] = (red[512 + 0 + k_] + red[512 + 2 + k_]) + (red[512 + 4 + k_] + red[512 + 6 + k_]);
            }
        }
    }
#endif
}

}  // namespace

// the descriptor's shape beyond fwd_min's test: 64 | every member, 8 | H
// (DVSOF_NO_DGRAD_MIN=1: the 4x4 stride-2 form on gconv2)
bool min9_dgrad_shape_ok(const int *C, int Cout, int H)
{
    static const bool off = getenv("DVSOF_NO_DGRAD_MIN") != nullptr;
    return !off && (C[0] & 63) == 0 && (C[1] & 63) == 0 && (H % DM_NR) == 0 && (Cout & 15) == 0;
}

int min9_prepare_dgrad(const float *w, float *wq, int Cout, int Ctot, hipStream_t st)
{
    dim3 grid((Ctot + 31) / 32, (Cout + 31) / 32, 9);
    hipLaunchKernelGGL(min9_dgrad_weights_kernel, grid, dim3(256), 0, st, w, wq, Cout, Ctot);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

// P as dvsof_conv2d_dgrad fills it for a sub-pixel layer (P.W = W'[9][Ctot][Cout] here)
int dgrad_min_launch(const GConvParams &P, hipStream_t st)
{
    const GSrc &S = P.src[0];
    if (!S.p || (reinterpret_cast<uintptr_t>(S.p) & 15) || ((S.sb | S.sy | S.sx) & 3) || S.sc != 1) return DVSOF_EINVAL;
    if (P.ndst != 2 || (reinterpret_cast<uintptr_t>(P.W) & 15)) return DVSOF_EINVAL;
    for (int i = 0; i < 2; ++i) {
        const GDst &D = P.dst[i];
        if (!D.p || D.sc != 1 || ((D.sb | D.sy | D.sx) & 3)) return DVSOF_EINVAL;
        if ((reinterpret_cast<uintptr_t>(D.p) | reinterpret_cast<uintptr_t>(D.addend) |
             reinterpret_cast<uintptr_t>(D.addend2) | reinterpret_cast<uintptr_t>(D.actsrc) |
             reinterpret_cast<uintptr_t>(D.head_w)) & 15)
            return DVSOF_EINVAL;
    }
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)dgrad_min_f32_kernel<1>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS));
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)dgrad_min_f32_kernel<2>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS));
        attr_set = true;
    }
    // items (pixel block, 64-channel tile) per workgroup.  Two (DVSOF_DGRAD_MIN_IPW=2; needs an even
    // chunk count: stage = chunk parity) hide the second item's first-chunk latency: alone on the
    // GPU the finest stage 151 -> 142 us, the others unchanged or worse (230 registers: 121 / 114 /
    // 108 against 121 / 110 / 107) -- and inside the step, beside the weight-gradient lane, a loss:
    // 2.538-2.547 against 2.501-2.513 ms.  Default: one.
    const long long total = (long long)P.B * (P.Ho / DM_NR) * (P.Wo / 16) * (P.N / 64);
    static const int ipw_env = getenv("DVSOF_DGRAD_MIN_IPW") ? atoi(getenv("DVSOF_DGRAD_MIN_IPW")) : 0;
    const int nchunks = P.Cin_tot / 16;
    int ipw = ipw_env == 2 ? 2 : 1;
    if ((nchunks & 1) || (total & 1)) ipw = 1;
    // resident weights (two chunks of K): 256 workgroups (whole channel-tile rows of them), >= 2 items each
    static const bool no_res = getenv("DVSOF_DGRAD_MIN_RESIDENT") && atoi(getenv("DVSOF_DGRAD_MIN_RESIDENT")) == 0;
    const int nct = P.N / 64;
    if (!no_res && !ipw_env && nchunks == 2 && 256 % nct == 0 && total >= 2 * 256) {
        static bool attr0 = false;
        if (!attr0) {
            DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)dgrad_min_f32_kernel<0>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, DM_LDS));
            attr0 = true;
        }
        hipLaunchKernelGGL(dgrad_min_f32_kernel<0>, dim3(256), dim3(DM_NT), DM_LDS, st, P, (int)total);
        DVSOF_LAUNCH_CHECK();
        return DVSOF_OK;
    }
    if (ipw == 2)
        hipLaunchKernelGGL(dgrad_min_f32_kernel<2>, dim3((unsigned)(total / 2)), dim3(DM_NT), DM_LDS, st, P, (int)total);
    else
        hipLaunchKernelGGL(dgrad_min_f32_kernel<1>, dim3((unsigned)total), dim3(DM_NT), DM_LDS, st, P, (int)total);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
