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

constexpr int WP_NS = 4;            // ring stages
constexpr int WP_PG = 2;            // 16-pixel groups per stage
constexpr int WP_XSLOT = 3 * 18;    // patch pixels per group
constexpr int WP_XP = 7;            // pieces of the two patches (108 of 112 slots used)
constexpr int WP_GP = 8;            // pieces of the gradient planes: 2 groups x 4 phases x 16 px
constexpr int WP_PIECES = 16;       // + 1 dummy so that every wave issues 4 loads per stage
constexpr int WP_LPW = WP_PIECES / 4;
constexpr int WP_STAGE = WP_PIECES * 1024;
constexpr unsigned WP_OOB = 0x80000000u;

__global__ __launch_bounds__(CONV_NT) void wgrad_patch_twins_kernel(const WGradParams P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pa = wave >> 1, pb = wave & 1;            // this wave's output phase

    const int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
    int s = 0;
    for (int i = 1; i < P.nsrc; ++i)
        if (bx >= P.tile_begin[i]) s = i;
    const GSrc &S = P.src[s];
    int coff = 0;
    for (int i = 0; i < s; ++i) coff += P.src[i].C;
    const int c0 = (bx - P.tile_begin[s]) * 32;          // channel tile of the member
    const int co0 = by * 32;
    const int kbeg = split * P.klen, kend = min(P.M, kbeg + P.klen);
    int groups_left = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int nsteps = (groups_left + WP_PG - 1) / WP_PG;
    const int H = P.Hv, W = P.Wv;                         // low-resolution frame

    // load slots: piece p = wave + 4 i; lanes 4 k .. 4 k + 3 fetch the four 16-byte chunks
    // of pixel slot 16 p + k
    unsigned v_off[WP_LPW];
    int v_dy[WP_LPW], v_dx[WP_LPW], v_g[WP_LPW];
#pragma unroll
    for (int i = 0; i < WP_LPW; ++i) {
        const int p = wave + 4 * i, q = lane & 3;
        v_off[i] = WP_OOB;
        v_dy[i] = v_dx[i] = 0;
        v_g[i] = -1;                                      // dummy / unused slot
        if (p < WP_XP) {
            const int n = 16 * p + (lane >> 2);
            if (n < WP_PG * WP_XSLOT) {
                const int g = n / WP_XSLOT, rem = n - g * WP_XSLOT, r = rem / 18, c = rem - 18 * r;
                v_g[i] = g;
                v_dy[i] = r - 1;
                v_dx[i] = c - 1;
                // the resource's base is shifted by (-1, -1): offsets stay non-negative
                v_off[i] = (unsigned)((r * S.sy + c * S.sx + c0 + 8 * q) * 2);
            }
        } else if (p < WP_XP + WP_GP) {
            const int n = 16 * (p - WP_XP) + (lane >> 2);
            const int g = n >> 6, ph = (n >> 4) & 3, c = n & 15;
            v_g[i] = 2 + g;                               // 2, 3: gradient planes of group 0, 1
            v_off[i] = (unsigned)(((ph >> 1) * P.g_py + (ph & 1) * P.g_px + c * P.g_sx + co0 + 8 * q) * 2);
        }
    }
    const __amdgpu_buffer_rsrc_t gres =
        __builtin_amdgcn_make_buffer_rsrc((void *)P.gout16, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S.p16 - ((long long)S.sy + S.sx)), 0, 0x7fffffff, 0x00020000);

    int g_ox = kbeg % P.Wo, g_oy = (kbeg / P.Wo) % P.Ho, g_b = kbeg / (P.Wo * P.Ho);
    auto issue = [&](int stage_idx) {
        int a_so[WP_PG], b_so[WP_PG], gy[WP_PG], gx[WP_PG];
        bool live[WP_PG];
#pragma unroll
        for (int g = 0; g < WP_PG; ++g) {
            live[g] = groups_left > 0;
            a_so[g] = __builtin_amdgcn_readfirstlane(
                (int)(((long long)g_b * P.g_sb + (long long)g_oy * P.g_sy + (long long)g_ox * P.g_sx) * 2));
            b_so[g] = __builtin_amdgcn_readfirstlane(
                (int)(((long long)g_b * S.sb + (long long)g_oy * S.sy + (long long)g_ox * S.sx) * 2));
            gy[g] = g_oy;
            gx[g] = g_ox;
            if (live[g]) {
                --groups_left;
                g_ox += BK;
                if (g_ox >= P.Wo) {
                    g_ox = 0;
                    if (++g_oy == P.Ho) {
                        g_oy = 0;
                        ++g_b;
                    }
                }
            }
        }
        unsigned char *st = smem + stage_idx * WP_STAGE;
#pragma unroll
        for (int i = 0; i < WP_LPW; ++i) {
            const int p = wave + 4 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            if (p < WP_XP) {
                // (a piece may hold slots of both groups: the group is per lane)
                const int g = v_g[i] == 1 ? 1 : 0;
                const int y = (g ? gy[1] : gy[0]) + v_dy[i], x = (g ? gx[1] : gx[0]) + v_dx[i];
                const bool ok = (v_g[i] >= 0) & (g ? live[1] : live[0]) & ((unsigned)y < (unsigned)H) &
                                ((unsigned)x < (unsigned)W);
                // per-lane group -> the group's base goes into the vector offset
                const unsigned base = (unsigned)(g ? b_so[1] : b_so[0]);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sres, dst, 16, ok ? v_off[i] + base : WP_OOB, 0, 0, 0);
            } else if (p < WP_XP + WP_GP) {
                const bool g1 = (p - WP_XP) >= WP_GP / 2;       // scalar: pieces 0-3 group 0, 4-7 group 1
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, (g1 ? live[1] : live[0]) ? v_off[i] : WP_OOB,
                                                         g1 ? a_so[1] : a_so[0], 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gres, dst, 16, WP_OOB, 0, 0, 0);   // the dummy piece
            }
        }
    };

    f32x16 acc[4];          // tap t = 2 p + q of this wave's phase: D[co 32][ci 32]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lrow = lane & 31, lh = lane >> 5;
    const bool do_bias = P.dbias != nullptr && bx == 0;
    float bsum = 0.f;
    // transposed fragment reads (wgrad2_twins_kernel): group tg = lane >> 4 covers channels
    // 16 (tg & 1) .. and pixels 8 (tg >> 1) + 4 rd ..; lane 4 tq + tp of the group addresses
    // pixel row tq, channels 4 tp .. 4 tp + 3.  Pixel rows are 64 bytes here.
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int frag = (8 * (tg >> 1) + tq) * 64 + 32 * (tg & 1) + 8 * tp;

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * WP_STAGE;
#pragma unroll
        for (int g = 0; g < WP_PG; ++g) {
            const unsigned char *ga = st + WP_XP * 1024 + ((g * 4 + wave) * 16) * 64 + frag;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)ga);
            const s16x4 ahi =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ga + 4 * 64));
            const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
            bf16x8 fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = pa + (t >> 1), c = pb + (t & 1);          // patch row / first patch column
                const unsigned char *xb = st + ((g * WP_XSLOT + r * 18 + c)) * 64 + frag;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)xb);
                const s16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(xb + 4 * 64));
                fb[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            if (do_bias) {
                const s16x8 v = __builtin_bit_cast(s16x8, fa);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += __builtin_bit_cast(float, (unsigned)(unsigned short)v[e] << 16);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[t], acc[t], 0, 0, 0);
        }
    };

#pragma unroll
    for (int u = 0; u < WP_NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int s0 = 0; s0 < nsteps; s0 += WP_NS) {
#pragma unroll
        for (int u = 0; u < WP_NS; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                if (st + WP_NS - 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WP_NS - 2) * WP_LPW) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (st + WP_NS - 1 < nsteps) issue((u + WP_NS - 1) % WP_NS);
                compute(u);
            }
        }
    }

    const int slab = wave * P.S + split;            // [phase][split], as wgrad2's blockIdx.z
    if (do_bias) {   // lanes l and l ^ 32 hold the two k halves of the same channel
        const float v = bsum + __shfl_xor(bsum, 32);
        if (lh == 0) P.dbias[(size_t)slab * P.Cout + co0 + lrow] = v;
    }
    const size_t wsize = (size_t)P.Cout * 4 * P.Cin_tot;
    float *dW = P.dW + (size_t)slab * wsize;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const size_t col = (size_t)t * P.Cin_tot + coff + c0 + lrow;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            dW[(size_t)co * 4 * P.Cin_tot + col] = acc[t][reg];
        }
    }
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
    if ((P.Cout & 31) || (P.Wo % BK) || P.Ho != P.Hv || P.Wo != P.Wv) return false;
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

bool wgrad_patch_eligible(const WGradParams &P)
{
    if (!P.twins || !P.gout16 || !wgrad_patch_shape_ok(P)) return false;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat && !P.src[s].p16) return false;
    return true;
}

// K splits for this kernel: enough workgroups for two per CU; a slab is a whole folded-size
// gradient, so no more splits than that takes (<= 64, the workspace bound of wgrad_splits)
int wgrad_patch_splits(const WGradParams &P)
{
    long long tiles = (long long)(P.Cout / 32);
    long long ct = 0;
    for (int s = 0; s < P.nsrc; ++s)
        if (!P.src[s].flat) ct += P.src[s].C / 32;
    tiles *= ct;
    const long long groups = ((long long)P.M + BK - 1) / BK;
    static const int target = getenv("DVSOF_WGRAD_PATCH_WGS") ? atoi(getenv("DVSOF_WGRAD_PATCH_WGS")) : 512;
    long long S = (target + tiles - 1) / tiles;
    const long long maxS = groups / (2 * WP_PG) > 0 ? groups / (2 * WP_PG) : 1;    // >= 2 stages per split
    if (S > maxS) S = maxS;
    // a slab is a whole phase-form gradient: at most ~32 MB of partial sums per layer
    const long long slab_bytes = 4LL * P.Cout * 4 * P.Cin_tot * 4;
    long long capS = (32LL << 20) / (slab_bytes > 0 ? slab_bytes : 1);
    static const int max_env = getenv("DVSOF_WGRAD_PATCH_MAXS") ? atoi(getenv("DVSOF_WGRAD_PATCH_MAXS")) : 64;
    if (capS > max_env) capS = max_env;
    if (capS < 1) capS = 1;
    if (S > capS) S = capS;
    if (S < 1) S = 1;
    return (int)S;
}

int wgrad_patch_launch(const WGradParams &P0, hipStream_t st)
{
    WGradParams P = P0;
    int nt = 0;
    for (int s = 0; s < P.nsrc; ++s) {
        P.tile_begin[s] = nt;
        if (!P.src[s].flat) nt += P.src[s].C / 32;
    }
    P.tile_begin[P.nsrc] = nt;
    constexpr size_t LDS = (size_t)WP_NS * WP_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_patch_twins_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    dim3 grid(nt, P.Cout / 32, P.S);
    hipLaunchKernelGGL(wgrad_patch_twins_kernel, grid, dim3(CONV_NT), LDS, st, P);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}
