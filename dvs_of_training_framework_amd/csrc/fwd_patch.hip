// Forward of the finest 2x up-sampling decoder stage (sub-pixel form) from the
// bf16 twins with the INPUT PATCH RESIDENT in LDS and the WEIGHTS RESIDENT IN
// REGISTERS.
//
//   y[2i+a][2j+b][co] = act( bias[co] + sum over taps (p,q), channels ci of
//        Wf[(a,b)][co][(p,q)][ci] * x[i-1+a+p][j-1+b+q][ci] )     a,b,p,q in {0,1}
//
// (the four 2x2 phase kernels of `up2 -> conv3x3` over cat[x, skip]; EV_FlowNet
// decoder, reference call site utils/training.py:158 through the absent
// EV_FlowNet.net).
//
// gconv2_kernel runs this stage as 4 phase GEMMs whose (phase, tap) views of
// the input are streamed through LDS one by one: 16 passes over the same
// pixels shifted by -1/0/+1 -- 218 MB fetched for 34 MB of input at batch 8,
// 256x256 (profiles/round3/b_traffic_pmc_bf16s.csv) and the launch is bound by
// exactly that (87-99 us; the matrix pipes are busy 6.5 % of the time).  Here
//
//   * a workgroup is PERSISTENT: it walks down a strip of blocks of
//     2 rows x 16 pixels (low resolution), all 4 phases, all 32 output
//     channels;
//   * wave w = phase (a,b).  Its weights Wf[(a,b)] -- 32 co x 512 K bf16 =
//     32 KiB -- are loaded ONCE into 128 VGPRs per lane, as the A operands of
//     v_mfma_f32_32x32x16_bf16 (D[co][pixel]);
//   * per block ONE 4 x 18-pixel patch of each member goes L2 -> LDS by LDS-DMA
//     (18 pieces of 1 KiB; the lane's global offset picks pixel and channel
//     chunk, so the LDS layout is ours: 128-byte pixel rows whose 16-byte
//     chunks are XOR-swizzled by (slot >> 1) & 7 -- a ds_read_b128 of 16
//     consecutive pixels then covers all 64 banks once); the 16 (phase, tap)
//     views are byte offsets into the patch.  4-stage ring;
//   * D[co][pixel]: a lane owns one pixel and 4 x 4 consecutive output
//     channels: bias (+ border-class bias of a folded constant member),
//     activation, one 16-byte store per quad (+ 8 bytes of the bf16 twin).
//
// Same products as gconv2_kernel in the bf16-twins mode (bf16 x bf16, f32
// accumulate); the f32 summation order differs.
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr int FP_CM = 64;                   // channels per member
constexpr int FP_N = 32;                    // output channels
constexpr int FP_SLOTS = 4 * 18;            // patch pixels: rows i-1 .. i+2, columns j0-1 .. j0+16
constexpr int FP_MEMB = FP_SLOTS * 128;     // bytes of one member's patch (9 pieces)
constexpr int FP_PIECES = 20;               // 18 + 2 padding pieces: every wave issues 5 loads
constexpr int FP_LPW = FP_PIECES / 4;
constexpr int FP_STAGE = FP_PIECES * 1024;
constexpr int FP_NS = 3;
constexpr int FP_XT = 4 * 4096;                // a 32 x 32 f32 output tile per wave (epilogue transpose)
constexpr unsigned FP_OOB = 0x80000000u;

__device__ __forceinline__ int fp_key(int slot) { return (slot >> 1) & 7; }

template <bool TWIN, bool ZOUT>
__global__ __launch_bounds__(CONV_NT) void fwd_patch_twins_kernel(const GConvParams P, int nblocks, int bpw)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pa = wave >> 1, pb = wave & 1;            // this wave's output phase
    const int H = P.Hv, W = P.Wv;                         // low-resolution frame
    const int nbx = W / 16, nby = H / 2;

    // linear id % 8 = XCD: an XCD owns a contiguous range of strips (neighbouring
    // strips share their halo columns, consecutive blocks of a strip their halo rows)
    unsigned wg = blockIdx.x;
    {
        const unsigned total = gridDim.x, x = wg & 7u, q = total >> 3, r = total & 7u;
        wg = x * q + min(x, r) + (wg >> 3);
    }
    const int blk0 = (int)wg * bpw;
    const int nsteps = max(0, min(nblocks, blk0 + bpw) - blk0);
    if (nsteps == 0) return;

    // ---- weights of this wave's phase: A operands, rows = co (lane & 31), k = 8 (lane >> 5) ..
    const int lrow = lane & 31, lh = lane >> 5;
    bf16x8 wf[32];      // [tap][member][kstep]
    {
        const unsigned short *wp = P.W16 + (long long)wave * P.w_phase_stride + (long long)lrow * 4 * P.Cin_tot + 8 * lh;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    wf[(t * 2 + m) * 4 + s] =
                        *(const bf16x8 *)(wp + t * P.Cin_tot + m * FP_CM + 16 * s);
    }

    // ---- load slots: piece p = wave + 4 i; pieces 0..8 member 0, 9..17 member 1, 18..19 padding
    unsigned v_off[FP_LPW];
    int v_dy[FP_LPW], v_dx[FP_LPW];
#pragma unroll
    for (int i = 0; i < FP_LPW; ++i) {
        const int p = wave + 4 * i;
        v_off[i] = FP_OOB;
        v_dy[i] = v_dx[i] = 0;
        if (p < 18) {
            const int m = p >= 9;
            const GSrc &S = P.src[m];
            const int slot = 8 * (p - 9 * m) + (lane >> 3), c = (lane & 7) ^ fp_key(slot);
            const int r = slot / 18, cc = slot - 18 * r;
            v_dy[i] = r - 1;
            v_dx[i] = cc - 1;
            // the resource's base is shifted by (-1, -1): offsets stay non-negative
            v_off[i] = (unsigned)((r * S.sy + cc * S.sx + 8 * c) * 2);
        }
    }
    const GSrc &S0 = P.src[0], &S1 = P.src[1];
    const __amdgpu_buffer_rsrc_t res0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S0.p16 - ((long long)S0.sy + S0.sx)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t res1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S1.p16 - ((long long)S1.sy + S1.sx)), 0, 0x7fffffff, 0x00020000);

    // blocks: idx = (b * nbx + bx) * nby + by -- a strip from top to bottom
    int k_by = blk0 % nby, k_bx = (blk0 / nby) % nbx, k_b = blk0 / (nby * nbx);
    auto issue = [&](int stage_idx) {
        const int oy = 2 * k_by, ox = 16 * k_bx;
        const int so0 = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S0.sb + (long long)oy * S0.sy + (long long)ox * S0.sx) * 2));
        const int so1 = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S1.sb + (long long)oy * S1.sy + (long long)ox * S1.sx) * 2));
        if (++k_by == nby) {
            k_by = 0;
            if (++k_bx == nbx) {
                k_bx = 0;
                ++k_b;
            }
        }
        unsigned char *st = smem + stage_idx * FP_STAGE;
#pragma unroll
        for (int i = 0; i < FP_LPW; ++i) {
            const int p = wave + 4 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            const bool ok = ((unsigned)(oy + v_dy[i]) < (unsigned)H) & ((unsigned)(ox + v_dx[i]) < (unsigned)W);
            const unsigned off = ok ? v_off[i] : FP_OOB;
            if (p < 9)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(res0, dst, 16, off, so0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(res1, dst, 16, p < 18 ? off : FP_OOB, so1, 0, 0);
        }
    };

    // ---- this lane's pixel of a block (B operand column / D column) and its output channels
    const int prr = lrow >> 4, pcc = lrow & 15;
    const GDst &D = P.dst[0];
    const int ecq = lane & 7;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias) bias4 = *(const f32x4 *)(P.bias + 4 * ecq);
    unsigned char *xtile = smem + FP_NS * FP_STAGE + wave * 4096;
    int c_by = blk0 % nby, c_bx = (blk0 / nby) % nbx, c_b = blk0 / (nby * nbx);

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * FP_STAGE;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // patch slot of this lane's pixel under tap (p, q) of phase (a, b)
            const int slot = (prr + pa + (t >> 1)) * 18 + pcc + pb + (t & 1);
            const unsigned char *row = st + slot * 128;
            const int key = fp_key(slot);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 xb = *(const bf16x8 *)(row + m * FP_MEMB + (((2 * s + lh) ^ key) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(t * 2 + m) * 4 + s], xb, acc, 0, 0, 0);
                }
        }
        // ---- epilogue through the wave's own LDS tile [pixel 32][channel 32] (16-byte chunks
        // XOR-swizzled by pixel & 7): acc[4 g + e] = channel 8 g + 4 lh + e of pixel lrow goes
        // in, lane = (pixel 8 r + (lane >> 3), chunk lane & 7) comes out -- 8 lanes store one
        // pixel's 128 bytes (and 64 bytes of the twin) instead of 64 lanes 16 bytes 256 B apart
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *(f32x4 *)(xtile + lrow * 128 + (((2 * g + lh) ^ (lrow & 7)) << 4)) = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = 8 * r + (lane >> 3);
            f32x4 v = *(const f32x4 *)(xtile + px * 128 + ((ecq ^ (px & 7)) << 4));
            const int oy = 2 * c_by + (px >> 4), ox = 16 * c_bx + (px & 15);
            const long long o = (long long)c_b * D.sb + (long long)oy * D.sy + (long long)ox * D.sx +
                                (long long)pa * D.ph_y + (long long)pb * D.ph_x + 4 * ecq;
            v += bias4;
            if (P.bias_cls) {
                const int Y = 2 * oy + pa, X = 2 * ox + pb;
                const int cls = 3 * (Y == 0 ? 1 : Y == P.out_H - 1 ? 2 : 0) + (X == 0 ? 1 : X == P.out_W - 1 ? 2 : 0);
                if (cls) v += *(const f32x4 *)(P.bias_cls + cls * FP_N + 4 * ecq);
            }
            if (ZOUT) *(f32x4 *)(P.zout + o) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], P.act);
            *(f32x4 *)(D.p + o) = v;
            if (TWIN) {
                s16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (short)bf16_bits(v[e]);
                *(s16x4 *)(D.p16 + o) = h;
            }
        }
        if (++c_by == nby) {
            c_by = 0;
            if (++c_bx == nbx) {
                c_bx = 0;
                ++c_b;
            }
        }
    };

    // ---- ring of 3.  vmcnt counts loads AND stores in order: behind the loads of stage s come
    // S(s-2) L(s+1) S(s-1) -- fewer stores for the first two stages
    constexpr int ST = (ZOUT ? 4 : 0) + 4 + (TWIN ? 4 : 0);
    static_assert(FP_NS == 3, "the waits below are written for a ring of 3");
#pragma unroll
    for (int u = 0; u < FP_NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int s0 = 0; s0 < nsteps; s0 += FP_NS) {
#pragma unroll
        for (int u = 0; u < FP_NS; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                if (st + 1 < nsteps) {
                    if (st >= 2) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP_LPW + 2 * ST) : "memory");
                    } else if (st == 1) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP_LPW + ST) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FP_LPW) : "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (st + FP_NS - 1 < nsteps) issue((u + FP_NS - 1) % FP_NS);
                compute(u);
            }
        }
    }
#endif
}


// ---------------------------------------------------------------------------
// The same stage in exact f32 (v_mfma_f32_32x32x2_f32, the benchmark's operand
// mode; gconv2_kernel<4,1,1,1> reaches 58 % of the f32 matrix peak here: its
// 128 x 32 tile leaves every MFMA with a fragment read of its own and streams
// the 16 views).  32 x 512 f32 weights per phase are 256 registers per lane:
// EIGHT waves, wave = (phase, member) -- 128 weight registers each, two waves
// per SIMD; the partial tiles of the two member waves of a phase meet in LDS
// after every block, transposed to pixel-major on the way: each wave then
// finishes 16 pixels, 8 lanes storing one pixel's 128 bytes.
// Patch: 72 slots x 2 members x 256 B, 16-byte chunks XOR-swizzled by the patch
// column & 15; a lane reads 8 bytes (two K steps: k = lane >> 5 is channel
// 4 i + n + 2 k in MFMA n of chunk i) -- ds_read_b64 of 2 rows x 16 pixels then
// meets 2 lanes per bank (ds_read_b32 banks mod 32 and DMA chunks of 16 bytes
// leave 4-way at best: measured 3.7x, the reads as long as the MFMAs);
// 3-stage ring of 40 KiB.
// ---------------------------------------------------------------------------
constexpr int FQ_NT = 512;
constexpr int FQ_MEMB = FP_SLOTS * 256;     // bytes of one member's patch (18 pieces)
constexpr int FQ_PIECES = 40;               // 36 + 4 padding pieces: every wave issues 5 loads
constexpr int FQ_LPW = FQ_PIECES / 8;
constexpr int FQ_STAGE = FQ_PIECES * 1024;
constexpr int FQ_NS = 3;
constexpr int FQ_XCH = 8 * 4096;            // exchange: a 32 x 32 partial tile per wave

template <bool ZOUT>
__global__ __launch_bounds__(FQ_NT) void fwd_patch_f32_kernel(const GConvParams P, int nblocks, int bpw)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave & 3, kh = wave >> 2;            // output phase; member whose K range is this wave's
    const int pa = ph >> 1, pb = ph & 1;
    const int H = P.Hv, W = P.Wv;
    const int nbx = W / 16, nby = H / 2;
    unsigned wg = blockIdx.x;
    {
        const unsigned total = gridDim.x, x = wg & 7u, q = total >> 3, r = total & 7u;
        wg = x * q + min(x, r) + (wg >> 3);
    }
    const int blk0 = (int)wg * bpw;
    const int nsteps = max(0, min(nblocks, blk0 + bpw) - blk0);
    if (nsteps == 0) return;

    // ---- weights: A operands, row co = lane & 31, k = lane >> 5
    const int lrow = lane & 31, lh = lane >> 5;
    float wf[128];      // [tap][2 i + n]: channel 4 i + n + 2 lh of member kh
    {
        // Through LDS, a tap at a time: a lane's values sit 2 KiB apart in memory (one row per
        // output channel) -- loaded straight into registers every instruction touches 32 lines
        // for 16 bytes each and the 8 waves evict each other's lines (measured: 36 us of
        // prologue).  LDS-DMA pieces of 4 rows x 256 B are coalesced; 16-byte chunks land
        // XOR-swizzled by the row so that the b64 reads of 32 rows meet 2 lanes per bank.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        unsigned char *wreg = smem + wave * 8192;       // ring space, not in use yet
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void *)P.W, 0, 0x7fffffff, 0x00020000);
        unsigned woff[8];
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int r = 4 * pc + (lane >> 4), c = (lane & 15) ^ (r & 15);
            woff[pc] = (unsigned)((r * 4 * P.Cin_tot + 4 * c) * 4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int so = __builtin_amdgcn_readfirstlane(
                (int)(((long long)ph * P.w_phase_stride + t * P.Cin_tot + kh * FP_CM) * 4));
#pragma unroll
            for (int pc = 0; pc < 8; ++pc)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    wres, (__attribute__((address_space(3))) void *)(wreg + pc * 1024), 16, woff[pc], so, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) {     // MFMA 2 i + n: k = lh is channel 4 i + n + 2 lh
                const f32x2 v = *(const f32x2 *)(wreg + lrow * 256 + ((i ^ (lrow & 15)) << 4) + 8 * lh);
                wf[t * 32 + 2 * i] = v[0];
                wf[t * 32 + 2 * i + 1] = v[1];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tap overwrites the slab
        }
        __builtin_amdgcn_s_barrier();   // every wave is done with the ring space
    }

    unsigned v_off[FQ_LPW];
    int v_dy[FQ_LPW], v_dx[FQ_LPW];
#pragma unroll
    for (int i = 0; i < FQ_LPW; ++i) {
        const int p = wave + 8 * i;
        v_off[i] = FP_OOB;
        v_dy[i] = v_dx[i] = 0;
        if (p < 36) {
            const int m = p >= 18;
            const GSrc &S = P.src[m];
            const int slot = 4 * (p - 18 * m) + (lane >> 4);
            const int r = slot / 18, cc = slot - 18 * r, c = (lane & 15) ^ (cc & 15);
            v_dy[i] = r - 1;
            v_dx[i] = cc - 1;
            v_off[i] = (unsigned)((r * S.sy + cc * S.sx + 4 * c) * 4);
        }
    }
    const GSrc &S0 = P.src[0], &S1 = P.src[1];
    const __amdgpu_buffer_rsrc_t res0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S0.p - ((long long)S0.sy + S0.sx)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t res1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(S1.p - ((long long)S1.sy + S1.sx)), 0, 0x7fffffff, 0x00020000);

    int k_by = blk0 % nby, k_bx = (blk0 / nby) % nbx, k_b = blk0 / (nby * nbx);
    auto issue = [&](int stage_idx) {
        const int oy = 2 * k_by, ox = 16 * k_bx;
        const int so0 = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S0.sb + (long long)oy * S0.sy + (long long)ox * S0.sx) * 4));
        const int so1 = __builtin_amdgcn_readfirstlane(
            (int)(((long long)k_b * S1.sb + (long long)oy * S1.sy + (long long)ox * S1.sx) * 4));
        if (++k_by == nby) {
            k_by = 0;
            if (++k_bx == nbx) {
                k_bx = 0;
                ++k_b;
            }
        }
        unsigned char *st = smem + stage_idx * FQ_STAGE;
#pragma unroll
        for (int i = 0; i < FQ_LPW; ++i) {
            const int p = wave + 8 * i;
            __attribute__((address_space(3))) void *dst =
                (__attribute__((address_space(3))) void *)(st + p * 1024);
            const bool ok = ((unsigned)(oy + v_dy[i]) < (unsigned)H) & ((unsigned)(ox + v_dx[i]) < (unsigned)W);
            const unsigned off = ok ? v_off[i] : FP_OOB;
            if (p < 18)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(res0, dst, 16, off, so0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(res1, dst, 16, p < 36 ? off : FP_OOB, so1, 0, 0);
        }
    };

    const int prr = lrow >> 4, pcc = lrow & 15;
    const GDst &D = P.dst[0];
    // epilogue roles: lane = (pixel 16 kh + 8 r + (lane >> 3), 16-byte channel chunk lane & 7) of
    // this wave's phase -- 8 lanes store one pixel's 128 bytes
    const int ecq = lane & 7;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (P.bias) bias4 = *(const f32x4 *)(P.bias + 4 * ecq);
    int c_by = blk0 % nby, c_bx = (blk0 / nby) % nbx, c_b = blk0 / (nby * nbx);
    int srow[4], skey[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int slot = (prr + pa + (t >> 1)) * 18 + pcc + pb + (t & 1);
        srow[t] = kh * FQ_MEMB + slot * 256 + 8 * lh;
        skey[t] = (pcc + pb + (t & 1)) & 15;
    }
    // partial tiles [pixel 32][channel 32] per wave, 16-byte chunks XOR-swizzled by pixel & 7
    unsigned char *xtile = smem + FQ_NS * FQ_STAGE + wave * 4096;
    const unsigned char *xt0 = smem + FQ_NS * FQ_STAGE + ph * 4096, *xt1 = xt0 + 4 * 4096;

    auto compute = [&](int u) {
        const unsigned char *st = smem + u * FQ_STAGE;
        f32x16 acc, acc1;      // two chains: a dependent MFMA waits for its predecessor's 16 passes
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
        // batches of 16 fragment reads (half the member's channels under one tap), the next
        // batch in flight under this batch's MFMAs
        float bv[2][16];
        auto fetch = [&](int buf, int batch) {
            const int t = batch >> 1, h = batch & 1;
            const unsigned char *row = st + srow[t];
            int key = skey[t];
            asm volatile("" : "+v"(key));    // keep the 64 swizzled offsets out of registers
#pragma unroll
            for (int jc = 0; jc < 8; ++jc) {    // channels 4 (8 h + jc) + 2 lh, + 1: one ds_read_b64
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 v = *(const f32x2 *)(row + (((8 * h + jc) ^ key) << 4));
                bv[buf][2 * jc] = v[0];
                bv[buf][2 * jc + 1] = v[1];
            }
        };
        const int dbg = DVSOF_DBG(P);   // probe build: 1 no exchange, 2 no MFMAs, 8 no stores, 16 no fragment reads
        if (!(dbg & 16)) fetch(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int batch = 0; batch < 8; ++batch) {
            const int buf = batch & 1;
            if (batch + 1 < 8 && !(dbg & 16)) fetch(buf ^ 1, batch + 1);
            if (!(dbg & 2))
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[batch * 16 + jj], bv[buf][jj], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[batch * 16 + jj + 1], bv[buf][jj + 1], acc1, 0, 0, 0);
            }
            // the scheduler otherwise pairs every read with its two MFMAs (read, wait, MFMA,
            // MFMA): the next batch's 8 reads first, then this batch's 16 MFMAs
            if (batch + 1 < 8) {
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        acc += acc1;
        // ---- both members' partial tiles meet in LDS, transposed to pixel-major on the way:
        // acc[4 g + e] = channel 8 g + 4 lh + e of pixel lrow
        if (!(dbg & 1)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                *(f32x4 *)(xtile + lrow * 128 + (((2 * g + lh) ^ (lrow & 7)) << 4)) = v;
            }
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int px = 16 * kh + 8 * r + (lane >> 3);
            const int xo = px * 128 + ((ecq ^ (px & 7)) << 4);
            f32x4 v = (dbg & 1) ? f32x4{acc[0], acc[1], acc[2], acc[3]}
                                : *(const f32x4 *)(xt0 + xo) + *(const f32x4 *)(xt1 + xo);
            const int oy = 2 * c_by + (px >> 4), ox = 16 * c_bx + (px & 15);
            const long long o = (long long)c_b * D.sb + (long long)oy * D.sy + (long long)ox * D.sx +
                                (long long)pa * D.ph_y + (long long)pb * D.ph_x + 4 * ecq;
            v += bias4;
            if (P.bias_cls) {
                const int Y = 2 * oy + pa, X = 2 * ox + pb;
                const int cls = 3 * (Y == 0 ? 1 : Y == P.out_H - 1 ? 2 : 0) + (X == 0 ? 1 : X == P.out_W - 1 ? 2 : 0);
                if (cls) v += *(const f32x4 *)(P.bias_cls + cls * FP_N + 4 * ecq);
            }
            if (ZOUT) *(f32x4 *)(P.zout + o) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], P.act);
            if (!(dbg & 8) || v[0] == 12345.678f) *(f32x4 *)(D.p + o) = v;
        }
        if (++c_by == nby) {
            c_by = 0;
            if (++c_bx == nbx) {
                c_bx = 0;
                ++c_b;
            }
        }
    };

    // ring of 3: behind the loads of stage s come S(s-2) L(s+1) S(s-1).  The ring barrier of
    // stage s + 1 also separates this block's exchange reads from the next block's writes.
    constexpr int ST = (ZOUT ? 2 : 0) + 2;
#pragma unroll
    for (int u = 0; u < FQ_NS - 1; ++u)
        if (u < nsteps) issue(u);
    for (int s0 = 0; s0 < nsteps; s0 += FQ_NS) {
#pragma unroll
        for (int u = 0; u < FQ_NS; ++u) {
            const int st = s0 + u;
            if (st < nsteps) {
                if (st + 1 < nsteps) {
                    if (st >= 2) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FQ_LPW + 2 * ST) : "memory");
                    } else if (st == 1) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FQ_LPW + ST) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FQ_LPW) : "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (st + FQ_NS - 1 < nsteps && !(DVSOF_DBG(P) & 4)) issue((u + FQ_NS - 1) % FQ_NS);
                if (!(DVSOF_DBG(P) & 32)) compute(u);
            }
        }
    }
#endif
}

}  // namespace

// The finest decoder stage: two vector members of 64 channels, 32 output channels, four
// sub-pixel phases of 2x2 taps, 16 | width, 2 | height -- in the bf16-twins mode and in exact
// f32 (DVSOF_NO_FWD_PATCH=1 / DVSOF_NO_FWD_PATCH_F32=1: gconv2)
static bool fp_f32(const GConvParams &P)
{
    static const bool off = getenv("DVSOF_NO_FWD_PATCH_F32") != nullptr;
    return !off && P.mfma_bf16 == 0;
}

bool fwd_patch_eligible(const GConvParams &P)
{
    static const bool off = getenv("DVSOF_NO_FWD_PATCH") != nullptr;
    const bool f32 = fp_f32(P);
    if (off || !(f32 || (P.mfma_bf16 == 3 && P.W16)) || !P.W) return false;
    if (P.nph != 4 || P.ks != 2 || P.stride != 1 || P.up != UP_NONE) return false;
    if (P.ph_pad != 1 || P.pad != 1 || P.src_ph_stride != 0 || P.quad || P.ph_exact) return false;
    if (P.nsrc != 2 || P.ndst != 1 || P.N != FP_N || P.Cin_tot != 2 * FP_CM) return false;
    if ((P.Wv % 16) || (P.Hv & 1) || P.Ho != P.Hv || P.Wo != P.Wv) return false;
    for (int s = 0; s < 2; ++s) {
        const GSrc &S = P.src[s];
        if (S.flat || S.sc != 1 || S.C != FP_CM) return false;
        if (f32 ? (!S.p || (reinterpret_cast<uintptr_t>(S.p) & 15) || ((S.sb | S.sy | S.sx) & 3))
                : (!S.p16 || ((S.sy | S.sx) & 7)))
            return false;
    }
    const GDst &D = P.dst[0];
    if (D.addend || D.addend2 || D.actsrc || D.sc != 1 || D.C != FP_N) return false;
    // 16-byte stores: every stride of the destination a multiple of 4 elements
    if ((D.sb | D.sy | D.sx | D.ph_y | D.ph_x) & 3) return false;
    if ((reinterpret_cast<uintptr_t>(D.p) & 15) || (reinterpret_cast<uintptr_t>(D.p16) & 7) ||
        (reinterpret_cast<uintptr_t>(P.zout) & 15))
        return false;
    if (f32 ? ((reinterpret_cast<uintptr_t>(P.W) & 15) || (P.w_phase_stride & 3))
            : ((reinterpret_cast<uintptr_t>(P.W16) & 15) || (P.w_phase_stride & 7)))
        return false;
    if (P.bias && (reinterpret_cast<uintptr_t>(P.bias) & 15)) return false;
    if (P.bias_cls && (reinterpret_cast<uintptr_t>(P.bias_cls) & 15)) return false;
    return true;
}

template <bool TWIN, bool ZOUT>
static int fp_launch(const GConvParams &P, int nblocks, int bpw, int grid, hipStream_t st)
{
    constexpr size_t LDS = (size_t)FP_NS * FP_STAGE + FP_XT;
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)fwd_patch_twins_kernel<TWIN, ZOUT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((fwd_patch_twins_kernel<TWIN, ZOUT>), dim3(grid), dim3(CONV_NT), LDS, st, P, nblocks, bpw);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

template <bool ZOUT>
static int fq_launch(const GConvParams &P, int nblocks, int bpw, int grid, hipStream_t st)
{
    constexpr size_t LDS = (size_t)FQ_NS * FQ_STAGE + FQ_XCH;
    static bool attr_set = false;
    if (!attr_set) {
        DVSOF_HIP_TRY(hipFuncSetAttribute((const void *)fwd_patch_f32_kernel<ZOUT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((fwd_patch_f32_kernel<ZOUT>), dim3(grid), dim3(FQ_NT), LDS, st, P, nblocks, bpw);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int fwd_patch_launch(const GConvParams &P, hipStream_t st)
{
    const int nby = P.Hv / 2, nbx = P.Wv / 16;
    const long long nblocks = (long long)P.B * nby * nbx;
    if (fp_f32(P)) {    // one workgroup of 8 waves per CU
#ifdef DVSOF_PROBES
        static const int dbg = getenv("DVSOF_FWD_PATCH_DBG") ? atoi(getenv("DVSOF_FWD_PATCH_DBG")) : 0;
        const_cast<GConvParams &>(P).dbg = dbg;
#endif
        static const int want = getenv("DVSOF_FWD_PATCH_F32_WGS") ? atoi(getenv("DVSOF_FWD_PATCH_F32_WGS")) : 256;
        long long bpw = (nblocks + want - 1) / want;
        if (bpw < 2) bpw = 2;
        const int grid = (int)((nblocks + bpw - 1) / bpw);
        return P.zout ? fq_launch<true>(P, (int)nblocks, (int)bpw, grid, st)
                      : fq_launch<false>(P, (int)nblocks, (int)bpw, grid, st);
    }
    // persistent workgroups: DVSOF_FWD_PATCH_WGS of them (default 512: two per CU), each at
    // least 4 blocks (the weights are loaded once per workgroup)
    static const int want = getenv("DVSOF_FWD_PATCH_WGS") ? atoi(getenv("DVSOF_FWD_PATCH_WGS")) : 512;
    long long bpw = (nblocks + want - 1) / want;
    if (bpw < 4) bpw = 4;
    const int grid = (int)((nblocks + bpw - 1) / bpw);
    const bool twin = P.dst[0].p16 != nullptr, z = P.zout != nullptr;
    if (twin) return z ? fp_launch<true, true>(P, (int)nblocks, (int)bpw, grid, st)
                       : fp_launch<true, false>(P, (int)nblocks, (int)bpw, grid, st);
    return z ? fp_launch<false, true>(P, (int)nblocks, (int)bpw, grid, st)
             : fp_launch<false, false>(P, (int)nblocks, (int)bpw, grid, st);
}
