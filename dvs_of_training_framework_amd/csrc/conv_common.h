// Shared definitions of the implicit-GEMM convolution kernels (gfx950, f32
// MFMA 32x32x2: exact-f32 matrix cores, 64 FLOP/clk/SIMD).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-byte vector with 4-byte alignment: gfx950 global loads/stores accept
// dword-aligned dwordx4 accesses (channel offsets such as 514*tap are only
// 8-byte aligned).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { UP_NONE = 0, UP_NEAREST = 1, UP_ZERO = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2 };

constexpr int CONV_NT = 256;  // 4 waves
constexpr int BK = 16;        // K elements per LDS stage
constexpr int LDK = BK + 4;   // row stride (floats): conflict-free ds_read_b128

// One member of the (virtual) channel concatenation a conv reads.
struct GSrc {
    const float *p;
    long long sb;     // batch stride (elements)
    int sy, sx, sc;   // row / pixel / channel strides (elements)
    int C;            // channels
    int flat;         // 1: scalar path, K flattened over (tap, channel)
    const unsigned short *p16;   // bf16 twin (same shape / strides in elements) or null
};

// One destination channel range of the output.
struct GDst {
    float *p;
    const float *addend;  // optional, same indexing as p: out = acc + addend
    const float *addend2; // optional second addend
    const float *actsrc;  // optional: out *= act'(actsrc) (ReLU: y, Mish: z)
    long long sb;
    int sy, sx, sc;
    int C;
    int ph_y, ph_x;       // element offsets of output phase (py, px)
    unsigned short *p16;  // optional bf16 twin of p, written with the same offsets
    const float *head_w;  // optional (dgrad_min.hip only): [2][C] weights of a flow head on this member
    const float *head_g;  // ... and the flow's gradient, planar [B][2][H][W]
    const float *head_x;  // ... optionally the head's input and the per-block partials of its weight / bias
    float *head_part;     //     gradient, [blocks][2 C + 2]
};

struct GConvParams {
    GSrc src[3];
    GDst dst[3];
    const float *W;     // [N][taps][Cin_tot], K contiguous
    const unsigned short *W16;   // bf16 twin of W (mfma_bf16 == 3)
    const float *bias;  // [N] or null
    const float *bias_cls;   // optional [9][N]: added to the rows of border class c = 1..8 of the
                             // OUTPUT frame (3 * (top 1 | bottom 2) + (left 1 | right 2)); class 0
                             // (interior) adds nothing.  A folded constant member's share (flowfold.hip)
    int out_H, out_W;        // output frame of the class test (bias_cls only)
    float *zout;        // optional pre-activation copy, indexed like dst[0]
    int nsrc, ndst;
    int B, Hv, Wv;      // virtual input size (after up-sampling)
    int up;             // UP_*
    int Ho, Wo;         // GEMM row grid
    int stride, pad, ks;
    int nph;            // 1, or 4 sub-pixel output phases (blockIdx.z)
    int ph_pad;         // pad of phase (py,px) = pad - py*ph_pad / pad - px*ph_pad
    long long w_phase_stride;  // elements between the weights of two phases
    long long src_ph_stride;   // elements between the source planes of two phases (0: phases share
                               // the sources; winograd.hip runs its 16 component GEMMs as phases)
    int N, Cin_tot, M;
    int quad;           // rows ordered (b,y,x,dy,dx); epilogue sums the 2x2 quad
    int act;            // forward activation (ACT_*), applied after bias+addend
    int bwd_act;        // activation kind for actsrc
    int mfma_bf16;      // 1: operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16
                        // (f32 accumulate; every tensor stays f32 in memory);
                        // 2: operands split hi + lo, three bf16 products (~2^-16 relative)
                        // 3: bf16 twins: the vector members' p16 and W16 stream through LDS as bf16
                        //    (gconv2 only; other kernels treat it as 1)
    int ph_exact;       // phased stride-2 dgrad: phase (py,px) only has (1+py) x (1+px) non-zero
                        // taps; a kernel MAY skip the others (they multiply zero weights)
    int xcd;            // 1: workgroups are re-mapped so that an XCD (linear id % 8) owns a contiguous
                        // range of (row tile, column tile, phase) with the phase fastest: the phases
                        // and column tiles of one row tile share its input rows in ONE L2
    int dbg;            // DVSOF_GCONV_DBG (timing probes): 1 = skip the epilogue, 2 = one K step, 4 = loads from one L2-resident KiB,
                        // 8 = no DMA in the loop, 16 = no fragment reads, 32 = no barrier,
                        // 128 = every K step re-reads chunk 0 (cache-resident footprint)
};

// winograd.hip: forms shared between consecutive Winograd layers of one frame (dvsof_conv_desc_t.
// winograd_pre / winograd_next / winograd_next_gout)
struct WinoChain {
    const float *v_pre;   // this call's transformed input, made by its producer (skip the input transform)
    float *v_next;        // also write B^T y B of this call's output (the consumer's transformed input)
    float *z_next;        // also write A y A^T (the gradient form of the weight gradient that reads y as gout)
};

// Weight-gradient problem (wgrad.hip)
struct WGradParams {
    GSrc src[3];
    const float *gout;  // [M][Cout]
    float *dW;          // [S][Cout][taps][Cin_tot]
    float *dbias;       // [S][Cout] or null
    int nsrc;
    int B, Hv, Wv, up, Ho, Wo, stride, pad, ks;
    int Cout, Cin_tot, M;
    int klen;             // pixels per K split (multiple of BK)
    int tile_begin[4];    // first column tile of each source; [nsrc] = total
    long long g_sb;       // gout strides (elements): batch, row, pixel
    int g_sy, g_sx;
    int g_py, g_px;       // gout offsets of output phase (py, px)
    int nph, ph_pad, S;   // phases (1|4), pad shift per phase, K splits
    int mfma_bf16;        // as in GConvParams
    long long src_ph_stride;   // elements between the source planes of two phases (winograd.hip)
    const unsigned short *gout16;   // bf16 twin of gout (mfma mode 3) or null
    int twins;            // 1: the vector members and gout are read from their bf16 twins (wgrad2_twins_kernel)
    int xcd;              // 1: XCD-aware workgroup order (an XCD owns a contiguous range of tiles, column
                          // tile fastest: the column tiles of a (row tile, K split) read the same
                          // gradient slab through one L2)
};

// Weight gradient of one flat concat member on the VALU (wgrad.hip), in the
// layer's ORIGINAL geometry (3x3 taps on the nearest-upsampled input).
struct FlatWG {
    GSrc S;
    const float *gout;   // dense [B*Ho*Wo][Cout]
    int B, Hv, Wv, up, Ho, Wo, stride, pad, ks, Cout, M, ncol;
    int coff, Cin_tot;   // column offset / total channels of the layer weight
};

// Mish = x * tanh(softplus(x)).  With n = e^x: tanh(log(1+n)) = t / (t + 2),
// t = n (n + 2) -- one v_exp_f32 and one v_rcp_f32 instead of libm's
// log1pf + tanhf (about 200 VALU instructions per element in an epilogue).
// x is clamped at 20 like torch's softplus threshold (t/(t+2) == 1.f there).
__device__ __forceinline__ float mish_t(float x, float &n)
{
    n = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.4426950408889634f);
    return n * (n + 2.f);
}

// f32 -> bf16 bits, round to nearest even
__device__ __forceinline__ unsigned short bf16_bits(float v)
{
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ float act_fwd(float v, int act)
{
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_MISH) {
        float n;
        const float t = mish_t(v, n);
        return v * (t * __builtin_amdgcn_rcpf(t + 2.f));
    }
    return v;
}

// d act / d pre-activation, given y (ReLU) or z (Mish)
__device__ __forceinline__ float act_bwd(float s, int act)
{
    if (act == ACT_RELU) return s > 0.f ? 1.f : 0.f;
    if (act == ACT_MISH) {
        // th + s (1 - th^2) sigmoid(s);  1 - th = 2 / (t + 2) without cancellation
        float n;
        const float t = mish_t(s, n);
        const float r = __builtin_amdgcn_rcpf(t + 2.f);
        const float th = t * r;
        const float sg = n * __builtin_amdgcn_rcpf(1.f + n);
        return th + s * ((2.f * r) * (1.f + th)) * sg;
    }
    return 1.f;
}

// One 16-wide K slice of a (TM*32) x (TN*32) wave tile from the lane's two
// k-quads per operand.  MODE 0: eight exact f32 MFMAs per block; 1: operands
// rounded to bf16, one 32x32x16 MFMA; 2: bf16 hi + lo split, three products.
template <int MODE, int TM, int TN>
__device__ __forceinline__ void mfma_k16(f32x16 (&acc)[TM][TN], const f32x4 (&a)[2][TM],
                                         const f32x4 (&b)[2][TN])
{
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][tm][i], b[j][tn][i],
                                                                           acc[tm][tn], 0, 0, 0);
    } else {
        bf16x8 ah[TM], bh[TN], al[TM], bl[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const f32x8 v = __builtin_shufflevector(a[0][t], a[1][t], 0, 1, 2, 3, 4, 5, 6, 7);
            ah[t] = __builtin_convertvector(v, bf16x8);
            if constexpr (MODE == 2)
                al[t] = __builtin_convertvector(v - __builtin_convertvector(ah[t], f32x8), bf16x8);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const f32x8 v = __builtin_shufflevector(b[0][t], b[1][t], 0, 1, 2, 3, 4, 5, 6, 7);
            bh[t] = __builtin_convertvector(v, bf16x8);
            if constexpr (MODE == 2)
                bl[t] = __builtin_convertvector(v - __builtin_convertvector(bh[t], f32x8), bf16x8);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if constexpr (MODE == 2) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                }
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
    }
}

// ---------------------------------------------------------------------------
// Epilogue shared by the MFMA convolution kernels.
//
// rowO[d * BM + r] = element offset of GEMM row r inside destination d (batch,
// row, pixel and output-phase terms), -1 for rows past M; the kernels fill it
// next to rowB/rowY/rowX.  The store sweep handles one 32x32 accumulator block
// at a time as: all offsets, all optional loads, all stores.  (A per-element
// load -> wait -> store chain serialises on memory latency: stores count in
// vmcnt on gfx9, so every "wait for my addend" also waits for the previous
// store.  Measured: 21 us of a 30 us launch at 4.2 M outputs.)
// acc[reg] <-> row (reg&3) + 8*(reg>>2) + 4*(lane>>5), column lane&31.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void conv_row_offsets(const GConvParams &P, long long *rowO, int *rowC, int BM,
                                                 int r, bool valid, int b, int oy, int ox, int phy,
                                                 int phx)
{
    if (P.bias_cls) {   // border class of the output pixel (phase problems: 2 * row + phase)
        const int Y = P.nph == 4 ? 2 * oy + phy : oy, X = P.nph == 4 ? 2 * ox + phx : ox;
        rowC[r] = valid ? 3 * (Y == 0 ? 1 : Y == P.out_H - 1 ? 2 : 0) + (X == 0 ? 1 : X == P.out_W - 1 ? 2 : 0)
                        : 0;
    }
    // a flow head folded into member 0's gradient (GDst.head_w; never together with bias_cls):
    // rowC = the row's offset in plane 0 of the flow gradient [B][2][Ho][Wo]
    if (P.dst[0].head_w) rowC[r] = valid ? (b * 2 * P.Ho + oy) * P.Wo + ox : -1;
    if (P.quad) {
        oy >>= 1;
        ox >>= 1;
    }
    for (int d = 0; d < P.ndst; ++d) {
        const GDst &D = P.dst[d];
        rowO[d * BM + r] = valid ? (long long)b * D.sb + (long long)oy * D.sy + (long long)ox * D.sx +
                                       (long long)phy * D.ph_y + (long long)phx * D.ph_x
                                 : -1;
    }
}

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const GConvParams &P, f32x16 (&acc)[TM][TN],
                                              const long long *rowO, const int *rowC, int BM, int n0,
                                              int wr, int wc, int lane)
{
    const int lrow = lane & 31;
    bool has_add = false, has_add2 = false, has_as = false;   // wave-uniform
    for (int d = 0; d < P.ndst; ++d) {
        has_add |= P.dst[d].addend != nullptr;
        has_add2 |= P.dst[d].addend2 != nullptr;
        has_as |= P.dst[d].actsrc != nullptr;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wc * TN + tn) * 32 + lrow;
        const bool col_ok = n < P.N;
        // destination (channel range) of this lane's column
        int dsel = 0, off = 0, run = 0;
        float *dp = P.dst[0].p;
        unsigned short *dp16 = P.dst[0].p16;
        const float *a1 = P.dst[0].addend, *a2 = P.dst[0].addend2, *as = P.dst[0].actsrc;
        int sc = P.dst[0].sc;
        for (int d = 1; d < P.ndst; ++d) {
            run += P.dst[d - 1].C;
            if (n >= run) {
                dsel = d;
                off = run;
                dp = P.dst[d].p;
                dp16 = P.dst[d].p16;
                a1 = P.dst[d].addend;
                a2 = P.dst[d].addend2;
                as = P.dst[d].actsrc;
                sc = P.dst[d].sc;
            }
        }
        const long long *ro = rowO + dsel * BM;
        const long long cpart = (long long)(n - off) * sc;
        const float bias = (P.bias && col_ok) ? P.bias[n] : 0.f;
        // flow head on member 0 (wave-uniform test; the lane's two head weights)
#ifdef DVSOF_NO_GENERAL_HEAD     // (variant build for A/B runs: tools/variant.sh)
        const bool has_head = false;
#else
        const bool has_head = P.dst[0].head_w != nullptr;
#endif
        float hw0 = 0.f, hw1 = 0.f;
        if (has_head && dsel == 0 && col_ok) {
            hw0 = P.dst[0].head_w[n];
            hw1 = P.dst[0].head_w[P.dst[0].C + n];
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int rbase = (wr * TM + tm) * 32 + 4 * (lane >> 5);
            if (!P.quad) {
                long long o[16];
                float v[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const long long r_ = ro[rbase + (reg & 3) + 8 * (reg >> 2)];
                    o[reg] = (col_ok && r_ >= 0) ? r_ + cpart : -1;
                    v[reg] = acc[tm][tn][reg] + bias;
                }
                if (P.bias_cls) {   // wave-uniform; border rows only load
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int c = rowC[rbase + (reg & 3) + 8 * (reg >> 2)];
                        if (c != 0 && col_ok) v[reg] += P.bias_cls[c * P.N + n];
                    }
                }
                if (has_add) {
                    float t[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) t[reg] = (a1 && o[reg] >= 0) ? a1[o[reg]] : 0.f;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) v[reg] = a1 ? v[reg] + t[reg] : v[reg];
                }
                if (has_add2) {
                    float t[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) t[reg] = (a2 && o[reg] >= 0) ? a2[o[reg]] : 0.f;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) v[reg] = a2 ? v[reg] + t[reg] : v[reg];
                }
                if (has_head) {     // + W_h^T g_flow at the row's pixel (dvsof_flow_head_bwd's data part)
                    const long long hwp = (long long)P.Ho * P.Wo;
                    float g0[16], g1[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rg = rowC[rbase + (reg & 3) + 8 * (reg >> 2)];
                        g0[reg] = rg >= 0 ? P.dst[0].head_g[rg] : 0.f;
                        g1[reg] = rg >= 0 ? P.dst[0].head_g[rg + hwp] : 0.f;
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) v[reg] += g0[reg] * hw0 + g1[reg] * hw1;
                }
                if (has_as) {
                    float t[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) t[reg] = (as && o[reg] >= 0) ? as[o[reg]] : 0.f;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        v[reg] = as ? v[reg] * act_bwd(t[reg], P.bwd_act) : v[reg];
                }
                if (P.zout) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (o[reg] >= 0) P.zout[o[reg]] = v[reg];
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] = act_fwd(v[reg], P.act);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    if (o[reg] >= 0) dp[o[reg]] = v[reg];
                if (dp16) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (o[reg] >= 0) dp16[o[reg]] = bf16_bits(v[reg]);
                }
            } else {   // quad rows: the lane's 4 consecutive registers are one low-res pixel
                long long o[4];
                float v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long long r_ = ro[rbase + 8 * g];
                    o[g] = (col_ok && r_ >= 0) ? r_ + cpart : -1;
                    v[g] = (acc[tm][tn][4 * g] + acc[tm][tn][4 * g + 1]) +
                           (acc[tm][tn][4 * g + 2] + acc[tm][tn][4 * g + 3]) + bias;
                }
                float t1[4], t2[4], t3[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    t1[g] = (a1 && o[g] >= 0) ? a1[o[g]] : 0.f;
                    t2[g] = (a2 && o[g] >= 0) ? a2[o[g]] : 0.f;
                    t3[g] = (as && o[g] >= 0) ? as[o[g]] : 0.f;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (a1) v[g] += t1[g];
                    if (a2) v[g] += t2[g];
                    if (as) v[g] *= act_bwd(t3[g], P.bwd_act);
                    if (o[g] >= 0) dp[o[g]] = v[g];
                    if (dp16 && o[g] >= 0) dp16[o[g]] = bf16_bits(v[g]);
                }
            }
        }
    }
}
