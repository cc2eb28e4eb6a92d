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
};

struct GConvParams {
    GSrc src[3];
    GDst dst[3];
    const float *W;     // [N][taps][Cin_tot], K contiguous
    const float *bias;  // [N] or null
    float *zout;        // optional pre-activation copy, indexed like dst[0]
    int nsrc, ndst;
    int B, Hv, Wv;      // virtual input size (after up-sampling)
    int up;             // UP_*
    int Ho, Wo;         // GEMM row grid
    int stride, pad, ks;
    int nph;            // 1, or 4 sub-pixel output phases (blockIdx.z)
    int ph_pad;         // pad of phase (py,px) = pad - py*ph_pad / pad - px*ph_pad
    long long w_phase_stride;  // elements between the weights of two phases
    int N, Cin_tot, M;
    int quad;           // rows ordered (b,y,x,dy,dx); epilogue sums the 2x2 quad
    int act;            // forward activation (ACT_*), applied after bias+addend
    int bwd_act;        // activation kind for actsrc
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
};

// Weight gradient of one flat concat member on the VALU (wgrad.hip), in the
// layer's ORIGINAL geometry (3x3 taps on the nearest-upsampled input).
struct FlatWG {
    GSrc S;
    const float *gout;   // dense [B*Ho*Wo][Cout]
    int B, Hv, Wv, up, Ho, Wo, stride, pad, ks, Cout, M, ncol;
    int coff, Cin_tot;   // column offset / total channels of the layer weight
};

__device__ __forceinline__ float act_fwd(float v, int act)
{
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_MISH) {
        // x * tanh(softplus(x)), softplus thresholded like torch (beta=1, 20)
        const float sp = v > 20.f ? v : log1pf(__expf(v));
        return v * tanhf(sp);
    }
    return v;
}

// d act / d pre-activation, given y (ReLU) or z (Mish)
__device__ __forceinline__ float act_bwd(float s, int act)
{
    if (act == ACT_RELU) return s > 0.f ? 1.f : 0.f;
    if (act == ACT_MISH) {
        const float sp = s > 20.f ? s : log1pf(__expf(s));
        const float th = tanhf(sp);
        const float sg = 1.f / (1.f + __expf(-s));
        return th + s * (1.f - th * th) * sg;
    }
    return 1.f;
}
