/*
 * dvsof.h -- C ABI of libdvsof_hip.so, the MI355X (gfx950) implementation of
 * the optical-flow training hot path of e-sha/dvs_of_training_framework.
 *
 * Conventions
 *   - every entry point returns int: 0 = ok, >0 = hipError_t of the failing
 *     HIP call, <0 = DVSOF_E* argument error.  No exceptions cross the ABI.
 *   - all pointers are DEVICE pointers unless a parameter is named host_*.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Calls only enqueue work: no allocation, no synchronisation, no
 *     host<->device copies, so a caller may capture them into a hipGraph.
 *   - scratch memory is provided by the caller (sizes from *_workspace_bytes);
 *     the library retains no pointer after a call returns.
 *   - tensors are dense, row-major, float32 unless stated.  "NCHW"/"NHWC"
 *     name the memory order.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the reference repository root).
 */
#ifndef DVSOF_H
#define DVSOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVSOF_VERSION 100 /* 0.1.0 */
#define DVSOF_MAX_SCALES 8

enum {
    DVSOF_OK = 0,
    DVSOF_EINVAL = -1,  /* bad shape / null pointer / unsupported size */
    DVSOF_ENOSPACE = -2, /* workspace too small */
    DVSOF_ECOMM = -3     /* RCCL missing or a collective / communicator call failed */
};

int dvsof_version(void);
/* static string for a code returned by any entry point */
const char *dvsof_error_string(int code);

/* ------------------------------------------------------------------ *
 * Event stream  ->  images / voxel grids
 * ------------------------------------------------------------------ */

/*
 * Per-pixel event count (integer histogram, idx = y*W + x).
 * Replaces get_count_image, utils/data.py:120-136 (numpy add.at, uint64).
 * out[H*W] uint32 is zeroed by the call.  Events outside the frame are
 * ignored (the reference raises; the Python wrapper checks and raises).
 */
int dvsof_count_image(const int64_t *x, const int64_t *y, int64_t n_events,
                      int H, int W, uint32_t *out, void *stream);

/*
 * Event batch -> polarity-signed, time-bilinear voxel grid out[B,C,H,W]
 * (NCHW, zeroed by the call), docs/VOXEL_SPEC.md.
 * Replaces EV_FlowNet Model.quantize / quantization_layer as called at
 * utils/training.py:59-64 and scripts/quantize_preprocessed.py:87-91; output
 * contract utils/dataset.py:436-448.  Event columns are the reference's wire
 * format (utils/dataset.py:961-1020): int64 x,y,polarity,sample_index and
 * float32 window-relative timestamp.  t0[b], t1[b] are the window of sample b.
 * n_events may be 0 (utils/loss.py:217-240 probes the model that way).
 * polarity is a SIGN: > 0 adds, < 0 subtracts, 0 contributes nothing.
 * bin0 (int32[n]) / lin0 (int64[n]) are optional debug outputs: the lower
 * temporal bin and the linear index of (b,bin0,y,x), -1 for dropped events.
 */
int dvsof_voxelize_fwd(const int64_t *x, const int64_t *y, const float *t,
                       const int64_t *polarity, const int64_t *sample_index,
                       int64_t n_events, const float *t0, const float *t1,
                       int B, int C, int H, int W, float *out, int32_t *bin0,
                       int64_t *lin0, void *stream);

/*
 * Same result through LDS-staged voxel tiles, two launches: events are
 * bucketed per 32x32 tile in one coalesced pass, each tile is accumulated in
 * LDS and stored with plain coalesced writes (no zero-fill pass, no global
 * float atomics).  Falls back to dvsof_voxelize_fwd for tiny inputs or when
 * the workspace is missing/too small.  Integer parts are identical; float
 * sums differ only in accumulation order.
 *
 * Workspace contract: the first dvsof_voxelize_control_bytes() bytes are
 * control words (bucket cursors, counters).  They must be ZERO when the call
 * starts executing and the call leaves them zero again (the kernels clean up
 * after themselves).  flags:
 *   0                   the call zero-fills the control words first (a fill
 *                       KERNEL: no entry point of this library enqueues
 *                       hipMemsetAsync, so a stream capture of it holds kernel
 *                       nodes only)
 *   DVSOF_VOX_WS_CLEAN  the caller guarantees they are zero (zero-filled once,
 *                       then only used by calls of this family, one at a time
 *                       in stream order): no fill is enqueued
 */
#define DVSOF_VOX_WS_CLEAN 1
size_t dvsof_voxelize_workspace_bytes(int64_t n_events, int B, int C, int H,
                                      int W);
/* 0 when the thread-per-event kernel serves this size (no control words) */
size_t dvsof_voxelize_control_bytes(int64_t n_events, int B, int C, int H,
                                    int W);
int dvsof_voxelize_tiled(const int64_t *x, const int64_t *y, const float *t,
                         const int64_t *polarity, const int64_t *sample_index,
                         int64_t n_events, const float *t0, const float *t1,
                         int B, int C, int H, int W, float *out, int32_t *bin0,
                         int64_t *lin0, void *workspace,
                         size_t workspace_bytes, int flags, void *stream);

/*
 * The same grid from the reference's ENCODED event columns (encode_batch,
 * utils/dataset.py:240-305: int16 x, int16 y, float32 timestamp, bool
 * polarity = 9 bytes/event instead of 44).  sample_event_offsets[B+1] = index
 * of the first event of every sample (from events_per_element /
 * elements_per_sample).  Workspace and flags: as dvsof_voxelize_tiled.
 */
int dvsof_voxelize_encoded(const int16_t *x, const int16_t *y, const float *t,
                           const uint8_t *polarity,
                           const int64_t *sample_event_offsets,
                           int64_t n_events, const float *t0, const float *t1,
                           int B, int C, int H, int W, float *out,
                           int32_t *bin0, int64_t *lin0, void *workspace,
                           size_t workspace_bytes, int flags, void *stream);

/* ------------------------------------------------------------------ *
 * Data augmentation on the device: horizontal
 * flip -> nearest-neighbour LUT rotation about the frame centre -> crop, the
 * order of utils/dataset.py:753-769.  Per-sample parameters are device
 * arrays: flip u8[B]; cos_sin f64[B][2] (cos, sin of the angle, computed on
 * the host in float64 as utils/data.py:181-186 does); box i32[B][4] =
 * (y0, x0, h, w) with one (h, w) for the whole batch.
 * ------------------------------------------------------------------ */

/* lut[b][source pixel] = rotated pixel that reads it (smallest index when
 * several do; >= H*W: none).  Replaces the (src, dst) index lists
 * RandomRotation hands to the native transformation.map, utils/data.py:199-215. */
int dvsof_augment_lut(const double *cos_sin, int B, int H, int W, int32_t *lut,
                      void *stream);

/* dst f32[D,h,w] = crop(rotate(flip(src[D,H,W]))); src is u8 or f32;
 * frame_sample i32[D] = sample of every frame.  utils/dataset.py:755-757,
 * utils/data.py:203-206 (rimages), utils/data.py:45-75 (crop). */
int dvsof_augment_frames(const void *src, int src_is_u8, int D, int H, int W,
                         const int32_t *frame_sample, const uint8_t *flip,
                         const double *cos_sin, const int32_t *box, int h,
                         int w, float *dst, void *stream);

/* Event coordinates through the same flip / rotation / crop
 * (utils/dataset.py:758, utils/data.py:208-213, utils/data.py:24-42).  lut
 * may be NULL (no rotation).  Removed events get x_out = y_out = -1 instead of
 * being compacted away (the voxeliser skips them). */
int dvsof_augment_events(const int64_t *x, const int64_t *y,
                         const int64_t *sample_index, int64_t n_events,
                         const uint8_t *flip, const int32_t *lut,
                         const int32_t *box, int B, int H, int W,
                         int64_t *x_out, int64_t *y_out, void *stream);

/* ------------------------------------------------------------------ *
 * Multi-scale warp / Charbonnier / smoothness / out-of-border loss
 * ------------------------------------------------------------------ */

/*
 * Bilinear resize with align_corners=True of n single-channel images.
 * Replaces utils/loss.py:20-21 (F.interpolate) as used for the CASCADED frame
 * pyramid at utils/loss.py:207-210.
 */
int dvsof_resize_bilinear_ac(const float *src, float *dst, int n, int hin,
                             int win, int hout, int wout, void *stream);

/*
 * The whole CASCADED pyramid of utils/loss.py:207-210 (level 0 resamples
 * images [D,H,W] to (hs[0],ws[0]); level k resamples level k-1) into
 * levels[k] = float[D*hs[k]*ws[k]].  One launch when the level sizes are
 * non-decreasing (the coarse -> fine order Losses uses), otherwise one resize
 * launch per level; bitwise the same values either way.
 */
int dvsof_loss_pyramid(const float *images, int D, int H, int W,
                       float *const *levels, const int *hs, const int *ws,
                       int num_levels, void *stream);

typedef struct {
    const float *frames; /* [D,h,w] all frames of the batch at this scale */
    const float *flow;   /* [N,2,h,w] predicted flow, pixels of this scale */
    float *grad_flow;    /* [N,2,h,w] written by *_bwd / *_fused, else NULL */
    int h, w;
} dvsof_loss_scale_t;

/* bytes of workspace dvsof_loss_* need for these scales (0 on bad input) */
size_t dvsof_loss_workspace_bytes(const dvsof_loss_scale_t *host_scales,
                                  int num_scales, int N);

/*
 * Forward of Losses.__call__ / Loss.__call__, utils/loss.py:121-171,179-214,
 * for all scales in one launch sequence.
 *   start_idx/stop_idx  int32[N]: frame index of the first/second image of
 *                       prediction n (resolved on the host from timestamps,
 *                       utils/loss.py:182-206)
 *   terms               float[3*num_scales]: smoothness_k, photometric_k,
 *                       outborder_k (row order as returned at utils/loss.py:171)
 *   oob_count           int32[num_scales*N]: out-of-border pixels per sample
 *                       (utils/loss.py:101), kept for the backward pass
 */
int dvsof_loss_fwd(const dvsof_loss_scale_t *host_scales, int num_scales,
                   int N, const int32_t *start_idx, const int32_t *stop_idx,
                   float *terms, int32_t *oob_count, void *workspace,
                   size_t workspace_bytes, void *stream);

/*
 * Backward: grad_flow_k = d(sum_t seeds[t,k] * term[t,k]) / d flow_k, i.e.
 * what loss.backward() (utils/training.py:158) yields through utils/loss.py.
 * seeds: device float[3*num_scales] (upstream gradients of the 12 scalars).
 * oob_count: as produced by dvsof_loss_fwd on the same inputs.
 */
int dvsof_loss_bwd(const dvsof_loss_scale_t *host_scales, int num_scales,
                   int N, const int32_t *start_idx, const int32_t *stop_idx,
                   const float *seeds, const int32_t *oob_count, void *stream);

/*
 * Training fast path: terms AND the gradient of
 *   loss = sum_t host_weights[t] * mean_k term[t,k] * loss_scale
 * (combined_loss, utils/training.py:12-24, with the 1/accumulation_steps of
 * utils/training.py:156 folded into loss_scale) in one sweep over the inputs.
 * Also writes the scalar loss to loss_out[0].
 */
int dvsof_loss_fused(const dvsof_loss_scale_t *host_scales, int num_scales,
                     int N, const int32_t *start_idx, const int32_t *stop_idx,
                     const float *host_weights /*[3]*/, float loss_scale,
                     float *terms, float *loss_out, int32_t *oob_count,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * dvsof_loss_fused that first builds the frame pyramid from images [D,H,W]
 * INTO host_scales[k].frames (as dvsof_loss_pyramid does), sharing a launch
 * with the out-of-border count: Losses.__call__'s interpolate cascade
 * (utils/loss.py:207-210) + terms + gradient in 4 launches.
 */
int dvsof_loss_fused_pyramid(const float *images, int D, int H, int W,
                             const dvsof_loss_scale_t *host_scales,
                             int num_scales, int N, const int32_t *start_idx,
                             const int32_t *stop_idx,
                             const float *host_weights /*[3]*/,
                             float loss_scale, float *terms, float *loss_out,
                             int32_t *oob_count, void *workspace,
                             size_t workspace_bytes, void *stream);


/* ------------------------------------------------------------------ *
 * EV_FlowNet predictor: convolution stack on the f32 matrix cores
 * (docs/MODEL_SPEC.md).  Replaces the ATen conv2d / upsample / cat /
 * activation ops and their autograd inside EV_FlowNet.net.Model.predictor as
 * called at utils/training.py:59-64 and differentiated at
 * utils/training.py:158 (source absent upstream; contract from
 * DummyNet/net.py:59-80 and tests/training/test_training.py:45-46).
 * ------------------------------------------------------------------ */

enum { DVSOF_ACT_NONE = 0, DVSOF_ACT_RELU = 1, DVSOF_ACT_MISH = 2 };
enum { DVSOF_NHWC = 0, DVSOF_NCHW = 1 };

typedef struct {
    const float *p; /* dense tensor [B,H,W,C] (NHWC) or [B,C,H,W] (NCHW) */
    int C;
    int layout; /* DVSOF_NHWC | DVSOF_NCHW */
    const void *p16; /* optional bf16 twin of an NHWC tensor (same shape, same
                        values rounded to bf16): read instead of p by the
                        matrix-core kernels in mfma mode 3 */
} dvsof_src_t;

/*
 * One conv layer y = act(conv(up(cat(src...)), W) + bias [+ residual]):
 * the input is the channel concatenation of nsrc tensors of spatial size
 * H x W, optionally 2x nearest-upsampled first; ksize x ksize taps, zero
 * padding `pad`, stride 1 or 2.  Output is NHWC [B,Ho,Wo,Cout] with
 * Ho = (H*(upsample ? 2 : 1) + 2*pad - ksize)/stride + 1.
 * Weights are [Cout][ksize][ksize][Ctot] (channels contiguous, Ctot = sum C).
 *
 * upsample = 2 makes the layer a TRANSPOSED convolution with stride 2 (the
 * north star's "strided conv / transposed-conv / Mish stack"; the reference's
 * network source is absent, docs/MODEL_SPEC.md keeps nearest up-sampling as
 * the decoder's default): zeros are inserted between the input pixels
 * (xz[2i][2j] = x[i][j]) before the 3x3 / pad-1 convolution, Ho = 2H.  With
 * W' = W flipped in both tap axes and its channel axes swapped this is
 * torch conv_transpose2d(x, W', stride 2, padding 1, output_padding 1).
 * Forward: four output-parity phases of (1+py)(1+px) taps on the
 * low-resolution input (prepared form [4][Cout][2][2][Ctot]); data gradient:
 * a stride-2 3x3 convolution of the output gradient with the flip-transposed
 * weights; weight gradient: that of the adjoint stride-2 layer, flip-
 * transposed; bias gradient: channel sums of the output gradient.
 */
typedef struct {
    dvsof_src_t src[3];
    int nsrc;
    int B, H, W;
    int upsample; /* 0 | 1 = 2x nearest | 2 = 2x zero insertion (transposed
                     convolution, stride 2: one NHWC source with C % 16 == 0,
                     ksize 3, pad 1, stride 1; see below) */
    int ksize, stride, pad;
    int Cout;
    int act; /* DVSOF_ACT_* */
    int mfma; /* 0: f32 matrix cores (exact products); 1: operands rounded to
                 bf16 in registers, v_mfma_f32_32x32x16_bf16, f32 accumulate;
                 2: operands split into bf16 hi + lo, products hi*hi + hi*lo +
                 lo*hi (the dropped lo*lo term is 2^-16 relative).
                 Every tensor stays f32 in memory in modes 0-2.
                 3: bf16 TWINS -- like 1, but the forward / data-gradient
                 kernels stream bf16 copies of their NHWC operands (src[i].p16,
                 gout16) and of the prepared weights (w16) through LDS: half
                 the LDS-DMA bytes mode 1 is bound by, no conversion in the K
                 loop.  Producers write the twin of their output next to the
                 f32 tensor (y16, dst[i].p16, dvsof_flow_head_bwd's gx16); the
                 weight gradient, the heads and the loss keep reading f32.
                 Needs every NHWC source channel count % 32 == 0, else the
                 call runs as mode 1. */
    void *scratch;        /* device scratch for layers that run as Winograd */
    size_t scratch_bytes; /* F(2x2,3x3) (dvsof_conv2d_scratch_bytes > 0); read by
                             dvsof_conv2d_fwd / _dgrad only, which return
                             DVSOF_ENOSPACE when it is missing or too small */
    const float *winograd_input; /* optional, read by dvsof_conv2d_wgrad only:
                             dvsof_conv2d_fwd leaves the transformed input of a
                             Winograd layer at the start of `scratch`; a caller
                             that kept that buffer intact passes it here and the
                             weight gradient skips its own input transform
                             (used when both run the same tile form) */
    /* mode 3 only (NULL otherwise) */
    void *y16;           /* dvsof_conv2d_fwd: bf16 twin of y, written */
    const void *w16;     /* bf16 twin of the weight argument of the call
                            (dvsof_to_bf16 of the prepared form) */
    const void *gout16;  /* dvsof_conv2d_dgrad / _wgrad: bf16 twin of gout */
    int flags;           /* DVSOF_CONV_* bits */
    const float *bias_cls; /* dvsof_conv2d_fwd, optional: [9][Cout] added to the
                            output rows of border class c = 3 * (top 1 |
                            bottom 2) + (left 1 | right 2); row 0 (interior)
                            is not read.  The position-dependent share of a
                            folded constant member (dvsof_flow_fold_weights) */
    /* Forms shared by CONSECUTIVE Winograd layers of one frame (the residual
     * chain), all optional; dvsof_conv2d_winograd_chain(desc, kind) says whether
     * this layer's forward / data gradient can write them (F(4x4) form, <= 64
     * tiles per image), DVSOF_EINVAL otherwise.  A producer then makes the
     * consumer's transforms in its own output transform (one launch and one pass
     * over the tensor instead of three; same bits):
     *   winograd_next       fwd / dgrad: also write the transformed input
     *                       [36][T][channels of the output] of the layer that
     *                       consumes this call's output (same B, H, W);
     *   winograd_pre        fwd / dgrad: this call's transformed input as a
     *                       producer wrote it -- the input transform is skipped
     *                       (fwd: it is also what winograd_input names for the
     *                       weight gradient);
     *   winograd_next_gout  dgrad: also write the gradient form [36][T][C] of
     *                       the output, for the weight gradient that takes it as
     *                       gout;
     *   winograd_gout       wgrad: that form of gout (used when the weight
     *                       gradient runs the F(4x4) form, else ignored). */
    const float *winograd_pre;
    float *winograd_next;
    float *winograd_next_gout;
    const float *winograd_gout;
} dvsof_conv_desc_t;
/* dvsof_conv2d_wgrad leaves the columns of the narrow planar members (the
 * 2-channel flow) unwritten: dvsof_flow_fold_grads fills them (below) */
#define DVSOF_CONV_WGRAD_SKIP_FLAT 1

/*
 * Bytes of scratch dvsof_conv2d_fwd / dvsof_conv2d_dgrad need for this layer:
 * non-zero for the wide 3x3 stride-1 layers (one NHWC source, channel counts
 * multiples of 64 and >= 256, even H and W, mfma != 1), which are evaluated as
 * Winograd convolutions -- F(4x4,3x3) when H and W are multiples of 4 (4x fewer
 * multiply-adds), else F(2x2,3x3) (2.25x fewer): input transform, NG = 36 | 16
 * component GEMMs on the matrix cores, output transform with the fused
 * epilogue.  Same result up to fp32 rounding of the transforms (~1e-5 | ~1e-6
 * of the output peak).  Their prepared weights are U[NG][Cout][Ctot] forward
 * and U'[NG][Ctot][Cout] backward; dvsof_conv2d_prepare makes either or both
 * from the raw weights (weight must not be NULL for these layers).  The
 * scratch holds the transformed input and the component products; it is only
 * used during the call.  The weight gradient of such a layer runs through the
 * same transforms inside its ordinary workspace.
 */
size_t dvsof_conv2d_scratch_bytes(const dvsof_conv_desc_t *desc);
/* 0: direct implicit GEMM; 2 | 4: Winograd output tile side used for this
 * problem (kind 0 forward, 1 data gradient, 2 weight gradient); for tools. */
int dvsof_conv2d_winograd_tile(const dvsof_conv_desc_t *desc, int kind);
/* 1: the forward (kind 0) / data gradient (kind 1) of this layer accepts
 * winograd_next [/ winograd_next_gout] (see dvsof_conv_desc_t) */
int dvsof_conv2d_winograd_chain(const dvsof_conv_desc_t *desc, int kind);

/*
 * Prepared weights.  An upsampled 3x3/pad-1 layer is evaluated as four 2x2
 * sub-pixel phase convolutions on the low-resolution input (16 instead of 36
 * tap-products per low-resolution pixel, same result up to fp32 summation
 * order): forward weights Wf[4][Cout][2][2][Ctot], data-gradient weights the
 * 4x4 stride-2 kernel Wd[Ctot][4][4][Cout] -- or, in exact f32 (mfma == 0)
 * with two NHWC members of 32 | C, 32 | Cout, 16 | W, 4 | H, by the minimal
 * bilinear algorithm of csrc/fwd_min.hip with NINE products: forward weights
 * Wt[9][Cout][Ctot] = G w G^T, and (64 | C, 8 | H) data-gradient weights
 * W'[9][Ctot][Cout] = G' w G'^T.  Which form a buffer holds is a function of
 * the descriptor alone (the same test picks the kernel at launch); the buffer
 * sizes are those of the sub-pixel forms either way.  Every other layer uses
 * the raw weights forward and their tap-flipped transpose backward.
 * dvsof_conv2d_prepare fills w_fwd (may be NULL when
 * dvsof_conv2d_fwd_weight_elems == Cout*k*k*Ctot: forward then takes the raw
 * weights) and w_dgrad (may be NULL) from the raw [Cout][k][k][Ctot] weights.
 * For a sub-pixel layer w_fwd == NULL with w_dgrad makes the data-gradient
 * form alone, from the raw weights (lets a caller make it later, e.g. on
 * another stream).  (Sixteen-product form only: weight == NULL means "w_fwd
 * already holds the phase kernels of an earlier call" and w_dgrad is derived
 * from them -- DVSOF_EINVAL for a layer whose forward form is Wt.)
 */
size_t dvsof_conv2d_fwd_weight_elems(const dvsof_conv_desc_t *desc);
size_t dvsof_conv2d_dgrad_weight_elems(const dvsof_conv_desc_t *desc);
int dvsof_conv2d_prepare(const dvsof_conv_desc_t *desc, const float *weight,
                         float *w_fwd, float *w_dgrad, void *stream);

/* weight = prepared forward weights (see above).
 * y (and z = pre-activation, optional, for Mish backward) are NHWC. */
int dvsof_conv2d_fwd(const dvsof_conv_desc_t *desc, const float *weight,
                     const float *bias, const float *residual, float *y,
                     float *z, void *stream);

typedef struct {
    float *p;            /* gradient w.r.t. desc->src[i], same layout/shape */
    const float *addend; /* optional: p = result + addend (may alias p) */
    const float *addend2; /* optional second addend */
    const float *actsrc; /* optional: p *= act'(actsrc), the producer's y
                            (ReLU) or z (Mish): yields its dz directly */
    void *p16;           /* optional bf16 twin of p, written (mode 3) */
    /* optional, only where dvsof_conv2d_dgrad_fuses_head(desc) says so (the
     * nine-product data gradient, csrc/dgrad_min.hip): the flow head that hangs
     * on this source is folded into the epilogue,
     *   p = (result + addends + head_w[0][c] gflow[b][0][y][x] + head_w[1][c] gflow[b][1][y][x]) * act'(actsrc)
     * head_w [2][C] (the head's 1x1 weights), head_gflow [B][2][H][W] planar (the
     * flow's loss gradient): what dvsof_flow_head_bwd computes as gx, without
     * the pass over the tensor in between (the head's own weight gradient:
     * dvsof_flow_head_bwd with gx == NULL).  DVSOF_EINVAL on any other kernel. */
    const float *head_w;
    const float *head_gflow;
    /* ... and, optionally with them, the head's OWN weight / bias gradient from
     * the same pass: head_x = the head's input (the tensor p is the gradient of;
     * may equal actsrc), head_part = [dvsof_conv2d_dgrad_head_rows(desc)][2 C + 2]
     * floats of per-block partial sums, every element written:
     * dvsof_flow_head_reduce(head_part, rows, C, dw, dbias) adds them up in a
     * fixed order (instead of dvsof_flow_head_bwd's second pass over x). */
    const float *head_x;
    float *head_part;
} dvsof_grad_dst_t;

/*
 * Data gradient: for every source i, dst[i].p = d loss / d src[i] given
 * gout = d loss / d (pre-activation output) [B,Ho,Wo,Cout] NHWC.
 * weight_t is the prepared data-gradient weight (dvsof_conv2d_prepare; for a
 * plain layer the tap-flipped transpose [Ctot][ksize][ksize][Cout]).  dst[i].p == NULL skips nothing (all sources
 * are computed together); pass a scratch buffer if a gradient is not needed.
 */
/* 1 when the data gradient of this layer accepts dvsof_grad_dst_t.head_w /
 * head_gflow (shape test alone; see above) */
int dvsof_conv2d_dgrad_fuses_head(const dvsof_conv_desc_t *desc);
/* rows of dvsof_grad_dst_t.head_part for this layer (0: not available) */
int dvsof_conv2d_dgrad_head_rows(const dvsof_conv_desc_t *desc);
/* dw [2][C], dbias [2] (may be NULL) = column sums of part [rows][2 C + 2] */
int dvsof_flow_head_reduce(const float *part, int rows, int C, float *dw, float *dbias, void *stream);
int dvsof_conv2d_dgrad(const dvsof_conv_desc_t *desc, const float *weight_t,
                       const float *gout, const dvsof_grad_dst_t *dst,
                       int bwd_act, void *stream);

size_t dvsof_conv2d_wgrad_workspace_bytes(const dvsof_conv_desc_t *desc);
/* dweight [Cout][k][k][Ctot], dbias [Cout] (optional); overwritten. */
int dvsof_conv2d_wgrad(const dvsof_conv_desc_t *desc, const float *gout,
                       float *dweight, float *dbias, void *workspace,
                       size_t workspace_bytes, void *stream);


/*
 * Which tile shape the library picks for this problem (for profiling tools):
 * kind 0 = forward, 1 = data gradient, 2 = weight gradient.  Returns the
 * tile id (1..5: 128x128, 128x64, 64x64, 256x32|64x128, 128x32|32x128) or <0.
 */
int dvsof_conv2d_tile_id(const dvsof_conv_desc_t *desc, int kind);
/* kernel generation serving the problem: 2 = LDS-DMA ring (gconv2/wgrad2),
 * 1 = register-staged (gconv/wgrad), 0 = VALU-only (flat members),
 * 3 = the first encoder layer's own kernels (csrc/first.hip: one planar source
 * of <= 16 channels, 3x3 stride 2 pad 1, 64 outputs, even frame sides; forward
 * and weight gradient, exact f32 in every operand mode). */
int dvsof_conv2d_kernel_generation(const dvsof_conv_desc_t *desc, int kind);
/* What the calling thread's LAST dvsof_conv2d_fwd (kind 0) / _dgrad (1) / _wgrad
 * (2) ran for a decoder stage: 0 the general kernels, 1 a patch-resident
 * sixteen-product kernel (csrc/fwd_patch.hip, csrc/wgrad_patch.hip), 2 the
 * nine-product minimal form (csrc/fwd_min.hip, dgrad_min.hip, wgrad_min.hip:
 * exact f32; 9 instead of 16 matrix products per low-resolution pixel).  The
 * choice depends on operand mode, twins and shape (profiling tools: bench.py
 * names its roofline groups and counts executed FLOPs with it) */
int dvsof_conv2d_last_patch(int kind);

/* wt[ci][k*k-1-tap][co] = w[co][tap][ci] */
int dvsof_weight_flip_transpose(const float *w, float *wt, int Cout, int ksize,
                                int Ctot, void *stream);

/*
 * Flow head: flow[b,j,y,x] = sum_c w[j][c] * x[b,y,x,c] + bias[j], j = 0,1.
 * x NHWC [B,H,W,C] (C % 4 == 0), flow NCHW [B,2,H,W] (the layout the loss
 * and the reference's prediction contract use, utils/training.py:59-64).
 */
int dvsof_flow_head_fwd(const float *x, const float *w, const float *bias,
                        float *flow, int B, int H, int W, int C, void *stream);
/* n <= 4 heads in one launch (host arrays of device pointers / sizes; bias may
 * be NULL or hold NULLs): the training forward computes all flows ahead of the
 * loss once nothing between the decoder stages reads them. */
int dvsof_flow_heads_fwd(int n, const float *const *host_x,
                         const float *const *host_w,
                         const float *const *host_bias, float *const *host_flow,
                         int B, const int *host_H, const int *host_W,
                         const int *host_C, void *stream);

size_t dvsof_flow_head_bwd_workspace_bytes(int B, int H, int W, int C);
/*
 * Backward of the head fused with the activation backward of the decoder
 * layer that produced x:
 *   gx = (gx_in + w^T gflow) * act'(actsrc)      (gx_in optional, may alias gx)
 *   dw[j][c] = sum gflow_j * x_c ,  dbias[j] = sum gflow_j
 */
int dvsof_flow_head_bwd(const float *x, const float *w, const float *gflow,
                        const float *gx_in, const float *actsrc, int act,
                        float *gx, float *dw, float *dbias, int B, int H, int W,
                        int C, void *workspace, size_t workspace_bytes,
                        void *gx16 /* optional bf16 twin of gx, written */,
                        void *stream);

/* dst[i] = bf16(src[i]) (round to nearest even), n elements: the bf16 twins of
 * the prepared weights for mfma mode 3 */
int dvsof_to_bf16(const float *src, void *dst, size_t n, void *stream);
/* up to 16 tensors in ONE launch (host arrays of device pointers / sizes) */
int dvsof_to_bf16_many(const float *const *host_src, void *const *host_dst,
                       const size_t *host_n, int count, void *stream);
/* dvsof_conv2d_prepare that also writes the bf16 twins of the forms it makes
 * (compute mode 3, "bf16 twins"), from the same kernels: w_fwd16 = twin of the
 * forward form -- of the RAW weight where that is the forward form --,
 * w_dgrad16 = twin of the data-gradient form; either may be NULL. */
int dvsof_conv2d_prepare16(const dvsof_conv_desc_t *desc, const float *weight,
                           float *w_fwd, float *w_dgrad, void *w_fwd16,
                           void *w_dgrad16, void *stream);

/* dz = dy * act'(actsrc), n elements (dz may alias dy) */
int dvsof_act_bwd(const float *dy, const float *actsrc, int act, float *dz,
                  size_t n, void *stream);


/* ------------------------------------------------------------------ *
 * Optimizer
 * ------------------------------------------------------------------ */

/* elements one workgroup of dvsof_adamw_step handles (chunk table unit) */
int dvsof_adamw_chunk_elems(void);

/*
 * Fused multi-tensor AdamW step (decoupled weight decay, optional amsgrad) for
 * one parameter group.  Replaces torch.optim.AdamW(amsgrad=True).step() as
 * built at train_flownet.py:57-75 and called at utils/training.py:164.
 *   ptrs    device uint64[5*T]: {param, grad, exp_avg, exp_avg_sq,
 *           max_exp_avg_sq} pointers of tensor t (float32, dense)
 *   sizes   device int64[T]: element counts
 *   chunks  device int32[2*num_chunks]: (tensor id, chunk index) per
 *           workgroup, chunk = dvsof_adamw_chunk_elems() elements
 *   step    1-based step count (bias correction)
 */
int dvsof_adamw_step(const uint64_t *ptrs, const int64_t *sizes,
                     const int32_t *chunks, int num_chunks, float lr,
                     float beta1, float beta2, float eps, float weight_decay,
                     int step, int amsgrad, void *stream);

/*
 * The same update for a step that is CAPTURED in a hipGraph (the train loop of
 * utils/training.py:138-167 replayed as one graph launch): kernel arguments
 * are frozen at capture, so what changes per step -- the scheduled learning
 * rate (LambdaLR, train_flownet.py:91-109) and the bias corrections -- is read
 * from `dyn`, device float[3] = {lr, lr/(1-beta1^t), sqrt(1-beta2^t)}.
 * dvsof_adamw_dynamic fills a HOST float[3] with exactly the values
 * dvsof_adamw_step would use (the caller puts it into `dyn` before each replay:
 * dvsof_adamw_set_dynamic),
 * so both entry points give bit-identical parameters.
 */
void dvsof_adamw_dynamic(float lr, float beta1, float beta2, int step,
                         float *host_out3);
/* host_values[n] -> dyn[n] (device) in stream order, the values travelling as
 * kernel arguments: no copy engine, no fence in front of the step's first
 * kernel; host_values may be reused as soon as the call returns */
int dvsof_adamw_set_dynamic(float *dyn, const float *host_values, int n,
                            void *stream);
int dvsof_adamw_step_dyn(const uint64_t *ptrs, const int64_t *sizes,
                         const int32_t *chunks, int num_chunks,
                         const float *dyn, float beta1, float beta2, float eps,
                         float weight_decay, int amsgrad, void *stream);


/*
 * Fused multi-tensor RAdam / Ranger step for one parameter group.  Replaces
 * RAdam.radam.RAdam and ranger.Ranger (train_flownet.py:62-71; un-vendored
 * submodules upstream -- arithmetic per Liu et al. 2020 and Lookahead, Zhang
 * et al. 2019; oracle/ref_optim.py).  Tables as for dvsof_adamw_step;
 * ptrs[5t+4] is the Lookahead slow buffer (Ranger) or unused (RAdam).
 *   degenerate_to_sgd  bit 0: take an un-rectified momentum step while the
 *                      variance is not tractable (both upstream defaults);
 *                      bit 1: RAdam's ">= threshold" rule instead of Ranger's ">"
 *   lookahead_now      1 on every k-th step of Ranger, else 0
 *   lookahead_alpha    slow-weight step (0 = no slow buffer: plain RAdam)
 */
int dvsof_radam_step(const uint64_t *ptrs, const int64_t *sizes,
                     const int32_t *chunks, int num_chunks, float lr,
                     float beta1, float beta2, float eps, float weight_decay,
                     int step, float nsma_threshold, int degenerate_to_sgd,
                     int lookahead_now, float lookahead_alpha, void *stream);

/* Gradient centralisation (Ranger): grad[r][:] -= mean(grad[r][:]). */
int dvsof_grad_centralize(float *grad, int rows, int row_len, void *stream);

/* ------------------------------------------------------------------ *
 * The flow member of a decoder stage in weight space (csrc/flowfold.hip).
 * Stage i convolves cat[x, skip, flow] with flow = Wh x + bh, the previous
 * stage's 1x1 flow head applied to the x member.  For the BACKWARD
 * (utils/training.py:158 through the network) the member is folded away:
 *   dvsof_flow_fold_weights   w_eff[Cout][9][Ctot-2] = the layer's weights
 *       without the two flow columns, x columns += Wflow . Wh: the data
 *       gradient then runs on cat[x, skip] alone and already contains the
 *       path through the flow head (the head's backward gets the flow's LOSS
 *       gradient only);
 *   dvsof_flow_fold_grads     after dvsof_conv2d_wgrad with
 *       DVSOF_CONV_WGRAD_SKIP_FLAT: writes the flow columns of dW from its x
 *       columns, Wh, bh, the bias gradient db_conv and the border sums of the
 *       output gradient g [B,H,W,Cout], and ADDS the stage's contribution to
 *       the head's gradients dwh [2][Cx], dbh [2].
 * cx_off / cf_off: first column of the x member / of the flow pair in a
 * weight row; 3x3 taps; the flow is the member convolved at the x member's
 * resolution (both up-sampled together).  ws: dvsof_flow_fold_workspace_bytes.
 * ------------------------------------------------------------------ */
int dvsof_flow_fold_weights(const float *w, int Cout, int Ctot, int cx_off,
                            int Cx, int cf_off, const float *wh, float *w_eff,
                            void *stream);
/* The same fold for the FORWARD: y = conv(up cat[x, skip], w_eff) + bias_eff
 * + bias_cls[class of the output pixel], where the head's bias bh reaches the
 * output through the flow taps that lie inside the frame:
 *   bias_eff[co]    = bias[co] + sum_{t,f} Wflow[co][t][f] bh[f]
 *   bias_cls[c][co] = - sum_{t outside for class c, f} Wflow[co][t][f] bh[f]
 * (3x3 taps, pad 1; rows of the first / last output line lose ky = 0 / 2,
 * columns kx = 0 / 2).  bias_cls is [9][Cout], row 0 zero. */
int dvsof_flow_fold_bias(const float *w, int Cout, int Ctot, int cf_off,
                         const float *bh, const float *bias, float *bias_eff,
                         float *bias_cls, void *stream);
size_t dvsof_flow_fold_workspace_bytes(int B, int Cout);
int dvsof_flow_fold_grads(float *dW, const float *w, int Cout, int Ctot,
                          int cx_off, int Cx, int cf_off, const float *wh,
                          const float *bh, const float *db_conv,
                          const float *g, int B, int H, int W, float *dwh,
                          float *dbh, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ *
 * Data-parallel gradient exchange (RCCL over xGMI).  The reference is single
 * process (no torch.distributed / NCCL use anywhere); this
 * is the one collective of the build: the average of a gradient bucket over
 * the ranks, in place, enqueued on `stream` (the caller's exchange stream,
 * gated by an event behind the bucket's last weight gradient and joined
 * before the optimizer step: parallel.GradReducer).  One process per GPU;
 * librccl is opened on first use (DVSOF_ECOMM when it is absent).
 *   rank 0:   dvsof_comm_unique_id(id)  -> send the 128 bytes to every rank
 *   all:      dvsof_comm_create(&comm, world, rank, id)   (current HIP device)
 *   per step: dvsof_allreduce_bucket(comm, bucket, n, stream) per bucket
 * ------------------------------------------------------------------ */
#define DVSOF_COMM_ID_BYTES 128
int dvsof_comm_unique_id(void *host_id128);
int dvsof_comm_create(void **comm, int world_size, int rank,
                      const void *host_id128);
int dvsof_comm_destroy(void *comm);
int dvsof_allreduce_bucket(void *comm, float *bucket, size_t n, void *stream);
/* dvsof_comm_info: ranks = what the communicator spans (ncclCommCount for an
 * RCCL communicator -- bench.py prints it as config.rccl_ranks), loopback = 1
 * for a loopback communicator, calls / elements = dvsof_allreduce_bucket calls
 * issued on it so far and the floats they carried (any pointer may be NULL).
 *
 * dvsof_comm_create_loopback: a communicator WITHOUT peers, for the ordering
 * tests of the exchange on one GPU: the other world_size - 1 ranks are
 * imaginary and contribute all-zero buckets, the wire is a spin of delay_us on
 * `stream`.  dvsof_allreduce_bucket then leaves bucket / world_size in place,
 * delay_us late -- an exchange that is neither the identity nor instantaneous
 * (a 1-rank RCCL group launches nothing at all), so a kernel that reads or
 * rewrites a bucket on the wrong side of its collective changes the result.
 * No RCCL involved; never used by a training run. */
int dvsof_comm_info(void *comm, int *ranks, int *loopback,
                    unsigned long long *calls, unsigned long long *elements);
int dvsof_comm_create_loopback(void **comm, int world_size, int delay_us);

/* ------------------------------------------------------------------ *
 * Step executor (csrc/exec.hip): the loop body of the reference's train()
 * (utils/training.py:138-167: model, loss, backward, optimizer.step) captured
 * once per batch signature as a hipGraph and replayed as plain kernel
 * launches from ONE C call, on the streams of the eager schedule.
 *   graph         hipGraph_t of a stream capture; kernel (and empty) nodes
 *                 only, else DVSOF_EINVAL; must outlive the executor
 *   side_streams  hipStream_t of lanes 1..n_side (lane 0 = the stream given
 *                 to dvsof_exec_launch); nodes are split into at most
 *                 1 + n_side chains, one event per dependency across chains
 *   dvsof_exec_calibrate  runs the step ONCE on `stream` alone with a timing
 *                       event per kernel, waits for it and re-plans the lanes
 *                       from the measured durations.  Three plans: "paths"
 *                       (lane 0 = longest path by time, the rest on the last
 *                       lane), "list" (list scheduling by remaining path
 *                       length, which also fixes the launch order; marks keep
 *                       their capture order) and "chain" (the greedy split in
 *                       effect before calibration).  The next 6 calls of
 *                       dvsof_exec_launch are real steps under each plan in
 *                       turn, timed on the device (the host waits for the
 *                       previous step before each of them); the fastest plan
 *                       stays.  DVSOF_EXEC_PLAN=paths|list|chain fixes one.
 *                       It IS a step: same kernels, same results as
 *                       dvsof_exec_launch -- under every plan
 *   dvsof_exec_plan     name of the plan in effect; settled = 0 while plans
 *                       are still being tried
 *   dvsof_exec_launch   side lanes start behind `stream`, `stream` continues
 *                       behind every lane; nothing else synchronises
 *   dvsof_exec_node     node i in launch order: lane, measured us, number of
 *                       cross-lane waits, kernel name (diagnostics)
 *   dvsof_exec_info     counts (any pointer may be NULL); lane_kernels[l] =
 *                       kernels of lane l for l < max_lanes
 *
 * Data parallelism and gradient accumulation (utils/training.py:156-167,
 * utils/options.py:318-325) under the executor.  The exchange is not a kernel:
 * while the step is captured the reducer leaves MARKS in the stream
 * (dvsof_exec_mark) where the eager loop issues a collective / joins the
 * exchange stream.  The executor never launches a mark; it acts on it:
 *   DVSOF_MARK_BUCKET  (bucket, n): the exchange stream waits for the lane's
 *                      progress, then dvsof_allreduce_bucket(comm, bucket, n)
 *                      on the exchange stream
 *                      Kernels captured behind a BUCKET mark do NOT depend on
 *                      it: they inherit what preceded the mark
 *   DVSOF_MARK_WAIT    (index of a BUCKET mark): the lane waits for THAT
 *                      collective (in front of the update of that bucket's
 *                      parameters when the optimizer runs inside the backward)
 *   DVSOF_MARK_JOIN    the lane waits for the exchange stream (in front of
 *                      the optimizer kernels)
 *   dvsof_exec_set_comm  communicator (dvsof_comm_create) and exchange stream
 *                      of the marks; comm NULL (or never set): marks are
 *                      skipped -- one GPU.  Call before the first replay.
 *   dvsof_exec_marks   number of marks found in the graph
 *   dvsof_exec_node_arg  copies the first nbytes of kernel argument `arg` of
 *                      node i (launch order) -- diagnostics: the pointer audit
 *                      of capture.CapturedTrainStep.audit.  UNCHECKED: a node
 *                      does not know its argument count or sizes (the HIP
 *                      graph API does not expose them); the caller takes both
 *                      from the code-object metadata of the SAME library file
 *                      that is loaded (_audit.py verifies that) -- a wrong
 *                      arg / nbytes is an out-of-bounds host read
 * Every rank replays the same graph, so the collectives are issued in the
 * same order everywhere.
 * ------------------------------------------------------------------ */
#define DVSOF_MARK_BUCKET 1
#define DVSOF_MARK_JOIN 2
#define DVSOF_MARK_WAIT 3
int dvsof_exec_mark(int kind, int index, float *bucket, size_t n, void *stream);
int dvsof_exec_set_comm(void *exec, void *comm, void *exchange_stream);
/* Stream of the UPDATE lane: kernels captured behind nothing but WAIT marks (a
 * bucket's optimizer update, captured on the exchange stream by
 * parallel.GradReducer) are replayed there, each behind its bucket's
 * collective, and no compute lane waits before the JOIN mark.  Without this
 * call they run on the exchange stream itself, between the collectives.  A
 * stream of its own hardware queue (first used before any communicator
 * exists: parallel.claim_streams). */
int dvsof_exec_set_update_stream(void *exec, void *update_stream);
int dvsof_exec_marks(void *exec, int *n_marks);
/* The exchange window of the k-th BUCKET mark (capture order): the kernels
 * captured behind the mark that are not behind its WAIT mark (or, without one,
 * the JOIN mark).  They inherit the mark's dependencies and may run beside
 * the collective, so none of them may touch [bucket, bucket + n).  nodes[] =
 * their positions in the launch order (at most cap; *count = how many there
 * are).  The caller checks the arguments (capture.py at recording time, with
 * the argument layouts of the code objects); DVSOF_EINVAL past the last mark. */
int dvsof_exec_mark_window(void *exec, int k, float **bucket, size_t *n,
                           int *index, int *nodes, int cap, int *count);
int dvsof_exec_node_arg(void *exec, int i, int arg, size_t nbytes, void *out);
int dvsof_exec_create(void *graph, void *const *side_streams, int n_side,
                      void **exec);
int dvsof_exec_info(void *exec, int *n_kernels, int *n_lanes, int *n_events,
                    int *n_waits, int *lane_kernels, int max_lanes);
int dvsof_exec_calibrate(void *exec, void *stream);
int dvsof_exec_launch(void *exec, void *stream);
int dvsof_exec_node(void *exec, int i, int *lane, float *us, int *n_waits,
                    char *name, int name_len);
int dvsof_exec_plan(void *exec, char *name, int name_len, int *settled);
int dvsof_exec_destroy(void *exec);

#ifdef __cplusplus
}
#endif
#endif /* DVSOF_H */
