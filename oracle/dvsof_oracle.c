/*
 * dvsof_oracle.c -- CPU restatement of the reference's optical-flow training
 * hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product package may
 * import, link or execute this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do (as the checker / the timed CPU port).
 *
 * Plain scalar C, one function per reference symbol, each citing the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * Pinned against: tests/loss/test_loss.py golden triples, the 10-fixture table
 *  and fresh outputs/gradients of the reference's own
 * utils.loss captured by tools/make_goldens.py (tests/golden/loss_reference.npz).
 * The voxeliser follows the build's VOXEL_SPEC (upstream source is an
 * un-vendored, un-pinned submodule: "parity unpinned");
 * orc_count_image follows utils/data.py:120-136.
 *
 * Sums are accumulated in double: the oracle is the "true value" the fp32
 * paths (reference and HIP) are compared against within a stated tolerance.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS 1e-3   /* utils/loss.py:26 */
#define ORC_ALPHA 0.45 /* utils/loss.py:25 */

/* charbonier_loss element, utils/loss.py:31: (d^2 + eps^2)^alpha */
static double rho(double d) { return pow(d * d + ORC_EPS * ORC_EPS, ORC_ALPHA); }
/* d rho / d d */
static double drho(double d)
{
    return 2.0 * ORC_ALPHA * d * pow(d * d + ORC_EPS * ORC_EPS, ORC_ALPHA - 1.0);
}

/* F.interpolate(mode='bilinear', align_corners=True), utils/loss.py:20-21.
 * src index = dst * (in-1)/(out-1) in float, taps clamped to the last row/col. */
/* OpenMP: the pragmas below are active only in libdvsof_oracle_omp.so (make omp:
 * -fopenmp), the flavour bench.py's cpu_baseline leg times on all host cores.  They split
 * work over IMAGES / SAMPLES whose outputs are disjoint, so results do not depend on
 * the thread count; the forward sums are added per sample and then in sample order.
 * The default library (the checker of tests/ and smoke()) is compiled without -fopenmp:
 * one thread, the code below as written. */
void orc_resize_bilinear_ac(const float *src, float *dst, int n, int hin,
                            int win, int hout, int wout)
{
    const float sh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
    const float sw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const float *s = src + (size_t)i * hin * win;
        float *d = dst + (size_t)i * hout * wout;
        for (int y = 0; y < hout; ++y) {
            const float fy = sh * (float)y;
            int y0 = (int)fy;
            if (y0 > hin - 1) y0 = hin - 1;
            const int y1 = y0 + (y0 < hin - 1 ? 1 : 0);
            const float ly = fy - (float)y0, hy = 1.f - ly;
            for (int x = 0; x < wout; ++x) {
                const float fx = sw * (float)x;
                int x0 = (int)fx;
                if (x0 > win - 1) x0 = win - 1;
                const int x1 = x0 + (x0 < win - 1 ? 1 : 0);
                const float lx = fx - (float)x0, hx = 1.f - lx;
                d[(size_t)y * wout + x] =
                    hy * (hx * s[(size_t)y0 * win + x0] + lx * s[(size_t)y0 * win + x1]) +
                    ly * (hx * s[(size_t)y1 * win + x0] + lx * s[(size_t)y1 * win + x1]);
            }
        }
    }
}

/* The normalised sampling grid of utils/loss.py:150-156, fp32 op for op:
 * g = (x + u); g /= (w-1)/2.; g -= 1.   (the divisor is a Python double that
 * ATen casts to float before an IEEE division). */
static void warp_grid(float x, float y, float u, float v, int h, int w,
                      float *gx, float *gy)
{
    const float dw = (float)((w - 1) / 2.0), dh = (float)((h - 1) / 2.0);
    *gx = (x + u) / dw - 1.f;
    *gy = (y + v) / dh - 1.f;
}

/* grid_sample(bilinear, zeros, align_corners=True), utils/loss.py:8-12,70:
 * unnormalise (g+1)*((size-1)/2); taps outside the frame contribute 0 to the
 * value AND to the coordinate gradient.  Returns the sample, fills d/dix,d/diy. */
static double sample_zeros(const float *img, int h, int w, float gx, float gy,
                           double *dix, double *diy)
{
    /* ATen's CPU kernel unnormalises as (g + 1) * ((size-1)/2) in float */
    const float ix = (gx + 1.f) * ((float)(w - 1) / 2.f);
    const float iy = (gy + 1.f) * ((float)(h - 1) / 2.f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    /* keep the int casts safe for wild flows */
    const double bx = fx0 < -4.f ? -4. : (fx0 > (float)w + 4.f ? (double)w + 4. : fx0);
    const double by = fy0 < -4.f ? -4. : (fy0 > (float)h + 4.f ? (double)h + 4. : fy0);
    const int x0 = (int)bx, y0 = (int)by, x1 = x0 + 1, y1 = y0 + 1;
    /* NaN or far-away coordinates: every tap is outside the frame */
    const int inside = (fx0 == (float)x0) && (fy0 == (float)y0);
    const double ax = (double)ix - fx0, ay = (double)iy - fy0; /* ix - ix_nw */
    const double cx = 1.0 - ax, cy = 1.0 - ay;                  /* ix_se - ix */
    double nw = 0, ne = 0, sw = 0, se = 0;
    if (inside) {
        if (y0 >= 0 && y0 < h && x0 >= 0 && x0 < w) nw = img[(size_t)y0 * w + x0];
        if (y0 >= 0 && y0 < h && x1 >= 0 && x1 < w) ne = img[(size_t)y0 * w + x1];
        if (y1 >= 0 && y1 < h && x0 >= 0 && x0 < w) sw = img[(size_t)y1 * w + x0];
        if (y1 >= 0 && y1 < h && x1 >= 0 && x1 < w) se = img[(size_t)y1 * w + x1];
    }
    *dix = -nw * cy + ne * cy - sw * ay + se * ay;
    *diy = -nw * cx - ne * ax + sw * cx + se * ax;
    return nw * cx * cy + ne * ax * cy + sw * cx * ay + se * ax * ay;
}

static int is_oob(float gx, float gy)
{ /* utils/loss.py:92-94, strict comparisons */
    return (gx < -1.f) || (gx > 1.f) || (gy < -1.f) || (gy > 1.f);
}

/*
 * One scale of Loss.__call__ (utils/loss.py:121-171).
 *   prev,next [N,1,h,w]   flow [N,2,h,w] (ch0 = u along x, ch1 = v along y)
 *   terms[3] = smoothness, photometric, outborder (the order returned at :171)
 *   oob_count[N] (may be NULL) = per-sample out-of-border pixel count (:101 / 2)
 */
void orc_loss_scale_fwd(const float *prev, const float *next, const float *flow,
                        int N, int h, int w, double *terms, int64_t *oob_count)
{
    const size_t hw = (size_t)h * w;
    double photo = 0, border = 0;
    double sm[4] = {0, 0, 0, 0};
#ifdef _OPENMP
    double (*part)[6] = (double (*)[6])calloc((size_t)N, sizeof(*part));
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int n = 0; n < N; ++n) {
        const float *U = flow + (size_t)n * 2 * hw, *V = U + hw;
        const float *I0 = prev + (size_t)n * hw, *I1 = next + (size_t)n * hw;
        int64_t cnt = 0;
        double bsum = 0;
#ifdef _OPENMP      /* this sample's sums; added in sample order below */
        double photo = 0, sm[4] = {0, 0, 0, 0};
#endif
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t p = (size_t)y * w + x;
                float gx, gy;
                double dx_, dy_;
                warp_grid((float)x, (float)y, U[p], V[p], h, w, &gx, &gy);
                /* photometric_loss, utils/loss.py:72-74 */
                photo += rho(sample_zeros(I1, h, w, gx, gy, &dx_, &dy_) - (double)I0[p]);
                /* outborder_regularization_loss, utils/loss.py:96-119 */
                if (is_oob(gx, gy)) {
                    ++cnt;
                    bsum += rho(U[p]) + rho(V[p]);
                }
                /* smoothness_loss, utils/loss.py:76-90 (4 directions, 2 channels) */
                for (int c = 0; c < 2; ++c) {
                    const float *F = c ? V : U;
                    if (x + 1 < w) sm[0] += rho((double)F[p + 1] - F[p]);
                    if (y + 1 < h) sm[1] += rho((double)F[p + w] - F[p]);
                    if (x + 1 < w && y + 1 < h) {
                        sm[2] += rho((double)F[p + w + 1] - F[p]);
                        sm[3] += rho((double)F[p + 1] - F[p + w]);
                    }
                }
            }
        if (oob_count) oob_count[n] = cnt;
#ifdef _OPENMP
        part[n][0] = photo;
        for (int i = 0; i < 4; ++i) part[n][1 + i] = sm[i];
        part[n][5] = cnt ? bsum / (2.0 * (double)cnt * N) : 0.0;
#else
        if (cnt) border += bsum / (2.0 * (double)cnt * N); /* :101,:113 */
#endif
    }
#ifdef _OPENMP
    for (int n = 0; n < N; ++n) {
        photo += part[n][0];
        for (int i = 0; i < 4; ++i) sm[i] += part[n][1 + i];
        border += part[n][5];
    }
    free(part);
#endif
    const double c0 = (double)N * 2 * h * (w - 1), c1 = (double)N * 2 * (h - 1) * w,
                 c2 = (double)N * 2 * (h - 1) * (w - 1);
    /* empty crops give 0 (utils/loss.py:29-30) */
    terms[0] = ((c0 > 0 ? sm[0] / c0 : 0) + (c1 > 0 ? sm[1] / c1 : 0) +
                (c2 > 0 ? sm[2] / c2 : 0) + (c2 > 0 ? sm[3] / c2 : 0)) / 4.0;
    terms[1] = photo / ((double)N * hw);
    terms[2] = border;
}

/*
 * d(g[0]*smooth + g[1]*photo + g[2]*border)/d flow for one scale: what
 * autograd produces through utils/loss.py:121-171 (gradient flows into flow
 * only; the out-of-border mask is no_grad, :98).  grad_flow is OVERWRITTEN.
 */
void orc_loss_scale_bwd(const float *prev, const float *next, const float *flow,
                        int N, int h, int w, const double *g, float *grad_flow)
{
    const size_t hw = (size_t)h * w;
    const double c0 = (double)N * 2 * h * (w - 1), c1 = (double)N * 2 * (h - 1) * w,
                 c2 = (double)N * 2 * (h - 1) * (w - 1);
    double *acc = (double *)calloc((size_t)N * 2 * hw, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        const float *U = flow + (size_t)n * 2 * hw, *V = U + hw;
        const float *I0 = prev + (size_t)n * hw, *I1 = next + (size_t)n * hw;
        double *GU = acc + (size_t)n * 2 * hw, *GV = GU + hw;
        int64_t cnt = 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float gx, gy;
                warp_grid((float)x, (float)y, U[(size_t)y * w + x], V[(size_t)y * w + x],
                          h, w, &gx, &gy);
                cnt += is_oob(gx, gy);
            }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t p = (size_t)y * w + x;
                float gx, gy;
                double dix, diy;
                warp_grid((float)x, (float)y, U[p], V[p], h, w, &gx, &gy);
                const double d = sample_zeros(I1, h, w, gx, gy, &dix, &diy) - (double)I0[p];
                const double s = g[1] * drho(d) / ((double)N * hw);
                /* grid_sampler backward scales by (size-1)/2, the grid
                 * normalisation divides by the same: pixel units remain */
                GU[p] += s * dix;
                GV[p] += s * diy;
                if (is_oob(gx, gy)) {
                    const double k = g[2] / (2.0 * (double)cnt * N);
                    GU[p] += k * drho(U[p]);
                    GV[p] += k * drho(V[p]);
                }
                for (int c = 0; c < 2; ++c) {
                    const float *F = c ? V : U;
                    double *G = c ? GV : GU;
                    double t;
                    if (x + 1 < w) {
                        t = g[0] * drho((double)F[p + 1] - F[p]) / (4.0 * c0);
                        G[p + 1] += t; G[p] -= t;
                    }
                    if (y + 1 < h) {
                        t = g[0] * drho((double)F[p + w] - F[p]) / (4.0 * c1);
                        G[p + w] += t; G[p] -= t;
                    }
                    if (x + 1 < w && y + 1 < h) {
                        t = g[0] * drho((double)F[p + w + 1] - F[p]) / (4.0 * c2);
                        G[p + w + 1] += t; G[p] -= t;
                        t = g[0] * drho((double)F[p + 1] - F[p + w]) / (4.0 * c2);
                        G[p + 1] += t; G[p + w] -= t;
                    }
                }
            }
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)N * 2 * hw; ++i) grad_flow[i] = (float)acc[i];
    free(acc);
}

/* get_count_image, utils/data.py:120-136: idx = y*W + x, integer histogram. */
void orc_count_image(const int64_t *x, const int64_t *y, int64_t n, int H, int W,
                     uint64_t *out)
{
    memset(out, 0, sizeof(uint64_t) * (size_t)H * W);
    for (int64_t i = 0; i < n; ++i) out[(size_t)y[i] * W + x[i]] += 1;
}

/*
 * VOXEL_SPEC (docs/VOXEL_SPEC.md): polarity-signed event volume, bilinear in
 * time, nearest in space.  For event i of sample b with window [t0_b, t1_b]:
 *   dt = t1-t0;  tn = dt > 0 ? ((t - t0) / dt) * (C-1) : 0          (all fp32)
 *   dropped if t < t0, t > t1, or x,y outside the frame
 *   c0 = (int)floorf(tn), f = tn - c0
 *   V[b,c0,y,x] += s*(1-f);   if (c0+1 < C) V[b,c0+1,y,x] += s*f,  s = sign(p)
 * bin0[i] receives c0 (or -1 when dropped) and lin0[i] the linear index of
 * (b,c0,y,x) -- the integer part that must be bit-exact on the GPU.
 * Callers: utils/training.py:59-64, scripts/quantize_preprocessed.py:87-91;
 * output contract utils/dataset.py:436-448 (B x C x H x W float32).
 */
void orc_voxelize(const int64_t *x, const int64_t *y, const float *t,
                  const int64_t *p, const int64_t *sample, int64_t n,
                  const float *t0, const float *t1, int B, int C, int H, int W,
                  float *out, int32_t *bin0, int64_t *lin0)
{
    const size_t total = (size_t)B * C * H * W;
    double *acc = (double *)calloc(total, sizeof(double));
#ifdef _OPENMP  /* thread k owns the samples b with b % T == k (and the events of no sample) */
#pragma omp parallel
#endif
    for (int64_t i = 0; i < n; ++i) {
        const int64_t b = sample[i];
#ifdef _OPENMP
        {
            const int T = omp_get_num_threads(), k = omp_get_thread_num();
            const int owner = (b < 0 || b >= B) ? 0 : (int)(b % T);
            if (owner != k) continue;
        }
#endif
        if (bin0) bin0[i] = -1;
        if (lin0) lin0[i] = -1;
        if (b < 0 || b >= B || x[i] < 0 || x[i] >= W || y[i] < 0 || y[i] >= H) continue;
        const float ts = t[i];
        if (ts < t0[b] || ts > t1[b]) continue;
        const float dt = t1[b] - t0[b];
        const float tn = dt > 0.f ? ((ts - t0[b]) / dt) * (float)(C - 1) : 0.f;
        int c0 = (int)floorf(tn);
        if (c0 > C - 1) c0 = C - 1;
        const float f = tn - (float)c0;
        const size_t base = (((size_t)b * C + c0) * H + (size_t)y[i]) * W + (size_t)x[i];
        const float pol = p[i] > 0 ? 1.f : (p[i] < 0 ? -1.f : 0.f); /* a sign */
        acc[base] += (double)(pol * (1.f - f));
        if (c0 + 1 < C) acc[base + (size_t)H * W] += (double)(pol * f);
        if (bin0) bin0[i] = c0;
        if (lin0) lin0[i] = (int64_t)base;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; ++i) out[i] = (float)acc[i];
    free(acc);
}
