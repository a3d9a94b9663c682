/*
 * TEST INFRASTRUCTURE — CPU oracle for the jpeg2png deblocking hot path.
 *
 * This file is a plain-C RESTATEMENT of the reference algorithm, written in the
 * same "gather + carried prob-state" form the HIP kernels use, so that every
 * intermediate of the device path has a CPU twin.  It is only ever used as a
 * checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the
 * product library never links or calls it.
 *
 * Parity status: PINNED.  The reference has no golden vectors,
 * so the pin is the compiled reference itself: tests/test_oracle.py
 * checks that this restatement is BIT-IDENTICAL to oracle/_ref/libj2p_ref.so
 * (the untouched reference sources built by oracle/Makefile) for 1- and
 * 3-channel, subsampled and padded inputs, and tests/golden/ holds outputs of
 * that reference build so the pin also holds where /root/reference is absent.
 *
 * Reference lines each piece follows (all in /root/reference):
 *   fdct8 / idct8 ............ ooura/dct.c:98-159 / :34-95 (double-promoted
 *                              constant products, float rounding per assignment)
 *   state init ............... compute.c:278-310 (aux_init)
 *   FISTA point .............. compute.c:430-440
 *   prob term ................ compute.c:38-70, compute_simd_step.c:7-62
 *   TV term .................. compute.c:73-125, compute_simd_step.c:64-153
 *   TGV2 term ................ compute.c:128-197, compute_simd_step.c:172-300
 *   norm + step .............. compute.c:200-216
 *   projection ............... compute.c:334-404, clamp :323-331
 *   driver scalars ........... compute.c:223-275, :407-465
 *
 * How the form differs from the reference while producing the same bits:
 *  - the reference SCATTERS each pixel's TV/TGV terms into up to 7 neighbours
 *    while scanning in raster order; here every target pixel GATHERS its terms
 *    from its neighbours in exactly the order the raster scan would have
 *    delivered them (SURVEY.md §8a "gather form"), so the float sums agree bit
 *    for bit;
 *  - the reference keeps the clamped DCT coefficients ("cos", compute.c:381)
 *    and turns them into the prob gradient at the start of the NEXT iteration
 *    (compute.c:47-51); here the projection emits that gradient block right
 *    away (same operations on the same values) and the next gradient pass just
 *    reads it.
 *
 * Build: gcc -std=c11 -O2 -msse2 -mfpmath=sse -ffp-contract=off (no fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#include "solver_oracle.h"

_Static_assert(FLT_EVAL_METHOD == 0, "oracle needs strict float evaluation");

/* ---- 8-point orthonormal DCT-II / DCT-III with the reference's rounding ---- */

/* sqrt(2/8)*cos/sin(pi k/16) and cos(pi/4); products with them are evaluated in
 * double and rounded to float at the assignment, like ooura/dct.c:24-31 */
static const double K1c = 0.49039264020161522456, K1s = 0.09754516100806413392;
static const double K2c = 0.46193976625564337806, K2s = 0.19134171618254488586;
static const double K3c = 0.41573480615127261854, K3s = 0.27778511650980111237;
static const double K4  = 0.35355339059327376220, KH  = 0.70710678118654752440;

static inline float mix_add(double ka, float a, double kb, float b) { return (float)(ka * (double)a + kb * (double)b); }
static inline float mix_sub(double ka, float a, double kb, float b) { return (float)(ka * (double)a - kb * (double)b); }
static inline float scale(double k, float a) { return (float)(k * (double)a); }

/* forward 1-D transform of v[0],v[s],...,v[7s] in place (one pass of dct8x8s) */
static void fdct8(float *v, int s)
{
        float e0 = v[0] + v[7*s], o0 = v[0] - v[7*s];
        float e1 = v[2*s] + v[5*s], o1 = v[2*s] - v[5*s];
        float e2 = v[4*s] + v[3*s], o2 = v[4*s] - v[3*s];
        float e3 = v[6*s] + v[1*s], o3 = v[6*s] - v[1*s];
        float p = e0 + e2, q = e1 + e3;
        v[0]   = scale(K4, p + q);
        v[4*s] = scale(K4, p - q);
        p = e0 - e2; q = e1 - e3;
        v[2*s] = mix_sub(K2c, p, K2s, q);
        v[6*s] = mix_add(K2c, q, K2s, p);
        float r  = scale(KH, o1 - o3);
        float t  = scale(KH, o1 + o3);
        float u3 = t - o2;
        float u1 = t + o2;
        float w3 = o0 - r;
        float w1 = o0 + r;
        v[1*s] = mix_sub(K1c, w1, K1s, u1);
        v[7*s] = mix_add(K1c, u1, K1s, w1);
        v[3*s] = mix_sub(K3c, w3, K3s, u3);
        v[5*s] = mix_add(K3c, u3, K3s, w3);
}

/* inverse 1-D transform (one pass of idct8x8s) */
static void idct8(float *v, int s)
{
        float a1 = mix_add(K1c, v[1*s], K1s, v[7*s]);
        float b1 = mix_sub(K1c, v[7*s], K1s, v[1*s]);
        float a3 = mix_add(K3c, v[3*s], K3s, v[5*s]);
        float b3 = mix_sub(K3c, v[5*s], K3s, v[3*s]);
        float dr = a1 - a3;
        float di = b1 + b3;
        a1 = a1 + a3;
        b3 = b3 - b1;
        b1 = scale(KH, dr + di);
        a3 = scale(KH, dr - di);
        float cr = mix_add(K2c, v[2*s], K2s, v[6*s]);
        float ci = mix_sub(K2c, v[6*s], K2s, v[2*s]);
        float s0 = scale(K4, v[0] + v[4*s]);
        float d0 = scale(K4, v[0] - v[4*s]);
        float m2r = s0 - cr;
        float m2i = d0 - ci;
        s0 = s0 + cr;
        d0 = d0 + ci;
        v[0]   = s0 + a1;
        v[7*s] = s0 - a1;
        v[2*s] = d0 + b1;
        v[5*s] = d0 - b1;
        v[4*s] = m2r - b3;
        v[3*s] = m2r + b3;
        v[6*s] = m2i - a3;
        v[1*s] = m2i + a3;
}

/* 8x8 block, natural order b[v*8+u]: pass over the first index for every column,
 * then over the second index for every row (ooura/dct.c:39,67 / :103,131) */
void oracle_fdct8x8(float b[64])
{
        for(int j = 0; j < 8; j++) { fdct8(b + j, 8); }
        for(int j = 0; j < 8; j++) { fdct8(b + 8*j, 1); }
}

void oracle_idct8x8(float b[64])
{
        for(int j = 0; j < 8; j++) { idct8(b + j, 8); }
        for(int j = 0; j < 8; j++) { idct8(b + 8*j, 1); }
}

/* decode_coefficients (jpeg.c:83-92) followed by unbox (box.c:5-19, as called at
 * jpeg2png.c:131-139): block-major int16 coefficients -> raster float plane */
void oracle_decode_plane(unsigned w, unsigned h, const int16_t *coef, const uint16_t *quant, float *out)
{
        const unsigned bw = w / 8, bh = h / 8;
        for(unsigned by = 0; by < bh; by++) {
                for(unsigned bx = 0; bx < bw; bx++) {
                        float b[64];
                        const int16_t *d = coef + ((size_t)by * bw + bx) * 64;
                        for(int j = 0; j < 64; j++) { b[j] = (float)((int)d[j] * (int)quant[j]); }
                        oracle_idct8x8(b);
                        for(int v = 0; v < 8; v++) {
                                for(int u = 0; u < 8; u++) {
                                        out[(size_t)(by*8+v) * w + bx*8+u] = b[v*8+u];
                                }
                        }
                }
        }
}

/* ---- solver state ---- */

typedef struct {
        unsigned cw, ch, ws, hs;      /* coefficient plane geometry             */
        const int16_t *d;             /* block-major quantised coefficients     */
        float q[64];                  /* quant table as float                   */
        float *x, *xp;                /* x_k and x_{k-1} (after the FISTA swap: */
                                      /* x = extrapolated point y_k)            */
        float *g;                     /* objective gradient                     */
        float *pg;                    /* carried prob-gradient state, cw*ch     */
        float *blk;                   /* scratch: one coefficient plane, blocks */
        float *sub;                   /* scratch: subsampled raster plane       */
} chan;

static inline float fsq(float v) { return v * v; }

/* first differences of the FISTA point with the reference's border rule
 * (compute.c:79,81): zero on the last column / last row */
static inline float dxf(const float *f, unsigned W, unsigned x, unsigned y)
{
        return x >= W - 1 ? 0.f : f[(size_t)y * W + x + 1] - f[(size_t)y * W + x];
}
static inline float dyf(const float *f, unsigned W, unsigned H, unsigned x, unsigned y)
{
        return y >= H - 1 ? 0.f : f[(size_t)(y + 1) * W + x] - f[(size_t)y * W + x];
}

/* TV norm of source pixel (x,y) over all channels (compute.c:84-89) */
static float tv_norm(unsigned nch, const chan *cs, unsigned W, unsigned H, unsigned x, unsigned y)
{
        float n = 0.f;
        for(unsigned c = 0; c < nch; c++) {
                n += fsq(dxf(cs[c].x, W, x, y));
                n += fsq(dyf(cs[c].x, W, H, x, y));
        }
        return sqrtf(n);
}

/* second differences at source pixel (x,y) of one channel (compute.c:136-146):
 * backward differences of the forward differences, zero on first column/row */
typedef struct { float xx, sym, yy; } hess;

static hess hessian(const float *f, unsigned W, unsigned H, unsigned x, unsigned y)
{
        float gx = dxf(f, W, x, y), gy = dyf(f, W, H, x, y);
        hess r;
        r.xx      = x == 0 ? 0.f : gx - dxf(f, W, x - 1, y);
        float gyx = x == 0 ? 0.f : gy - dyf(f, W, H, x - 1, y);
        float gxy = y == 0 ? 0.f : gx - dxf(f, W, x, y - 1);
        r.yy      = y == 0 ? 0.f : gy - dyf(f, W, H, x, y - 1);
        r.sym = (float)((double)(gxy + gyx) / 2.);   /* (g_xy + g_yx) / 2. */
        return r;
}

static float tgv_norm(unsigned nch, const chan *cs, unsigned W, unsigned H, unsigned x, unsigned y)
{
        float n = 0.f;
        for(unsigned c = 0; c < nch; c++) {
                hess h = hessian(cs[c].x, W, H, x, y);
                n += fsq(h.xx) + 2 * fsq(h.sym) + fsq(h.yy);
        }
        return sqrtf(n);
}

/* ---- one iteration pieces ---- */

/* gradient of all three terms for every pixel, gather form; returns the log sums */
static void gradient_pass(unsigned nch, chan *cs, unsigned W, unsigned H,
                          float weight, const float *pweight,
                          double *tv_sum, double *tv2_sum)
{
        const float a_tv = (float)(1. / (double)sqrtf((float)nch));     /* compute.c:90 */
        const float alpha = weight / sqrtf((float)(4 / 2));             /* compute.c:258 */
        const float a_tgv = (float)((double)alpha * 1. / (double)sqrtf((float)nch)); /* :154 */
        const int tgv_on = weight != 0.f;

        /* per-source norms are shared by all channels: tabulate once */
        float *n1 = malloc(sizeof(float) * (size_t)W * H);
        float *n2 = tgv_on ? malloc(sizeof(float) * (size_t)W * H) : NULL;
        double tv = 0., tv2 = 0.;
        for(unsigned y = 0; y < H; y++) {
                for(unsigned x = 0; x < W; x++) {
                        float n = tv_norm(nch, cs, W, H, x, y);
                        n1[(size_t)y * W + x] = n;
                        tv += a_tv * n;
                }
        }
        if(tgv_on) {
                for(unsigned y = 0; y < H; y++) {
                        for(unsigned x = 0; x < W; x++) {
                                float n = tgv_norm(nch, cs, W, H, x, y);
                                n2[(size_t)y * W + x] = n;
                                tv2 += a_tgv * n;
                        }
                }
        }
        *tv_sum = tv;
        *tv2_sum = tv2;

        for(unsigned c = 0; c < nch; c++) {
                chan *k = &cs[c];
                const float *f = k->x;
                const float p_alpha = pweight[c] * 2 * 255 * sqrtf(2);   /* compute.c:245 */
                for(unsigned y = 0; y < H; y++) {
                        for(unsigned x = 0; x < W; x++) {
                                float g = 0.f;                           /* compute.c:239-241 */
                                /* prob term, replicated over the sample's footprint (compute.c:53-66) */
                                if(pweight[c] != 0.f && x < k->cw * k->ws && y < k->ch * k->hs) {
                                        g += p_alpha * k->pg[(size_t)(y / k->hs) * k->cw + x / k->ws];
                                }
                                /* TV: sources (x,y-1) -> its "below", (x-1,y) -> its "right", own */
                                float n;
                                if(y > 0 && (n = n1[(size_t)(y-1) * W + x]) != 0.f) {
                                        g += a_tv * dyf(f, W, H, x, y-1) / n;
                                }
                                if(x > 0 && (n = n1[(size_t)y * W + x-1]) != 0.f) {
                                        g += a_tv * dxf(f, W, x-1, y) / n;
                                }
                                if((n = n1[(size_t)y * W + x]) != 0.f) {
                                        g += a_tv * -(dxf(f, W, x, y) + dyf(f, W, H, x, y)) / n;
                                }
                                if(tgv_on) {
                                        hess h;
                                        /* raster order of the sources that write into (x,y):
                                         * (x,y-1) down, (x+1,y-1) down-left, (x-1,y) right, own,
                                         * (x+1,y) left, (x-1,y+1) up-right, (x,y+1) up */
                                        if(y > 0 && (n = n2[(size_t)(y-1) * W + x]) != 0.f) {
                                                h = hessian(f, W, H, x, y-1);
                                                g += a_tgv * ((h.yy + h.sym) / n);
                                        }
                                        if(y > 0 && x < W-1 && (n = n2[(size_t)(y-1) * W + x+1]) != 0.f) {
                                                h = hessian(f, W, H, x+1, y-1);
                                                g += a_tgv * ((-h.sym) / n);
                                        }
                                        if(x > 0 && (n = n2[(size_t)y * W + x-1]) != 0.f) {
                                                h = hessian(f, W, H, x-1, y);
                                                g += a_tgv * ((h.sym + h.xx) / n);
                                        }
                                        if((n = n2[(size_t)y * W + x]) != 0.f) {
                                                h = hessian(f, W, H, x, y);
                                                g += a_tgv * (-(2 * h.xx + 2 * h.sym + 2 * h.yy) / n);
                                        }
                                        if(x < W-1 && (n = n2[(size_t)y * W + x+1]) != 0.f) {
                                                h = hessian(f, W, H, x+1, y);
                                                g += a_tgv * ((h.sym + h.xx) / n);
                                        }
                                        if(y < H-1 && x > 0 && (n = n2[(size_t)(y+1) * W + x-1]) != 0.f) {
                                                h = hessian(f, W, H, x-1, y+1);
                                                g += a_tgv * ((-h.sym) / n);
                                        }
                                        if(y < H-1 && (n = n2[(size_t)(y+1) * W + x]) != 0.f) {
                                                h = hessian(f, W, H, x, y+1);
                                                g += a_tgv * ((h.yy + h.sym) / n);
                                        }
                                }
                                k->g[(size_t)y * W + x] = g;
                        }
                }
        }
        free(n1);
        free(n2);
}

/* f -= step * g/||g||   (compute.c:200-216) */
static void descend(chan *k, size_t n, float step)
{
        double acc = 0.;
        for(size_t i = 0; i < n; i++) { acc += fsq(k->g[i]); }
        float norm = sqrtf((float)acc);
        if(norm != 0.f) {
                for(size_t i = 0; i < n; i++) {
                        k->x[i] = k->x[i] - step * (k->g[i] / norm);
                }
        }
}

/* projection onto the quantisation box + next prob state (compute.c:334-404, :41-51).
 * returns sum over coefficients of ((clamped - d*q)/q)^2 for the log */
static double project(chan *k, unsigned W, unsigned H, int want_prob)
{
        (void)H;
        const unsigned bw = k->cw / 8, bh = k->ch / 8;
        const int resample = !(k->cw == W && k->ch == H);
        float *plane = k->x;
        if(resample) {
                /* block mean down-sample, keep the residual in x (compute.c:348-370) */
                for(unsigned cy = 0; cy < k->ch; cy++) {
                        for(unsigned cx = 0; cx < k->cw; cx++) {
                                float mean = 0.f;
                                for(unsigned sy = 0; sy < k->hs; sy++) {
                                        for(unsigned sx = 0; sx < k->ws; sx++) {
                                                mean += k->x[(size_t)(cy * k->hs + sy) * W + cx * k->ws + sx];
                                        }
                                }
                                mean /= (float)(k->ws * k->hs);
                                k->sub[(size_t)cy * k->cw + cx] = mean;
                                for(unsigned sy = 0; sy < k->hs; sy++) {
                                        for(unsigned sx = 0; sx < k->ws; sx++) {
                                                k->x[(size_t)(cy * k->hs + sy) * W + cx * k->ws + sx] -= mean;
                                        }
                                }
                        }
                }
                plane = k->sub;
        }
        double dist = 0.;
        for(unsigned by = 0; by < bh; by++) {
                for(unsigned bx = 0; bx < bw; bx++) {
                        float b[64], e[64];
                        const int16_t *d = k->d + ((size_t)by * bw + bx) * 64;
                        for(int v = 0; v < 8; v++) {
                                for(int u = 0; u < 8; u++) {
                                        b[v*8+u] = plane[(size_t)(by*8+v) * k->cw + bx*8+u];
                                }
                        }
                        oracle_fdct8x8(b);
                        for(int j = 0; j < 64; j++) {
                                float lo = ((float)d[j] - 0.5f) * k->q[j];       /* compute.c:326-327 */
                                float hi = ((float)d[j] + 0.5f) * k->q[j];
                                float v = b[j];
                                v = v > hi ? hi : (v < lo ? lo : v);
                                b[j] = v;
                                if(want_prob) {
                                        float t = v - (float)d[j] * k->q[j];     /* compute.c:47 */
                                        dist += fsq(t / k->q[j]);                /* simd:22-26 */
                                        e[j] = t / fsq(k->q[j]);                 /* compute.c:49 */
                                }
                        }
                        oracle_idct8x8(b);
                        for(int v = 0; v < 8; v++) {
                                for(int u = 0; u < 8; u++) {
                                        plane[(size_t)(by*8+v) * k->cw + bx*8+u] = b[v*8+u];
                                }
                        }
                        if(want_prob) {
                                oracle_idct8x8(e);
                                for(int v = 0; v < 8; v++) {
                                        for(int u = 0; u < 8; u++) {
                                                k->pg[(size_t)(by*8+v) * k->cw + bx*8+u] = e[v*8+u];
                                        }
                                }
                        }
                }
        }
        if(resample) {
                /* add the new means back onto the residual (compute.c:390-403) */
                for(unsigned cy = 0; cy < k->ch; cy++) {
                        for(unsigned cx = 0; cx < k->cw; cx++) {
                                float mean = k->sub[(size_t)cy * k->cw + cx];
                                for(unsigned sy = 0; sy < k->hs; sy++) {
                                        for(unsigned sx = 0; sx < k->ws; sx++) {
                                                k->x[(size_t)(cy * k->hs + sy) * W + cx * k->ws + sx] += mean;
                                        }
                                }
                        }
                }
        }
        return dist;
}

void oracle_canvas_size(unsigned nch, const oracle_plane *pl, unsigned *W, unsigned *H)
{
        unsigned w = 0, h = 0;
        for(unsigned c = 0; c < nch; c++) {
                if(pl[c].w * pl[c].w_samp > w) { w = pl[c].w * pl[c].w_samp; }
                if(pl[c].h * pl[c].h_samp > h) { h = pl[c].h * pl[c].h_samp; }
        }
        *W = w;
        *H = h;
}

/* debugging aid for tools/trace_vs_oracle.py: when set, every iteration appends, per channel, the
 * gradient (W*H floats) and then the new iterate (W*H floats) to this buffer */
static float *g_trace;
void oracle_set_trace(float *buf) { g_trace = buf; }

int oracle_compute(unsigned nch, const oracle_plane *pl, float weight, const float *pweight,
                   unsigned iterations, float *const *out, double *log_rows)
{
        if(nch == 0 || nch > 3) { return -1; }
        unsigned W, H;
        oracle_canvas_size(nch, pl, &W, &H);
        const size_t n = (size_t)W * H;
        chan cs[3];
        memset(cs, 0, sizeof(cs));
        for(unsigned c = 0; c < nch; c++) {
                chan *k = &cs[c];
                k->cw = pl[c].w; k->ch = pl[c].h; k->ws = pl[c].w_samp; k->hs = pl[c].h_samp;
                k->d = pl[c].coef;
                for(int j = 0; j < 64; j++) { k->q[j] = (float)pl[c].quant[j]; }
                k->x = malloc(sizeof(float) * n);
                k->xp = malloc(sizeof(float) * n);
                k->g = malloc(sizeof(float) * n);
                k->pg = calloc((size_t)k->cw * k->ch, sizeof(float));  /* cos = d*q => zero prob gradient */
                k->sub = malloc(sizeof(float) * (size_t)k->cw * k->ch);
                if(!k->x || !k->xp || !k->g || !k->pg || !k->sub) { return -1; }
                /* replicate-upsample with edge clamp (compute.c:295-303) */
                for(unsigned y = 0; y < H; y++) {
                        for(unsigned x = 0; x < W; x++) {
                                unsigned cy = y / k->hs, cx = x / k->ws;
                                if(cy > k->ch - 1) { cy = k->ch - 1; }
                                if(cx > k->cw - 1) { cx = k->cw - 1; }
                                k->x[(size_t)y * W + x] = pl[c].pixels[(size_t)cy * k->cw + cx];
                        }
                }
                memcpy(k->xp, k->x, sizeof(float) * n);
        }

        const float radius = sqrtf((float)H * (float)W) / 2;             /* compute.c:425 */
        const float step = radius / sqrtf((float)(1 + iterations));      /* compute.c:443 */
        float t = 1;
        double carried_prob = 0.;   /* prob distance of the state entering this iteration */
        for(unsigned it = 0; it < iterations; it++) {
                float tnext = (1 + sqrtf(1 + 4 * fsq(t))) / 2;           /* compute.c:431-432 */
                float factor = (t - 1) / tnext;
                for(unsigned c = 0; c < nch; c++) {
                        chan *k = &cs[c];
                        for(size_t j = 0; j < n; j++) {
                                k->xp[j] = k->x[j] + factor * (k->x[j] - k->xp[j]);
                        }
                        float *sw = k->x; k->x = k->xp; k->xp = sw;
                }
                t = tnext;

                double tv, tv2;
                gradient_pass(nch, cs, W, H, weight, pweight, &tv, &tv2);

                /* log values exactly as the reference's default (SIMD) build prints them:
                 * prob_dist = sum_c 0.5*sum((cos-dq)/q)^2 WITHOUT alpha (compute_simd_step.c:61) */
                float total_alpha = 0.f;
                for(unsigned c = 0; c < nch; c++) {
                        if(pweight[c] != 0.f) { total_alpha += pweight[c] * 2 * 255 * sqrtf(2); }
                }
                total_alpha += nch;
                if(weight != 0.f) { total_alpha += (weight / sqrtf((float)(4 / 2))) * nch; }
                if(log_rows) {
                        log_rows[4*it + 0] = (tv + tv2 + carried_prob) / total_alpha;
                        log_rows[4*it + 1] = carried_prob;
                        log_rows[4*it + 2] = tv;
                        log_rows[4*it + 3] = tv2;
                }

                if(g_trace) {
                        for(unsigned c = 0; c < nch; c++) {
                                memcpy(g_trace + ((size_t)(it * nch + c) * 2) * n, cs[c].g, sizeof(float) * n);
                        }
                }
                for(unsigned c = 0; c < nch; c++) { descend(&cs[c], n, step); }
                carried_prob = 0.;
                for(unsigned c = 0; c < nch; c++) {
                        double dsum = project(&cs[c], W, H, pweight[c] != 0.f);
                        if(pweight[c] != 0.f) { carried_prob += 0.5 * dsum; }
                }
                if(g_trace) {
                        for(unsigned c = 0; c < nch; c++) {
                                memcpy(g_trace + ((size_t)(it * nch + c) * 2 + 1) * n, cs[c].x, sizeof(float) * n);
                        }
                }
        }
        for(unsigned c = 0; c < nch; c++) {
                memcpy(out[c], cs[c].x, sizeof(float) * n);
                free(cs[c].x); free(cs[c].xp); free(cs[c].g); free(cs[c].pg); free(cs[c].sub);
        }
        return 0;
}
