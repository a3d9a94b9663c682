/* TEST INFRASTRUCTURE — interface of the CPU oracle (see solver_oracle.c). */
#ifndef J2P_SOLVER_ORACLE_H
#define J2P_SOLVER_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one colour component as the reference's compute() receives it
 * (struct coef, /root/reference/jpeg2png.h:7-20), with const inputs */
typedef struct {
        unsigned w, h;            /* coefficient plane size, multiples of 8  */
        unsigned w_samp, h_samp;  /* subsampling factors                     */
        const int16_t *coef;      /* block-major [h/8][w/8][64], natural     */
        const uint16_t *quant;    /* 64 entries, natural order               */
        const float *pixels;      /* decoded plane, raster w*h               */
} oracle_plane;

void oracle_fdct8x8(float b[64]);
void oracle_idct8x8(float b[64]);
void oracle_decode_plane(unsigned w, unsigned h, const int16_t *coef, const uint16_t *quant, float *out);
void oracle_canvas_size(unsigned nch, const oracle_plane *pl, unsigned *W, unsigned *H);

/* restatement of compute() (compute.c:407): out[c] receives the W*H canvas plane;
 * log_rows (optional) receives iterations x {objective, prob_dist, tv, tv2} */
void oracle_set_trace(float *buf);   /* debugging aid, see solver_oracle.c */
int oracle_compute(unsigned nch, const oracle_plane *pl, float weight, const float *pweight,
                   unsigned iterations, float *const *out, double *log_rows);

#ifdef __cplusplus
}
#endif
#endif
