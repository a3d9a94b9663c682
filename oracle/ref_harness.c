/*
 * TEST INFRASTRUCTURE — not part of the product path.
 *
 * Thin flat-C wrapper that lets the test-suite and bench.py's cpu_baseline leg
 * drive the UNMODIFIED reference solver.  It is compiled together with the
 * reference's own translation units straight from /root/reference (see
 * oracle/Makefile, target _ref/libj2p_ref.so); no reference source is copied
 * into this repository.  The wrapper only
 *   - defines the `main_progressbar` global the reference's utils.c expects
 *     (declared extern in jpeg2png.h:22, normally defined in jpeg2png.c:175),
 *   - marshals plain arrays into the reference's `struct coef`
 *     (jpeg2png.h:7-20) with the ownership rules of compute() (compute.c:304-305,
 *     455-461: the incoming fdata is freed, a new W*H plane is handed back),
 *   - optionally routes the reference's CSV logger (logger.c:13,23) to a file so
 *     tests can compare the per-iteration objective trace.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "jpeg2png.h"
#include "compute.h"
#include "logger.h"
#include "utils.h"

struct progressbar *main_progressbar = NULL;

/* Run the reference compute() (compute.c:407) on caller-owned arrays.
 *   cw/ch/wsamp/hsamp : per-channel coefficient-plane geometry
 *   data[c]           : int16 block-major coefficients  (ch*cw values)
 *   fdata[c]          : decoded raster plane            (ch*cw floats)
 *   quant[c]          : 64 uint16
 *   out[c]            : receives the W*H canvas plane (caller allocates >= W*H floats)
 *   csv_path          : NULL or file that receives the reference CSV log
 *   seconds           : if non-NULL receives wall time spent inside compute()
 * returns 0, or -1 on allocation failure.  outW/outH receive the canvas size. */
int ref_compute(unsigned nchannel,
                const unsigned *cw, const unsigned *ch,
                const unsigned *wsamp, const unsigned *hsamp,
                const int16_t *const *data, const float *const *fdata,
                const uint16_t *const *quant,
                float weight, const float *pweight, unsigned iterations,
                float *const *out, unsigned *outW, unsigned *outH,
                const char *csv_path, double *seconds)
{
        if(nchannel == 0 || nchannel > 3) { return -1; }
        struct coef coefs[3];
        float pw[3];
        for(unsigned c = 0; c < nchannel; c++) {
                size_t n = (size_t)cw[c] * ch[c];
                coefs[c].w = cw[c];
                coefs[c].h = ch[c];
                coefs[c].w_samp = wsamp[c];
                coefs[c].h_samp = hsamp[c];
                coefs[c].data = malloc(n * sizeof(int16_t));
                coefs[c].fdata = alloc_simd(n * sizeof(float));
                if(!coefs[c].data) { return -1; }
                memcpy(coefs[c].data, data[c], n * sizeof(int16_t));
                memcpy(coefs[c].fdata, fdata[c], n * sizeof(float));
                memcpy(coefs[c].quant_table, quant[c], 64 * sizeof(uint16_t));
                pw[c] = pweight[c];
        }
        struct logger log;
        FILE *csv = NULL;
        if(csv_path) {
                csv = fopen(csv_path, "w");
                if(!csv) { return -1; }
        }
        logger_start(&log, csv);
        log.filename = "ref";
        log.channel = nchannel == 3 ? 3 : 0;

        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        compute(nchannel, coefs, &log, NULL, weight, pw, iterations);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if(seconds) {
                *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        }
        if(csv) { fclose(csv); }

        for(unsigned c = 0; c < nchannel; c++) {
                /* compute() rewrote w,h to the canvas size (compute.c:459-460) */
                size_t n = (size_t)coefs[c].w * coefs[c].h;
                memcpy(out[c], coefs[c].fdata, n * sizeof(float));
                *outW = coefs[c].w;
                *outH = coefs[c].h;
                free_simd(coefs[c].fdata);
                free(coefs[c].data);
        }
        return 0;
}
