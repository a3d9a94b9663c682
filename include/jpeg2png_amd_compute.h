/*
 * Drop-in layer of libjpeg2png_amd.so: the reference's solver entry point with
 * the reference's own types, so that the library links in place of compute.o
 * (reference Makefile:33) without touching jpeg2png.c.
 *
 * The three structs restate the reference's interface types field for field —
 * `struct coef` jpeg2png.h:7-20, `struct logger` logger.h:6-11,
 * `struct progressbar` progressbar.h:4-7 — because they cross the boundary by
 * value/pointer.  When this header is included AFTER the reference's own
 * headers (a build inside the reference tree) the guards below skip the
 * re-definitions.
 */
#ifndef JPEG2PNG_AMD_COMPUTE_H
#define JPEG2PNG_AMD_COMPUTE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef JPEG2PNG_PROGRESSBAR_H
#define JPEG2PNG_PROGRESSBAR_H
struct progressbar {
        unsigned current;
        unsigned max;
};
void progressbar_inc(struct progressbar *pb);        /* progressbar.c:53-55; provided by the host program */
#endif

#ifndef JPEG2PNG_LOGGER_H
#define JPEG2PNG_LOGGER_H
struct logger {
        FILE *f;
        const char *filename;
        unsigned channel;
        unsigned iteration;
};
/* logger.c:20-28; provided by the host program */
void logger_log(struct logger *log, double objective, double prob_dist, double tv, double tv2);
#endif

#ifndef JPEG2PNG_JPEG2PNG_H
#define JPEG2PNG_JPEG2PNG_H
struct coef {
        unsigned h;
        unsigned w;
        unsigned h_samp;          /* vertical subsampling factor   */
        unsigned w_samp;          /* horizontal subsampling factor */
        int16_t *data;            /* DCT coefficients              */
        float *fdata;             /* image data                    */
        uint16_t quant_table[64];
};
#endif

/* compute.h:8 / compute.c:407.  Same contract as the reference:
 *  - consumes the incoming coef->fdata (aligned_alloc'd, compute.c:304-305) and hands back a
 *    16-byte aligned W*H plane the caller frees in coef->fdata, rewriting coef->w/h to the canvas
 *    size (compute.c:455-461) — the incoming buffer itself where it already has that size;
 *  - sets log->iteration and calls logger_log() once per iteration (compute.c:428,272),
 *    progressbar_inc() once per iteration when pb != NULL (compute.c:449-452);
 *  - re-entrant and thread-safe (callers: jpeg2png.c:144, :147-152 inside omp parallel);
 *  - errors: prints "jpeg2png: <message>" to stderr and exit(EXIT_FAILURE), like die()
 *    (utils.c:20-28).  There is NO CPU fallback: without a gfx950 device it dies.
 * Device selection: environment variable J2P_DEVICE (default 0); or J2P_DEVICES=a,b,...: a canvas of at least
 * 48 rows per listed GPU is then cut into row bands, one per GPU (j2p_compute_tiled), with results that do not
 * depend on the number of GPUs; smaller canvases run on the first GPU of the list. */
void compute(unsigned nchannel, struct coef coefs[], struct logger *log, struct progressbar *pb,
             float weight, float pweight[], unsigned iterations);

/* same, with an explicit device and an error code instead of exit(); 0 on success.
 * On ANY error return the caller still owns its inputs: coefs[c].fdata, w and h are untouched (the inputs are released
 * only behind a successful download), so the call can be retried — on another device, for instance.  (A channel whose
 * plane has the canvas's size keeps its buffer — the result is downloaded into it; only a failing download, a device
 * fault, can leave such a plane partly overwritten.) */
int j2p_compute(int device, unsigned nchannel, struct coef coefs[], struct logger *log,
                struct progressbar *pb, float weight, const float pweight[], unsigned iterations);

/* where the wall time of the calling thread's last successful compute() / j2p_compute() / j2p_compute_tiled() went (ms):
 * create = upload + aux_init (compute.c:278-310) — beside it, on helper threads, the output planes are allocated and their
 * pages touched (housekeeping: how long that took; create ends when both are done) —, issue = queueing the iteration loop,
 * wait = until the last iteration has finished, download (compute.c:455-461), destroy = freeing the inputs and the solver;
 * total = create + issue + wait + download + destroy.  J2P_ESTATE when the thread has no successful call behind it. */
typedef struct j2p_compute_times {
        double create_ms, issue_ms, housekeeping_ms, wait_ms, download_ms, destroy_ms, total_ms;
} j2p_compute_times;
int j2p_compute_timing(j2p_compute_times *out);


/* compute() with the canvas cut into `nband` row bands, band i on devices[i] (ids may repeat: several bands on
 * one GPU).  One process, one host thread per band; the bands exchange their edge rows and gradient-norm row
 * sums over peer access every iteration (reference loop: compute.c:427-453).  Same contract and — bit for bit —
 * the same planes as j2p_compute(); 0 on success. */
int j2p_compute_tiled(unsigned nband, const int devices[], unsigned nchannel, struct coef coefs[], struct logger *log,
                      struct progressbar *pb, float weight, const float pweight[], unsigned iterations);

#ifdef __cplusplus
}
#endif
#endif
