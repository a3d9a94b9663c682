/*
 * Host side of the drop-in: compute() with the reference's signature
 * (compute.h:8, compute.c:407-465) implemented on top of the C-ABI shim.
 * Plain C like the reference's host code; the device work is entirely behind
 * j2p_solver_* (include/jpeg2png_amd.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <assert.h>
#include <pthread.h>

#include "jpeg2png_amd.h"
#include "jpeg2png_amd_compute.h"

/* iterations per device round-trip: keeps the progress bar and the CSV log moving
 * without a host sync per iteration (SURVEY.md §7 hard part 8) */
#define J2P_CHUNK 32u

/* stands in for `omp critical(progressbar)` (compute.c:450): compute() may be entered from
 * several host threads at once (jpeg2png.c:147,330) and they share one progress bar */
static pthread_mutex_t progress_lock = PTHREAD_MUTEX_INITIALIZER;

/* the host program's callbacks (logger.c:20, progressbar.c:53).  Weak so that the
 * library also loads into processes that do not provide them (tests, bench): with
 * log->f == NULL and pb == NULL the reference never observably calls them either. */
extern void logger_log(struct logger *log, double objective, double prob_dist, double tv, double tv2) __attribute__((weak));
extern void progressbar_inc(struct progressbar *pb) __attribute__((weak));

int j2p_compute(int device, unsigned nchannel, struct coef coefs[], struct logger *log,
                struct progressbar *pb, float weight, const float pweight[], unsigned iterations)
{
        assert(FLT_ROUNDS == 1);                               /* compute.c:408 */
        if(nchannel == 0 || nchannel > J2P_MAX_CHANNELS || !coefs || !pweight) { return J2P_EINVAL; }
        j2p_plane planes[J2P_MAX_CHANNELS];
        for(unsigned c = 0; c < nchannel; c++) {
                planes[c].w = coefs[c].w;
                planes[c].h = coefs[c].h;
                planes[c].w_samp = coefs[c].w_samp;
                planes[c].h_samp = coefs[c].h_samp;
                planes[c].data = coefs[c].data;
                planes[c].fdata = coefs[c].fdata;
                planes[c].quant_table = coefs[c].quant_table;
        }
        j2p_solver *s = NULL;
        j2p_band whole = {0, 0};
        int rc = j2p_solver_create(&s, device, NULL, nchannel, planes, weight, pweight, iterations, whole, 0);
        if(rc != J2P_OK) { return rc; }
        /* aux_init frees the input planes as soon as they are up-sampled (compute.c:304-305) */
        for(unsigned c = 0; c < nchannel; c++) {
                free(coefs[c].fdata);
                coefs[c].fdata = NULL;
        }
        const int want_log = log && log->f && logger_log;
        j2p_log_row rows[J2P_CHUNK];
        unsigned done = 0;
        while(done < iterations) {
                unsigned n = iterations - done;
                if(n > J2P_CHUNK) { n = J2P_CHUNK; }
                rc = j2p_solver_run(s, n, want_log ? rows : NULL);
                if(rc == J2P_OK && !want_log && pb) { rc = j2p_solver_sync(s); }
                if(rc != J2P_OK) { j2p_solver_destroy(s); return rc; }
                for(unsigned i = 0; i < n; i++) {
                        if(log) { log->iteration = done + i; }                     /* compute.c:428 */
                        if(want_log) { logger_log(log, rows[i].objective, rows[i].prob_dist, rows[i].tv, rows[i].tv2); }
                        if(pb && progressbar_inc) {
                                pthread_mutex_lock(&progress_lock);
                                progressbar_inc(pb);                               /* compute.c:449-452 */
                                pthread_mutex_unlock(&progress_lock);
                        }
                }
                done += n;
        }
        unsigned W = 0, H = 0;
        j2p_solver_canvas(s, &W, &H);
        for(unsigned c = 0; c < nchannel; c++) {
                size_t bytes = sizeof(float) * (size_t)W * H;
                float *plane = aligned_alloc(16, (bytes + 15) & ~(size_t)15);      /* alloc_simd, utils.h:89-98 */
                if(!plane) { j2p_solver_destroy(s); return J2P_ENOMEM; }
                rc = j2p_solver_download(s, c, plane);
                if(rc != J2P_OK) { free(plane); j2p_solver_destroy(s); return rc; }
                coefs[c].fdata = plane;                                            /* compute.c:458 */
                coefs[c].w = W;                                                    /* compute.c:459-460 */
                coefs[c].h = H;
        }
        j2p_solver_destroy(s);
        return J2P_OK;
}

void compute(unsigned nchannel, struct coef coefs[], struct logger *log, struct progressbar *pb,
             float weight, float pweight[], unsigned iterations)
{
        int device = 0;
        const char *env = getenv("J2P_DEVICE");
        if(env && *env) { device = atoi(env); }
        int rc = j2p_compute(device, nchannel, coefs, log, pb, weight, pweight, iterations);
        if(rc != J2P_OK) {
                const char *msg = j2p_last_error();
                /* die(), utils.c:20-28 */
                fprintf(stderr, "jpeg2png: %s\n", (msg && *msg) ? msg : "GPU solver failed");
                exit(EXIT_FAILURE);
        }
}
