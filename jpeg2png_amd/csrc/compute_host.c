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
/* likewise the first half of the host's die() (utils.c:11-17): it wipes the progress bar off the line before the
 * `jpeg2png: ` prefix goes out.  Used when the host has it, so that a failure in here reads like one of its own. */
extern void die_message_start(void) __attribute__((weak));

/* The loop of compute.c:427-453 over either engine: one j2p_solver (whole canvas on one GPU) or one j2p_tiled
 * (row bands over several GPUs).  Same chunking, callbacks and hand-back either way. */
static int compute_on(unsigned nband, const int devices[], unsigned nchannel, struct coef coefs[], struct logger *log,
                      struct progressbar *pb, float weight, const float pweight[], unsigned iterations)
{
        assert(FLT_ROUNDS == 1);                               /* compute.c:408 */
        if(nchannel == 0 || nchannel > J2P_MAX_CHANNELS || !coefs || !pweight || !devices || nband == 0) { return J2P_EINVAL; }
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
        j2p_tiled *t = NULL;
        int rc;
        if(nband > 1) {
                rc = j2p_tiled_create(&t, nband, devices, NULL, nchannel, planes, weight, pweight, iterations);
        } else {
                j2p_band whole = {0, 0};
                rc = j2p_solver_create(&s, devices[0], NULL, nchannel, planes, weight, pweight, iterations, whole, 0);
        }
        if(rc != J2P_OK) { return rc; }
        const int want_log = log && log->f && logger_log;
        j2p_log_row rows[J2P_CHUNK];
        float *outp[J2P_MAX_CHANNELS] = {NULL, NULL, NULL};
        unsigned W = 0, H = 0;
        if(t) { j2p_tiled_canvas(t, &W, &H, NULL); } else { j2p_solver_canvas(s, &W, &H); }
        const size_t out_bytes = (sizeof(float) * (size_t)W * H + 15) & ~(size_t)15;
        unsigned done = 0;
        int housekeeping = iterations == 0;    /* (no iterations: nothing to hide it behind, done below) */
        while(done < iterations) {
                unsigned n = iterations - done;
                if(n > J2P_CHUNK) { n = J2P_CHUNK; }
                rc = t ? j2p_tiled_run(t, n, want_log ? rows : NULL) : j2p_solver_run(s, n, want_log ? rows : NULL);
                if(rc == J2P_OK && !housekeeping) {
                        /* While the GPU works on the first chunk (the run call only queues it unless log rows are wanted):
                         * aux_init frees the input planes as soon as they are up-sampled (compute.c:304-305) — they are
                         * on the device since create — and the planes compute() hands back (compute.c:455-461) are
                         * allocated and their pages touched, so that the download at the end writes into mapped memory */
                        housekeeping = 1;
                        for(unsigned c = 0; c < nchannel; c++) {
                                free(coefs[c].fdata);
                                coefs[c].fdata = NULL;
                                outp[c] = aligned_alloc(16, out_bytes);                /* alloc_simd, utils.h:89-98 */
                                if(!outp[c]) { rc = J2P_ENOMEM; goto out; }
                                for(size_t off = 0; off < out_bytes; off += 4096) { ((volatile char *)outp[c])[off] = 0; }
                        }
                }
                if(rc == J2P_OK && !want_log && pb) { rc = t ? j2p_tiled_sync(t) : j2p_solver_sync(s); }
                if(rc != J2P_OK) { goto out; }
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
        for(unsigned c = 0; c < nchannel; c++) {
                if(!outp[c]) {                                                     /* iterations == 0 */
                        free(coefs[c].fdata);
                        coefs[c].fdata = NULL;
                        outp[c] = aligned_alloc(16, out_bytes);
                        if(!outp[c]) { rc = J2P_ENOMEM; goto out; }
                }
                rc = t ? j2p_tiled_download(t, c, outp[c]) : j2p_solver_download(s, c, outp[c]);
                if(rc != J2P_OK) { goto out; }
        }
        for(unsigned c = 0; c < nchannel; c++) {
                coefs[c].fdata = outp[c];                                          /* compute.c:458 */
                outp[c] = NULL;
                coefs[c].w = W;                                                    /* compute.c:459-460 */
                coefs[c].h = H;
        }
out:
        for(unsigned c = 0; c < nchannel; c++) { free(outp[c]); }
        if(t) { j2p_tiled_destroy(t); }
        if(s) { j2p_solver_destroy(s); }
        return rc;
}

int j2p_compute(int device, unsigned nchannel, struct coef coefs[], struct logger *log,
                struct progressbar *pb, float weight, const float pweight[], unsigned iterations)
{
        return compute_on(1, &device, nchannel, coefs, log, pb, weight, pweight, iterations);
}

int j2p_compute_tiled(unsigned nband, const int devices[], unsigned nchannel, struct coef coefs[], struct logger *log,
                      struct progressbar *pb, float weight, const float pweight[], unsigned iterations)
{
        return compute_on(nband, devices, nchannel, coefs, log, pb, weight, pweight, iterations);
}

/* rows a band must at least have before compute() spreads a canvas over the GPUs of J2P_DEVICES: three 16-row
 * gradient segments, so that every band has an interior to hide the halo exchange behind */
#define J2P_MIN_BAND_ROWS (3u * J2P_TILE_ROWS)

void compute(unsigned nchannel, struct coef coefs[], struct logger *log, struct progressbar *pb,
             float weight, float pweight[], unsigned iterations)
{
        /* J2P_DEVICE=n: that GPU.  J2P_DEVICES=a,b,...: the canvas is row-tiled over those GPUs when it is tall
         * enough, otherwise (and for the other calls of a multi-threaded host) the first one is used. */
        int devs[32];
        unsigned ndev = 0;
        const char *list = getenv("J2P_DEVICES");
        if(list && *list) {
                const char *p = list;
                while(*p && ndev < 32) {
                        char *end = NULL;
                        long v = strtol(p, &end, 10);
                        if(end == p) { break; }
                        devs[ndev++] = (int)v;
                        p = *end == ',' ? end + 1 : end;
                }
        }
        if(ndev == 0) {
                const char *env = getenv("J2P_DEVICE");
                devs[ndev++] = (env && *env) ? atoi(env) : 0;
        }
        unsigned nband = 1;
        if(ndev > 1 && coefs && nchannel >= 1 && nchannel <= J2P_MAX_CHANNELS) {
                unsigned H = 0, align = J2P_TILE_ROWS;
                for(unsigned c = 0; c < nchannel; c++) {
                        if(coefs[c].h * coefs[c].h_samp > H) { H = coefs[c].h * coefs[c].h_samp; }
                        while(coefs[c].h_samp && align % (8 * coefs[c].h_samp)) { align += J2P_TILE_ROWS; }
                }
                unsigned per = align > J2P_MIN_BAND_ROWS ? align : J2P_MIN_BAND_ROWS;
                per = (per + align - 1) / align * align;
                nband = H / per;
                if(nband > ndev) { nband = ndev; }
                if(nband < 1) { nband = 1; }
        }
        int rc = compute_on(nband, devs, nchannel, coefs, log, pb, weight, pweight, iterations);
        if(rc != J2P_OK) {
                const char *msg = j2p_last_error();
                /* die(), utils.c:20-28 */
                if(die_message_start) { die_message_start(); } else { fprintf(stderr, "jpeg2png: "); }
                fprintf(stderr, "%s\n", (msg && *msg) ? msg : "GPU solver failed");
                exit(EXIT_FAILURE);
        }
}
