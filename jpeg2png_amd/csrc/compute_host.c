/*
 * Host side of the drop-in: compute() with the reference's signature
 * (compute.h:8, compute.c:407-465) implemented on top of the C-ABI shim.
 * Plain C like the reference's host code; the device work is entirely behind
 * j2p_solver_* (include/jpeg2png_amd.h).
 */
#define _DEFAULT_SOURCE                 /* clock_gettime, madvise under -std=c11 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <assert.h>
#include <pthread.h>
#include <time.h>
#include <sys/mman.h>

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

/* Where a call's wall time went: create = upload + aux_init issue, issue = queueing the iteration loop, housekeeping =
 * preparing the output planes and freeing the inputs on a helper thread BESIDE the loop, wait = until the last iteration
 * has finished, download, destroy.  Kept per calling thread for j2p_compute_timing() — the host-to-host figure of
 * bench.py taken apart; J2P_COMPUTE_TIMING=1 also prints one line per call on stderr. */
/* (internal, j2p_solver.hip) the calling thread's j2p_last_error() text */
extern void j2p_set_last_error(const char *msg);
extern int j2p_tiled_exchange_forced(void);          /* (internal, j2p_tiled.hip) */

static double now_ms(void)
{
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
static _Thread_local j2p_compute_times last_times;
static _Thread_local int last_times_valid;

int j2p_compute_timing(j2p_compute_times *out)
{
        if(!out || !last_times_valid) { return J2P_ESTATE; }
        *out = last_times;
        return J2P_OK;
}

/* What the host owes the caller besides the solve, done by a helper thread while the GPU iterates: the planes compute()
 * hands back (compute.c:455-461) are allocated and their pages touched, so that the download at the end writes into
 * mapped memory, and the input planes are freed (aux_init frees them as soon as they are up-sampled, compute.c:304-305 —
 * they are on the device since create).  In line, between two chunks of iterations, those ~12 ms for a 64 MiB plane left
 * the GPU idle: it runs 32 iterations in 4.
 * The inputs go only once the main thread says so (`verdict`): after EVERY iteration has been queued without an error —
 * a call that fails before that returns with coefs[c].fdata untouched, so that the caller of j2p_compute() can retry,
 * on another device for instance (tests/test_capi_gpu.py). */
struct housekeeping {
        unsigned nchannel;
        struct coef *coefs;
        size_t out_bytes;
        float *out[J2P_MAX_CHANNELS];
        int failed;
        double ms;
        pthread_mutex_t lock;
        pthread_cond_t cv;
        int verdict;               /* 0: undecided, 1: the loop is queued, free the inputs, -1: keep them */
};

static void housekeeping_outputs(struct housekeeping *h)
{
        for(unsigned c = 0; c < h->nchannel; c++) {
                /* alloc_simd (utils.h:89-98) is aligned_alloc(16, ...); 2 MiB alignment + MADV_HUGEPAGE lets a kernel with
                 * transparent huge pages map the plane with 32 faults per 64 MiB instead of 16384 (free() takes either) */
                const size_t big = (size_t)2 << 20;
                h->out[c] = h->out_bytes >= 4 * big ? aligned_alloc(big, (h->out_bytes + big - 1) & ~(big - 1)) : aligned_alloc(16, h->out_bytes);
                if(!h->out[c]) { h->failed = 1; continue; }
#ifdef MADV_HUGEPAGE
                if(h->out_bytes >= 4 * big) { (void)madvise(h->out[c], h->out_bytes, MADV_HUGEPAGE); }
#endif
                for(size_t off = 0; off < h->out_bytes; off += 4096) { ((volatile char *)h->out[c])[off] = 0; }
        }
}

static void housekeeping_inputs(struct housekeeping *h)
{
        for(unsigned c = 0; c < h->nchannel; c++) {
                free(h->coefs[c].fdata);                                           /* compute.c:304-305 */
                h->coefs[c].fdata = NULL;
        }
}

static void *housekeeping_main(void *arg)
{
        struct housekeeping *h = arg;
        const double t0 = now_ms();
        housekeeping_outputs(h);
        pthread_mutex_lock(&h->lock);
        while(h->verdict == 0) { pthread_cond_wait(&h->cv, &h->lock); }
        const int go = h->verdict > 0 && !h->failed;
        pthread_mutex_unlock(&h->lock);
        const double t1 = now_ms();
        if(go) { housekeeping_inputs(h); }
        h->ms = (t1 - t0) + (now_ms() - t1);        /* (the time spent waiting for the verdict is the main thread's) */
        return NULL;
}

/* The loop of compute.c:427-453 over either engine: one j2p_solver (whole canvas on one GPU) or one j2p_tiled
 * (row bands over several GPUs).  Same chunking, callbacks and hand-back either way. */
static int compute_on(unsigned nband, const int devices[], unsigned nchannel, struct coef coefs[], struct logger *log,
                      struct progressbar *pb, float weight, const float pweight[], unsigned iterations)
{
        assert(FLT_ROUNDS == 1);                               /* compute.c:408 */
        last_times_valid = 0;
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
        const char *timing_env = getenv("J2P_COMPUTE_TIMING");
        const int timing = timing_env && atoi(timing_env) != 0;
        double t_mark[6] = {0., 0., 0., 0., 0., 0.}, t_house = 0.;
        t_mark[0] = now_ms();
        if(nband > 1) {
                rc = j2p_tiled_create(&t, nband, devices, NULL, nchannel, planes, weight, pweight, iterations);
                if((rc == J2P_EDEVICE || rc == J2P_ENOMEM) && !j2p_tiled_exchange_forced()) {
                        /* these GPUs cannot be tiled over (no peer access and no RCCL, or no exchange that reproduces the
                         * one-GPU solve on them): the first of them solves the canvas alone — same bits, nothing has run yet */
                        fprintf(stderr, "jpeg2png_amd: not row-tiling this canvas over %u GPUs (%s); solving it on GPU %d\n", nband, j2p_last_error(), devices[0]);
                        j2p_band whole = {0, 0};
                        rc = j2p_solver_create(&s, devices[0], NULL, nchannel, planes, weight, pweight, iterations, whole, 0);
                }
        } else {
                j2p_band whole = {0, 0};
                rc = j2p_solver_create(&s, devices[0], NULL, nchannel, planes, weight, pweight, iterations, whole, 0);
        }
        if(rc != J2P_OK) { return rc; }
        t_mark[1] = now_ms();
        const int want_log = log && log->f && logger_log;
        j2p_log_row rows[J2P_CHUNK];
        unsigned W = 0, H = 0;
        if(t) { j2p_tiled_canvas(t, &W, &H, NULL); } else { j2p_solver_canvas(s, &W, &H); }
        struct housekeeping hk;
        hk.nchannel = nchannel;
        hk.coefs = coefs;
        hk.out_bytes = (sizeof(float) * (size_t)W * H + 15) & ~(size_t)15;
        hk.failed = 0;
        hk.ms = 0.;
        hk.verdict = 0;
        pthread_mutex_init(&hk.lock, NULL);
        pthread_cond_init(&hk.cv, NULL);
        for(unsigned c = 0; c < J2P_MAX_CHANNELS; c++) { hk.out[c] = NULL; }
        pthread_t hk_thread;
        const int hk_started = pthread_create(&hk_thread, NULL, housekeeping_main, &hk) == 0;
        unsigned done = 0;
        while(done < iterations) {
                unsigned n = iterations - done;
                /* nothing to report between chunks: the whole loop goes to the device queue at once */
                if(n > J2P_CHUNK && (want_log || pb)) { n = J2P_CHUNK; }
                rc = t ? j2p_tiled_run(t, n, want_log ? rows : NULL) : j2p_solver_run(s, n, want_log ? rows : NULL);
                if(rc == J2P_OK && !want_log && pb) { rc = t ? j2p_tiled_sync(t) : j2p_solver_sync(s); }
                if(rc != J2P_OK) { break; }
                for(unsigned i = 0; i < n && (log || pb); i++) {
                        if(log) { log->iteration = done + i; }                     /* compute.c:428 */
                        if(want_log) { logger_log(log, rows[i % J2P_CHUNK].objective, rows[i % J2P_CHUNK].prob_dist, rows[i % J2P_CHUNK].tv, rows[i % J2P_CHUNK].tv2); }
                        if(pb && progressbar_inc) {
                                pthread_mutex_lock(&progress_lock);
                                progressbar_inc(pb);                               /* compute.c:449-452 */
                                pthread_mutex_unlock(&progress_lock);
                        }
                }
                done += n;
        }
        /* every iteration is queued (or one failed): the inputs may go (or stay) */
        if(hk_started) {
                pthread_mutex_lock(&hk.lock);
                hk.verdict = rc == J2P_OK ? 1 : -1;
                pthread_cond_signal(&hk.cv);
                pthread_mutex_unlock(&hk.lock);
                t_mark[2] = now_ms();
                pthread_join(hk_thread, NULL);
        } else {
                t_mark[2] = now_ms();
                const double h0 = now_ms();
                housekeeping_outputs(&hk);
                if(rc == J2P_OK && !hk.failed) { housekeeping_inputs(&hk); }
                hk.ms = now_ms() - h0;
        }
        pthread_mutex_destroy(&hk.lock);
        pthread_cond_destroy(&hk.cv);
        t_house = hk.ms;
        float **outp = hk.out;
        if(rc == J2P_OK && hk.failed) { j2p_set_last_error("out of host memory for the output planes"); rc = J2P_ENOMEM; }
        t_mark[3] = t_mark[2];
        if(rc != J2P_OK) { goto out; }
        rc = t ? j2p_tiled_sync(t) : j2p_solver_sync(s);
        if(rc != J2P_OK) { goto out; }
        t_mark[3] = now_ms();
        for(unsigned c = 0; c < nchannel; c++) {
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
        t_mark[4] = now_ms();
        for(unsigned c = 0; c < nchannel; c++) { free(hk.out[c]); }
        if(t) { j2p_tiled_destroy(t); }
        if(s) { j2p_solver_destroy(s); }
        t_mark[5] = now_ms();
        if(rc == J2P_OK) {
                last_times.create_ms = t_mark[1] - t_mark[0];
                last_times.issue_ms = t_mark[2] - t_mark[1];
                last_times.housekeeping_ms = t_house;
                last_times.wait_ms = t_mark[3] - t_mark[2];
                last_times.download_ms = t_mark[4] - t_mark[3];
                last_times.destroy_ms = t_mark[5] - t_mark[4];
                last_times.total_ms = t_mark[5] - t_mark[0];
                last_times_valid = 1;
                if(timing) {
                        fprintf(stderr, "j2p compute timing (ms): create %.2f, issue %.2f (beside it, on a helper thread: housekeeping %.2f), wait %.2f, download %.2f, destroy %.2f, total %.2f\n",
                                last_times.create_ms, last_times.issue_ms, t_house, last_times.wait_ms, last_times.download_ms,
                                last_times.destroy_ms, last_times.total_ms);
                }
        }
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
