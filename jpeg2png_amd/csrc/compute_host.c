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

/* Iterations per device round trip WHEN SOMEBODY IS WATCHING (a progress bar or a CSV log: compute.c:428,449-452 tick
 * once per iteration, in real time).  A host sync per iteration would cost a small image most of its speed and a
 * fixed chunk moves the bar of the default `-i 50` twice; so chunks follow the clock: one iteration each at first,
 * then a sixth of the iterations done so far (every chunk ~1/6 of the time elapsed: the bar of `-i 50` moves ~24
 * times, 4096^2 `-i 500` syncs ~36 times = under 1 % of its 60 ms), never more than ~50 ms worth or J2P_CHUNK_MAX
 * (the row buffer).  With neither a bar nor a log the whole loop goes to the device queue at once. */
#define J2P_CHUNK_MAX 256u
#define J2P_CHUNK_MS 50.0

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

/* What the host owes the caller besides the solve: the planes compute() hands back (compute.c:455-461) allocated with
 * their pages in place, so that the download writes into mapped memory (faulted in by the download itself, 64 MiB cost
 * 12 ms), and the input planes freed (compute.c:304-305).
 * WHEN it does so is dictated by a measurement (round 5, profiles/r05_host_to_host.jsonl): a change to the process's address
 * space — mmap, munmap, the page faults of a first touch — WHILE the solver's kernels run makes one launch of the loop take
 * 12-16 ms instead of 53 us (rocprofv3: one k_gradient per call; the GPU waits, not the host), in processes that have
 * solved before: 80-98 instead of 66 ms per 4096^2 call, whichever thread does it.  So none of it happens while the loop
 * runs: the output planes are prepared by helper threads BESIDE THE UPLOAD (create), the loop is issued when they are
 * done, and the inputs are freed behind the download — which also means a call that fails returns with the caller's
 * planes still the caller's (tests/test_fineprint_gpu.py).
 * And most of it does not happen at all: a full-resolution channel's input plane (w x h floats from alloc_simd,
 * jpeg.c:83-92) has exactly the size of the canvas plane compute() hands back (compute.c:455-461) whenever the channel
 * covers the canvas — every Y-only call, the luma of unpadded images, all of 4:4:4 — and its content is on the device
 * since create: the download goes INTO it and the pointer stays.  To the caller that is the contract to the letter (the
 * incoming plane is gone, an aligned W x H plane it must free is there); to the process it is 64 MiB less to map, fault,
 * copy around and unmap per 4096^2 call.  (Only a failing download — a device fault — can then leave such a plane with
 * part of its old content overwritten.) */
#define J2P_TOUCHERS 8u
struct toucher {
        char *base;
        size_t bytes;
};
static void *toucher_main(void *arg)
{
        const struct toucher *t = arg;
        for(size_t off = 0; off < t->bytes; off += 4096) { ((volatile char *)t->base)[off] = 0; }
        return NULL;
}

struct housekeeping {
        unsigned nchannel;
        size_t out_bytes;
        float *out[J2P_MAX_CHANNELS];
        int reuse[J2P_MAX_CHANNELS];    /* the channel's INPUT plane has the canvas's size: it becomes the output plane */
        int failed;
        double ms;
};

static void *housekeeping_main(void *arg)
{
        struct housekeeping *h = arg;
        const double t0 = now_ms();
        const size_t page = 4096;
        for(unsigned c = 0; c < h->nchannel; c++) {
                if(h->reuse[c]) { continue; }
                /* alloc_simd (utils.h:89-98) is aligned_alloc(16, ...); page-aligned here, and kept out of transparent huge
                 * pages (a huge-page fault next to pinned user pages is the most expensive form of the effect above) */
                h->out[c] = aligned_alloc(page, (h->out_bytes + page - 1) & ~(page - 1));
                if(!h->out[c]) { h->failed = 1; continue; }
#ifdef MADV_NOHUGEPAGE
                (void)madvise(h->out[c], h->out_bytes, MADV_NOHUGEPAGE);
#endif
                /* first touch, split over a few threads (page faults of one address space scale that far) */
                pthread_t th[J2P_TOUCHERS];
                struct toucher part[J2P_TOUCHERS];
                unsigned started = 0;
                const size_t slice = ((h->out_bytes / J2P_TOUCHERS) + page - 1) & ~(page - 1);
                for(unsigned k = 0; k < J2P_TOUCHERS; k++) {
                        const size_t lo = (size_t)k * slice;
                        if(lo >= h->out_bytes) { break; }
                        part[k].base = (char *)h->out[c] + lo;
                        part[k].bytes = h->out_bytes - lo < slice ? h->out_bytes - lo : slice;
                        if(k + 1 < J2P_TOUCHERS && h->out_bytes > (size_t)8 << 20 && pthread_create(&th[started], NULL, toucher_main, &part[k]) == 0) { started++; }
                        else { toucher_main(&part[k]); }
                }
                for(unsigned k = 0; k < started; k++) { pthread_join(th[k], NULL); }
        }
        h->ms = now_ms() - t0;
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
        unsigned W = 0, H = 0;
        for(unsigned c = 0; c < nchannel; c++) {
                planes[c].w = coefs[c].w;
                planes[c].h = coefs[c].h;
                planes[c].w_samp = coefs[c].w_samp;
                planes[c].h_samp = coefs[c].h_samp;
                planes[c].data = coefs[c].data;
                planes[c].fdata = coefs[c].fdata;
                planes[c].quant_table = coefs[c].quant_table;
                if(coefs[c].w * coefs[c].w_samp > W) { W = coefs[c].w * coefs[c].w_samp; }     /* compute.c:410-416 */
                if(coefs[c].h * coefs[c].h_samp > H) { H = coefs[c].h * coefs[c].h_samp; }
        }
        j2p_solver *s = NULL;
        j2p_tiled *t = NULL;
        int rc;
        const char *timing_env = getenv("J2P_COMPUTE_TIMING");
        const int timing = timing_env && atoi(timing_env) != 0;
        double t_mark[6] = {0., 0., 0., 0., 0., 0.};
        t_mark[0] = now_ms();
        /* the output planes, beside the upload */
        struct housekeeping hk;
        hk.nchannel = nchannel;
        hk.out_bytes = (sizeof(float) * (size_t)W * H + 15) & ~(size_t)15;
        hk.failed = 0;
        hk.ms = 0.;
        unsigned fresh = 0;
        for(unsigned c = 0; c < J2P_MAX_CHANNELS; c++) {
                hk.out[c] = NULL;
                hk.reuse[c] = c < nchannel && coefs[c].fdata && (size_t)coefs[c].w * coefs[c].h == (size_t)W * H;
                if(c < nchannel && !hk.reuse[c]) { fresh++; }
        }
        pthread_t hk_thread;
        const int hk_started = fresh && W && H && pthread_create(&hk_thread, NULL, housekeeping_main, &hk) == 0;
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
        if(hk_started) { pthread_join(hk_thread, NULL); } else if(fresh && W && H) { housekeeping_main(&hk); }
        float **outp = hk.out;
        if(rc == J2P_OK && hk.failed) { j2p_set_last_error("out of host memory for the output planes"); rc = J2P_ENOMEM; }
        t_mark[1] = t_mark[2] = t_mark[3] = now_ms();
        if(rc != J2P_OK) { goto out; }
        {
                unsigned cw = 0, ch = 0;
                if(t) { j2p_tiled_canvas(t, &cw, &ch, NULL); } else { j2p_solver_canvas(s, &cw, &ch); }
                if(cw != W || ch != H) { j2p_set_last_error("canvas size mismatch between the host and the solver"); rc = J2P_ESTATE; goto out; }
        }
        const int want_log = log && log->f && logger_log;
        j2p_log_row rows[J2P_CHUNK_MAX];
        unsigned done = 0;
        const double t_loop = now_ms();
        while(done < iterations) {
                unsigned n = iterations - done;
                /* nothing to report between chunks: the whole loop goes to the device queue at once */
                if(want_log || pb) {
                        unsigned chunk = done / 6;
                        if(done) {
                                const double per_it = (now_ms() - t_loop) / (double)done;
                                const double most = per_it > 0. ? J2P_CHUNK_MS / per_it : (double)J2P_CHUNK_MAX;
                                if((double)chunk > most) { chunk = (unsigned)most; }
                        }
                        if(chunk > J2P_CHUNK_MAX) { chunk = J2P_CHUNK_MAX; }
                        if(chunk < 1) { chunk = 1; }
                        if(n > chunk) { n = chunk; }
                }
                rc = t ? j2p_tiled_run(t, n, want_log ? rows : NULL) : j2p_solver_run(s, n, want_log ? rows : NULL);
                if(rc == J2P_OK && !want_log && pb) { rc = t ? j2p_tiled_sync(t) : j2p_solver_sync(s); }
                if(rc != J2P_OK) { break; }
                for(unsigned i = 0; i < n && (log || pb); i++) {
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
        t_mark[2] = t_mark[3] = now_ms();
        if(rc != J2P_OK) { goto out; }
        rc = t ? j2p_tiled_sync(t) : j2p_solver_sync(s);
        if(rc != J2P_OK) { goto out; }
        t_mark[3] = now_ms();
        for(unsigned c = 0; c < nchannel; c++) {
                float *dst = hk.reuse[c] ? coefs[c].fdata : outp[c];
                rc = t ? j2p_tiled_download(t, c, dst) : j2p_solver_download(s, c, dst);
                if(rc != J2P_OK) { goto out; }
        }
        t_mark[4] = now_ms();
        /* everything has succeeded: the inputs go (compute.c:304-305 frees them at aux_init; here nothing frees memory
         * while kernels run, see above) and the new planes change hands */
        for(unsigned c = 0; c < nchannel; c++) {
                if(!hk.reuse[c]) {
                        free(coefs[c].fdata);
                        coefs[c].fdata = outp[c];                                  /* compute.c:458 */
                        outp[c] = NULL;
                }
                coefs[c].w = W;                                                    /* compute.c:459-460 */
                coefs[c].h = H;
        }
out:
        if(rc != J2P_OK) { t_mark[4] = now_ms(); }
        for(unsigned c = 0; c < nchannel; c++) { free(hk.out[c]); }
        if(t) { j2p_tiled_destroy(t); }
        if(s) { j2p_solver_destroy(s); }
        t_mark[5] = now_ms();
        if(rc == J2P_OK) {
                last_times.create_ms = t_mark[1] - t_mark[0];
                last_times.issue_ms = t_mark[2] - t_mark[1];
                last_times.housekeeping_ms = hk.ms;
                last_times.wait_ms = t_mark[3] - t_mark[2];
                last_times.download_ms = t_mark[4] - t_mark[3];
                last_times.destroy_ms = t_mark[5] - t_mark[4];
                last_times.total_ms = t_mark[5] - t_mark[0];
                last_times_valid = 1;
                if(timing) {
                        fprintf(stderr, "j2p compute timing (ms): create %.2f (upload + aux_init; beside it the output planes: %.2f), issue %.2f, wait %.2f, download %.2f, free + destroy %.2f, total %.2f\n",
                                last_times.create_ms, hk.ms, last_times.issue_ms, last_times.wait_ms, last_times.download_ms,
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
