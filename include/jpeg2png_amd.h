/*
 * jpeg2png_amd — MI355X (gfx950) implementation of jpeg2png's deblocking solver.
 *
 * C-ABI of libjpeg2png_amd.so.  Plain pointers and sizes only; every function
 * except j2p_last_error()/j2p_version() returns 0 on success and a negative
 * J2P_E* code on failure (message via j2p_last_error()); nothing here ever
 * calls exit().  Citations are into the reference tree (victorvde/jpeg2png).
 *
 * Two layers:
 *
 *  1. Drop-in layer (declared in jpeg2png_amd_compute.h, C host code):
 *       void compute(unsigned nchannel, struct coef coefs[], struct logger *log,
 *                    struct progressbar *pb, float weight, float pweight[],
 *                    unsigned iterations);
 *     same symbol, signature, ownership and callback behaviour as
 *     compute.h:8 / compute.c:407-465, linked in place of compute.o.
 *
 *  2. Shim layer (this file): what that host code — or any other FFI (ctypes,
 *     cgo, JNI) — binds.  A `j2p_solver` owns the device-resident working set of
 *     ONE compute() call (the reference's `struct aux`, compute.c:21-34, for all
 *     channels) on one GPU.  A solver may hold just a horizontal BAND of rows of
 *     a taller plane (row tiling over several GPUs); the caller then exchanges
 *     the band-edge rows and the per-row-of-tiles gradient norms between the two
 *     phase calls of every iteration.
 */
#ifndef JPEG2PNG_AMD_H
#define JPEG2PNG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define J2P_OK          0
#define J2P_EINVAL     -1   /* bad argument / precondition of compute() violated   */
#define J2P_ENOMEM     -2   /* host or device allocation failed                    */
#define J2P_EDEVICE    -3   /* HIP runtime error, no usable gfx950 device          */
#define J2P_ESTATE     -4   /* call sequence error                                 */

#define J2P_MAX_CHANNELS 3  /* ASSUME(nchannel <= 3), compute.c:118                */
#define J2P_HALO_ROWS    2  /* TGV2 gradient reach in rows, compute.c:137-143,165-183 */
#define J2P_TILE_ROWS   16  /* rows per gradient tile = granularity of the norm partials */

/* One colour component as compute() receives it — the fields of `struct coef`
 * (jpeg2png.h:7-20) that the solver reads.  All pointers are HOST pointers. */
typedef struct j2p_plane {
        unsigned w, h;              /* coefficient-plane size; multiples of 8 (box.c:6-7)   */
        unsigned w_samp, h_samp;    /* max_samp / comp_samp (jpeg.c:57-58), >= 1            */
        const int16_t *data;        /* quantised DCT coefficients, block-major
                                       [h/8][w/8][64], natural order (jpeg.c:68-77)         */
        const float *fdata;         /* decoded plane, row-major w*h (jpeg.c:83-92 +
                                       jpeg2png.c:131-139); may be NULL: the library then
                                       decodes `data` itself on the device                  */
        const uint16_t *quant_table;/* 64 entries, natural order, all non-zero (jpeg.c:41-46) */
} j2p_plane;

/* Row band of the canvas this solver owns, in canvas rows.  {0,0} = whole canvas.
 * row_begin/row_end must be multiples of lcm(8*h_samp) over channels and of
 * J2P_TILE_ROWS so that no DCT block and no gradient tile straddles two GPUs. */
typedef struct j2p_band {
        unsigned row_begin, row_end;
} j2p_band;

typedef struct j2p_solver j2p_solver;

/* per-iteration log row = arguments of logger_log() (logger.c:20, compute.c:271-272) */
typedef struct j2p_log_row {
        double objective, prob_dist, tv, tv2;
} j2p_log_row;

const char *j2p_version(void);
const char *j2p_last_error(void);            /* thread-local */
int j2p_device_count(int *count);

/* aux_init (compute.c:278-310) on the device: uploads the planes (for a band:
 * only the coefficient rows the band covers; `planes` always describes the
 * WHOLE image, host arrays may be band-local when bit 0 of band_local_arrays is set;
 * bit 1, J2P_BAND_EVEN_IF_WHOLE: a band covering every row is still driven with the
 * phase calls and exchanges like any other band — one-band runs of the row tiling),
 * allocates x_k, x_{k-1}, gradient, prob state, and sets x_k = x_{k-1} =
 * replicate-upsample(fdata).  `iterations` fixes the step size
 * radius/sqrtf(1+iterations) (compute.c:443).  `stream` is a hipStream_t
 * (NULL = the solver creates its own). */
int j2p_solver_create(j2p_solver **out, int device, void *stream,
                      unsigned nchannel, const j2p_plane planes[],
                      float weight, const float pweight[], unsigned iterations,
                      j2p_band band, int band_local_arrays);
void j2p_solver_destroy(j2p_solver *s);
#define J2P_BAND_LOCAL_ARRAYS  1
#define J2P_BAND_EVEN_IF_WHOLE 2

/* Device memory of destroyed solvers is cached per process and handed to the next solver on the same device
 * (hipMalloc / hipFree cost milliseconds and hipFree stalls every stream of the device, which matters when the
 * host calls compute() from several threads, jpeg2png.c:147,330).  Bounded PER DEVICE: 16 blocks and 8 GiB
 * (environment J2P_POOL_MIB = another limit in MiB, 0 = cache nothing); any device allocation of the library
 * that runs out of memory drops the cache and retries.  j2p_pool_trim() releases the cache; j2p_batch_destroy()
 * and j2p_tiled_destroy() call it. */
void j2p_pool_trim(void);

/* diagnostics: schedule switches that change speed, never results (A/B timing and the parity tests of both
 * schedules).  Only between iterations. */
#define J2P_OPT_NORM_FOLD     1   /* 1 (default for band solvers and for whole canvases up to 2.5 Mpixel): level 1 of the
                                     ||g|| reduction runs inside the gradient kernel (its last-arriving wavefronts);
                                     0: separate reduction kernel — same bits */
#define J2P_OPT_JOINT_INWAVE  2   /* 1: all channels of a joint image in one wavefront; 0 (default): one wavefront
                                     per channel.  Environment J2P_JOINT_INWAVE sets the default at create time */
#define J2P_OPT_NORM_IN_PROJECT 4 /* 1 (needs NORM_FOLD): the gradient kernel leaves per-tile-row sums and every wavefront of
                                     the projection kernel runs the final tree itself: no reduction launch in between */
#define J2P_OPT_NT_GRADIENT 5     /* 0..3: which streams are accessed non-temporally (1: the gradient plane, 2: + the prob
                                     state, 3: + the coefficients); negative = default: by the working sets of ALL
                                     solvers live on the device against its Infinity Cache (256 MiB), re-evaluated at
                                     create and reset (environment J2P_NT_SCOPE=solver: this solver's own only) */
#define J2P_OPT_MIXED_PROJECT 6   /* 1 (default): canvases up to 1 Mpixel project all channels in ONE launch whatever their
                                     sampling; 0: one launch per sampling class, as large canvases do */
#define J2P_OPT_NARROW_COEFFICIENTS 7 /* 1 (default): a channel whose quantised coefficients all lie in [-127, 127] (found at create)
                                     keeps them resident as one byte each and the projection reads those: 21 instead of
                                     22 bytes per pixel; 0: the int16 form (jpeg2png.h:14) for every channel — same floats, same bits */
int j2p_solver_debug_option(j2p_solver *s, int option, int value);
/* bytes per quantised coefficient the projection kernel reads for channel c: 1 or 2 (J2P_OPT_NARROW_COEFFICIENTS) */
int j2p_solver_coefficient_bytes(const j2p_solver *s, unsigned c, unsigned *bytes);

/* The checked build (the analogue of the reference's DEBUG=1, whose pixel indexer p() asserts every access,
 * utils.h:68-81): compiled with -DJ2P_DEBUG, every global load and store of the two phase kernels is compared
 * with the byte range it is meant to stay in; violations are counted on the device.  j2p_debug_build() tells
 * which build this is; j2p_solver_debug_violations() returns the count, the code of the first offending site
 * (1xx = gradient phase, 2xx = projection phase, see j2p_kernels.hip.h) and its offset; J2P_ESTATE in a release
 * build, where the checks are compiled out. */
int j2p_debug_build(void);
/* Test hook (no device needed): the map from k_gradient's workgroups to (strip, rows) — j2p_kernels.hip.h: grad_item —
 * evaluated on the host for a W x rows plane with `rows_per_tile`-row tile rows, one wavefront per strip
 * (channel_wavefronts 1) or per strip and channel (2, 3), the given shares (1/256 of every XCD's run) of double / half /
 * quarter tile-row items, top-down or bottom-up.  items: up to max_items records of five unsigned {strip, first row,
 * rows, tile row, kind (0 whole, 1 half, 2 quarter, 3 double)}; *n_items: how many the launch has; *workgroups: its grid. */
int j2p_debug_grad_items(unsigned W, unsigned rows, unsigned rows_per_tile, unsigned channel_wavefronts, unsigned zone_d, unsigned zone_b,
                         unsigned zone_c, int reverse, unsigned *items, unsigned max_items, unsigned *n_items, unsigned *workgroups);
int j2p_solver_debug_violations(j2p_solver *s, unsigned long long *count, unsigned *site, unsigned long long *offset);

/* Timing tool (builds with -DJ2P_TRACE only, tools/wave_trace.py; J2P_ESTATE otherwise): while on, every wavefront
 * of the two phase kernels appends {start, first data, end (10 ns ticks), id} — four 64-bit words — to a device
 * buffer; a call with host_out copies up to max_records records out and empties the buffer. */
int j2p_solver_trace(j2p_solver *s, int on, unsigned long long *host_out, unsigned max_records, unsigned *n);

/* canvas geometry (compute.c:410-416) and band bookkeeping */
int j2p_solver_canvas(const j2p_solver *s, unsigned *W, unsigned *H);
int j2p_solver_band(const j2p_solver *s, unsigned *row_begin, unsigned *row_end);

/* kernel launches per iteration of an unlogged j2p_solver_run(): 2 = gradient and projection with ||g|| reduced inside
 * them, 3 = with a reduction launch in between */
int j2p_solver_launches_per_iteration(const j2p_solver *s, unsigned *n);

/* back to iteration 0 from the inputs that are already resident in HBM */
int j2p_solver_reset(j2p_solver *s);

/* The whole iteration loop (compute.c:427-453) for a solver that owns the whole
 * canvas.  Runs `n` further iterations asynchronously on the solver's stream;
 * if `rows` is non-NULL it receives n log rows (this synchronises at the end). */
int j2p_solver_run(j2p_solver *s, unsigned n, j2p_log_row *rows);

/* The same loop split at its two device-wide dependencies, for row-tiled runs.
 *   phase_gradient : FISTA point (compute.c:430-440) + prob/TV/TGV2 gradient
 *                    (compute.c:239-261) for the band; leaves one double per
 *                    channel per row-of-tiles (sum of g*g) in the partials buffer.
 *   [caller all-gathers the partials of all bands into norm_partials_all]
 *   phase_project  : norm (compute.c:200-207, fixed summation order over the
 *                    GLOBAL partial array => GPU-count invariant), step
 *                    (:209-216), projection (:334-404), next prob state.
 *   [caller sends the band's first/last J2P_HALO_ROWS rows of the new iterate
 *    to the neighbouring bands' halo rows]
 */
int j2p_solver_phase_gradient(j2p_solver *s);
int j2p_solver_phase_project(j2p_solver *s);

/* phase_gradient in two parts, to hide the halo exchange behind compute: the INTERIOR part
 * (every 16-row segment except the band's first and last) reads no halo row and can be issued
 * straight after phase_project; the EDGES part needs the neighbours' rows and may go to another
 * stream (NULL = the solver's).  Once the solver's stream has been made to wait for the EDGES
 * part, j2p_solver_phase_rowsums() finishes the phase (partials_local is valid after it). */
/* (Both split forms measured slower than whole phases wherever tried — DESIGN.md section 10 — and answer only in the
 * experiments build, -DJ2P_EXPERIMENTS: j2p_experiments_build() == 1; the release library returns J2P_ESTATE.) */
int j2p_experiments_build(void);
#define J2P_GRADIENT_INTERIOR 1
#define J2P_GRADIENT_EDGES    2
int j2p_solver_phase_gradient_part(j2p_solver *s, int part, void *stream);
int j2p_solver_phase_rowsums(j2p_solver *s);

/* phase_project in two parts, for the same purpose: the BOUNDARY part computes the norm and projects the
 * band's first and last block row of every channel — the rows send_top / send_bottom point into — so the
 * halo exchange can start while the INTERIOR part (everything else; ends the iteration) is still running. */
#define J2P_PROJECT_BOUNDARY 1
#define J2P_PROJECT_INTERIOR 2
int j2p_solver_phase_project_part(j2p_solver *s, int part);

/* Device addresses the caller needs for the exchanges (all on the solver's device).
 *   partials_local : nchannel * local_tile_rows doubles written by phase_gradient
 *   partials_all   : nchannel * global_tile_rows doubles read by phase_project;
 *                    for a whole-canvas solver both are the same buffer. Layout
 *                    [tile_row][channel], so the bands of consecutive GPUs concatenate.
 *   halo addresses : for channel c, the rows of the CURRENT iterate x_k:
 *                    send_top  = first J2P_HALO_ROWS own rows, recv_top = the halo
 *                    rows above them (same for bottom); each J2P_HALO_ROWS*W floats.
 *   log_local      : NULL unless j2p_solver_set_logging() is on; then 2 + J2P_MAX_CHANNELS doubles:
 *                    the band's tv and tv2 sums (valid after the gradient phase) and its prob distance
 *                    per channel for the state the projection phase leaves. */
typedef struct j2p_exchange {
        double *partials_local;  unsigned local_tile_rows;
        double *partials_all;    unsigned global_tile_rows; unsigned first_tile_row;
        float *send_top[J2P_MAX_CHANNELS], *recv_top[J2P_MAX_CHANNELS];
        float *send_bottom[J2P_MAX_CHANNELS], *recv_bottom[J2P_MAX_CHANNELS];
        size_t halo_floats;
        double *log_local;
} j2p_exchange;
int j2p_solver_exchange_info(j2p_solver *s, j2p_exchange *info);

/* Pieces of a row-tiled run inside ONE process (what j2p_tiled below is made of; peers' device pointers must be
 * accessible from the solver's device: same GPU, or peer access enabled).
 *   stream            : the hipStream_t the solver launches on
 *   halo_rows         : the send/recv row addresses of j2p_exchange for x buffer 0 or 1 — the iterate produced by
 *                       iteration k (0-based) lives in buffer (k + 1) & 1 — independent of the solver's state
 *   norm_from_bands   : between the two phases: ||g|| from every band's level-1 sums (partials_local), read in
 *                       place; replaces the all-gather into partials_all.  nout == 0: the result is this solver's
 *                       norm; nout > 0: it is stored into every norm_out[i] (j2p_solver_norm_ptr() of the bands,
 *                       this solver's included) — ONE band reduces for all, the others call norm_external
 *   norm_external     : between the two phases: another band's norm_from_bands has stored (or will have stored, by
 *                       the time the solver's stream gets there — the caller orders the streams) this solver's norm
 *   alternate_rowsums : band solvers: from now on iteration k leaves its level-1 sums in buffers[k & 1] (returned;
 *                       buffers[0] is partials_local), so that a band may start its next gradient phase while
 *                       slower bands are still reading this iteration's sums
 *   copy_rows         : n row blocks copied on the solver's stream (the neighbours' edge rows into its halo rows) */
int j2p_solver_stream(j2p_solver *s, void **stream);
int j2p_solver_halo_rows(j2p_solver *s, int buffer, j2p_exchange *info);
int j2p_solver_alternate_rowsums(j2p_solver *s, const double *buffers[2]);
int j2p_solver_norm_from_bands(j2p_solver *s, unsigned nband, const double *const rowsums[],
                               const unsigned first_tile_row[], const unsigned tile_rows[],
                               unsigned nout, float *const norm_out[]);
int j2p_solver_norm_ptr(j2p_solver *s, float **norm);              /* device address of the solver's [channel] norms */
int j2p_solver_norm_external(j2p_solver *s);
int j2p_solver_copy_rows(j2p_solver *s, unsigned n, float *const dst[], const float *const src[], size_t floats);

/* Bands that can WRITE each other's memory (same GPU, or peer access): the two per-iteration exchanges without a
 * kernel of their own.  After link_bands a band's projection phase stores its first / last J2P_HALO_ROWS rows of the
 * new iterate ALSO into the neighbouring solvers' halo rows (no halo copy or send), and its gradient phase stores
 * its level-1 sums of ||g||^2 into EVERY band's own copy of the global [tile row][channel] array (no reduction
 * launch on a root band, no norm event); the projection phase then reduces ||g|| from its own copy inside k_project.
 * A band iteration is two launches, and all traffic between GPUs is posted writes.
 *   up_halo / down_halo : [x buffer 0 / 1][channel] — j2p_solver_halo_rows(neighbour above, buffer).recv_bottom /
 *                         (neighbour below, buffer).recv_top; all NULL at the canvas's first / last band
 *   push                : [iteration parity][band] — every band's j2p_solver_global_rowsums() arrays (this band's
 *                         own included), npush of them
 * The CALLER orders the streams: a band's gradient phase of iteration k behind its neighbours' projection of k - 1;
 * its projection of k behind EVERY band's gradient phase of k — which is also what makes the stores into the
 * neighbours' halo rows safe: they replace the halo of x_{k-1}, which the neighbours' gradient phase of k still read. */
typedef struct j2p_band_links {
        float *up_halo[2][J2P_MAX_CHANNELS];
        float *down_halo[2][J2P_MAX_CHANNELS];
        unsigned npush;
        double *push[2][32];
        /* optional (ncount 0 = none): counters in memory every GPU can update atomically (coherent pinned host memory);
         * the gradient launch adds 1 to each when the band's last tile row has been pushed, so that a band's stream can
         * wait for "every band's gradient has finished" with ONE hipStreamWaitValue64 (nband counts per iteration) */
        unsigned ncount;
        unsigned long long *count[32];
} j2p_band_links;
int j2p_solver_global_rowsums(j2p_solver *s, double *arrays[2]);   /* band solvers: [global tile row][channel], even / odd iterations */
int j2p_solver_link_bands(j2p_solver *s, const j2p_band_links *links);   /* NULL: back to halo rows of its own */

/* One plane set row-tiled over several GPUs from one process: nband solvers, band i on devices[i] (ids may
 * repeat), one host thread per band.  cuts = nband + 1 row boundaries from 0 to the canvas height, aligned to
 * lcm(16, 8 * h_samp), or NULL for near-equal bands.  run / sync / download mirror the j2p_solver calls; the
 * planes are bit-identical to a whole-canvas solver's whatever the cut.  (Reference loop: compute.c:427-453.)
 * nband == 1 is a plain whole-canvas solver behind the same calls.  How the bands exchange their row sums of g^2 and
 * their edge rows is chosen at create time (j2p_tiled_exchange() names it).  Environment J2P_TILED_EXCHANGE / J2P_TILED_WAIT
 * NAME one (then nothing else is tried and a failure is an error).  Unnamed, and with every band on a GPU of its own, the
 * first create on a device list VERIFIES the candidates there: a scratch canvas cut from the job's first rows (three
 * 16-row tile rows per band) is solved whole by one plain solver — no exchange: the truth — and as bands through each
 * candidate; a candidate whose planes differ in one bit is demoted (one line on stderr), the fastest of the rest is kept
 * for that device list for the life of the process (J2P_TILED_VERIFY=0: never; =1: also when bands share a GPU; =2: also
 * print the timings).  Bands that share a GPU get "direct" with event waits.  The exchanges:
 *   "direct" (needs every GPU to be able to write every other's memory): both exchanges ride on the two phase kernels
 *            as posted peer writes (j2p_solver_link_bands) — two launches per band and iteration; J2P_TILED_WAIT=all
 *            (default) | root | collector | counter: how a band's projection learns that every band's gradient has
 *            finished (events, or — counter — a value in host memory the kernels count up, hipStreamWaitValue64);
 *   "copy"   round 3's schedule — a copy kernel pulls the neighbours' edge rows, one band reduces ||g|| for all
 *            (experiments build only, J2P_TILED_NORM=all: every band for itself) — kept as the cross-check of "direct" and for canvases taller
 *            than 16384 rows;
 *   "rccl"   (the candidate without peer access) ncclAllGather + grouped ncclSend / ncclRecv on the band's own stream, one
 *            communicator per band from ncclCommInitAll, librccl loaded with dlopen (J2P_RCCL_LIBRARY names another
 *            copy); needs one GPU per band.
 * J2P_EDEVICE when none of them works — or verifies — on the devices given.  host_cpu_seconds: user + system time the band
 * threads have spent issuing work so far. */
typedef struct j2p_tiled j2p_tiled;
int j2p_tiled_create(j2p_tiled **out, unsigned nband, const int devices[], const unsigned cuts[], unsigned nchannel,
                     const j2p_plane planes[], float weight, const float pweight[], unsigned iterations);
void j2p_tiled_destroy(j2p_tiled *t);
int j2p_tiled_canvas(const j2p_tiled *t, unsigned *W, unsigned *H, unsigned *nband);
int j2p_tiled_band(const j2p_tiled *t, unsigned band, int *device, unsigned *row_begin, unsigned *row_end, j2p_solver **solver);
int j2p_tiled_reset(j2p_tiled *t);                                 /* back to iteration 0 from the resident inputs */
int j2p_tiled_run(j2p_tiled *t, unsigned n, j2p_log_row *rows);   /* asynchronous unless rows != NULL */
int j2p_tiled_sync(j2p_tiled *t);
int j2p_tiled_download(j2p_tiled *t, unsigned c, float *out);      /* W * H floats */
int j2p_tiled_host_cpu_seconds(const j2p_tiled *t, double *seconds);
/* the librccl the "rccl" exchange would use (dlopen, J2P_RCCL_LIBRARY): ncclGetVersion's code (e.g. 22203), J2P_EDEVICE
 * when no usable library is found — the exchange SURVEY.md §8e names (RCCL halo send/recv + norm all-gather) */
int j2p_rccl_version(int *version);
int j2p_tiled_exchange(const j2p_tiled *t, const char **name);    /* "direct" ("direct, wait root" / "…collector"), "copy", "rccl"; "none": one plain band */

/* CSV logging for band solvers (the "+3 doubles when logging" of the norm exchange): with logging on, the
 * phase calls also leave the band's tv / tv2 / prob sums in j2p_exchange.log_local; the caller adds the bands'
 * sums up (any fixed order: the values only feed the log) and turns n iterations' worth of
 * {tv, tv2, prob[J2P_MAX_CHANNELS]} into the reference's log rows (compute.c:226-272, logger.c:20-28):
 * row i reports the prob distance of the state ENTERING iteration i (0 at iteration 0). */
int j2p_solver_set_logging(j2p_solver *s, int on);
int j2p_log_rows_from_sums(unsigned nchannel, float weight, const float pweight[], unsigned n,
                           const double *sums /* n * (2 + J2P_MAX_CHANNELS) */, j2p_log_row *rows);

/* band-local arrays only: after the caller has exchanged the halo rows of the INITIAL
 * iterate, copy them into x_{k-1}'s halo rows too (fista = copy(fdata), compute.c:307-309) */
int j2p_solver_commit_initial_halo(j2p_solver *s);

/* copy channel c's band rows of the current iterate to host (W * band_rows floats);
 * this is what compute() hands back in coef->fdata (compute.c:455-461) */
int j2p_solver_download(j2p_solver *s, unsigned c, float *out);

/* diagnostics: channel c's objective gradient as the last gradient phase left it (W * band_rows
 * floats; compute.c's aux->obj_gradient before the step) */
int j2p_solver_download_gradient(j2p_solver *s, unsigned c, float *out);

/* device pointer to channel c's current iterate (own rows), for on-device consumers */
int j2p_solver_plane_ptr(j2p_solver *s, unsigned c, float **dev_ptr);

int j2p_solver_sync(j2p_solver *s);

/* average device time of the two phase kernels since the last reset, measured
 * with HIP events on the solver's stream (bench.py's roofline leg) */
int j2p_solver_kernel_times(j2p_solver *s, double *gradient_ms, double *project_ms, unsigned *samples);
int j2p_solver_enable_timing(j2p_solver *s, int every);   /* 0 = off, k = sample every k-th iteration */
/* what a bracket of two event records measures with nothing in between on the solver's stream (calibrated when timing is
 * switched on): the scale of what the event brackets add to j2p_solver_kernel_times() — reported, not subtracted */
int j2p_solver_timing_overhead(j2p_solver *s, double *event_pair_ms);

/* decode_coefficients + unbox (jpeg.c:83-92, box.c:5-19) on the device:
 * out[h*w] raster floats from block-major int16 coefficients.  Host pointers. */
int j2p_decode_plane(int device, unsigned w, unsigned h, const int16_t *data,
                     const uint16_t *quant_table, float *out);

/* 8x8 transforms on the device for n blocks of 64 floats, in place (host pointers);
 * dct8x8s / idct8x8s of ooura/dct.c:98 / :34 — exposed for the parity tests */
int j2p_dct8x8_blocks(int device, float *blocks, size_t n, int inverse);

/* The colour conversion of the PNG writer on the device (png.c:37-62: YCbCr -> RGB, clamp,
 * 8/16-bit samples, crop to w x h) including the luma +128 fix-up of jpeg2png.c:156-159,
 * from the current iterates of three (solver, channel) pairs on one device — the same solver
 * three times after a joint compute(), three solvers after `-s`.  out_host receives h*w*3
 * bytes (bits == 8) or h*w*6 bytes, big-endian samples (bits == 16), ready for libpng rows. */
typedef struct j2p_plane_ref {
        j2p_solver *solver;
        unsigned channel;
} j2p_plane_ref;
int j2p_planes_to_rgb(const j2p_plane_ref planes[3], unsigned w, unsigned h, unsigned bits, uint8_t *out_host);
/* the same for image rows [row_begin, row_end) that three solvers on one device all hold — whole-canvas or band
 * solvers (the bands of a row-tiled image convert their own rows on their own GPU); out_host receives
 * (row_end - row_begin) * w * 3 (or 6) bytes */
int j2p_planes_rows_to_rgb(const j2p_plane_ref planes[3], unsigned w, unsigned row_begin, unsigned row_end, unsigned bits,
                           uint8_t *out_host);

/* Image batches (BASELINE configs[4]; the file loop jpeg2png.c:330-337): a batch owns slots_per_device worker
 * threads per GPU, each driving one image at a time on streams of its own, so that the uploads, solves and
 * downloads of different images overlap; device memory is recycled between images (no hipMalloc / hipFree per
 * image).  A job is what decode_file() does between read_jpeg() and write_png() (jpeg2png.c:141-161). */
typedef struct j2p_batch j2p_batch;
typedef struct j2p_job {
        unsigned nchannel;                     /* 1..3 */
        j2p_plane planes[J2P_MAX_CHANNELS];    /* host arrays; must stay valid until j2p_batch_wait() returns */
        int separate;                          /* 0: one joint compute(nchannel, ...) (jpeg2png.c:144) with weight[0],
                                                  iterations[0]; 1: one compute(1, ...) per component (jpeg2png.c:147-152) */
        float weight[J2P_MAX_CHANNELS];
        float pweight[J2P_MAX_CHANNELS];
        unsigned iterations[J2P_MAX_CHANNELS];
        /* output: out_bits 8 / 16 = RGB samples (png.c:37-62 incl. the luma +128 of jpeg2png.c:156-159), cropped to
         * out_w x out_h, into out_rgb (h*w*3 or h*w*6 bytes); out_bits 0 = the float canvas planes into
         * out_planes[c] (NULL entries are skipped): W*H floats of the joint canvas, or — separate — of component
         * c's own canvas, w*w_samp x h*h_samp (compute.c:410-416 per call) */
        /* (hand over output memory that is already MAPPED — arrays reused between jobs: mapping or first-touching host memory
         * while other jobs' kernels run stalls a launch of theirs each time; 229 against 196-201 images/s at 1080p on one GPU,
         * profiles/r05_batch_prealloc.jsonl) */
        unsigned out_bits, out_w, out_h;
        uint8_t *out_rgb;
        float *out_planes[J2P_MAX_CHANNELS];
        /* optional callbacks, called on the worker thread: log rows of `n` iterations starting at `first`
         * (channel = component, or 3 for a joint solve: jpeg2png.c:143,149) and progress ticks (iterations done) */
        void (*on_rows)(void *user, unsigned channel, unsigned first, unsigned n, const j2p_log_row *rows);
        void (*on_progress)(void *user, unsigned n);
        void *user;
        /* nonzero: ONE image over SEVERAL of the batch's devices — every solve of the job is row-tiled (j2p_tiled), one
         * band per device, each band converting its own rows to RGB; for the case of fewer images than GPUs
         * (decode_file -> compute of one large image, jpeg2png.c:141-152).  Same bits either way.  The devices are
         * entries [tile_first, tile_first + tile_count) of the list given to j2p_batch_create (tile_count 0 = all of
         * them), so that a few large images can each have a share of the GPUs.  How many bands there are is the
         * library's decision: a band costs its GPU ~35 us per iteration of cross-band scheduling whatever its size
         * (profiles/r03_band_alone.jsonl), so a band gets at least 2 Mpixel per channel (tile_min_band_pixels: another
         * gate; (size_t)-1 = none: tests with small images) and at least 48 rows; an image too small for two such bands —
         * and any image when the devices cannot be tiled over (no peer access and no RCCL, or no exchange that verifies on
         * them) — is solved on ONE GPU instead */
        int tile;
        unsigned tile_first, tile_count;
        size_t tile_min_band_pixels;           /* 0 = the library's gate (2 Mpixel) */
} j2p_job;
int j2p_batch_create(j2p_batch **out, unsigned ndev, const int devices[], unsigned slots_per_device);
void j2p_batch_destroy(j2p_batch *b);                       /* finishes queued jobs first */
int j2p_batch_submit(j2p_batch *b, const j2p_job *job, int *ticket);
int j2p_batch_wait(j2p_batch *b, int ticket);               /* the job's status; its error text in j2p_last_error() */

/* test hook: n > 0: the n-th j2p_solver_run() / j2p_tiled_run() call from now on (any thread) fails with J2P_EDEVICE
 * before it queues anything — how the tests make a solve fail after its create phase (j2p_compute()'s error contract);
 * n < 0: in the |n|-th j2p_tiled_run() of a multi-band solver the LAST band fails halfway through its iterations, with
 * the other bands' work queued behind it (the teardown of a failed row-tiled run); 0 disarms. */
void j2p_debug_fail_run_after(int n);

/* test hook: compares the kernels' fast division / square root (the compiler's IEEE
 * sequences without range scaling) with `/` and sqrtf() on n pseudo-random operand pairs
 * inside the range the kernels screen for; both counters must come back 0 */
int j2p_math_selftest(int device, size_t n, unsigned seed, unsigned long long *div_mismatches,
                      unsigned long long *sqrt_mismatches);

/* test hook: the two packed square-root sequences of the gradient kernel against sqrtf() on EVERY
 * float in [2^-100, 2^127) (about 1.9e9 values); both counters must come back 0 */
int j2p_sqrt_exhaustive(int device, unsigned long long *rsq_mismatches, unsigned long long *fast_mismatches);

/* test hook: exhaustive checks of the SHORT division (one residual correction, j2p_kernels.hip.h: div_exact_recip).
 *   pass 3: denominators first .. first + count - 1 of the 2^23 floats of [1, 2), reciprocal by IEEE division (what
 *           phase B uses), each against all 2^23 numerator mantissas: the quotient against `/`  — all 2^23 denominators
 *           = 2^46 quotients, the proof by enumeration the kernels rely on (report[0] must be 0);
 *   pass 1: every norm phase A can meet: its reciprocal refined from the v_rsq_f32 seed by two Newton steps against
 *           1.f / n;  pass 2: radicands first .. of the 2^24 floats of [1, 4) x all numerators with THAT reciprocal —
 *           the form phase A would use; both fail exactly where theory says (norms with an all-ones mantissa), which
 *           is why phase A keeps the long form.
 * report[0] = number of mismatches, report[1..8] = the first offenders' bit patterns. */
int j2p_division_exhaustive(int device, int pass, unsigned first, unsigned count, unsigned long long report[9]);

#ifdef __cplusplus
}
#endif
#endif
