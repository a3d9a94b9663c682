// jpeg2png_amd — image batches over streams and GPUs (BASELINE.json configs[4]; the file loop jpeg2png.c:330-337
// and decode_file's compute calls, jpeg2png.c:141-152).
//
// A j2p_batch owns `slots_per_device` worker threads per GPU.  A job is one image — what decode_file() does
// between read_jpeg() and write_png(): one joint compute(3, ...) or three separate compute(1, ...) calls, then the
// planes handed back as floats or, converted on the device (png.c:37-62), as RGB samples.  Every worker drives its
// jobs on streams of its own, so while one slot's image is being solved the next slot's coefficients go up and a
// third one's pixels come down: H2D / solve / D2H overlap without any of them knowing about the others.  Device
// memory comes from the library's pool (one arena per solver, recycled between jobs): after the first few images
// a job performs no hipMalloc / hipFree — the device-wide synchronisation inside hipFree is what used to
// serialise concurrent compute() calls.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpeg2png_amd.h"
#include "j2p_internal.h"

extern "C" int j2p_tiled_exchange_forced(void);     // j2p_tiled.hip

namespace {

struct Job {
        j2p_job desc;
        int ticket = 0;
        int rc = J2P_OK;
        bool finished = false;
        char err[256] = "";
};

}  // namespace

struct j2p_batch {
        std::vector<int> devices;           // as given to j2p_batch_create
        std::vector<int> worker_device;
        std::vector<std::thread> workers;
        std::mutex lock;
        std::condition_variable work, finished;
        std::deque<Job *> queue;
        std::map<int, Job *> jobs;          // every job not yet collected by j2p_batch_wait
        int next_ticket = 1;
        bool quit = false;
};

namespace {

// iterations per round trip when a job wants progress or log rows: as compute() does it (compute_host.c) — one iteration
// each at first, then a sixth of the iterations done so far, never more than ~50 ms worth or kChunkMax (the row buffer):
// the bar of the default `-i 50` moves two dozen times, not twice (compute.c:449-452 ticks once per iteration)
constexpr unsigned kChunkMax = 256;
constexpr double kChunkMs = 50.;
unsigned next_chunk(unsigned done, unsigned left, std::chrono::steady_clock::time_point t_loop)
{
        unsigned chunk = done / 6;
        if(done) {
                const double per_it = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count() / done;
                const double most = per_it > 0. ? kChunkMs / per_it : (double)kChunkMax;
                if((double)chunk > most) { chunk = (unsigned)most; }
        }
        if(chunk > kChunkMax) { chunk = kChunkMax; }
        if(chunk < 1) { chunk = 1; }
        return left < chunk ? left : chunk;
}

#define JOB_TRY(expr)                                                                              \
        do {                                                                                       \
                rc = (expr);                                                                       \
                if(rc != J2P_OK) { goto out; }                                                     \
        } while(0)

unsigned gcd_u(unsigned a, unsigned b) { return b ? gcd_u(b, a % b) : a; }

// what both paths of a job check before they touch anything
int validate_job(const j2p_job &d)
{
        if(d.nchannel == 0 || d.nchannel > J2P_MAX_CHANNELS) { return j2p_fail(J2P_EINVAL, "job: nchannel must be 1..3"); }
        if(d.out_bits != 0 && d.out_bits != 8 && d.out_bits != 16) { return j2p_fail(J2P_EINVAL, "job: out_bits must be 0, 8 or 16"); }
        if(d.out_bits && (!d.out_rgb || d.nchannel != 3)) { return j2p_fail(J2P_EINVAL, "job: RGB output needs three channels and out_rgb"); }
        return J2P_OK;
}

// pixels per channel a band must at least have: the cross-band schedule costs every band ~35 us per iteration
// whatever its size (profiles/r03_band_alone.jsonl: 276 vs 245 us for a 2048-row band of a 16384-wide plane), which
// is more than a whole 1080p iteration takes on one GPU
size_t tile_min_band_pixels(const j2p_job &d)
{
        return d.tile_min_band_pixels ? (d.tile_min_band_pixels == (size_t)-1 ? 0 : d.tile_min_band_pixels) : (size_t)2 << 20;
}

// One image over several of the batch's devices (j2p_job::tile): every solve of the job becomes a j2p_tiled with band
// b on devices[b]; the solves of `-s` share their cuts so that band b of the three components meets on one GPU for
// the colour conversion.  Returns J2P_OK with *handled = false when the image should be solved on one GPU after all:
// too small for two bands (rows, or pixels per band), or the devices cannot be tiled over (the create phase failed:
// no peer access and no RCCL, or no room for the band arenas) — the caller then takes the single-solver path.
// Failures after the iterations have started stay failures.
int run_job_tiled(const j2p_job &d, const std::vector<int> &devices, bool *handled)
{
        *handled = false;
        const unsigned nsolve = d.separate ? d.nchannel : 1;
        unsigned H[J2P_MAX_CHANNELS] = {0, 0, 0}, Wc[J2P_MAX_CHANNELS] = {0, 0, 0}, align = J2P_TILE_ROWS, hmin = ~0u;
        for(unsigned c = 0; c < d.nchannel; c++) {
                const j2p_plane &p = d.planes[c];
                if(p.h_samp == 0 || p.h == 0 || p.w_samp == 0 || p.w == 0) { return j2p_fail(J2P_EINVAL, "job: channel %u: empty plane", c); }
                align = align / gcd_u(align, 8 * p.h_samp) * (8 * p.h_samp);
                const unsigned k = d.separate ? c : 0;
                if(p.h * p.h_samp > H[k]) { H[k] = p.h * p.h_samp; }
                if(p.w * p.w_samp > Wc[k]) { Wc[k] = p.w * p.w_samp; }
        }
        size_t pixels_min = ~(size_t)0;
        for(unsigned k = 0; k < nsolve; k++) {
                if(H[k] < hmin) { hmin = H[k]; }
                if((size_t)Wc[k] * H[k] < pixels_min) { pixels_min = (size_t)Wc[k] * H[k]; }
        }
        // at least three 16-row gradient segments per band
        unsigned per = 3 * J2P_TILE_ROWS;
        per = (per + align - 1) / align * align;
        unsigned nband = hmin / per;
        if(tile_min_band_pixels(d) && pixels_min / tile_min_band_pixels(d) < nband) { nband = (unsigned)(pixels_min / tile_min_band_pixels(d)); }
        if(nband > devices.size()) { nband = (unsigned)devices.size(); }
        if(nband > 32) { nband = 32; }
        if(nband < 2) { return J2P_OK; }
        // near-equal bands of the shortest canvas in units of the alignment; the last band ends where each canvas ends
        unsigned cuts[33];
        {
                const unsigned units = hmin / align;          // whole units; the remainder goes to the last band
                unsigned start = 0;
                for(unsigned b = 0; b < nband; b++) {
                        cuts[b] = start * align;
                        start += units / nband + (b < units % nband ? 1 : 0);
                }
        }
        j2p_tiled *t[J2P_MAX_CHANNELS] = {nullptr, nullptr, nullptr};
        unsigned its[J2P_MAX_CHANNELS] = {0, 0, 0}, done[J2P_MAX_CHANNELS] = {0, 0, 0};
        int rc = J2P_OK;
        const bool chunked = d.on_rows || d.on_progress;
        for(unsigned k = 0; k < nsolve; k++) {
                cuts[nband] = H[k];
                its[k] = d.iterations[k];
                if(d.separate) {
                        rc = j2p_tiled_create(&t[k], nband, devices.data(), cuts, 1, &d.planes[k], d.weight[k], &d.pweight[k], its[k]);
                } else {
                        rc = j2p_tiled_create(&t[k], nband, devices.data(), cuts, d.nchannel, d.planes, d.weight[0], d.pweight, its[k]);
                }
                if((rc == J2P_EDEVICE || rc == J2P_ENOMEM) && !j2p_tiled_exchange_forced()) {
                        // these GPUs cannot be tiled over (no peer access and no RCCL, no exchange that verifies on them, or
                        // no room for the band arenas): the image is solved on one of them, as it would have been without
                        // `tile`.  Nothing has run yet.  (An exchange NAMED through J2P_TILED_EXCHANGE / J2P_TILED_WAIT that
                        // cannot be had is an error: the caller asked for that one.)
                        fprintf(stderr, "jpeg2png_amd: not row-tiling this image over %u GPUs (%s); solving it on one\n", nband, j2p_last_error());
                        rc = J2P_OK;
                        goto out;
                }
                if(rc != J2P_OK) { goto out; }
        }
        *handled = true;
        if(!chunked) {
                for(unsigned k = 0; k < nsolve; k++) { JOB_TRY(j2p_tiled_run(t[k], its[k], nullptr)); }
        } else {
                j2p_log_row rows[kChunkMax];
                const auto t_loop = std::chrono::steady_clock::now();
                for(;;) {
                        bool any = false;
                        for(unsigned k = 0; k < nsolve; k++) {
                                const unsigned left = its[k] - done[k];
                                const unsigned step = next_chunk(done[k], left, t_loop);
                                if(!step) { continue; }
                                any = true;
                                JOB_TRY(j2p_tiled_run(t[k], step, d.on_rows ? rows : nullptr));
                                if(d.on_rows) { d.on_rows(d.user, d.separate ? k : 3u, done[k], step, rows); }
                                else { JOB_TRY(j2p_tiled_sync(t[k])); }
                                if(d.on_progress) { d.on_progress(d.user, step); }
                                done[k] += step;
                        }
                        if(!any) { break; }
                }
        }
        if(d.out_bits) {
                const size_t row_bytes = (size_t)d.out_w * (d.out_bits == 8 ? 3 : 6);
                for(unsigned b = 0; b < nband; b++) {
                        const unsigned y0 = cuts[b];
                        unsigned y1 = b + 1 < nband ? cuts[b + 1] : d.out_h;
                        if(y1 > d.out_h) { y1 = d.out_h; }
                        if(y0 >= y1) { continue; }                       // band below the image (canvas padding only)
                        j2p_plane_ref ref[3];
                        for(unsigned c = 0; c < 3; c++) {
                                j2p_solver *bs = nullptr;
                                JOB_TRY(j2p_tiled_band(d.separate ? t[c] : t[0], b, nullptr, nullptr, nullptr, &bs));
                                ref[c].solver = bs;
                                ref[c].channel = d.separate ? 0 : c;
                        }
                        JOB_TRY(j2p_planes_rows_to_rgb(ref, d.out_w, y0, y1, d.out_bits, d.out_rgb + (size_t)y0 * row_bytes));
                }
        } else {
                for(unsigned c = 0; c < d.nchannel; c++) {
                        if(!d.out_planes[c]) { continue; }
                        JOB_TRY(j2p_tiled_download(d.separate ? t[c] : t[0], d.separate ? 0 : c, d.out_planes[c]));
                }
        }
out:
        for(unsigned k = 0; k < J2P_MAX_CHANNELS; k++) {
                if(t[k]) { j2p_tiled_destroy(t[k]); }
        }
        return rc;
}

int run_job(const j2p_job &d, int device)
{
        j2p_solver *s[J2P_MAX_CHANNELS] = {nullptr, nullptr, nullptr};
        unsigned nsolver = 0;
        unsigned its[J2P_MAX_CHANNELS] = {0, 0, 0}, done[J2P_MAX_CHANNELS] = {0, 0, 0};
        int rc = J2P_OK;
        const j2p_band whole = {0, 0};
        const bool chunked = d.on_rows || d.on_progress;
        if(d.separate) {
                // jpeg2png.c:147-152: one compute(1, ...) per component, each with its own weight and iteration count
                nsolver = d.nchannel;
                for(unsigned c = 0; c < nsolver; c++) {
                        its[c] = d.iterations[c];
                        JOB_TRY(j2p_solver_create(&s[c], device, nullptr, 1, &d.planes[c], d.weight[c], &d.pweight[c], its[c], whole, 0));
                }
        } else {
                // jpeg2png.c:144: compute(3, ...) with the first weight and iteration count
                nsolver = 1;
                its[0] = d.iterations[0];
                JOB_TRY(j2p_solver_create(&s[0], device, nullptr, d.nchannel, d.planes, d.weight[0], d.pweight, its[0], whole, 0));
        }
        if(!chunked) {
                for(unsigned c = 0; c < nsolver; c++) { JOB_TRY(j2p_solver_run(s[c], its[c], nullptr)); }
        } else {
                // compute.c:427-453 in chunks so that the caller's bar and CSV keep moving; without log rows the
                // chunks of the (up to three) solvers are issued back to back and overlap on the GPU
                j2p_log_row rows[kChunkMax];
                const auto t_loop = std::chrono::steady_clock::now();
                for(;;) {
                        unsigned step[J2P_MAX_CHANNELS] = {0, 0, 0};
                        bool any = false;
                        for(unsigned c = 0; c < nsolver; c++) {
                                const unsigned left = its[c] - done[c];
                                step[c] = next_chunk(done[c], left, t_loop);
                                if(!step[c]) { continue; }
                                any = true;
                                JOB_TRY(j2p_solver_run(s[c], step[c], d.on_rows ? rows : nullptr));
                                if(d.on_rows) { d.on_rows(d.user, d.separate ? c : 3u, done[c], step[c], rows); }   // channel 3 = joint, jpeg2png.c:143
                        }
                        if(!any) { break; }
                        for(unsigned c = 0; c < nsolver; c++) {
                                if(!step[c]) { continue; }
                                if(!d.on_rows) { JOB_TRY(j2p_solver_sync(s[c])); }
                                if(d.on_progress) { d.on_progress(d.user, step[c]); }
                                done[c] += step[c];
                        }
                }
        }
        if(d.out_bits) {
                j2p_plane_ref ref[3];
                for(unsigned c = 0; c < 3; c++) {
                        ref[c].solver = d.separate ? s[c] : s[0];
                        ref[c].channel = d.separate ? 0 : c;
                }
                JOB_TRY(j2p_planes_to_rgb(ref, d.out_w, d.out_h, d.out_bits, d.out_rgb));
        } else {
                for(unsigned c = 0; c < d.nchannel; c++) {
                        if(!d.out_planes[c]) { continue; }
                        JOB_TRY(j2p_solver_download(d.separate ? s[c] : s[0], d.separate ? 0 : c, d.out_planes[c]));
                }
        }
out:
        for(unsigned c = 0; c < J2P_MAX_CHANNELS; c++) {
                if(s[c]) { j2p_solver_destroy(s[c]); }
        }
        return rc;
}

void worker_main(j2p_batch *b, int device)
{
        (void)hipSetDevice(device);
        for(;;) {
                Job *job = nullptr;
                {
                        std::unique_lock<std::mutex> g(b->lock);
                        b->work.wait(g, [&] { return b->quit || !b->queue.empty(); });
                        if(b->queue.empty()) { return; }          // quit, and nothing left to do
                        job = b->queue.front();
                        b->queue.pop_front();
                }
                int rc = validate_job(job->desc);
                bool tiled = false;
                int single = device;
                if(rc == J2P_OK && job->desc.tile) {
                        // the job's share of the batch's devices, in the order given (a tiled job puts one band on each)
                        const size_t first = job->desc.tile_first, nall = b->devices.size();
                        const size_t count = job->desc.tile_count ? job->desc.tile_count : nall;
                        if(first >= nall || count > nall - first) { rc = j2p_fail(J2P_EINVAL, "job: tile devices [%zu, %zu) of %zu", first, first + count, nall); }
                        else {
                                const std::vector<int> share(b->devices.begin() + (ptrdiff_t)first, b->devices.begin() + (ptrdiff_t)(first + count));
                                if(job->desc.tile_count) { single = share[0]; }     // untiled after all: on a GPU of its share
                                if(share.size() > 1) { rc = run_job_tiled(job->desc, share, &tiled); }
                        }
                }
                if(rc == J2P_OK && !tiled) {
                        if(single != device) { (void)hipSetDevice(single); }
                        rc = run_job(job->desc, single);
                        if(single != device) { (void)hipSetDevice(device); }
                }
                {
                        std::lock_guard<std::mutex> g(b->lock);
                        job->rc = rc;
                        if(rc != J2P_OK) { strncpy(job->err, j2p_last_error(), sizeof(job->err) - 1); }
                        job->finished = true;
                }
                b->finished.notify_all();
        }
}

}  // namespace

extern "C" {

int j2p_batch_create(j2p_batch **out, unsigned ndev, const int devices[], unsigned slots_per_device)
{
        if(!out || !devices) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        *out = nullptr;
        if(ndev == 0 || ndev > 64 || slots_per_device == 0 || slots_per_device > 16) {
                return j2p_fail(J2P_EINVAL, "batch: 1..64 devices, 1..16 slots per device");
        }
        int have = 0;
        if(hipGetDeviceCount(&have) != hipSuccess || have <= 0) {
                return j2p_fail(J2P_EDEVICE, "no HIP device available: the jpeg2png_amd solver has no CPU fallback");
        }
        for(unsigned i = 0; i < ndev; i++) {
                if(devices[i] < 0 || devices[i] >= have) { return j2p_fail(J2P_EINVAL, "device %d out of range (0..%d)", devices[i], have - 1); }
        }
        j2p_batch *b = new(std::nothrow) j2p_batch();
        if(!b) { return j2p_fail(J2P_ENOMEM, "host allocation failed"); }
        b->devices.assign(devices, devices + ndev);
        // slot-major order: the first ndev workers sit on different GPUs, so few images spread out first
        for(unsigned k = 0; k < slots_per_device; k++) {
                for(unsigned i = 0; i < ndev; i++) { b->worker_device.push_back(devices[i]); }
        }
        for(int dev : b->worker_device) { b->workers.emplace_back(worker_main, b, dev); }
        *out = b;
        return J2P_OK;
}

void j2p_batch_destroy(j2p_batch *b)
{
        if(!b) { return; }
        {
                std::lock_guard<std::mutex> g(b->lock);
                b->quit = true;
        }
        b->work.notify_all();
        for(std::thread &t : b->workers) { t.join(); }          // queued jobs are finished first
        for(auto &kv : b->jobs) { delete kv.second; }
        delete b;
        j2p_pool_trim();        // the arenas the images recycled go back to the device with the batch
}

int j2p_batch_submit(j2p_batch *b, const j2p_job *job, int *ticket)
{
        if(!b || !job || !ticket) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        Job *j = new(std::nothrow) Job();
        if(!j) { return j2p_fail(J2P_ENOMEM, "host allocation failed"); }
        j->desc = *job;
        {
                std::lock_guard<std::mutex> g(b->lock);
                if(b->quit) { delete j; return j2p_fail(J2P_ESTATE, "batch is shutting down"); }
                j->ticket = b->next_ticket++;
                b->jobs[j->ticket] = j;
                b->queue.push_back(j);
                *ticket = j->ticket;
        }
        b->work.notify_one();
        return J2P_OK;
}

int j2p_batch_wait(j2p_batch *b, int ticket)
{
        if(!b) { return j2p_fail(J2P_EINVAL, "batch is NULL"); }
        Job *j = nullptr;
        {
                std::unique_lock<std::mutex> g(b->lock);
                auto it = b->jobs.find(ticket);
                if(it == b->jobs.end()) { return j2p_fail(J2P_EINVAL, "unknown ticket %d", ticket); }
                j = it->second;
                b->finished.wait(g, [&] { return j->finished; });
                b->jobs.erase(it);
        }
        const int rc = j->rc;
        if(rc != J2P_OK) { j2p_fail(rc, "%s", j->err); }
        delete j;
        return rc;
}

}  // extern "C"
