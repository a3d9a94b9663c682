// jpeg2png_amd — one plane set row-tiled over several GPUs from ONE process (BASELINE.json configs[3]).
//
// The canvas is cut into contiguous row bands (multiples of 8*h_samp and of the 16-row gradient tile, so that no
// DCT block and no gradient tile straddles two bands); band i is a j2p_solver on devices[i] — device ids may
// repeat, several bands then share a GPU.  One host thread per band issues that band's launches, so the eight
// GPUs of a node are fed in parallel and there is no interpreter in the iteration loop.  Per iteration the bands
// meet twice (reference loop compute.c:427-453): ||g|| (compute.c:200-207) needs every band's
// gradient, and the next gradient reaches 2 rows into the neighbouring bands (TGV2, compute.c:137-143,165-183).
// Three ways of carrying those two exchanges — same arithmetic, same bits, chosen at create time
// (J2P_TILED_EXCHANGE=direct|copy|rccl; default: direct where every GPU can write every other's memory, else rccl):
//
//   direct  The exchanges ride on the two phase kernels; a band's iteration is TWO launches and all traffic between
//           GPUs is posted writes (j2p_solver_link_bands).  k_gradient's last-arriving wavefront of every 16-row tile
//           row stores that row's sum of g^2 into EVERY band's copy of the global array; k_project reduces ||g|| from
//           its own band's copy (the same fixed tree over the same global array as the single-GPU solver, so the
//           result does not depend on the number of bands) and stores the band's first / last two rows of the new
//           iterate also into the neighbours' halo rows.  Ordering by HIP events only: gradient(k) waits for the
//           neighbours' projection(k - 1); projection(k) waits for EVERY band's gradient(k) — two cross-stream waits
//           per band and iteration, at most two sequential cross-device hops.
//   copy    Round 3's schedule, kept as the cross-check of `direct` on real multi-GPU hardware (bench.py --gpus N times
//           both and compares their results): the neighbours' edge rows are PULLED by a small copy kernel in front of
//           the gradient launch, and ONE band — the root — waits for the others' gradient events, reduces all bands'
//           row sums (read in place) and stores the float norm into every band's norm word; the others wait for
//           that event (experiments build, J2P_TILED_NORM=all: every band reduces for itself).  Four launches per band and iteration, three
//           sequential hops.  Also what canvases taller than 16384 rows use (k_project's in-kernel tree holds 1024 rows).
//   rccl    GPUs without peer access, or on request: ncclAllGather of the bands' row sums between the phases and one
//           ncclGroupStart/End of ncclSend/ncclRecv for the 2 + 2 edge rows behind the projection, on the band's own
//           stream, one communicator per band (ncclCommInitAll); librccl is dlopen()ed, so the library loads and runs
//           without it.  What north_star names.
//
// Host side: a band thread that needs another band's event sleeps on a condition variable until that event has
// been RECORDED (hipStreamWaitEvent on an event not yet recorded would be a no-op); it never waits for the GPU.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/resource.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <new>
#include <thread>
#include <vector>
#include <stdint.h>
#include <string.h>

#include "jpeg2png_amd.h"
#include "j2p_internal.h"

namespace {

constexpr unsigned kLogCols = 2 + J2P_MAX_CHANNELS;
constexpr unsigned kTreeRowsInProject = 1024;   // tile rows k_project's in-kernel tree handles (kWaveTreeMax)

enum Exchange { kDirect = 0, kCopy = 1, kRccl = 2 };
const char *const kExchangeName[3] = {"direct", "copy", "rccl"};
// `direct`: how a band's projection learns that EVERY band's gradient launch has finished (J2P_TILED_WAIT).  On one GPU a
// stream pays ~13 us per event of another stream it waits for between two of its kernels, whatever the kernels touch
// (tools/ubench/event_waits.hip, profiles/r04_event_waits.json: 7 waits +94 us, through a collecting stream +62);
// what a wait for another GPU's event costs cannot be measured on this pool's one-GPU boxes, so all three exist and
// bench.py --gpus N times them:
//   all        the band's stream waits for the other N - 1 gradient events itself (one hop, N - 1 barrier packets)
//   root       band 0 waits for them and records one event, the others wait for that (two hops, one packet each)
//   collector  a helper stream of the band waits for the N - 1 events and records one event the band's stream waits for
//   counter    no events at all: k_gradient itself adds one count to every band's counter (coherent pinned host memory)
//              when the band's last tile row has been pushed, and the band's stream waits for ONE value —
//              hipStreamWaitValue64(counter >= N (k + 1)) — in front of projection(k); behind it the stream writes the
//              iteration number into the band's own flag (hipStreamWriteValue64), which the neighbours' streams wait for
//              in front of gradient(k + 1).  One wait whatever N (on one GPU: +15 us at n = 7 against +94 for events,
//              profiles/r04_wait_value.json), and the host threads never sleep on each other.
enum WaitMode { kWaitAll = 0, kWaitRoot = 1, kWaitCollector = 2, kWaitCounter = 3 };
const char *const kWaitName[4] = {"all", "root", "collector", "counter"};

// ---------------------------------------------------------------------------------------------------------------
// librccl through dlopen: the C host reaches RCCL without linking against it (the library must load on hosts that
// have none), and uses the copy the process already has (torch's bundled one under Python) when there is one.
// Types as in rccl.h: ncclComm_t is a pointer, ncclResult_t 0 = success, ncclFloat32 = 7, ncclFloat64 = 8.
// ---------------------------------------------------------------------------------------------------------------
struct Rccl {
        void *handle = nullptr;
        int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
        int (*CommDestroy)(void *comm) = nullptr;
        int (*CommAbort)(void *comm) = nullptr;             // optional: how a failed run leaves its queued collectives
        int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st) = nullptr;
        int (*Broadcast)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t st) = nullptr;
        int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
        int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
        int (*GroupStart)() = nullptr;
        int (*GroupEnd)() = nullptr;
        const char *(*GetErrorString)(int) = nullptr;
        int (*GetVersion)(int *version) = nullptr;          // optional
        char why[256] = "";
};
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8;

Rccl *rccl_load()
{
        static std::mutex lock;
        static Rccl *lib = nullptr;
        static bool tried = false;
        std::lock_guard<std::mutex> g(lock);
        if(tried) { return lib; }
        tried = true;
        Rccl *r = new(std::nothrow) Rccl();
        if(!r) { return nullptr; }
        const char *env = getenv("J2P_RCCL_LIBRARY");
        const char *names[] = {env && *env ? env : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        // a copy already in the process first (RTLD_NOLOAD), then by name
        for(int pass = 0; pass < 2 && !r->handle; pass++) {
                for(const char *n : names) {
                        if(!n) { continue; }
                        r->handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                        if(r->handle) { break; }
                }
        }
        if(!r->handle) {
                snprintf(r->why, sizeof(r->why), "librccl not found (%s)", dlerror() ? "dlopen failed; J2P_RCCL_LIBRARY names another" : "no such library");
                lib = r;
                return lib;
        }
#define RCCL_SYM(field, name)                                                                      \
        do {                                                                                       \
                *reinterpret_cast<void **>(&r->field) = dlsym(r->handle, name);                    \
                if(!r->field && !r->why[0]) { snprintf(r->why, sizeof(r->why), "librccl lacks %s", name); } \
        } while(0)
        RCCL_SYM(CommInitAll, "ncclCommInitAll");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        *reinterpret_cast<void **>(&r->CommAbort) = dlsym(r->handle, "ncclCommAbort");
        RCCL_SYM(AllGather, "ncclAllGather");
        RCCL_SYM(Broadcast, "ncclBroadcast");
        RCCL_SYM(Send, "ncclSend");
        RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
        *reinterpret_cast<void **>(&r->GetVersion) = dlsym(r->handle, "ncclGetVersion");
        lib = r;
        return lib;
}
bool rccl_usable(const Rccl *r) { return r && r->handle && !r->why[0]; }

struct Band {
        int device = 0;
        j2p_solver *solver = nullptr;
        hipStream_t stream = nullptr;
        unsigned row0 = 0, row1 = 0;
        hipEvent_t ev_grad[2] = {nullptr, nullptr};    // behind the gradient phase of iteration it (slot it & 1)
        hipEvent_t ev_edge[2] = {nullptr, nullptr};    // behind the projection of iteration it
        hipEvent_t ev_norm[2] = {nullptr, nullptr};    // root band only: behind the norm (`copy`) / every band's gradient (`direct`, wait root) of iteration it
        hipStream_t collector = nullptr;               // `direct`, wait collector: the stream that waits for the other bands' gradient events
        hipEvent_t ev_all[2] = {nullptr, nullptr};     // ... and what it records behind them
        // iterations whose event has been recorded (guarded by j2p_tiled::seq_lock)
        uint64_t grad_recorded = 0, edge_recorded = 0, norm_recorded = 0;
        j2p_exchange rows[2];                  // halo / edge row addresses of x buffer 0 and 1
        const double *rowsum[2] = {nullptr, nullptr};   // `copy`: level-1 sums of even / odd iterations
        double *global_rows[2] = {nullptr, nullptr};    // the band's copies of the global [tile row][channel] array
        double *rowsum_local = nullptr;        // `rccl`: what the band contributes to the all-gather
        float *norm = nullptr;                 // the band solver's norm word(s), [channel]
        unsigned first_tr = 0, ntr = 0;
        void *comm = nullptr;                  // `rccl`: this band's communicator (rank = band index)
        double *log_dev = nullptr;             // the band's {tv, tv2, prob[3]} of the iteration just finished
        double *log_host = nullptr;            // pinned: [chunk][kLogCols]
        unsigned log_cap = 0;
        std::thread thread;
        double cpu_seconds = 0.;               // user + system time of the band thread inside band_iterations
        int rc = J2P_OK;
        char err[256] = "";
};

}  // namespace

struct j2p_tiled {
        unsigned nband = 0, nch = 0, W = 0, H = 0;
        float weight = 0.f, pweight[J2P_MAX_CHANNELS] = {0.f, 0.f, 0.f};
        std::vector<Band *> bands;
        uint64_t iter = 0;                     // iterations issued so far
        double carried[J2P_MAX_CHANNELS] = {0., 0., 0.};
        bool carried_valid = true;             // false after iterations run without logging (their prob sums were not kept)
        bool logging = false;                  // the band solvers currently run their logging kernels
        Exchange exchange = kDirect;
        WaitMode wait = kWaitAll;
        bool threaded = false;                 // band threads exist (every run but the plain one-band one)
        bool norm_by_root = true;              // `copy`: one band reduces ||g|| for all (default); false: every band for itself
        bool self_neighbours = false;          // test hook (J2P_TILED_SELF_NEIGHBOURS=1, one band, rccl): the band exchanges with itself
        bool equal_counts = true;              // `rccl`: every band has as many tile rows (one ncclAllGather; else grouped broadcasts)
        unsigned root = 0;
        Rccl *rccl = nullptr;
        // wait counter: one 64-byte line per band and kind in coherent pinned host memory
        unsigned long long *signals = nullptr;          // [2 * nband][8]: gradient counts, then projection flags
        unsigned long long *grad_count(unsigned b) const { return signals + 8 * (size_t)b; }
        unsigned long long *edge_flag(unsigned b) const { return signals + 8 * ((size_t)nband + b); }
        // command hand-over to the band threads
        std::mutex lock;
        std::condition_variable wake, done;
        uint64_t generation = 0;
        unsigned cmd_n = 0;
        bool cmd_log = false, quit = false;
        unsigned finished = 0;
        std::atomic<bool> abort{false};
        // "band p has recorded its event of iteration it": sequence numbers in Band, one lock and one condition
        // variable for all of them (a band thread sleeps here; nothing spins)
        std::mutex seq_lock;
        std::condition_variable seq_cv;
};

namespace {

#define BAND_TRY(expr)                                                                             \
        do {                                                                                       \
                int rc_ = (expr);                                                                  \
                if(rc_ != J2P_OK) { return rc_; }                                                  \
        } while(0)
#define BAND_HIP(expr)                                                                             \
        do {                                                                                       \
                hipError_t e_ = (expr);                                                            \
                if(e_ != hipSuccess) { return j2p_fail(J2P_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
        } while(0)
#define BAND_NCCL(t, expr)                                                                         \
        do {                                                                                       \
                int e_ = (expr);                                                                   \
                if(e_ != 0) { return j2p_fail(J2P_EDEVICE, "%s failed: %s", #expr, (t)->rccl->GetErrorString(e_)); } \
        } while(0)

enum Which { kGrad, kEdge, kNorm };

uint64_t &seq_of(Band *p, Which w) { return w == kGrad ? p->grad_recorded : (w == kEdge ? p->edge_recorded : p->norm_recorded); }

// record band me's event of iteration `it` on its stream and tell the waiting band threads
int record(j2p_tiled *t, Band *me, Which w, uint64_t it)
{
        hipEvent_t ev = w == kGrad ? me->ev_grad[it & 1] : (w == kEdge ? me->ev_edge[it & 1] : me->ev_norm[it & 1]);
        BAND_HIP(hipEventRecord(ev, me->stream));
        {
                std::lock_guard<std::mutex> g(t->seq_lock);
                seq_of(me, w) = it + 1;
        }
        t->seq_cv.notify_all();
        return J2P_OK;
}

// sleep (host) until band p has recorded its event of iteration `it`, then make `me`'s stream (or `on`) wait for it
int wait_for(j2p_tiled *t, Band *me, Band *p, Which w, uint64_t it, hipStream_t on = nullptr)
{
        {
                std::unique_lock<std::mutex> g(t->seq_lock);
                t->seq_cv.wait(g, [&] { return seq_of(p, w) > it || t->abort.load(std::memory_order_relaxed); });
                if(seq_of(p, w) <= it) { return j2p_fail(J2P_ESTATE, "another band failed"); }
        }
        hipEvent_t ev = w == kGrad ? p->ev_grad[it & 1] : (w == kEdge ? p->ev_edge[it & 1] : p->ev_norm[it & 1]);
        BAND_HIP(hipStreamWaitEvent(on ? on : me->stream, ev, 0));
        return J2P_OK;
}

// `copy`: the neighbours' edge rows of the iterate produced by iteration `it` into this band's halo rows
int pull_halos(j2p_tiled *t, unsigned b, uint64_t it)
{
        Band *me = t->bands[b];
        Band *up = b > 0 ? t->bands[b - 1] : nullptr, *down = b + 1 < t->nband ? t->bands[b + 1] : nullptr;
        if(!up && !down) { return J2P_OK; }
        const int buf = (int)((it + 1) & 1);
        float *dst[2 * J2P_MAX_CHANNELS];
        const float *src[2 * J2P_MAX_CHANNELS];
        unsigned n = 0;
        if(up) { BAND_TRY(wait_for(t, me, up, kEdge, it)); }
        if(down) { BAND_TRY(wait_for(t, me, down, kEdge, it)); }
        for(unsigned c = 0; c < t->nch; c++) {
                if(up) { dst[n] = me->rows[buf].recv_top[c]; src[n++] = up->rows[buf].send_bottom[c]; }
                if(down) { dst[n] = me->rows[buf].recv_bottom[c]; src[n++] = down->rows[buf].send_top[c]; }
        }
        return j2p_solver_copy_rows(me->solver, n, dst, src, me->rows[buf].halo_floats);
}

// `rccl`: every band's row sums of this iteration into every band's global array (between the two phases)
int rccl_gather_rowsums(j2p_tiled *t, unsigned b)
{
        Band *me = t->bands[b];
        const Rccl *r = t->rccl;
        if(t->equal_counts) {
                BAND_NCCL(t, r->AllGather(me->rowsum_local, me->global_rows[0], (size_t)me->ntr * t->nch, kNcclFloat64, me->comm, me->stream));
                return J2P_OK;
        }
        // bands of different heights: one broadcast per band, grouped into one launch
        BAND_NCCL(t, r->GroupStart());
        for(unsigned p = 0; p < t->nband; p++) {
                const Band *pb = t->bands[p];
                BAND_NCCL(t, r->Broadcast(me->rowsum_local, me->global_rows[0] + (size_t)pb->first_tr * t->nch, (size_t)pb->ntr * t->nch,
                                          kNcclFloat64, (int)p, me->comm, me->stream));
        }
        BAND_NCCL(t, r->GroupEnd());
        return J2P_OK;
}

// `rccl`: the band's edge rows of the iterate produced by iteration `it` to the neighbours, theirs into its halo rows
int rccl_exchange_halos(j2p_tiled *t, unsigned b, uint64_t it)
{
        Band *me = t->bands[b];
        const Rccl *r = t->rccl;
        int up = b > 0 ? (int)b - 1 : -1, down = b + 1 < t->nband ? (int)b + 1 : -1;
        if(t->self_neighbours) { up = down = (int)b; }
        if(up < 0 && down < 0) { return J2P_OK; }
        const j2p_exchange &e = me->rows[(it + 1) & 1];
        BAND_NCCL(t, r->GroupStart());
        for(unsigned c = 0; c < t->nch; c++) {
                if(up >= 0) {
                        BAND_NCCL(t, r->Send(e.send_top[c], e.halo_floats, kNcclFloat32, up, me->comm, me->stream));
                        BAND_NCCL(t, r->Recv(e.recv_top[c], e.halo_floats, kNcclFloat32, up, me->comm, me->stream));
                }
                if(down >= 0) {
                        BAND_NCCL(t, r->Send(e.send_bottom[c], e.halo_floats, kNcclFloat32, down, me->comm, me->stream));
                        BAND_NCCL(t, r->Recv(e.recv_bottom[c], e.halo_floats, kNcclFloat32, down, me->comm, me->stream));
                }
        }
        BAND_NCCL(t, r->GroupEnd());
        return J2P_OK;
}

int band_iterations(j2p_tiled *t, unsigned b, unsigned n, bool log)
{
        Band *me = t->bands[b];
        BAND_HIP(hipSetDevice(me->device));
        Band *up = b > 0 ? t->bands[b - 1] : nullptr, *down = b + 1 < t->nband ? t->bands[b + 1] : nullptr;
        // `copy`: a band may run ahead of the others by up to one gradient phase, so the row sums alternate between two
        // buffers: iteration it + 2 overwrites those of iteration it only after every reader's norm(it) has run
        const double *rowsums[2][32];
        unsigned first[32], count[32];
        float *norm_out[32];
        for(unsigned p = 0; p < t->nband; p++) {
                rowsums[0][p] = t->bands[p]->rowsum[0];
                rowsums[1][p] = t->bands[p]->rowsum[1];
                first[p] = t->bands[p]->first_tr;
                count[p] = t->bands[p]->ntr;
                norm_out[p] = t->bands[p]->norm;
        }
        const bool by_root = t->norm_by_root;
        Band *root = t->bands[t->root];
        const bool doomed = b + 1 == t->nband && j2p_injected_band_failure();      // (test hook, j2p_debug_fail_run_after(-n))
        for(unsigned i = 0; i < n; i++) {
                const uint64_t it = t->iter + i;
                if(doomed && i == n / 2) { return j2p_fail(J2P_EDEVICE, "injected band failure (j2p_debug_fail_run_after)"); }
                switch(t->exchange) {
                case kDirect:
                        if(t->wait == kWaitCounter) {
                                // no events: values in host memory every GPU sees (see WaitMode)
                                if(it > 0) {
                                        if(up) { BAND_HIP(hipStreamWaitValue64(me->stream, t->edge_flag(b - 1), it, hipStreamWaitValueGte, ~0ull)); }
                                        if(down) { BAND_HIP(hipStreamWaitValue64(me->stream, t->edge_flag(b + 1), it, hipStreamWaitValueGte, ~0ull)); }
                                }
                                BAND_TRY(j2p_solver_phase_gradient(me->solver));
                                BAND_HIP(hipStreamWaitValue64(me->stream, t->grad_count(b), (uint64_t)t->nband * (it + 1), hipStreamWaitValueGte, ~0ull));
                                BAND_TRY(j2p_solver_phase_project(me->solver));
                                BAND_HIP(hipStreamWriteValue64(me->stream, t->edge_flag(b), it + 1, 0));
                                break;
                        }
                        // ---- phase A behind the neighbours' projection of the previous iteration (their edge rows are
                        // in this band's halo rows when that launch has finished) ----
                        if(it > 0) {
                                if(up) { BAND_TRY(wait_for(t, me, up, kEdge, it - 1)); }
                                if(down) { BAND_TRY(wait_for(t, me, down, kEdge, it - 1)); }
                        }
                        BAND_TRY(j2p_solver_phase_gradient(me->solver));
                        BAND_TRY(record(t, me, kGrad, it));
                        // ---- phase B behind EVERY band's gradient launch (their row sums are in this band's global
                        // array; nobody still reads the halo rows this band's projection is about to overwrite) ----
                        if(t->wait == kWaitAll || (t->wait == kWaitRoot && me == root)) {
                                for(unsigned p = 0; p < t->nband; p++) {
                                        if(p != b) { BAND_TRY(wait_for(t, me, t->bands[p], kGrad, it)); }
                                }
                                if(t->wait == kWaitRoot) { BAND_TRY(record(t, me, kNorm, it)); }      // "every gradient launch has finished"
                        } else if(t->wait == kWaitRoot) {
                                BAND_TRY(wait_for(t, me, root, kNorm, it));
                        } else {
                                for(unsigned p = 0; p < t->nband; p++) {
                                        if(p != b) { BAND_TRY(wait_for(t, me, t->bands[p], kGrad, it, me->collector)); }
                                }
                                BAND_HIP(hipEventRecord(me->ev_all[it & 1], me->collector));
                                BAND_HIP(hipStreamWaitEvent(me->stream, me->ev_all[it & 1], 0));
                        }
                        BAND_TRY(j2p_solver_phase_project(me->solver));
                        BAND_TRY(record(t, me, kEdge, it));
                        break;
                case kCopy:
                        if(it > 0) { BAND_TRY(pull_halos(t, b, it - 1)); }
                        BAND_TRY(j2p_solver_phase_gradient(me->solver));
                        BAND_TRY(record(t, me, kGrad, it));
                        // ---- the global norm: every band's row sums, one fixed tree ----
                        if(!by_root) {
                                for(unsigned p = 0; p < t->nband; p++) {
                                        if(p != b) { BAND_TRY(wait_for(t, me, t->bands[p], kGrad, it)); }
                                }
                                BAND_TRY(j2p_solver_norm_from_bands(me->solver, t->nband, rowsums[it & 1], first, count, 0, nullptr));
                        } else if(me == root) {
                                for(unsigned p = 0; p < t->nband; p++) {
                                        if(p != b) { BAND_TRY(wait_for(t, me, t->bands[p], kGrad, it)); }
                                }
                                // ... and the result goes into every band's own norm word (peer stores)
                                BAND_TRY(j2p_solver_norm_from_bands(me->solver, t->nband, rowsums[it & 1], first, count, t->nband, norm_out));
                                BAND_TRY(record(t, me, kNorm, it));
                        } else {
                                BAND_TRY(wait_for(t, me, root, kNorm, it));
                                BAND_TRY(j2p_solver_norm_external(me->solver));
                        }
                        BAND_TRY(j2p_solver_phase_project(me->solver));
                        BAND_TRY(record(t, me, kEdge, it));
                        break;
                case kRccl:
                        // everything in the band's own stream: gradient, all-gather, projection, send/recv of the edge rows
                        BAND_TRY(j2p_solver_phase_gradient(me->solver));
                        BAND_TRY(rccl_gather_rowsums(t, b));
                        BAND_TRY(j2p_solver_phase_project(me->solver));
                        BAND_TRY(rccl_exchange_halos(t, b, it));
                        break;
                }
                if(log) {
                        BAND_HIP(hipMemcpyAsync(me->log_host + (size_t)i * kLogCols, me->log_dev, kLogCols * sizeof(double),
                                                hipMemcpyDeviceToHost, me->stream));
                }
        }
        return J2P_OK;
}

double thread_cpu_seconds()
{
        struct rusage u;
        if(getrusage(RUSAGE_THREAD, &u) != 0) { return 0.; }
        return (double)u.ru_utime.tv_sec + (double)u.ru_stime.tv_sec + 1e-6 * ((double)u.ru_utime.tv_usec + (double)u.ru_stime.tv_usec);
}

// wait counter: a band that failed never counts or writes its flag, and the other bands' streams would sit in their
// hipStreamWaitValue64 for ever — and with them every j2p_tiled_sync() and the destroy.  Let every waiter through (what
// they then compute is thrown away: the solver is unusable once `abort` is set).
void release_value_waiters(j2p_tiled *t)
{
        if(!t->signals) { return; }
        for(unsigned i = 0; i < 2 * t->nband; i++) {
                __atomic_store_n(t->signals + 8 * (size_t)i, (unsigned long long)1 << 62, __ATOMIC_SEQ_CST);
        }
}

void band_main(j2p_tiled *t, unsigned b)
{
        uint64_t seen = 0;
        for(;;) {
                unsigned n;
                bool log;
                {
                        std::unique_lock<std::mutex> g(t->lock);
                        t->wake.wait(g, [&] { return t->quit || t->generation != seen; });
                        if(t->quit) { return; }
                        seen = t->generation;
                        n = t->cmd_n;
                        log = t->cmd_log;
                }
                Band *me = t->bands[b];
                const double cpu0 = thread_cpu_seconds();
                me->rc = band_iterations(t, b, n, log);
                me->cpu_seconds += thread_cpu_seconds() - cpu0;
                if(me->rc != J2P_OK) {
                        strncpy(me->err, j2p_last_error(), sizeof(me->err) - 1);
                        {
                                std::lock_guard<std::mutex> g(t->seq_lock);
                                t->abort.store(true);
                        }
                        t->seq_cv.notify_all();
                        release_value_waiters(t);
                }
                {
                        std::lock_guard<std::mutex> g(t->lock);
                        t->finished++;
                }
                t->done.notify_one();
        }
}

unsigned gcd_u(unsigned a, unsigned b) { return b ? gcd_u(b, a % b) : a; }

// can every band's GPU write every other band's memory?  (same device: yes)
bool peers_reachable(unsigned nband, const int devices[], char *why, size_t why_len)
{
        for(unsigned a = 0; a < nband; a++) {
                for(unsigned b = 0; b < nband; b++) {
                        if(devices[a] == devices[b]) { continue; }
                        int can = 0;
                        if(hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) {
                                (void)hipGetLastError();
                                snprintf(why, why_len, "device %d cannot access device %d's memory (no peer access)", devices[a], devices[b]);
                                return false;
                        }
                }
        }
        return true;
}

int enable_peer_access(unsigned nband, const int devices[])
{
        for(unsigned a = 0; a < nband; a++) {
                for(unsigned b = 0; b < nband; b++) {
                        const int da = devices[a], db = devices[b];
                        if(da == db) { continue; }
                        if(hipSetDevice(da) != hipSuccess) { return j2p_fail(J2P_EDEVICE, "hipSetDevice(%d) failed", da); }
                        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                        (void)hipGetLastError();
                        if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                                return j2p_fail(J2P_EDEVICE, "hipDeviceEnablePeerAccess(%d -> %d): %s", da, db, hipGetErrorString(e));
                        }
                }
        }
        return J2P_OK;
}

bool devices_distinct(unsigned nband, const int devices[])
{
        for(unsigned a = 0; a < nband; a++) {
                for(unsigned b = a + 1; b < nband; b++) {
                        if(devices[a] == devices[b]) { return false; }
                }
        }
        return true;
}

}  // namespace

extern "C" {

void j2p_tiled_destroy(j2p_tiled *t)
{
        if(!t) { return; }
        {
                std::lock_guard<std::mutex> g(t->lock);
                t->quit = true;
        }
        t->wake.notify_all();
        for(Band *b : t->bands) {
                if(b->thread.joinable()) { b->thread.join(); }
        }
        int prev = -1;
        (void)hipGetDevice(&prev);
        if(t->abort.load()) {
                release_value_waiters(t);
                // rccl: a band that failed mid-run leaves the other bands' all-gather / send / recv kernels queued and waiting
                // for it; synchronising those streams would never return.  ncclCommAbort tears the communicators down with
                // their kernels (where the library has it; without it the streams are NOT drained below)
                if(t->rccl && t->rccl->CommAbort) {
                        for(Band *b : t->bands) {
                                if(b->comm) { (void)t->rccl->CommAbort(b->comm); b->comm = nullptr; }
                        }
                }
        }
        bool stuck = false;
        // (every aborted solver, not only the value form: a candidate of the verification that timed out on event waits or on
        // copies has no values to release, but its streams may be just as stuck — the polling loop with its limit decides
        // whether they can be synchronised at all; hipStreamSynchronize on a stream that never drains would hang the process)
        if(t->abort.load()) {
                // the release has to be REPEATED until every band stream is idle: the streams still hold hipStreamWriteValue64
                // operations of the iterations that were queued before the failure, and each of them puts a small value back
                // over the released one (seen: the failed band's own flag fell back to its last iteration and the other
                // band's stream sat in front of it for ever).  Finitely many are queued, every pass lets the streams get
                // further: this terminates.
                const auto t0 = std::chrono::steady_clock::now();
                for(;;) {
                        if(t->signals) { release_value_waiters(t); }
                        bool idle = true;
                        for(Band *b : t->bands) {
                                (void)hipSetDevice(b->device);
                                if(b->stream && hipStreamQuery(b->stream) == hipErrorNotReady) { idle = false; }
                        }
                        (void)hipGetLastError();
                        if(idle) { break; }
                        // (streams that do not drain although every value has been released are stuck on something else:
                        // better a leak than a process that never returns)
                        if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.) { stuck = true; break; }
                        std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
        }
        const bool undrainable = stuck || (t->abort.load() && t->exchange == kRccl && !(t->rccl && t->rccl->CommAbort));
        for(Band *b : t->bands) {
                (void)hipSetDevice(b->device);
                if(b->stream && !undrainable) { (void)hipStreamSynchronize(b->stream); }
                if(b->comm && t->rccl && !undrainable) { (void)t->rccl->CommDestroy(b->comm); }
        }
        for(Band *b : t->bands) {
                (void)hipSetDevice(b->device);
                // (undrainable: what the stuck streams may still touch — solvers, events, the collector, pinned buffers — is
                // leaked rather than waited for or pulled from under them)
                if(b->solver && !undrainable) { j2p_solver_destroy(b->solver); }        // synchronises the band's stream first
                for(int k = 0; k < 2 && !undrainable; k++) {
                        if(b->ev_grad[k]) { (void)hipEventDestroy(b->ev_grad[k]); }
                        if(b->ev_edge[k]) { (void)hipEventDestroy(b->ev_edge[k]); }
                        if(b->ev_norm[k]) { (void)hipEventDestroy(b->ev_norm[k]); }
                        if(b->ev_all[k]) { (void)hipEventDestroy(b->ev_all[k]); }
                }
                if(b->collector && !undrainable) { (void)hipStreamSynchronize(b->collector); (void)hipStreamDestroy(b->collector); }
                if(b->log_host && !undrainable) { (void)hipHostFree(b->log_host); }
                delete b;
        }
        if(prev >= 0) { (void)hipSetDevice(prev); }
        if(t->signals && !undrainable) { (void)hipHostFree(t->signals); }
        delete t;
        j2p_pool_trim();        // band arenas are large and rarely reused at the same size: back to the device
}

}  // extern "C"

namespace {

// how the bands of one tiled solver exchange: -1 = this layer decides (see j2p_tiled_create)
struct Plan {
        int exchange = -1;      // Exchange
        int wait = -1;          // WaitMode (direct only)
        bool forced = false;    // named by the caller's environment: no fallback to another one
};

int tiled_create_impl(j2p_tiled **out, unsigned nband, const int devices[], const unsigned cuts[], unsigned nchannel,
                      const j2p_plane planes[], float weight, const float pweight[], unsigned iterations, Plan plan)
{
        if(!out || !devices || !planes || !pweight) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        *out = nullptr;
        if(nband == 0 || nband > 32) { return j2p_fail(J2P_EINVAL, "1..32 bands, got %u", nband); }
        if(nchannel == 0 || nchannel > J2P_MAX_CHANNELS) { return j2p_fail(J2P_EINVAL, "nchannel must be 1..3 (compute.c:118)"); }
        unsigned W = 0, H = 0, align = J2P_TILE_ROWS;
        for(unsigned c = 0; c < nchannel; c++) {
                const j2p_plane &p = planes[c];
                if(p.w_samp == 0 || p.h_samp == 0 || p.w == 0 || p.h == 0) { return j2p_fail(J2P_EINVAL, "channel %u: empty plane", c); }
                if(p.w * p.w_samp > W) { W = p.w * p.w_samp; }
                if(p.h * p.h_samp > H) { H = p.h * p.h_samp; }
                align = align / gcd_u(align, 8 * p.h_samp) * (8 * p.h_samp);
        }
        // band boundaries: the caller's, or near-equal multiples of the alignment
        std::vector<unsigned> edge(nband + 1);
        if(cuts) {
                for(unsigned b = 0; b <= nband; b++) { edge[b] = cuts[b]; }
                if(edge[0] != 0 || edge[nband] != H) { return j2p_fail(J2P_EINVAL, "cuts must run from 0 to the canvas height %u", H); }
        } else {
                const unsigned units = (H + align - 1) / align;
                if(units < nband) { return j2p_fail(J2P_EINVAL, "a canvas of %u rows has only %u bands of %u rows for %u devices", H, units, align, nband); }
                unsigned start = 0;
                for(unsigned b = 0; b < nband; b++) {
                        edge[b] = start * align;
                        start += units / nband + (b < units % nband ? 1 : 0);
                }
                edge[nband] = H;
        }
        for(unsigned b = 0; b < nband; b++) {
                if(edge[b] >= edge[b + 1] || edge[b] % align) { return j2p_fail(J2P_EINVAL, "band %u: rows [%u,%u) not aligned to %u", b, edge[b], edge[b + 1], align); }
        }
        j2p_tiled *t = new(std::nothrow) j2p_tiled();
        if(!t) { return j2p_fail(J2P_ENOMEM, "host allocation failed"); }
        t->nband = nband;
        t->nch = nchannel;
        t->W = W;
        t->H = H;
        t->weight = weight;
        for(unsigned c = 0; c < nchannel; c++) { t->pweight[c] = pweight[c]; }
        int prev = -1;
        (void)hipGetDevice(&prev);
        int rc = J2P_OK;
        // ---- how the bands exchange (see the head of this file): the plan names it, or the devices decide ----
        {
                const char *env = j2p_exp_env("J2P_TILED_NORM");
                t->norm_by_root = !(env && strcmp(env, "all") == 0);
                t->root = 0;
                env = j2p_exp_env("J2P_TILED_SELF_NEIGHBOURS");
                t->self_neighbours = nband == 1 && env && atoi(env) != 0;
                if(plan.wait >= 0) { t->wait = (WaitMode)plan.wait; }
                const int want = plan.exchange;
                char why[200] = "";
                const bool reach = peers_reachable(nband, devices, why, sizeof(why));
                const unsigned rows_of_tiles = (H + J2P_TILE_ROWS - 1) / J2P_TILE_ROWS;
                if(want == kRccl || (want < 0 && !reach)) {
                        // one communicator per band: RCCL wants a GPU per rank
                        Rccl *r = rccl_load();
                        if(!rccl_usable(r)) {
                                rc = j2p_fail(J2P_EDEVICE, "%s%s%s", reach ? "" : why, reach ? "" : ", and RCCL is not available: ", r ? r->why : "out of memory");
                        } else if(!devices_distinct(nband, devices)) {
                                rc = j2p_fail(J2P_EDEVICE, "the rccl exchange needs one GPU per band (a device is listed twice)");
                        } else {
                                t->rccl = r;
                                t->exchange = kRccl;
                        }
                } else if(!reach) {
                        rc = j2p_fail(J2P_EDEVICE, "%s: the %s exchange needs it", why, kExchangeName[want]);
                } else if(want == kCopy || rows_of_tiles > kTreeRowsInProject) {
                        t->exchange = kCopy;        // (also: canvases whose row sums k_project's in-kernel tree cannot hold;
                                                    // decided again below from the first band's own tile-row count)
                } else {
                        t->exchange = kDirect;
                }
        }
        // a single band is a whole-canvas solver: j2p_tiled_run hands it to j2p_solver_run (no thread, no exchange) —
        // unless the one-band RCCL self-test asks for the real machinery
        const bool plain_single = nband == 1 && !(t->exchange == kRccl && t->self_neighbours);
        t->threaded = !plain_single;
        // direct / copy: every band writes (reads) other bands' memory: peer access first, so that every allocation the
        // band solvers make below is mapped for the peers from the start
        if(rc == J2P_OK && t->exchange != kRccl) { rc = enable_peer_access(nband, devices); }
        for(unsigned b = 0; b < nband && rc == J2P_OK; b++) {
                Band *bd = new(std::nothrow) Band();
                if(!bd) { rc = j2p_fail(J2P_ENOMEM, "host allocation failed"); break; }
                t->bands.push_back(bd);
                bd->device = devices[b];
                bd->row0 = edge[b];
                bd->row1 = edge[b + 1];
                const j2p_band band = {edge[b], edge[b + 1]};
                rc = j2p_solver_create(&bd->solver, bd->device, nullptr, nchannel, planes, weight, pweight, iterations,
                                       plain_single ? j2p_band{0, 0} : band, plain_single ? 0 : J2P_BAND_EVEN_IF_WHOLE);
                if(rc != J2P_OK) { break; }
                if(hipSetDevice(bd->device) != hipSuccess) { rc = j2p_fail(J2P_EDEVICE, "hipSetDevice(%d) failed", bd->device); break; }
                void *st = nullptr;
                j2p_solver_stream(bd->solver, &st);
                bd->stream = (hipStream_t)st;
                j2p_exchange e;
                j2p_solver_exchange_info(bd->solver, &e);
                bd->rowsum[0] = bd->rowsum[1] = e.partials_local;
                bd->rowsum_local = e.partials_local;
                bd->global_rows[0] = bd->global_rows[1] = e.partials_all;
                bd->first_tr = e.first_tile_row;
                bd->ntr = e.local_tile_rows;
                // (small canvases have tile rows of 8 or 4 image rows: what counts is the solver's own number of them)
                if(b == 0 && t->exchange == kDirect && e.global_tile_rows > kTreeRowsInProject) { t->exchange = kCopy; }
                if(b > 0 && bd->ntr != t->bands[0]->ntr) { t->equal_counts = false; }
                if(t->threaded && t->exchange == kCopy) {
                        rc = j2p_solver_alternate_rowsums(bd->solver, bd->rowsum);
                        if(rc != J2P_OK) { break; }
                }
                if(t->threaded && t->exchange == kDirect) {
                        rc = j2p_solver_global_rowsums(bd->solver, bd->global_rows);
                        if(rc != J2P_OK) { break; }
                }
                rc = j2p_solver_norm_ptr(bd->solver, &bd->norm);
                if(rc != J2P_OK) { break; }
                j2p_solver_halo_rows(bd->solver, 0, &bd->rows[0]);
                j2p_solver_halo_rows(bd->solver, 1, &bd->rows[1]);
                // (J2P_TILED_EVENT_FLAGS=<hex>: extra hipEventCreateWithFlags bits, a timing experiment — e.g. 0x20000000
                // hipEventDisableSystemFence.  Honoured ONLY when all bands share one GPU: between GPUs the system-scope
                // release of the record is what makes a band's stores into its peers' memory visible to them)
                unsigned evflags = hipEventDisableTiming;
                if(const char *env = j2p_exp_env("J2P_TILED_EVENT_FLAGS")) {
                        bool one_device = true;
                        for(unsigned k = 1; k < nband; k++) { one_device = one_device && devices[k] == devices[0]; }
                        if(one_device) { evflags |= (unsigned)strtoul(env, nullptr, 16); }
                }
                for(int k = 0; k < 2 && rc == J2P_OK; k++) {
                        if(hipEventCreateWithFlags(&bd->ev_grad[k], evflags) != hipSuccess ||
                           hipEventCreateWithFlags(&bd->ev_edge[k], evflags) != hipSuccess ||
                           hipEventCreateWithFlags(&bd->ev_norm[k], evflags) != hipSuccess ||
                           hipEventCreateWithFlags(&bd->ev_all[k], evflags) != hipSuccess) {
                                rc = j2p_fail(J2P_EDEVICE, "hipEventCreate failed");
                        }
                }
                if(rc == J2P_OK && t->wait == kWaitCollector && hipStreamCreateWithFlags(&bd->collector, hipStreamNonBlocking) != hipSuccess) {
                        rc = j2p_fail(J2P_EDEVICE, "hipStreamCreate failed");
                }
        }
        // ---- direct, wait counter: the counters and flags, and stream memory operations on every device ----
        // A stream that waits for a value blocks the hardware queue it is mapped to, and a device's streams share a few
        // hardware queues (four by default): a waiting band can sit in front of the band it waits for — seen as a hang with
        // five bands on one GPU.  (Event waits do not have the problem: the runtime knows those dependencies.)  The value form
        // is for bands on GPUs of their own, and that is the only place the picker (pick_plan) offers it; named through
        // J2P_TILED_WAIT it is also taken with two bands on a device (the one-GPU rehearsal of the tests, where it has been
        // running since round 4 — whether two band streams share a hardware queue is the runtime's choice, not a promise),
        // and with more it silently becomes the event form.
        if(rc == J2P_OK && t->threaded && t->exchange == kDirect && t->wait == kWaitCounter) {
                for(unsigned a = 0; a < nband; a++) {
                        unsigned same = 0;
                        for(unsigned b = 0; b < nband; b++) { same += devices[a] == devices[b]; }
                        if(same > 2) { t->wait = kWaitAll; }
                }
        }
        if(rc == J2P_OK && t->threaded && t->exchange == kDirect && t->wait == kWaitCounter) {
                for(unsigned b = 0; b < nband && rc == J2P_OK; b++) {
                        int can = 0;
                        if(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, devices[b]) != hipSuccess || !can) {
                                rc = j2p_fail(J2P_EDEVICE, "J2P_TILED_WAIT=counter: device %d has no hipStreamWaitValue", devices[b]);
                        }
                }
                if(rc == J2P_OK && hipHostMalloc((void **)&t->signals, (size_t)2 * nband * 64, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
                        (void)hipGetLastError();
                        rc = j2p_fail(J2P_ENOMEM, "J2P_TILED_WAIT=counter: no pinned host memory for the counters");
                }
                if(rc == J2P_OK) { memset(t->signals, 0, (size_t)2 * nband * 64); }
        }
        // ---- direct: tell every band where its neighbours' halo rows and everybody's global arrays are ----
        if(rc == J2P_OK && t->threaded && t->exchange == kDirect) {
                for(unsigned b = 0; b < nband && rc == J2P_OK; b++) {
                        Band *bd = t->bands[b];
                        j2p_band_links l;
                        memset(&l, 0, sizeof(l));
                        for(int buf = 0; buf < 2; buf++) {
                                for(unsigned c = 0; c < nchannel; c++) {
                                        if(b > 0) { l.up_halo[buf][c] = t->bands[b - 1]->rows[buf].recv_bottom[c]; }
                                        if(b + 1 < nband) { l.down_halo[buf][c] = t->bands[b + 1]->rows[buf].recv_top[c]; }
                                }
                                for(unsigned p = 0; p < nband; p++) { l.push[buf][p] = t->bands[p]->global_rows[buf]; }
                        }
                        l.npush = nband;
                        if(t->wait == kWaitCounter) {
                                l.ncount = nband;
                                for(unsigned p = 0; p < nband; p++) { l.count[p] = t->grad_count(p); }
                        }
                        rc = j2p_solver_link_bands(bd->solver, &l);
                }
        }
        // ---- rccl: one communicator per band, all in this process ----
        if(rc == J2P_OK && t->exchange == kRccl) {
                std::vector<void *> comms(nband, nullptr);
                const int e = t->rccl->CommInitAll(comms.data(), (int)nband, devices);
                if(e != 0) { rc = j2p_fail(J2P_EDEVICE, "ncclCommInitAll over %u GPUs failed: %s", nband, t->rccl->GetErrorString(e)); }
                else {
                        for(unsigned b = 0; b < nband; b++) { t->bands[b]->comm = comms[b]; }
                }
        }
        // every band's initial state (and, direct, nothing else) must be in place before any band's first phase reads
        // or writes a neighbour: create is not a hot path, drain
        for(unsigned b = 0; b < t->bands.size() && rc == J2P_OK; b++) { rc = j2p_solver_sync(t->bands[b]->solver); }
        if(prev >= 0) { (void)hipSetDevice(prev); }
        if(rc != J2P_OK) {
                // (the error text survives the destroy: it is the calling thread's)
                char keep[512];
                strncpy(keep, j2p_last_error(), sizeof(keep) - 1);
                keep[sizeof(keep) - 1] = 0;
                j2p_tiled_destroy(t);
                return j2p_fail(rc, "%s", keep);
        }
        if(t->threaded) {
                for(unsigned b = 0; b < nband; b++) { t->bands[b]->thread = std::thread(band_main, t, b); }
        }
        *out = t;
        return J2P_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Which exchange a set of GPUs gets is MEASURED AND CHECKED on those GPUs, once per process and device list, before the
// first job runs on them (the loop being replaced: compute.c:427-453).  None of the exchanges can be assumed right on
// hardware this code has not met: posted peer writes, waits on values in host memory and RCCL all depend on what the
// runtime and the fabric of THAT node do.  So: a scratch canvas of the job's width, three 16-row tile rows per band,
// cut from the job's own first rows, is solved for kVerifyIterations iterations (a) whole, by ONE plain solver on the
// first GPU — no exchange at all: the truth — and (b) as nband bands through every candidate exchange; a candidate
// whose canvas differs from (a) in a single bit is DEMOTED (one line on stderr), and the fastest of the others is the
// plan for this device list from then on.  The scratch canvas is small on purpose: its planes stay in the caches
// between iterations, which is where a missing release / acquire shows (a full-size band evicts its own stale lines).
// J2P_TILED_EXCHANGE / J2P_TILED_WAIT name an exchange and skip all of this.  Devices that are listed twice share
// caches and queues — nothing to find out — and get `direct` with event waits; J2P_TILED_VERIFY=1 runs the procedure
// there too (tests; =2 also prints what was measured), J2P_TILED_VERIFY=0 never runs it.
// ---------------------------------------------------------------------------------------------------------------
constexpr unsigned kVerifyIterations = 8, kVerifyTimedIterations = 24;
// per device list: the verified plan — or the fact that NOTHING verifies there (kept too: a node on which every candidate
// fails must not repeat a truth solve, five scratch solvers, an RCCL init and their deadlines for every image of a batch)
// — or a marker that a thread is at it right now: others asking for the same list wait for that thread instead of
// verifying beside it on the same GPUs, which would also distort the timings the choice is made from
struct PlanEntry {
        bool done = false;      // false: being measured by some thread
        bool none = false;      // done, and no exchange reproduces the one-GPU solve on these GPUs
        Plan plan;
        std::string why;        // none: the error text
};
std::mutex g_plan_lock;
std::condition_variable g_plan_cv;
std::map<std::vector<int>, PlanEntry> g_plans;

struct Candidate {
        Plan plan;
        const char *name;
};

// j2p_tiled_sync with a deadline: a candidate exchange that never completes on this hardware (a value that is never
// counted up, a collective that never matches) must cost its candidacy, not the process.  On time-out the solver is marked
// failed — which releases value waiters and lets the destroy abort RCCL communicators — and J2P_EDEVICE comes back.
int sync_within(j2p_tiled *t, double seconds)
{
        const auto t0 = std::chrono::steady_clock::now();
        for(;;) {
                bool idle = true;
                for(Band *b : t->bands) {
                        if(hipSetDevice(b->device) != hipSuccess) { return j2p_fail(J2P_EDEVICE, "hipSetDevice(%d) failed", b->device); }
                        const hipError_t e = b->stream ? hipStreamQuery(b->stream) : hipSuccess;
                        if(e == hipErrorNotReady) { idle = false; }
                        else if(e != hipSuccess) { (void)hipGetLastError(); return j2p_fail(J2P_EDEVICE, "band stream on device %d: %s", b->device, hipGetErrorString(e)); }
                }
                (void)hipGetLastError();
                if(idle) { return J2P_OK; }
                if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
                        {
                                std::lock_guard<std::mutex> g(t->seq_lock);
                                t->abort.store(true);
                        }
                        t->seq_cv.notify_all();
                        release_value_waiters(t);
                        return j2p_fail(J2P_EDEVICE, "the bands did not finish within %.0f s", seconds);
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
}
constexpr double kVerifyDeadlineSeconds = 20.;

// one scratch run: 0 = matches `truth` (seconds per timed iteration in *secs), 1 = ran and differs, 2 = could not run
int verify_candidate(const Candidate &cand, unsigned nband, const int devices[], const unsigned cuts[], unsigned nchannel,
                     const j2p_plane planes[], float weight, const float pweight[], size_t canvas_floats,
                     const std::vector<std::vector<float>> &truth, double *secs, char *why, size_t why_len)
{
        j2p_tiled *t = nullptr;
        int rc = tiled_create_impl(&t, nband, devices, cuts, nchannel, planes, weight, pweight, kVerifyIterations, cand.plan);
        int verdict = 2;
        std::vector<float> got(canvas_floats);
        if(rc == J2P_OK) { rc = j2p_tiled_run(t, kVerifyIterations, nullptr); }
        if(rc == J2P_OK) { rc = sync_within(t, kVerifyDeadlineSeconds); }
        if(rc == J2P_OK) {
                verdict = 0;
                for(unsigned c = 0; c < nchannel && rc == J2P_OK; c++) {
                        rc = j2p_tiled_download(t, c, got.data());
                        if(rc == J2P_OK && memcmp(got.data(), truth[c].data(), canvas_floats * sizeof(float)) != 0) {
                                size_t bad = 0, first = canvas_floats;
                                for(size_t i = 0; i < canvas_floats; i++) {
                                        if(memcmp(&got[i], &truth[c][i], sizeof(float)) != 0) { bad++; if(first == canvas_floats) { first = i; } }
                                }
                                snprintf(why, why_len, "channel %u: %zu of %zu pixels differ from the one-GPU solve after %u iterations (first at pixel %zu)",
                                         c, bad, canvas_floats, kVerifyIterations, first);
                                verdict = 1;
                                break;
                        }
                }
        }
        if(rc == J2P_OK && verdict == 0) {
                // timed: the same canvas again, more iterations (the first run paid for first launches and page mappings)
                rc = j2p_tiled_reset(t);
                const auto t0 = std::chrono::steady_clock::now();
                if(rc == J2P_OK) { rc = j2p_tiled_run(t, kVerifyTimedIterations, nullptr); }
                if(rc == J2P_OK) { rc = sync_within(t, kVerifyDeadlineSeconds); }
                *secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / kVerifyTimedIterations;
        }
        if(rc != J2P_OK) {
                snprintf(why, why_len, "%s", j2p_last_error());
                verdict = 2;
        }
        if(t) { j2p_tiled_destroy(t); }
        return verdict;
}

void device_list_text(unsigned nband, const int devices[], char *out, size_t len)
{
        size_t at = 0;
        out[0] = 0;
        for(unsigned b = 0; b < nband && at + 8 < len; b++) { at += (size_t)snprintf(out + at, len - at, b ? ",%d" : "%d", devices[b]); }
}

// J2P_TILED_EXCHANGE / J2P_TILED_WAIT: the caller names the exchange (plan->forced) — no verification, no fallback
int plan_from_environment(Plan *plan)
{
        *plan = Plan();
        const char *env = getenv("J2P_TILED_WAIT");
        if(env && *env) {
                for(int k = 0; k < 4; k++) { if(strcmp(env, kWaitName[k]) == 0) { plan->wait = k; } }
                if(plan->wait < 0) { return j2p_fail(J2P_EINVAL, "J2P_TILED_WAIT=%s: all, root, collector or counter", env); }
                plan->forced = true;
        }
        env = getenv("J2P_TILED_EXCHANGE");
        if(env && *env) {
                for(int k = 0; k < 3; k++) { if(strcmp(env, kExchangeName[k]) == 0) { plan->exchange = k; } }
                if(plan->exchange < 0) { return j2p_fail(J2P_EINVAL, "J2P_TILED_EXCHANGE=%s: direct, copy or rccl", env); }
                plan->forced = true;
        }
        return J2P_OK;
}

// the plan for this device list: from the cache, or by the procedure above.  Returns J2P_OK with plan->exchange < 0 when
// there is nothing to decide (the caller takes the defaults), an error when no exchange works on these GPUs.
// decided: 0 = nothing to decide for THIS job (canvas too short for the exercise, or an error that is the job's own /
// transient: not remembered), 1 = *plan is the verified choice, 2 = nothing verifies on these GPUs (error returned)
int measure_plan(unsigned nband, const int devices[], unsigned nchannel, const j2p_plane planes[], float weight, const float pweight[], bool verbose, Plan *plan, int *decided)
{
        *decided = 0;
        // ---- the scratch canvas: the job's first rows, three tile rows per band ----
        unsigned align = J2P_TILE_ROWS, H = 0;
        for(unsigned c = 0; c < nchannel; c++) {
                const j2p_plane &p = planes[c];
                if(p.w_samp == 0 || p.h_samp == 0 || p.w == 0 || p.h == 0) { return J2P_OK; }     // (the create proper reports it)
                align = align / gcd_u(align, 8 * p.h_samp) * (8 * p.h_samp);
                if(p.h * p.h_samp > H) { H = p.h * p.h_samp; }
        }
        const unsigned per_band = (3 * J2P_TILE_ROWS + align - 1) / align * align;
        const unsigned rows = per_band * nband;
        if(H < rows) { return J2P_OK; }                       // a canvas this short is not worth the exercise: defaults
        j2p_plane scratch[J2P_MAX_CHANNELS];
        unsigned W = 0;
        for(unsigned c = 0; c < nchannel; c++) {
                scratch[c] = planes[c];                        // data / fdata: the first rows ARE the arrays' prefixes
                const unsigned h = rows / planes[c].h_samp;
                if(scratch[c].h > h) { scratch[c].h = h; }
                if(scratch[c].w * scratch[c].w_samp > W) { W = scratch[c].w * scratch[c].w_samp; }
        }
        unsigned Hs = 0;
        for(unsigned c = 0; c < nchannel; c++) { if(scratch[c].h * scratch[c].h_samp > Hs) { Hs = scratch[c].h * scratch[c].h_samp; } }
        if(Hs != rows) { return J2P_OK; }
        unsigned cuts[33];
        for(unsigned b = 0; b <= nband; b++) { cuts[b] = b * per_band; }
        const size_t canvas_floats = (size_t)W * rows;
        char devtext[160];
        device_list_text(nband, devices, devtext, sizeof(devtext));
        // ---- (a) the truth: one plain solver, no exchange ----
        std::vector<std::vector<float>> truth(nchannel, std::vector<float>(canvas_floats));
        {
                j2p_solver *s = nullptr;
                int rc = j2p_solver_create(&s, devices[0], nullptr, nchannel, scratch, weight, pweight, kVerifyIterations, j2p_band{0, 0}, 0);
                if(rc == J2P_OK) { rc = j2p_solver_run(s, kVerifyIterations, nullptr); }
                for(unsigned c = 0; c < nchannel && rc == J2P_OK; c++) { rc = j2p_solver_download(s, c, truth[c].data()); }
                if(s) { j2p_solver_destroy(s); }
                if(rc != J2P_OK) { return rc; }
        }
        // ---- (b) the candidates ----
        char why[200] = "";
        const bool reach = peers_reachable(nband, devices, why, sizeof(why));
        const bool own_gpus = devices_distinct(nband, devices);
        unsigned most = 0;
        for(unsigned a = 0; a < nband; a++) {
                unsigned same = 0;
                for(unsigned b = 0; b < nband; b++) { same += devices[a] == devices[b]; }
                if(same > most) { most = same; }
        }
        std::vector<Candidate> cands;
        auto add = [&](int ex, int wait, const char *name) {
                Candidate c;
                c.plan.exchange = ex;
                c.plan.wait = wait;
                c.name = name;
                cands.push_back(c);
        };
        if(reach) {
                bool value_waits = most <= 2;                  // (two per device only ever under J2P_TILED_VERIFY=1: the rehearsal)
                for(unsigned b = 0; b < nband && value_waits; b++) {
                        int can = 0;
                        value_waits = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, devices[b]) == hipSuccess && can;
                }
                (void)hipGetLastError();
                if(value_waits) { add(kDirect, kWaitCounter, "direct, wait counter"); }
                if(nband > 2) { add(kDirect, kWaitCollector, "direct, wait collector"); }
                add(kDirect, kWaitAll, "direct");
                add(kCopy, -1, "copy");
        }
        if(own_gpus && (!reach || rccl_usable(rccl_load()))) { add(kRccl, -1, "rccl"); }
        int best = -1;
        double best_secs = 0.;
        char line[512];
        size_t at = 0;
        line[0] = 0;
        for(size_t k = 0; k < cands.size(); k++) {
                // (RCCL costs seconds to initialise: only tried when nothing that writes peers' memory has been verified)
                if(cands[k].plan.exchange == kRccl && best >= 0) { continue; }
                double secs = 0.;
                char cwhy[320] = "";
                const int v = verify_candidate(cands[k], nband, devices, cuts, nchannel, scratch, weight, pweight, canvas_floats, truth, &secs, cwhy, sizeof(cwhy));
                if(v == 1) {
                        fprintf(stderr, "jpeg2png_amd: row tiling over GPUs %s: exchange '%s' DEMOTED, it does not reproduce the one-GPU solve (%s)\n", devtext, cands[k].name, cwhy);
                } else if(v == 2) {
                        fprintf(stderr, "jpeg2png_amd: row tiling over GPUs %s: exchange '%s' not available (%s)\n", devtext, cands[k].name, cwhy);
                } else {
                        if(at + 64 < sizeof(line)) { at += (size_t)snprintf(line + at, sizeof(line) - at, "%s'%s' %.1f us", at ? ", " : "", cands[k].name, secs * 1e6); }
                        if(best < 0 || secs < best_secs) { best = (int)k; best_secs = secs; }
                }
        }
        if(best < 0) {
                *decided = 2;
                return j2p_fail(J2P_EDEVICE, "row tiling over GPUs %s: no exchange reproduces the one-GPU solve on them (see stderr)%s%s", devtext,
                                reach ? "" : "; ", reach ? "" : why);
        }
        *plan = cands[(size_t)best].plan;
        *decided = 1;
        if(verbose) {
                fprintf(stderr, "jpeg2png_amd: row tiling over GPUs %s: verified per scratch iteration %s -> '%s'\n", devtext, line, cands[(size_t)best].name);
        }
        return J2P_OK;
}

int pick_plan(unsigned nband, const int devices[], unsigned nchannel, const j2p_plane planes[], float weight, const float pweight[], bool verbose, Plan *plan)
{
        *plan = Plan();
        const std::vector<int> key(devices, devices + nband);
        {
                std::unique_lock<std::mutex> g(g_plan_lock);
                for(;;) {
                        auto it = g_plans.find(key);
                        if(it == g_plans.end()) { break; }
                        if(!it->second.done) { g_plan_cv.wait(g); continue; }        // (the entry may be gone afterwards: look again)
                        if(it->second.none) { return j2p_fail(J2P_EDEVICE, "%s", it->second.why.c_str()); }
                        *plan = it->second.plan;
                        return J2P_OK;
                }
                g_plans[key] = PlanEntry();                    // ours to measure
        }
        int decided = 0;
        const int rc = measure_plan(nband, devices, nchannel, planes, weight, pweight, verbose, plan, &decided);
        {
                std::lock_guard<std::mutex> g(g_plan_lock);
                if(decided == 0) { g_plans.erase(key); }
                else {
                        PlanEntry &e = g_plans[key];
                        e.done = true;
                        e.none = decided == 2;
                        e.plan = *plan;
                        if(e.none) { e.why = j2p_last_error(); }
                }
        }
        g_plan_cv.notify_all();
        return rc;
}

}  // namespace

extern "C" {

// (internal, for compute_host.c and j2p_batch.hip) nonzero when the environment names the exchange: a job that cannot be
// row-tiled THAT way fails instead of quietly running on one GPU
int j2p_tiled_exchange_forced(void)
{
        Plan plan;
        return plan_from_environment(&plan) != J2P_OK || plan.forced;
}

int j2p_tiled_create(j2p_tiled **out, unsigned nband, const int devices[], const unsigned cuts[], unsigned nchannel,
                     const j2p_plane planes[], float weight, const float pweight[], unsigned iterations)
{
        if(!out || !devices || !planes || !pweight) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        *out = nullptr;
        if(nband == 0 || nband > 32) { return j2p_fail(J2P_EINVAL, "1..32 bands, got %u", nband); }
        if(nchannel == 0 || nchannel > J2P_MAX_CHANNELS) { return j2p_fail(J2P_EINVAL, "nchannel must be 1..3 (compute.c:118)"); }
        Plan plan;
        const char *env = nullptr;
        {
                const int rc = plan_from_environment(&plan);
                if(rc != J2P_OK) { return rc; }
        }
        if(!plan.forced && nband > 1) {
                env = getenv("J2P_TILED_VERIFY");
                const int verify = env && *env ? atoi(env) : -1;          // -1: where it can matter
                if(verify > 0 || (verify < 0 && devices_distinct(nband, devices))) {
                        int ndev = 0;
                        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
                                return j2p_fail(J2P_EDEVICE, "no HIP device available: the jpeg2png_amd solver has no CPU fallback");
                        }
                        for(unsigned b = 0; b < nband; b++) {
                                if(devices[b] < 0 || devices[b] >= ndev) { return j2p_fail(J2P_EINVAL, "device %d out of range (0..%d)", devices[b], ndev - 1); }
                        }
                        int prev = -1;
                        (void)hipGetDevice(&prev);
                        const int rc = pick_plan(nband, devices, nchannel, planes, weight, pweight, verify >= 2, &plan);
                        if(prev >= 0) { (void)hipSetDevice(prev); }
                        if(rc != J2P_OK) { return rc; }
                }
        }
        return tiled_create_impl(out, nband, devices, cuts, nchannel, planes, weight, pweight, iterations, plan);
}

int j2p_rccl_version(int *version)
{
        if(!version) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        *version = 0;
        Rccl *r = rccl_load();
        if(!rccl_usable(r)) { return j2p_fail(J2P_EDEVICE, "%s", r ? r->why : "out of host memory"); }
        if(!r->GetVersion || r->GetVersion(version) != 0) { return j2p_fail(J2P_EDEVICE, "librccl has no ncclGetVersion"); }
        return J2P_OK;
}

int j2p_tiled_exchange(const j2p_tiled *t, const char **name)
{
        if(!t || !name) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        static const char *const direct_names[4] = {"direct", "direct, wait root", "direct, wait collector", "direct, wait counter"};
        *name = !t->threaded ? "none" : (t->exchange == kDirect ? direct_names[t->wait] : kExchangeName[t->exchange]);
        return J2P_OK;
}

int j2p_tiled_canvas(const j2p_tiled *t, unsigned *W, unsigned *H, unsigned *nband)
{
        if(!t) { return j2p_fail(J2P_EINVAL, "tiled solver is NULL"); }
        if(W) { *W = t->W; }
        if(H) { *H = t->H; }
        if(nband) { *nband = t->nband; }
        return J2P_OK;
}

int j2p_tiled_band(const j2p_tiled *t, unsigned band, int *device, unsigned *row_begin, unsigned *row_end, j2p_solver **solver)
{
        if(!t || band >= t->nband) { return j2p_fail(J2P_EINVAL, "bad band index"); }
        const Band *b = t->bands[band];
        if(device) { *device = b->device; }
        if(row_begin) { *row_begin = b->row0; }
        if(row_end) { *row_end = b->row1; }
        if(solver) { *solver = b->solver; }
        return J2P_OK;
}

int j2p_tiled_sync(j2p_tiled *t)
{
        if(!t) { return j2p_fail(J2P_EINVAL, "tiled solver is NULL"); }
        // (after a band's failure the other bands' streams may hold collectives that can never complete: do not wait on them)
        if(t->abort.load()) { return j2p_fail(J2P_ESTATE, "a band failed earlier; the tiled solver is unusable"); }
        for(Band *b : t->bands) { BAND_TRY(j2p_solver_sync(b->solver)); }
        return J2P_OK;
}

int j2p_tiled_reset(j2p_tiled *t)
{
        if(!t) { return j2p_fail(J2P_EINVAL, "tiled solver is NULL"); }
        if(t->abort.load()) { return j2p_fail(J2P_ESTATE, "a band failed earlier; the tiled solver is unusable"); }
        // a band's reset must neither overtake a neighbour still reading its edge rows nor be overtaken by one still
        // writing into its halo rows: drain first, reset, drain again — this is not a hot path
        BAND_TRY(j2p_tiled_sync(t));
        // the band threads are idle between run() calls; every band back to iteration 0 from its resident inputs
        for(Band *b : t->bands) { BAND_TRY(j2p_solver_reset(b->solver)); }
        {
                std::lock_guard<std::mutex> g(t->seq_lock);
                for(Band *b : t->bands) { b->grad_recorded = b->edge_recorded = b->norm_recorded = 0; }
        }
        BAND_TRY(j2p_tiled_sync(t));
        if(t->signals) { memset(t->signals, 0, (size_t)2 * t->nband * 64); }     // (every stream is idle: nobody waits or counts)
        t->iter = 0;
        for(unsigned c = 0; c < J2P_MAX_CHANNELS; c++) { t->carried[c] = 0.; }
        t->carried_valid = true;
        return J2P_OK;
}

int j2p_tiled_run(j2p_tiled *t, unsigned n, j2p_log_row *rows)
{
        if(!t) { return j2p_fail(J2P_EINVAL, "tiled solver is NULL"); }
        if(n == 0) { return J2P_OK; }
        if(t->abort.load()) { return j2p_fail(J2P_ESTATE, "a band failed earlier; the tiled solver is unusable"); }
        if(t->threaded && j2p_injected_failure()) { return j2p_fail(J2P_EDEVICE, "injected failure (j2p_debug_fail_run_after)"); }
        if(!t->threaded) {
                // one band = a whole-canvas solver: its own loop (its norm reduction is not the band solvers')
                BAND_TRY(j2p_solver_run(t->bands[0]->solver, n, rows));
                t->iter += n;
                return J2P_OK;
        }
        const bool log = rows != nullptr;
        if(log != t->logging) {
                // the logging kernels run only while somebody reads their sums
                for(Band *b : t->bands) { BAND_TRY(j2p_solver_set_logging(b->solver, log ? 1 : 0)); }
                t->logging = log;
        }
        if(log) {
                int prev = -1;
                (void)hipGetDevice(&prev);
                for(Band *b : t->bands) {
                        (void)hipSetDevice(b->device);
                        if(!b->log_dev) {
                                j2p_exchange e;
                                BAND_TRY(j2p_solver_exchange_info(b->solver, &e));
                                b->log_dev = e.log_local;
                        }
                        if(b->log_cap < n) {
                                if(b->log_host) { (void)hipHostFree(b->log_host); b->log_host = nullptr; }
                                BAND_HIP(hipHostMalloc((void **)&b->log_host, (size_t)n * kLogCols * sizeof(double), hipHostMallocDefault));
                                b->log_cap = n;
                        }
                }
                if(prev >= 0) { (void)hipSetDevice(prev); }
        }
        {
                std::lock_guard<std::mutex> g(t->lock);
                t->cmd_n = n;
                t->cmd_log = log;
                t->finished = 0;
                t->generation++;
        }
        t->wake.notify_all();
        {
                std::unique_lock<std::mutex> g(t->lock);
                t->done.wait(g, [&] { return t->finished == t->nband; });
        }
        t->iter += n;
        // (the band that failed first, not one that stopped because of it)
        for(int pass = 0; pass < 2; pass++) {
                for(Band *b : t->bands) {
                        if(b->rc != J2P_OK && (pass == 1 || !strstr(b->err, "another band failed"))) {
                                return j2p_fail(b->rc, "band [%u,%u) on device %d: %s", b->row0, b->row1, b->device, b->err);
                        }
                }
        }
        if(log) {
                BAND_TRY(j2p_tiled_sync(t));
                // the bands' sums in band order (any fixed order: the values only feed the log)
                std::vector<double> sums((size_t)n * kLogCols, 0.);
                for(Band *b : t->bands) {
                        for(size_t k = 0; k < sums.size(); k++) { sums[k] += b->log_host[k]; }
                }
                // the prob distance entering the first of these iterations is only known if the previous ones were
                // logged too (or there were none): NaN otherwise, as j2p_solver_run reports it
                j2p_rows_from_sums_carry(t->nch, t->weight, t->pweight, n, sums.data(), t->carried, t->carried_valid, rows);
                t->carried_valid = true;
        } else {
                t->carried_valid = false;
        }
        return J2P_OK;
}

int j2p_tiled_host_cpu_seconds(const j2p_tiled *t, double *seconds)
{
        if(!t || !seconds) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        double s = 0.;
        for(const Band *b : t->bands) { s += b->cpu_seconds; }
        *seconds = s;
        return J2P_OK;
}

int j2p_tiled_download(j2p_tiled *t, unsigned c, float *out)
{
        if(!t || !out) { return j2p_fail(J2P_EINVAL, "NULL argument"); }
        for(Band *b : t->bands) { BAND_TRY(j2p_solver_download(b->solver, c, out + (size_t)b->row0 * t->W)); }
        return J2P_OK;
}

}  // extern "C"
