// jpeg2png_amd — C-ABI shim over the gfx950 kernels (include/jpeg2png_amd.h).
//
// A j2p_solver is the device-resident twin of the reference's per-call working
// set (`struct aux` x nchannel, compute.c:21-34) plus the iteration scalars of
// compute()/compute_step() (compute.c:425-443, :245, :258), which are evaluated
// here on the host in float exactly as the reference does and handed to the
// kernels as arguments.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "jpeg2png_amd.h"
#include "j2p_internal.h"
#include "j2p_kernels.hip.h"

using namespace j2p;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
        va_list l;
        va_start(l, fmt);
        vsnprintf(g_err, sizeof(g_err), fmt, l);
        va_end(l);
        return code;
}

}  // namespace

int j2p_fail(int code, const char *fmt, ...)
{
        va_list l;
        va_start(l, fmt);
        vsnprintf(g_err, sizeof(g_err), fmt, l);
        va_end(l);
        return code;
}

namespace {

#define HIP_TRY(expr)                                                                              \
        do {                                                                                       \
                hipError_t e_ = (expr);                                                            \
                if(e_ != hipSuccess) {                                                             \
                        return fail(e_ == hipErrorOutOfMemory ? J2P_ENOMEM : J2P_EDEVICE,          \
                                    "%s failed: %s", #expr, hipGetErrorString(e_));                \
                }                                                                                  \
        } while(0)

struct ChanHost {
        unsigned cw = 0, ch = 0, ws = 1, hs = 1;
        unsigned crow0 = 0, crows = 0;       // coefficient rows of d / pg held on this device
        unsigned frow0 = 0, frows = 0;       // coefficient rows of the decoded input held (init only)
        float *xbuf[2] = {nullptr, nullptr}; // allocation bases, (rows + 2*halo) * W floats
        float *grad = nullptr;
        float *pg = nullptr;
        int16_t *d = nullptr;
        uint8_t *d8 = nullptr;               // d + 128 in bytes (k_narrow_coefficients); read by k_project instead of d when `narrow`
        bool narrow_fits = false;            // every |d| of the channel's rows held here is <= 127
        bool narrow = false;                 // ... and the projection reads the bytes (J2P_OPT_NARROW_COEFFICIENTS, default on)
        float *q = nullptr;
        float *decoded = nullptr;            // frows * cw floats
        float *scratch_f = nullptr;          // decode scratch of bands too short to lend their x buffers (create only)
        int16_t *scratch_d = nullptr;
        float pweight = 0.f;
};

}  // namespace

struct j2p_solver {
        int device = 0;
        hipStream_t stream = nullptr;
        bool own_stream = false;
        unsigned nch = 0;
        unsigned W = 0, H = 0;
        unsigned row0 = 0, rows = 0;
        bool whole = true;
        bool band_local = false;
        ChanHost ch[kMaxCh];
        float weight = 0.f;
        unsigned iterations = 0;
        // iteration state
        unsigned iter = 0;
        float t = 1.f;
        float factor = 0.f;      // of the iteration whose gradient phase ran last
        int cur = 0;             // xbuf[cur] is x_k
        bool grad_done = false;
        bool proj_boundary_done = false;   // between the two parts of a split projection phase
        void *arena = nullptr;   // the one device allocation everything below is carved from (pooled, see pool_take)
        size_t arena_bytes = 0;
        // reductions
        bool fold = false;       // norm reduction folded into k_gradient (J2P_OPT_NORM_FOLD); default: band solvers only
        unsigned zone_d = 0, zone_b = 0, zone_c = 0;   // shares (1/256) of a gradient launch dealt as double / half / quarter tile rows (grad_item)
        bool grad_reverse = false;         // the gradient launch walks the canvas bottom-up (Geo::reverse)
        bool joint_inwave = false;   // J2P_OPT_JOINT_INWAVE
        bool norm_in_project = false;   // J2P_OPT_NORM_IN_PROJECT (with fold): level 2 of the norm inside k_project
        int nip_form = 1;               // ... by every wavefront (1: small canvases) or by the workgroup's first (2), see project_strip
        int nt = 0;                     // 0..3: streams with the non-temporal hint (nt_policy; J2P_OPT_NT_GRADIENT)
        bool nt_forced = false;         // set through J2P_OPT_NT_GRADIENT: the policy no longer touches it
        bool live_registered = false;   // this solver's bytes are part of the device's live total (nt_policy)
        size_t live_ws = 0, live_g = 0, live_planes = 0, live_d = 0;
        bool phase_log = false;         // the gradient phase of the running iteration was issued with logging
        bool mixed_project = true;      // small canvases: all samplings in one projection launch (J2P_OPT_MIXED_PROJECT)
        unsigned *d_maxabs = nullptr;                 // [channel] largest |d| (k_narrow_coefficients, create only)
        unsigned long long *dbg_counters = nullptr;   // J2P_DEBUG builds: [0] address violations, [1] first site, [2] first offset
        unsigned long long *trace = nullptr;          // J2P_TRACE builds: wave records (tools/wave_trace.py)
        unsigned trace_cap = 0, trace_used = 0;      // records reserved by the launches so far
        bool trace_on = false;
        bool norm_ready = false; // the gradient launch of this iteration also produced norm[]
        bool norm_by_project = false;   // ... or left level-1 row sums that k_project reduces itself
        unsigned *tickets = nullptr;     // device: [ntr_local] per-tile-row arrival counters + [1] finished-rows counter
        unsigned rpw = 16;
        unsigned px = 2;                // columns per lane of the gradient strips (2: 128-column strips, 1: 64-column strips)
        bool interior_done = false;
        bool rowsums_pending = false;
        unsigned ntx = 0, nseg = 0, ntr_local = 0, ntr_global = 0, first_tr = 0;   // strips per row, row segments
        double *part_g2 = nullptr;       // [c][ntr_local][ntx]
        double *rowsum_odd = nullptr;    // band solvers: second level-1 buffer, used by odd iterations once rowsum_alternate is on
        bool rowsum_alternate = false;
        double *rowsum_local = nullptr;  // [ntr_local][c]
        double *rowsum_all = nullptr;    // [ntr_global][c]  (== rowsum_local when whole)
        double *rowsum_all_odd = nullptr;   // band solvers: the global array of odd iterations once the bands are linked
        bool linked = false;                // j2p_solver_link_bands: neighbours' rows read in place, row sums pushed
        j2p_band_links links;
        RowsumPush *push_dev = nullptr;     // [2]: the push lists of even / odd iterations, in device memory (GradArgs::push)
        bool band_nip = true;               // band solvers: ||g|| from the global row sums inside k_project (NIP 2) instead of k_norm_finish
        float *norm = nullptr;           // [c]
        // logging
        double *part_tv = nullptr;       // [ntiles][2]
        double *part_prob = nullptr;     // [c][strips]
        unsigned strips_stride = 0;
        double *logsums = nullptr;       // [iter chunk][2 + kMaxCh]
        unsigned logsums_cap = 0;
        double carried_prob[kMaxCh] = {0., 0., 0.};
        bool carried_valid = true;
        bool log_phases = false;         // the phase calls run the logging kernels and fill log_band
        bool bandlog_pending = false;    // split gradient phase: band sums still to be launched (phase_rowsums)
        double *log_band = nullptr;      // [2 + kMaxCh]: tv, tv2 of the last gradient phase, prob per channel of the last projection
        // timing
        unsigned timing = 0;     // 0 = off, k = time every k-th iteration
        std::vector<hipEvent_t> ev;      // triples: before gradient, after gradient/before reduce.., see record()
        size_t ev_used = 0;
        double acc_grad_ms = 0., acc_proj_ms = 0.;
        unsigned acc_samples = 0;
        double ev_pair_ms = 0.;  // what two event records with NOTHING between them measure on this stream (enable_timing)
};

namespace {

struct DeviceGuard {
        int prev = -1;
        bool ok = true;
        explicit DeviceGuard(int dev)
        {
                if(hipGetDevice(&prev) != hipSuccess) { prev = -1; }
                if(prev != dev) { ok = hipSetDevice(dev) == hipSuccess; }
        }
        ~DeviceGuard()
        {
                if(prev >= 0) { (void)hipSetDevice(prev); }
        }
};

// ---------------------------------------------------------------------------
// Device-memory pool.  hipMalloc / hipFree cost milliseconds and hipFree synchronises the whole device, which
// serialises the concurrent compute() calls of a multi-threaded host (jpeg2png.c:147,330) far more than the
// solves themselves at 1080p.  A solver therefore makes ONE allocation (its arena) and returns it here when it
// is destroyed; the next solver on that device takes the smallest cached block that fits.  Bounded: at most
// kPoolBlocks blocks / kPoolBytes bytes stay cached per process, the rest is released; j2p_pool_trim() drops all.
// ---------------------------------------------------------------------------
struct PoolBlock {
        int device;
        void *ptr;
        size_t bytes;
};
std::mutex g_pool_lock;
std::vector<PoolBlock> g_pool;
constexpr size_t kPoolBlocks = 16;                     // per device
constexpr size_t kPoolBytesDefault = (size_t)8 << 30;  // per device; J2P_POOL_MIB overrides (0 = no caching)

size_t pool_cap_bytes()
{
        static const size_t cap = [] {
                const char *env = getenv("J2P_POOL_MIB");
                if(env && *env) { return (size_t)strtoull(env, nullptr, 10) << 20; }
                return kPoolBytesDefault;
        }();
        return cap;
}

void pool_drop_all()
{
        std::vector<PoolBlock> drop;
        {
                std::lock_guard<std::mutex> g(g_pool_lock);
                drop.swap(g_pool);
        }
        for(const PoolBlock &b : drop) {
                DeviceGuard guard(b.device);
                (void)hipFree(b.ptr);
        }
}

// hipMalloc for everything that does not go through the pool (log buffers, the stand-alone decode / DCT calls):
// on out-of-memory the cached arenas go back to the device and the allocation is tried once more
hipError_t dev_malloc(void **out, size_t bytes)
{
        hipError_t e = hipMalloc(out, bytes);
        if(e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                pool_drop_all();
                e = hipMalloc(out, bytes);
        }
        return e;
}

hipError_t pool_take(int device, size_t bytes, void **out, size_t *got)
{
        {
                std::lock_guard<std::mutex> g(g_pool_lock);
                size_t best = g_pool.size();
                for(size_t i = 0; i < g_pool.size(); i++) {
                        const PoolBlock &b = g_pool[i];
                        // a block more than twice the size asked for stays for a larger customer
                        if(b.device == device && b.bytes >= bytes && b.bytes <= 2 * bytes + (1u << 20) &&
                           (best == g_pool.size() || b.bytes < g_pool[best].bytes)) { best = i; }
                }
                if(best != g_pool.size()) {
                        *out = g_pool[best].ptr;
                        *got = g_pool[best].bytes;
                        g_pool.erase(g_pool.begin() + (ptrdiff_t)best);
                        return hipSuccess;
                }
        }
        *got = bytes;
        return dev_malloc(out, bytes);
}

void pool_give(int device, void *ptr, size_t bytes)
{
        if(!ptr) { return; }
        {
                std::lock_guard<std::mutex> g(g_pool_lock);
                size_t total = bytes, blocks = 0;
                for(const PoolBlock &b : g_pool) {
                        if(b.device == device) { total += b.bytes; blocks++; }
                }
                if(blocks < kPoolBlocks && total <= pool_cap_bytes()) {
                        g_pool.push_back(PoolBlock{device, ptr, bytes});
                        return;
                }
        }
        (void)hipFree(ptr);
}

// ---------------------------------------------------------------------------
// What is live on each device, for the non-temporal policy (nt_policy): the Infinity Cache is shared by every
// solver iterating on the GPU — the images of a batch, the components of `-s`, the bands of a tiled run that
// share a device — so the policy looks at the sum of their working sets, not at one solver's.
// ---------------------------------------------------------------------------
struct LiveBytes {
        size_t working_set = 0, g = 0, planes = 0, d = 0;
};
constexpr int kMaxDevices = 64;
LiveBytes g_live[kMaxDevices];          // guarded by g_pool_lock

void live_add(int device, const LiveBytes &b, int sign)
{
        if(device < 0 || device >= kMaxDevices) { return; }
        std::lock_guard<std::mutex> g(g_pool_lock);
        LiveBytes &l = g_live[device];
        if(sign > 0) { l.working_set += b.working_set; l.g += b.g; l.planes += b.planes; l.d += b.d; }
        else { l.working_set -= b.working_set; l.g -= b.g; l.planes -= b.planes; l.d -= b.d; }
}
LiveBytes live_on(int device)
{
        if(device < 0 || device >= kMaxDevices) { return LiveBytes{}; }
        std::lock_guard<std::mutex> g(g_pool_lock);
        return g_live[device];
}

// bump allocator over the arena: pass 1 (base == nullptr) only adds the sizes up
struct Carver {
        char *base = nullptr;
        size_t used = 0;
        template <typename T>
        void take(T *&p, size_t count)
        {
                used = (used + 255) & ~(size_t)255;
                p = base ? reinterpret_cast<T *>(base + used) : nullptr;
                used += count * sizeof(T);
        }
};

constexpr size_t kNtWorkingSet = (size_t)260 << 20;      // see nt_policy in j2p_solver_create
constexpr size_t kNormInProjectPixels = (size_t)5 << 19; // whole canvases up to this size (2.5 Mpixel) reduce ||g|| without a launch of its own
constexpr size_t kMixedProjectPixels = (size_t)1 << 20;  // canvases up to this size project all channels in one launch

// whole canvases from this size on (and at most kFoldMaxRows tile rows) reduce ||g|| entirely inside k_gradient: on wide
// planes the k_norm_whole launch stages 17 K partials through one CU (9 us at W = 16384) — 16384x2048 232.1 -> 229.4 us per
// iteration, 8192^2 521.6 -> 511.3; at 4096^2 the launch (4.7 us) is the cheaper one, 117.6 vs 119.4
// (profiles/r04_fold_without_ack_waits.jsonl)
constexpr size_t kFoldWholePixels = (size_t)1 << 25;
// fewer gradient wavefronts than this (128-column, 16-row strips) -> 64-column strips.  0 = never: measured on canvases from
// 0.26 to 16.8 Mpixel (profiles/r03_px_rpw_sweep.jsonl) the one-column-per-lane form is nowhere faster than the packed one
// with the same rows per strip (512x512 4:2:0 29.2 vs 27.5 us per iteration, 1080p Y 31.1 vs 28.7, 2048^2 45.5 vs 44.4):
// twice the wavefronts do not shorten the launch, because each still walks as many rows, one row trip at a time
constexpr unsigned long long kPx1Waves = 0;
// rows per gradient strip: 16; 8, then 4, while the strips make fewer wavefronts than half the chip's 4096 slots (the
// launch is then one generation whose length is the busiest SIMD's: shorter strips balance it, at 25 / 50 instead of 12.5 %
// redundant rows).  A limit of 4096 for the first step was measured too (profiles/r03_px_rpw_sweep.jsonl,
// r03_rpw_concurrency.json): a single 2048^2 Y plane or 1080p 4:2:0 image gains 2 %, eight concurrent 1080p 4:2:0 images —
// the batch case, where the chip is full anyway — lose 5.7 %; not taken
constexpr unsigned long long kHalfStripWaves = 2048;
constexpr unsigned long long kShortStripWaves = 2048;
// half / quarter items at the end of a gradient launch (see j2p_solver_create): launches of fewer strips than this, and
// the shares (1/256) of every XCD's run dealt that way
constexpr unsigned long long kZoneMaxWaves = 3 * 4096;
constexpr unsigned kZoneB = 32, kZoneC = 10;
constexpr unsigned kBigZoneD = 200, kBigZoneB = 24, kBigZoneC = 8;

unsigned gcd_u(unsigned a, unsigned b) { return b ? gcd_u(b, a % b) : a; }
unsigned lcm_u(unsigned a, unsigned b) { return a / gcd_u(a, b) * b; }

// nt_policy: which streams of the iteration get the non-temporal hint, so that what stays without it can live in
// the 256 MiB Infinity Cache.  Per byte and iteration x_k and x_{k-1} are touched 2-3 times, g and the prob state
// twice, d once: keep the planes, then d and the prob state if they fit beside them, g last.
// Measured on single Y planes (us per iteration; none / level 1 / level 2 / level 3):
//   4096x3584 (252 MiB) 112.3 / 114.1            4096x4096 (288 MiB) 135.8 / 127.0
//   4096x5120 (360 MiB) 175.0 / 163.7 / 159.6 / 163.4
//   16384x2048 (576 MiB) 295 / 292.7 / 250.8 / 240.2     8192x8192 (1152 MiB) - / 541.5 / 528.3 / 527.0
// The cache is the DEVICE's: the sums run over every live solver of the device (the images of a batch, the
// components of `-s`, bands sharing a GPU), re-evaluated at create and at reset.  J2P_NT_SCOPE=solver: this
// solver's own bytes only (round 2's policy; A/B).
int nt_policy(const j2p_solver *s)
{
        static const bool own_only = [] {
                const char *env = j2p_exp_env("J2P_NT_SCOPE");
                return env && strcmp(env, "solver") == 0;
        }();
        LiveBytes l = live_on(s->device);
        if(own_only || !s->live_registered) { l = LiveBytes{s->live_ws, s->live_g, s->live_planes, s->live_d}; }
        if(l.working_set <= kNtWorkingSet) { return 0; }                        // everything fits
        if(l.working_set - l.g <= kNtWorkingSet) { return 1; }                  // everything but g fits
        if(l.planes + l.d <= kNtWorkingSet) { return 2; }                       // planes and d fit
        return 3;
}

// the bytes of the coefficients the projection actually reads (one or two per coefficient, ChanHost::narrow) in the
// solver's share of the device's live total, and the policy again
void account_coefficient_bytes(j2p_solver *s)
{
        size_t d_bytes = 0;
        for(unsigned c = 0; c < s->nch; c++) {
                const ChanHost &h = s->ch[c];
                d_bytes += (size_t)(h.crows ? h.crows : 1) * h.cw * (h.narrow ? sizeof(uint8_t) : sizeof(int16_t));
        }
        if(d_bytes == s->live_d) { return; }
        if(s->live_registered) { live_add(s->device, LiveBytes{s->live_ws, s->live_g, s->live_planes, s->live_d}, -1); }
        s->live_ws = s->live_ws - s->live_d + d_bytes;
        s->live_d = d_bytes;
        if(s->live_registered) { live_add(s->device, LiveBytes{s->live_ws, s->live_g, s->live_planes, s->live_d}, +1); }
        if(!s->nt_forced) { s->nt = nt_policy(s); }
}

ChanDev chan_dev(const j2p_solver *s, unsigned c)
{
        const ChanHost &h = s->ch[c];
        ChanDev k;
        const size_t halo = (size_t)kHalo * s->W;
        k.xcur = h.xbuf[s->cur] + halo;
        k.xprev = h.xbuf[s->cur ^ 1] + halo;
        k.grad = h.grad;
        k.pg = h.pg;
        k.d = h.d;
        k.d8 = h.narrow ? h.d8 : nullptr;
        k.q = h.q;
        k.cw = h.cw;
        k.ch = h.ch;
        k.ws = h.ws;
        k.hs = h.hs;
        k.crow0 = h.crow0;
        k.crows = h.crows;
        k.p_alpha = h.pweight * 2 * 255 * sqrtf(2);       // compute.c:245
        k.prob_on = h.pweight != 0.f;
#ifdef J2P_DEBUG
        {
                // what each access of the phase kernels is meant to stay inside (see DbgChan)
                const long above = s->row0 < (unsigned)kHalo ? (long)s->row0 : (long)kHalo;
                const unsigned below_rows = s->H - s->row0 - s->rows;
                const long below = below_rows < (unsigned)kHalo ? (long)below_rows : (long)kHalo;
                const float *own[2] = {k.xcur, k.xprev};
                for(int i = 0; i < 2; i++) {
                        k.dbg.x_read[i] = {reinterpret_cast<const char *>(own[i] - above * (long)s->W),
                                           reinterpret_cast<const char *>(own[i] + ((long)s->rows + below) * (long)s->W)};
                        k.dbg.x_own[i] = {reinterpret_cast<const char *>(own[i]), reinterpret_cast<const char *>(own[i] + (size_t)s->rows * s->W)};
                }
                const size_t cells = (size_t)(h.crows ? h.crows : 1) * h.cw;
                k.dbg.grad = {reinterpret_cast<const char *>(h.grad), reinterpret_cast<const char *>(h.grad + (size_t)s->rows * s->W)};
                k.dbg.pg = {reinterpret_cast<const char *>(h.pg), reinterpret_cast<const char *>(h.pg + cells)};
                if(h.narrow) { k.dbg.d = {reinterpret_cast<const char *>(h.d8), reinterpret_cast<const char *>(h.d8 + cells)}; }
                else { k.dbg.d = {reinterpret_cast<const char *>(h.d), reinterpret_cast<const char *>(h.d + cells)}; }
                k.dbg.counters = s->dbg_counters;
        }
#endif
        return k;
}

Geo geo_of(const j2p_solver *s)
{
        Geo g;
        g.W = s->W;
        g.H = s->H;
        g.row0 = s->row0;
        g.rows = s->rows;
        g.ntx = s->ntx;
        g.rpw = s->rpw;
        g.seg_off = 0;
        g.seg_mul = 1;
        g.units = 0;            // (do_phase_gradient fills in the launch's own)
        g.ntr_launch = 0;
        g.zone_d = g.zone_b = g.zone_c = 0;
        g.reverse = 0;
#ifdef J2P_TRACE
        g.trace = s->trace_on ? s->trace : nullptr;
        g.trace_cap = s->trace_cap;
        g.trace_seq = s->iter;
        g.trace_base = s->trace_used;
#endif
        return g;
}

// where the gradient phase of iteration `iter` leaves its level-1 row sums [tile row][channel]: band solvers whose sums
// are read in place by other bands (j2p_solver_alternate_rowsums) alternate between two arrays
double *rowsums_of(const j2p_solver *s, unsigned iter)
{
        return (s->rowsum_alternate && (iter & 1)) ? s->rowsum_odd : s->rowsum_local;
}

// units of a gradient launch over `ntr` tile rows (grad_item): pairs of tile rows x 4 strips (256-thread workgroups), or —
// joint images, one wavefront per channel — x one strip
unsigned grad_units(const j2p_solver *s, unsigned ntr)
{
        const unsigned positions = s->ntx * ((ntr + 1) / 2);
        return s->nch == 1 || s->joint_inwave ? (positions + 3) / 4 : positions;
}
ZoneShares zone_shares(const Geo &g) { return ZoneShares{g.zone_d, g.zone_b, g.zone_c}; }

template <int NCH, int J, int PX = 2>
void launch_gradient_n(const GradArgs &a, hipStream_t st, bool tgv, bool log, int nt)
{
        // J == 1: 4 strips per 256-thread workgroup; J > 1: one strip per workgroup of J wavefronts
        const dim3 grid(grad_grid(a.geo.units, zone_shares(a.geo)));
        const dim3 block = J == 1 ? dim3(256) : dim3(64 * J);
        if constexpr(NCH == 1 && PX == 2) {
                // non-temporal g / prob state (see nt_policy): the one-channel-per-wavefront kernels without logging
                if(nt >= 1 && !log) {
                        if(nt >= 2) {
                                if(tgv) { hipLaunchKernelGGL((k_gradient<NCH, true, false, J, 2>), grid, block, 0, st, a); }
                                else { hipLaunchKernelGGL((k_gradient<NCH, false, false, J, 2>), grid, block, 0, st, a); }
                        } else {
                                if(tgv) { hipLaunchKernelGGL((k_gradient<NCH, true, false, J, 1>), grid, block, 0, st, a); }
                                else { hipLaunchKernelGGL((k_gradient<NCH, false, false, J, 1>), grid, block, 0, st, a); }
                        }
                        return;
                }
        }
        // (one column per lane is for canvases that leave wavefront slots empty: they fit the caches, no hint)
        if(tgv) {
                if(log) { hipLaunchKernelGGL((k_gradient<NCH, true, true, J, 0, PX>), grid, block, 0, st, a); }
                else { hipLaunchKernelGGL((k_gradient<NCH, true, false, J, 0, PX>), grid, block, 0, st, a); }
        } else {
                if(log) { hipLaunchKernelGGL((k_gradient<NCH, false, true, J, 0, PX>), grid, block, 0, st, a); }
                else { hipLaunchKernelGGL((k_gradient<NCH, false, false, J, 0, PX>), grid, block, 0, st, a); }
        }
}

hipEvent_t next_event(j2p_solver *s)
{
        if(s->ev_used == s->ev.size()) {
                hipEvent_t e;
                if(hipEventCreate(&e) != hipSuccess) { return nullptr; }
                s->ev.push_back(e);
        }
        return s->ev[s->ev_used++];
}

void mark(j2p_solver *s)
{
        if(!s->timing || s->iter % s->timing) { return; }
        hipEvent_t e = next_event(s);
        if(e) { (void)hipEventRecord(e, s->stream); }
}

// fold recorded events (groups of 4: gradient begin/end, project begin/end) into the accumulators
int flush_timing(j2p_solver *s)
{
        if(s->ev_used == 0) { return J2P_OK; }
        HIP_TRY(hipStreamSynchronize(s->stream));
        for(size_t i = 0; i + 3 < s->ev_used; i += 4) {
                float g = 0.f, p = 0.f;
                HIP_TRY(hipEventElapsedTime(&g, s->ev[i], s->ev[i + 1]));
                HIP_TRY(hipEventElapsedTime(&p, s->ev[i + 2], s->ev[i + 3]));
                // (a pair of records brackets the kernel AND the records' own packets.  What an EMPTY bracket measures is
                // calibrated when timing is switched on (ev_pair_ms, 4.8 us on MI355X) and reported, but NOT taken off: with
                // a kernel in between part of it overlaps, and the corrected figures came out below rocprofv3's — 50.5 vs
                // 53.2 us — which is the wrong side to err on)
                s->acc_grad_ms += g;
                s->acc_proj_ms += p;
                s->acc_samples++;
        }
        s->ev_used = 0;
        return J2P_OK;
}

// part: 0 = all segments, 1 = interior segments only (they never read halo rows), 2 = the first and
// last segment (after the halo rows have arrived).  1 then 2 make one gradient phase; `st` is the stream
// the kernel goes to (part 2 may use a side stream so that it overlaps part 1).
// band-level sums for the CSV row (tv, tv2 after a gradient phase; prob distance after a projection)
static void launch_band_log(j2p_solver *s, int which)
{
        if(which == 0) {
                hipLaunchKernelGGL(k_log_sums, dim3(1), dim3(256), 0, s->stream, (const double *)s->part_tv, s->ntx * s->nseg,
                                   (const double *)s->part_prob, 0u, s->strips_stride, s->nch, s->log_band, 0);
        } else {
                hipLaunchKernelGGL(k_log_sums, dim3(1), dim3(256), 0, s->stream, (const double *)s->part_tv, 0u,
                                   (const double *)s->part_prob, s->strips_stride, s->strips_stride, s->nch, s->log_band, 1);
        }
}

// per-tile-row sums of the band's norm partials: blocks of up to 256 tile rows, as many as stage in LDS
static void launch_rowsums(j2p_solver *s)
{
        const unsigned total = s->ntr_local * s->nch;
        unsigned per_block = kStageDoubles / s->ntx;
        if(per_block > 256) { per_block = 256; }
        hipLaunchKernelGGL(k_rowsums, dim3((total + per_block - 1) / per_block), dim3(256), 0, s->stream,
                           (const double *)s->part_g2, s->rowsum_local, s->ntx, s->ntr_local, s->nch, per_block);
}

int do_phase_gradient(j2p_solver *s, bool log, int part = 0, hipStream_t st = nullptr)
{
        if(s->grad_done) { return fail(J2P_ESTATE, "phase_gradient called twice without phase_project"); }
        if(part == 2 && !s->interior_done) { return fail(J2P_ESTATE, "gradient boundary part before the interior part"); }
        if(part != 2 && s->interior_done) { return fail(J2P_ESTATE, "gradient interior part issued twice"); }
        if(part != 0 && s->nseg < 3) { return fail(J2P_ESTATE, "band too short to split the gradient phase"); }
        if(!st) { st = s->stream; }
        if(part != 2) {
                s->phase_log = log;
                // FISTA scalars in float, as compute.c:431-432,440
                const float tnext = (1 + sqrtf(1 + 4 * (s->t * s->t))) / 2;
                s->factor = (s->t - 1) / tnext;
                s->t = tnext;
        }
        unsigned nseg_launch = s->nseg;
        unsigned seg_off = 0, seg_mul = 1;
        if(part == 1) { nseg_launch = s->nseg - 2; seg_off = 1; }
        if(part == 2) { nseg_launch = 2; seg_mul = s->nseg - 1; }

        GradArgs a;
        for(unsigned c = 0; c < s->nch; c++) { a.ch[c] = chan_dev(s, c); }
        a.geo = geo_of(s);
        a.geo.seg_off = seg_off;
        a.geo.seg_mul = seg_mul;
        a.geo.units = grad_units(s, nseg_launch);
        a.geo.ntr_launch = nseg_launch;
        // (half / quarter items: whole phases of one channel per workgroup wavefront, see the policy in j2p_solver_create)
        if(part == 0 && s->nch == 1 && !s->joint_inwave) { a.geo.zone_d = s->zone_d; a.geo.zone_b = s->zone_b; a.geo.zone_c = s->zone_c; }
        a.geo.reverse = part == 0 && s->grad_reverse ? 1u : 0u;
        a.factor = s->factor;
        a.a_tv = (float)(1. / (double)sqrtf((float)s->nch));                    // compute.c:90
        const float alpha = s->weight / sqrtf((float)(4 / 2));                  // compute.c:258
        a.a_tgv = (float)((double)alpha * 1. / (double)sqrtf((float)s->nch));   // compute.c:154
        a.part_g2 = s->part_g2;
        a.part_tv = s->part_tv;
        // the norm reduction rides on this launch: level 1 (row sums) always, level 2 (tree -> norm) when the
        // launch covers the whole canvas
        const bool nip = s->norm_in_project && s->fold && s->whole && part == 0 && s->ntr_global <= kWaveTreeMax && !log;
        const bool fold_norm = !nip && s->fold && s->whole && part == 0 && s->ntr_global <= kFoldMaxRows;
        a.row_ticket = s->fold ? s->tickets : nullptr;
        a.done_ticket = s->tickets + s->ntr_local;
        a.rowsum = rowsums_of(s, s->iter);
        a.push = nullptr;
        if(s->linked) {
                if(part != 0) { return fail(J2P_ESTATE, "linked bands run whole phases (there is no exchange to hide)"); }
                a.push = s->push_dev + (s->iter & 1);
        }
        a.norm_out = fold_norm ? s->norm : nullptr;
        a.nch_total = s->nch;
        a.fold_phase = s->iter & 1;
        a.fold_rows = s->ntr_local;
        a.ntr_global = s->ntr_global;
        const bool tgv = s->weight != 0.f;
        if(part != 2) { mark(s); }     // event timing covers the main launch only (the edge part runs on another stream)
        // joint images: one wavefront per channel, the norms exchanged through LDS (three times the
        // wavefronts at 4 per SIMD), beats all channels in one wavefront (248 VGPRs, 2 per SIMD) at every
        // size measured: 198 vs 293 us at 12 Mpixel 4:2:0, 494 vs 717 us at 36 Mpixel.  J2P_JOINT_INWAVE=1
        // selects the in-wavefront kernel (kept: it is the same arithmetic in another schedule, and tested).
#ifdef J2P_EXPERIMENTS
        const bool inwave = s->joint_inwave;
        switch(s->nch) {
        case 1:
                if(s->px == 1) { launch_gradient_n<1, 1, 1>(a, st, tgv, log, 0); }
                else { launch_gradient_n<1, 1>(a, st, tgv, log, s->nt); }
                break;
        case 2:
                if(inwave) { launch_gradient_n<2, 1>(a, st, tgv, log, 0); }
                else if(s->px == 1) { launch_gradient_n<1, 2, 1>(a, st, tgv, log, 0); }
                else { launch_gradient_n<1, 2>(a, st, tgv, log, s->nt); }
                break;
        default:
                if(inwave) { launch_gradient_n<3, 1>(a, st, tgv, log, 0); }
                else if(s->px == 1) { launch_gradient_n<1, 3, 1>(a, st, tgv, log, 0); }
                else { launch_gradient_n<1, 3>(a, st, tgv, log, s->nt); }
                break;
        }
#else
        // (release build: one wavefront per channel, two columns per lane — the schedules that won everywhere)
        switch(s->nch) {
        case 1: launch_gradient_n<1, 1>(a, st, tgv, log, s->nt); break;
        case 2: launch_gradient_n<1, 2>(a, st, tgv, log, s->nt); break;
        default: launch_gradient_n<1, 3>(a, st, tgv, log, s->nt); break;
        }
#endif
        if(part != 2) { mark(s); }
        HIP_TRY(hipGetLastError());
#ifdef J2P_TRACE
        if(s->trace_on) {       // one record per wavefront of the launch (J == 1: 4 strips per workgroup; joint: one strip)
                const bool per_channel = s->nch > 1 && !s->joint_inwave;
                s->trace_used += grad_grid(a.geo.units, zone_shares(a.geo)) * (per_channel ? s->nch : 4u);
        }
#endif
        if(part == 1) {
                s->interior_done = true;
                return J2P_OK;
        }
        s->interior_done = false;
        s->norm_ready = fold_norm;
        s->norm_by_project = nip;
        if(part == 0 && !s->whole && !s->fold) {
                launch_rowsums(s);
                HIP_TRY(hipGetLastError());
        }
        s->grad_done = true;
        s->rowsums_pending = part == 2 && !s->whole && !s->fold;
        // band sums for the CSV row: behind the last launch of the phase (for a split phase that is
        // j2p_solver_phase_rowsums(), once the solver's stream has joined the edge part)
        s->bandlog_pending = false;
        if(log && s->log_phases) {
                if(part == 0) { launch_band_log(s, 0); }
                else { s->bandlog_pending = true; }
        }
        return J2P_OK;
}

// after a split gradient phase: per-tile-row sums on the solver's stream (the caller has made that
// stream wait for the boundary part)
int do_rowsums(j2p_solver *s)
{
        if(!s->grad_done) { return fail(J2P_ESTATE, "rowsums without a finished gradient phase"); }
        if(s->bandlog_pending) {
                launch_band_log(s, 0);
                s->bandlog_pending = false;
        }
        if(!s->rowsums_pending) { return J2P_OK; }     // whole-canvas solver: folded into the norm kernel
        launch_rowsums(s);
        HIP_TRY(hipGetLastError());
        s->rowsums_pending = false;
        return J2P_OK;
}

// the instantiations of k_project by what the launch needs: NT = non-temporal level (nt_policy), NIP = who reduces
// ||g|| (0: read from memory, 1: every wavefront, 2: the workgroup's first wavefront — bands), PTR = pointer form of the
// row loads (rows >= 64 KiB apart)
template <int NT, int NIP>
void launch_project_unit_nt(bool ptr, dim3 grid, hipStream_t st, const ProjArgs &a)
{
        if(ptr) { hipLaunchKernelGGL((k_project<false, 1, 1, NT, NIP, true>), grid, dim3(256), 0, st, a); }
        else { hipLaunchKernelGGL((k_project<false, 1, 1, NT, NIP, false>), grid, dim3(256), 0, st, a); }
}
void launch_project_unit(int nt, int nip, bool ptr, dim3 grid, hipStream_t st, const ProjArgs &a)
{
        if(nip == 2) {
                switch(nt) {
                case 0: launch_project_unit_nt<0, 2>(ptr, grid, st, a); break;
                case 1: launch_project_unit_nt<1, 2>(ptr, grid, st, a); break;
                case 2: launch_project_unit_nt<2, 2>(ptr, grid, st, a); break;
                default: launch_project_unit_nt<3, 2>(ptr, grid, st, a); break;
                }
        } else {
                switch(nt) {
                case 0: launch_project_unit_nt<0, 0>(ptr, grid, st, a); break;
                case 1: launch_project_unit_nt<1, 0>(ptr, grid, st, a); break;
                case 2: launch_project_unit_nt<2, 0>(ptr, grid, st, a); break;
                default: launch_project_unit_nt<3, 0>(ptr, grid, st, a); break;
                }
        }
}
// every other case by sampling class: logging, subsampled channels, the per-wavefront tree of small canvases
template <bool LOG, int NIP>
void launch_project_sampled(unsigned ws, unsigned hs, dim3 grid, hipStream_t st, const ProjArgs &a)
{
        if(ws == 1 && hs == 1) { hipLaunchKernelGGL((k_project<LOG, 1, 1, 0, NIP>), grid, dim3(256), 0, st, a); }
        else if(ws == 2 && hs == 2) { hipLaunchKernelGGL((k_project<LOG, 2, 2, 0, NIP>), grid, dim3(256), 0, st, a); }
        else if(ws == 2 && hs == 1) { hipLaunchKernelGGL((k_project<LOG, 2, 1, 0, NIP>), grid, dim3(256), 0, st, a); }
        else if(ws == 1 && hs == 2) { hipLaunchKernelGGL((k_project<LOG, 1, 2, 0, NIP>), grid, dim3(256), 0, st, a); }
        else { hipLaunchKernelGGL((k_project<LOG, 0, 0, 0, NIP>), grid, dim3(256), 0, st, a); }
}

// part: 0 = whole phase; J2P_PROJECT_BOUNDARY (1) = norm + the band's first and last block row of every
// channel (they hold the rows the neighbours need); J2P_PROJECT_INTERIOR (2) = the rest, ends the iteration
int do_phase_project(j2p_solver *s, bool log, int part = 0)
{
        const hipStream_t st = s->stream;
        if(!s->grad_done) { return fail(J2P_ESTATE, "phase_project called before phase_gradient"); }
        if(s->rowsums_pending) { return fail(J2P_ESTATE, "phase_project before j2p_solver_phase_rowsums"); }
        // the two phases of an iteration must agree on logging: where the norm is reduced depends on it
        if(log != s->phase_log) { return fail(J2P_ESTATE, "phase_project: logging differs from this iteration's gradient phase"); }
        if(part == 2 && !s->proj_boundary_done) { return fail(J2P_ESTATE, "interior part of phase_project before the boundary part"); }
        if(part != 2 && s->proj_boundary_done) { return fail(J2P_ESTATE, "boundary part of phase_project issued twice"); }
        unsigned P = 1;
        while(P < s->ntr_global) { P <<= 1; }
        // the global [tile row][channel] sums a band solver finishes ||g|| from: gathered by the caller, or — linked
        // bands — stored there by every band's gradient launch, even and odd iterations in two arrays
        const double *global_rows = s->whole ? rowsums_of(s, s->iter) : ((s->linked && (s->iter & 1)) ? s->rowsum_all_odd : s->rowsum_all);
        bool nip2 = false;
        if(part == 2 || s->norm_ready || s->norm_by_project) {
                // the norm is already there, or every wavefront of k_project reduces the row sums itself
        } else if(!s->whole && s->band_nip && part == 0 && s->ntr_global <= kWaveTreeMax) {
                // band solvers: the first wavefront of every workgroup of k_project runs the tree (NIP 2) — no launch
                nip2 = true;
        } else if(s->fold) {
                // level 1 came with the gradient launch (band solvers: the caller has gathered all bands' row sums)
                hipLaunchKernelGGL(k_norm_finish, dim3(s->nch), dim3(256), P * sizeof(double), st,
                                   global_rows, s->ntr_global, s->nch, s->norm);
        } else if(s->whole) {
                // stage as many of the partials at once as the CU's LDS holds (P <= 4096)
                unsigned stage = 0;                                  // narrow canvases: direct form (4.6 vs 5.4 us at 4096^2)
                if(s->ntx > 48) {
                        stage = s->ntr_local * s->ntx;
                        if((P + stage) * sizeof(double) > kNormLdsBytes) { stage = kNormLdsBytes / sizeof(double) - P; }
                }
                hipLaunchKernelGGL(k_norm_whole, dim3(s->nch), dim3(256), (P + stage) * sizeof(double), st,
                                   (const double *)s->part_g2, s->ntx, s->ntr_local, s->nch, s->norm, stage);
        } else {
                hipLaunchKernelGGL(k_norm_finish, dim3(s->nch), dim3(256), P * sizeof(double), st,
                                   global_rows, s->ntr_global, s->nch, s->norm);
        }
        ProjArgs a;
        for(unsigned c = 0; c < s->nch; c++) { a.ch[c] = chan_dev(s, c); }
        a.geo = geo_of(s);
        a.factor = s->factor;
        const float radius = sqrtf((float)s->H * (float)s->W) / 2;             // compute.c:425
        a.step = radius / sqrtf((float)(1 + s->iterations));                    // compute.c:443
        a.norm = s->norm;
        a.part_prob = s->part_prob;
        a.strips_per_chan = s->strips_stride;
        a.norm_rowsums = nip2 ? global_rows : (s->norm_by_project ? rowsums_of(s, s->iter) : nullptr);
        a.norm_rows = s->ntr_global;
        a.norm_nch = s->nch;
        const int nip = nip2 ? 2 : (s->norm_by_project ? s->nip_form : 0);
        for(unsigned c = 0; c < kMaxCh; c++) { a.halo_up[c] = a.halo_down[c] = nullptr; }
        if(s->linked) {
                // the band's edge rows of x_{k+1} also go into the neighbours' halo rows of the buffer being written
                // (only the NIP 2 instantiations store them: a linked band always takes those)
                if(!nip2) { return fail(J2P_ESTATE, "linked bands: ||g|| must be reduced inside k_project (whole phases, at most %u tile rows, J2P_BAND_NIP not 0)", kWaveTreeMax); }
                for(unsigned c = 0; c < s->nch; c++) {
                        a.halo_up[c] = s->links.up_halo[s->cur ^ 1][c];
                        a.halo_down[c] = s->links.down_halo[s->cur ^ 1][c];
                }
        }
        if(part != 1) { mark(s); }
        const bool inwave_nt_off = s->nch > 1 && s->joint_inwave;      // those gradient kernels have no non-temporal form
        auto block_rows = [&](unsigned hs, unsigned z) -> unsigned {
                const unsigned brows = (s->rows + 8 * hs - 1) / (8 * hs);
                if(part == 0) { a.by_offset[z] = 0; a.by_mul[z] = 1; a.nby[z] = brows; }
                else if(part == 1) { a.by_offset[z] = 0; a.by_mul[z] = brows > 1 ? brows - 1 : 1; a.nby[z] = brows < 2 ? brows : 2; }
                else { a.by_offset[z] = 1; a.by_mul[z] = 1; a.nby[z] = brows > 2 ? brows - 2 : 0; }
                return a.nby[z];
        };
        bool mixed = false;
        for(unsigned c = 1; c < s->nch; c++) { mixed = mixed || s->ch[c].ws != s->ch[0].ws || s->ch[c].hs != s->ch[0].hs; }
        if(mixed && s->mixed_project && nip != 2 && (size_t)s->W * s->rows <= kMixedProjectPixels) {
                // (k_project_mixed has the per-wavefront tree only; canvases this small take that form anyway)
                // small canvas, several samplings: one launch for all channels (k_project_mixed)
                unsigned max_strips = 0;
                for(unsigned c = 0; c < s->nch; c++) {
                        a.chan_of_z[c] = c;
                        const unsigned ws = s->ch[c].ws, hs = s->ch[c].hs;
                        const unsigned strips = ((s->W + 64 * ws - 1) / (64 * ws)) * block_rows(hs, c);
                        if(strips > max_strips) { max_strips = strips; }
                }
                if(max_strips) {
                        const dim3 grid((max_strips + 3) / 4, 1, s->nch);
                        if(log) { hipLaunchKernelGGL((k_project_mixed<true, false>), grid, dim3(256), 0, st, a); }
                        else if(nip == 1) { hipLaunchKernelGGL((k_project_mixed<false, true>), grid, dim3(256), 0, st, a); }
                        else { hipLaunchKernelGGL((k_project_mixed<false, false>), grid, dim3(256), 0, st, a); }
#ifdef J2P_TRACE
                        if(s->trace_on) { s->trace_used += grid.x * grid.z * 4; }
#endif
                }
        } else {
        // one launch per sampling class present (usually: luma 1x1, both chroma 2x2)
        bool done[kMaxCh] = {false, false, false};
        for(unsigned c0 = 0; c0 < s->nch; c0++) {
                if(done[c0]) { continue; }
                const unsigned ws = s->ch[c0].ws, hs = s->ch[c0].hs;
                unsigned nz = 0;
                for(unsigned c = c0; c < s->nch; c++) {
                        if(!done[c] && s->ch[c].ws == ws && s->ch[c].hs == hs) {
                                block_rows(hs, nz);
                                a.chan_of_z[nz++] = c;
                                done[c] = true;
                        }
                }
                if(a.nby[0] == 0) { continue; }
                const unsigned strips = ((s->W + 64 * ws - 1) / (64 * ws)) * a.nby[0];
                dim3 grid((strips + 3) / 4, 1, nz);
                if(ws == 1 && hs == 1 && !log && nip != 1) {
                        // the 1x1 instantiations: g is read non-temporally exactly when the gradient launch wrote it that
                        // way (nt levels, nt_policy); rows >= 64 KiB apart take the pointer form of the 24 loads (see
                        // project_strip); bands reduce ||g|| themselves (NIP 2)
                        const bool far_rows = (size_t)s->W * sizeof(float) >= 65536;
                        const int nt = inwave_nt_off ? 0 : s->nt;
                        launch_project_unit(nt, nip, far_rows, grid, st, a);
                }
                else if(log && nip == 2) { launch_project_sampled<true, 2>(ws, hs, grid, st, a); }
                else if(log) { launch_project_sampled<true, 0>(ws, hs, grid, st, a); }
                else if(nip == 2) { launch_project_sampled<false, 2>(ws, hs, grid, st, a); }
                else if(nip == 1) { launch_project_sampled<false, 1>(ws, hs, grid, st, a); }
                else { launch_project_sampled<false, 0>(ws, hs, grid, st, a); }
#ifdef J2P_TRACE
                if(s->trace_on) {
                        s->trace_used += grid.x * grid.z * 4;
                        a.geo.trace_base = s->trace_used;       // the next sampling class's launch
                }
#endif
        }
        }
        if(part != 1) { mark(s); }
        HIP_TRY(hipGetLastError());
        if(part == 1) {
                s->proj_boundary_done = true;
                return J2P_OK;
        }
        s->proj_boundary_done = false;
        if(log && s->log_phases) { launch_band_log(s, 1); }
        s->cur ^= 1;        // SWAP(fdata, fista) of compute.c:438: the buffer just written is x_{k+1}
        s->iter++;
        s->grad_done = false;
        return J2P_OK;
}

int upload(void *dst, const void *src, size_t bytes, hipStream_t st)
{
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return J2P_OK;
}

int launch_init(j2p_solver *s)
{
        for(unsigned c = 0; c < s->nch; c++) {
                const ChanHost &h = s->ch[c];
                ChanDev k = chan_dev(s, c);
                k.crow0 = h.frow0;          // the decoded input has its own row window
                const int fill_halo = s->band_local ? 0 : 1;
                hipLaunchKernelGGL(k_init_state, dim3(2048), dim3(256), 0, s->stream, k, geo_of(s),
                                   (const float *)h.decoded, fill_halo);
                if(h.crows) {
                        hipLaunchKernelGGL(k_fill_zero, dim3(1024), dim3(256), 0, s->stream, h.pg, (size_t)h.crows * h.cw);
                }
        }
        HIP_TRY(hipGetLastError());
        // the partials of a folding gradient launch carry the iteration's parity in their sign bit (fold_tile_row): what
        // the slots hold before iteration 0 must carry the other one — all bits set
        HIP_TRY(hipMemsetAsync(s->part_g2, 0xff, (size_t)s->ntx * s->ntr_local * s->nch * sizeof(double), s->stream));
        s->iter = 0;
        s->t = 1.f;
        s->cur = 0;
        s->grad_done = false;
        s->interior_done = false;
        s->rowsums_pending = false;
        s->norm_ready = false;
        s->norm_by_project = false;
        s->proj_boundary_done = false;
        s->bandlog_pending = false;
        for(unsigned c = 0; c < kMaxCh; c++) { s->carried_prob[c] = 0.; }
        s->carried_valid = true;
        return J2P_OK;
}

}  // namespace

extern "C" {

const char *j2p_version(void) { return "jpeg2png_amd 0.1 (gfx950)"; }
const char *j2p_last_error(void) { return g_err; }
void j2p_set_last_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }   // (for compute_host.c; not in the header)

// test hook: the n-th j2p_solver_run / j2p_tiled_run from now on fails with J2P_EDEVICE before it queues anything
// (0 = disarm) — how tests/test_capi_gpu.py makes a solve fail after create
static std::atomic<int> g_fail_run{0}, g_fail_band{0};
void j2p_debug_fail_run_after(int n)
{
        g_fail_run.store(n > 0 ? n : 0);
        g_fail_band.store(n < 0 ? -n : 0);
}

int j2p_device_count(int *count)
{
        if(!count) { return fail(J2P_EINVAL, "count is NULL"); }
        int n = 0;
        if(hipGetDeviceCount(&n) != hipSuccess) { n = 0; }
        *count = n;
        return J2P_OK;
}

void j2p_solver_destroy(j2p_solver *s)
{
        if(!s) { return; }
        DeviceGuard guard(s->device);
        if(s->stream) { (void)hipStreamSynchronize(s->stream); }
        if(s->live_registered) { live_add(s->device, LiveBytes{s->live_ws, s->live_g, s->live_planes, s->live_d}, -1); }
        pool_give(s->device, s->arena, s->arena_bytes);
        (void)hipFree(s->logsums);
        (void)hipFree(s->log_band);
        (void)hipFree(s->trace);
        for(hipEvent_t e : s->ev) { (void)hipEventDestroy(e); }
        if(s->own_stream && s->stream) { (void)hipStreamDestroy(s->stream); }
        delete s;
}

void j2p_pool_trim(void) { pool_drop_all(); }

int j2p_solver_create(j2p_solver **out, int device, void *stream, unsigned nchannel, const j2p_plane planes[],
                      float weight, const float pweight[], unsigned iterations, j2p_band band, int band_local_arrays)
{
        if(!out || !planes || !pweight) { return fail(J2P_EINVAL, "NULL argument"); }
        *out = nullptr;
        if(nchannel == 0 || nchannel > kMaxCh) { return fail(J2P_EINVAL, "nchannel must be 1..3 (compute.c:118), got %u", nchannel); }
        unsigned W = 0, H = 0, align = (unsigned)J2P_TILE_ROWS;
        for(unsigned c = 0; c < nchannel; c++) {
                const j2p_plane &p = planes[c];
                if(p.w == 0 || p.h == 0 || (p.w & 7) || (p.h & 7)) {
                        return fail(J2P_EINVAL, "channel %u: coefficient plane %ux%u is not a positive multiple of 8 (box.c:6-7)", c, p.w, p.h);
                }
                if(p.w_samp == 0 || p.h_samp == 0) { return fail(J2P_EINVAL, "channel %u: zero sampling factor", c); }
                if(!p.data || !p.quant_table) { return fail(J2P_EINVAL, "channel %u: data/quant_table is NULL", c); }
                for(int j = 0; j < 64; j++) {
                        if(p.quant_table[j] == 0) { return fail(J2P_EINVAL, "channel %u: invalid quantization table (jpeg.c:41-45)", c); }
                }
                if(p.w * p.w_samp > W) { W = p.w * p.w_samp; }     // compute.c:410-416
                if(p.h * p.h_samp > H) { H = p.h * p.h_samp; }
                align = lcm_u(align, 8 * p.h_samp);
        }
        if(H > (unsigned)kMaxTileRows * kTY) { return fail(J2P_EINVAL, "canvas height %u exceeds %u", H, kMaxTileRows * kTY); }   // (shorter tile rows: only far below)
        bool whole = band.row_begin == 0 && (band.row_end == 0 || band.row_end >= H);
        if(whole && (band_local_arrays & J2P_BAND_EVEN_IF_WHOLE) && band.row_end >= H) { whole = false; band.row_end = H; }
        unsigned row0 = whole ? 0 : band.row_begin, row1 = whole ? H : band.row_end;
        if(!whole) {
                if(row0 >= row1 || row1 > H) { return fail(J2P_EINVAL, "bad band [%u,%u) for canvas height %u", row0, row1, H); }
                if(row0 % align || (row1 % align && row1 != H)) {
                        return fail(J2P_EINVAL, "band [%u,%u) must be aligned to %u rows", row0, row1, align);
                }
        }
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
                return fail(J2P_EDEVICE, "no HIP device available: the jpeg2png_amd solver has no CPU fallback");
        }
        if(device < 0 || device >= ndev) { return fail(J2P_EINVAL, "device %d out of range (0..%d)", device, ndev - 1); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }

        // k_norm_whole (only the A/B baseline of the folded reduction, J2P_OPT_NORM_FOLD = 0) stages the norm
        // partials in up to 156 KiB of dynamic LDS (per device: idempotent)
        if(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_norm_whole), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)kNormLdsBytes) != hipSuccess) {
                return fail(J2P_EDEVICE, "hipFuncSetAttribute(k_norm_whole, %u bytes of LDS) failed", kNormLdsBytes);
        }
        j2p_solver *s = new(std::nothrow) j2p_solver();
        if(!s) { return fail(J2P_ENOMEM, "host allocation failed"); }
        s->device = device;
        s->nch = nchannel;
        s->W = W;
        s->H = H;
        s->row0 = row0;
        s->rows = row1 - row0;
        s->whole = whole;
        // the in-kernel reduction is worth its serial tail only for band solvers, where it replaces a launch AND lets
        // the row sums alternate between two buffers (measured on whole canvases: 4096^2 140.0 us per iteration either
        // way, 512^2 4:2:0 42.0 vs 40.3 us — the tail costs what the k_norm_whole launch did)
        s->fold = !whole;
        // ... except on whole canvases small enough to be bound by the number of dependent launches: there the
        // gradient kernel leaves the per-tile-row sums and every wavefront of k_project runs the final tree itself
        // (512x512 4:2:0: 27.7 -> 27.2 us per iteration, 1024^2 Y: 22.1 -> 21.7, 1080p and 1536^2 Y: equal;
        // 2048^2: 48.9 -> 50.8, 4096^2: 130 -> 134, hence the limit)
        if(whole && (size_t)W * H <= kNormInProjectPixels) {
                s->fold = true;
                s->norm_in_project = true;
        }
        // ... and on whole canvases large enough for the reduction launch to cost more than the fold (kFoldWholePixels)
        if(whole && (size_t)W * H >= kFoldWholePixels && (H + kTY - 1) / kTY <= kFoldMaxRows) { s->fold = true; }
        s->band_local = !whole && (band_local_arrays & J2P_BAND_LOCAL_ARRAYS) != 0;
        s->weight = weight;
        s->iterations = iterations;
        {
                // the one schedule switch tests reach through the environment (read once, here): all channels of a
                // joint image inside one wavefront instead of one wavefront per channel — same bits, slower
                const char *env = j2p_exp_env("J2P_JOINT_INWAVE");
                s->joint_inwave = env && atoi(env) != 0;
                // ... and: one projection launch per sampling class also on small canvases (J2P_OPT_MIXED_PROJECT)
                env = j2p_exp_env("J2P_MIXED_PROJECT");
                if(env) { s->mixed_project = atoi(env) != 0; }
                // ... and (A/B timing): band solvers finish ||g|| with a k_norm_finish launch instead of inside k_project
                env = j2p_exp_env("J2P_BAND_NIP");
                if(env) { s->band_nip = atoi(env) != 0; }
        }
        int rc = J2P_OK;
#define CREATE_TRY(expr)                                                                           \
        do {                                                                                       \
                hipError_t e_ = (expr);                                                            \
                if(e_ != hipSuccess) {                                                             \
                        rc = fail(e_ == hipErrorOutOfMemory ? J2P_ENOMEM : J2P_EDEVICE,            \
                                  "%s failed: %s", #expr, hipGetErrorString(e_));                  \
                        j2p_solver_destroy(s);                                                     \
                        return rc;                                                                 \
                }                                                                                  \
        } while(0)
        if(stream) { s->stream = (hipStream_t)stream; }
        else {
                CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
                s->own_stream = true;
        }

        // ---- geometry of every buffer ----
        for(unsigned c = 0; c < nchannel; c++) {
                const j2p_plane &p = planes[c];
                ChanHost &h = s->ch[c];
                h.cw = p.w; h.ch = p.h; h.ws = p.w_samp; h.hs = p.h_samp;
                h.pweight = pweight[c];
                // coefficient rows of the band (block aligned because the band is)
                unsigned c0 = row0 / h.hs, c1 = (row1 + h.hs - 1) / h.hs;
                if(c0 > h.ch) { c0 = h.ch; }
                if(c1 > h.ch) { c1 = h.ch; }
                h.crow0 = c0;
                h.crows = c1 - c0;
                // rows of the decoded input the init kernel touches (own rows + halo, clamped like compute.c:298)
                if(s->band_local) {
                        if(h.ch * h.hs < H) {
                                rc = fail(J2P_EINVAL, "band-local arrays need every channel to cover the canvas height");
                                j2p_solver_destroy(s);
                                return rc;
                        }
                        h.frow0 = h.crow0;
                        h.frows = h.crows;
                } else {
                        const unsigned y0 = row0 >= (unsigned)kHalo ? row0 - kHalo : 0;
                        const unsigned y1 = row1 + kHalo < H ? row1 + kHalo : H;
                        unsigned f0 = y0 / h.hs, f1 = (y1 - 1) / h.hs + 1;
                        if(f0 > h.ch - 1) { f0 = h.ch - 1; }
                        if(f1 > h.ch) { f1 = h.ch; }
                        if(f1 <= f0) { f1 = f0 + 1; }
                        h.frow0 = f0;
                        h.frows = f1 - f0;
                }
        }
        // reductions: tile rows are counted on the canvas, the band owns a contiguous range
        // Gradient strips: columns per lane (px) and rows per strip = rows per norm partial ("tile row", rpw).
        // A canvas that fills the chip: 128-column strips (two columns per lane, packed arithmetic) of 16 rows (32 / 48 / 64
        // measured no faster).  A smaller canvas leaves wavefront slots empty and is bound by how long ONE
        // wavefront takes to walk its rows (wave timelines, profiles/r03_wave_trace.jsonl: ~0.9 us per row trip whatever
        // the SIMD's load), so it gets shorter strips (8 or 4 rows: fewer trips per wavefront).  64-column strips (one
        // column per lane: the same kernel instantiated on float instead of float2) exist behind J2P_PX=1 and do not
        // pay (kPx1Waves).  Functions of the CANVAS only (never of the band), so that every band of a tiled run — and
        // the whole-canvas solver — reduce ||g|| over the same partials in the same order.
        {
                auto strips = [&](unsigned px) { return W <= 4 ? 1u : (W - 4 + (64 * px - 4) - 1) / (64 * px - 4); };
                auto waves = [&](unsigned px, unsigned g) { return (unsigned long long)strips(px) * nchannel * ((H + g - 1) / g); };
                unsigned px = 2, g = kTY;
                // (limits measured, profiles/r03_px_rpw_sweep.jsonl)
                if(waves(2, kTY) < kPx1Waves && !s->joint_inwave) { px = 1; }
                if(waves(px, g) < kHalfStripWaves) { g = 8; }
                if(g == 8 && waves(px, g) < kShortStripWaves) { g = 4; }
                // (timing experiments: J2P_PX = 1 / 2, J2P_RPW = 2 ... 64 for every solver of the process; band solvers take
                // only the divisors of the band alignment, 16 — tools/rpw_fine.py sweeps the rest on whole canvases)
                if(const char *env = j2p_exp_env("J2P_PX")) {
                        const int v = atoi(env);
                        if((v == 1 && !s->joint_inwave) || v == 2) { px = (unsigned)v; }
                }
                if(const char *env = j2p_exp_env("J2P_RPW")) {
                        const int v = atoi(env);
                        if(v >= 2 && v <= 64 && (s->whole || kTY % v == 0)) { g = (unsigned)v; }
                }
                s->px = px;
                s->rpw = g;
                s->ntx = strips(px);
                // The LAST wavefronts of a gradient launch march half and quarter tile rows (grad_item): a launch ends with
                // its last wavefront, and a whole 16-row item dispatched last keeps a few SIMDs busy for a wavefront life
                // (17 us of 53 at 4096^2, profiles/r06_wave_trace.jsonl) while the rest of the chip drains.  Shares in
                // 1/256 of every XCD's run; one channel per workgroup wavefront.  Who marches a row never changes a bit
                // (march_rows), so the choice may depend on the BAND: measured (profiles/r06_zones_mid_sizes.jsonl,
                // r06_zones_by_size.jsonl; us per iteration without / with) 1080p 30.1 / 28.7, 2048^2 45.5 / 42.2,
                // 4096x2048 70.9 / 67.8, 4096x3072 94.2 / 92.0, 4096^2 120.0 / 118.7; nothing from three wavefront
                // generations on (8192x4096 235.3 / 235.9, 16384x2048 230.8 / 231.1, 8192^2 515.7 / 515.8).
                const unsigned long long launch_waves = (unsigned long long)strips(px) * ((s->rows + g - 1) / g);
                if(nchannel == 1 && g >= 8 && launch_waves < kZoneMaxWaves) {
                        s->zone_b = kZoneB;
                        s->zone_c = g >= 16 ? kZoneC : 0;
                } else if(nchannel == 1 && g >= 16) {
                        // ... and from three generations on the FIRST workgroups march two tile rows at once (34 row trips for
                        // 32 rows: fewer source rows recomputed and re-read), the tail shares smaller: 16384x2048 229.8 -> 227.0
                        // us per iteration, 8192^2 476.0 -> 471.1; below that doubles cost more at the end of the launch than they
                        // save (2048^2 43.2 -> 46.0, 4096x2048 69.3 -> 72.4, 4096^2 +-0: profiles/r06_doubles.jsonl)
                        s->zone_d = kBigZoneD;
                        s->zone_b = kBigZoneB;
                        s->zone_c = kBigZoneC;
                }
                // (halves need two groups of four rows per tile row, quarters four: march_rows)
                if(const char *env = j2p_exp_env("J2P_ZONE_B")) { if(nchannel == 1 && g >= 8) { s->zone_b = (unsigned)atoi(env); } }
                if(const char *env = j2p_exp_env("J2P_ZONE_C")) { if(nchannel == 1 && g >= 16) { s->zone_c = (unsigned)atoi(env); } }
                if(const char *env = j2p_exp_env("J2P_ZONE_D")) { if(nchannel == 1 && g >= 8) { s->zone_d = (unsigned)atoi(env); } }
                if(s->zone_b > 256) { s->zone_b = 256; }
                if(s->zone_b + s->zone_c > 256) { s->zone_c = 256 - s->zone_b; }
                if(s->zone_d + s->zone_b + s->zone_c > 256) { s->zone_d = 256 - s->zone_b - s->zone_c; }
        }
        s->nseg = (s->rows + s->rpw - 1) / s->rpw;
        s->ntr_local = s->nseg;
        s->ntr_global = (H + s->rpw - 1) / s->rpw;
        s->first_tr = row0 / s->rpw;
        const size_t ntiles = (size_t)s->ntx * s->ntr_local;
        unsigned max_strips = 0;
        for(unsigned c = 0; c < nchannel; c++) {
                const ChanHost &h = s->ch[c];
                const unsigned strips = ((W + 64 * h.ws - 1) / (64 * h.ws)) * ((s->rows + 8 * h.hs - 1) / (8 * h.hs));
                if(strips > max_strips) { max_strips = strips; }
        }
        s->strips_stride = max_strips;
        // ---- one arena for everything (sizes first, then the pointers) ----
        const size_t plane_floats = (size_t)(s->rows + 2 * kHalo) * W;
        float *q_all = nullptr;
        Carver carve;
        for(int pass = 0; pass < 2; pass++) {
                if(pass == 1) {
                        CREATE_TRY(pool_take(device, carve.used + 256, &s->arena, &s->arena_bytes));
                        carve.base = static_cast<char *>(s->arena);
                        carve.used = 0;
                }
                for(unsigned c = 0; c < nchannel; c++) {
                        ChanHost &h = s->ch[c];
                        carve.take(h.xbuf[0], plane_floats);
                        carve.take(h.xbuf[1], plane_floats);
                        carve.take(h.grad, (size_t)s->rows * W);
                        // the prob state always has at least one (zero) row: the gradient kernel reads it unconditionally
                        carve.take(h.pg, (size_t)(h.crows ? h.crows : 1) * h.cw);
                        carve.take(h.decoded, (size_t)h.frows * h.cw);
                        carve.take(h.d, (size_t)(h.crows ? h.crows : 1) * h.cw);
                        carve.take(h.d8, (size_t)(h.crows ? h.crows : 1) * h.cw);
                        // device-side decode of a band's input window (own rows + halo rows, rounded out to whole
                        // block rows) normally borrows the two x buffers as scratch; a band of only a few rows is
                        // smaller than that window, and gets scratch of its own
                        if(!planes[c].fdata) {
                                const size_t cells = (size_t)((h.frow0 + h.frows + 7) / 8 - h.frow0 / 8) * 8 * h.cw;
                                if(cells > plane_floats) {
                                        carve.take(h.scratch_f, cells);
                                        carve.take(h.scratch_d, cells);
                                }
                        }
                }
                carve.take(q_all, 64 * kMaxCh);
                carve.take(s->part_g2, ntiles * nchannel);
                carve.take(s->rowsum_local, (size_t)s->ntr_local * nchannel);
                if(whole) {
                        s->rowsum_all = s->rowsum_local;
                } else {
                        carve.take(s->rowsum_all, (size_t)s->ntr_global * nchannel);
                        carve.take(s->rowsum_all_odd, (size_t)s->ntr_global * nchannel);
                        carve.take(s->push_dev, 2);
                        carve.take(s->rowsum_odd, (size_t)s->ntr_local * nchannel);
                }
                carve.take(s->norm, kMaxCh);
                carve.take(s->tickets, (size_t)s->ntr_local + 1);
                carve.take(s->dbg_counters, 3);
                carve.take(s->d_maxabs, kMaxCh);
                carve.take(s->part_tv, ntiles * 2);
                carve.take(s->part_prob, (size_t)max_strips * nchannel);
        }

        // ---- nt_policy (see nt_policy() above): this solver's bytes join the device's live total ----
        {
                for(unsigned c = 0; c < nchannel; c++) {
                        const ChanHost &h = s->ch[c];
                        const size_t cells = (size_t)(h.crows ? h.crows : 1) * h.cw;
                        s->live_ws += (2 * plane_floats + (size_t)s->rows * W + cells) * sizeof(float) + cells * sizeof(int16_t);
                        s->live_planes += 2 * plane_floats * sizeof(float);
                        s->live_d += cells * sizeof(int16_t);
                }
                s->live_g = (size_t)nchannel * s->rows * W * sizeof(float);
                live_add(device, LiveBytes{s->live_ws, s->live_g, s->live_planes, s->live_d}, +1);
                s->live_registered = true;
                s->nt = nt_policy(s);
                // canvases whose planes x_k, x_{k-1} do not both fit the Infinity Cache: the gradient phase walks bottom-up
                // (k_project walks top-down), so that each phase starts on the rows the one before touched last (Geo::reverse).
                // Schedule only: the bits do not depend on who marches a row when.
                s->grad_reverse = s->live_planes > kNtWorkingSet;
                if(const char *env = j2p_exp_env("J2P_GRAD_REVERSE")) { s->grad_reverse = atoi(env) != 0; }
        }

        // ---- uploads (host arrays: whole-image unless band_local) ----
        float qf[64 * kMaxCh];
        for(unsigned c = 0; c < nchannel; c++) {
                s->ch[c].q = q_all + 64 * c;
                for(int j = 0; j < 64; j++) { qf[64 * c + j] = (float)planes[c].quant_table[j]; }
        }
        CREATE_TRY(hipMemcpyAsync(q_all, qf, sizeof(float) * 64 * nchannel, hipMemcpyHostToDevice, s->stream));
        CREATE_TRY(hipMemsetAsync(s->tickets, 0, ((size_t)s->ntr_local + 1) * sizeof(unsigned), s->stream));
        CREATE_TRY(hipMemsetAsync(s->part_prob, 0, (size_t)max_strips * nchannel * sizeof(double), s->stream));
        CREATE_TRY(hipMemsetAsync(s->dbg_counters, 0, 3 * sizeof(unsigned long long), s->stream));
        CREATE_TRY(hipMemsetAsync(s->d_maxabs, 0, kMaxCh * sizeof(unsigned), s->stream));
        for(unsigned c = 0; c < nchannel; c++) {
                const j2p_plane &p = planes[c];
                ChanHost &h = s->ch[c];
                const size_t host_row0 = s->band_local ? h.crow0 : 0;
                if(h.crows) {
                        // block-major: coefficient row r lives in block row r/8; rows are block aligned
                        const int16_t *src = p.data + (size_t)(h.crow0 - host_row0) * h.cw;
                        CREATE_TRY(hipMemcpyAsync(h.d, src, (size_t)h.crows * h.cw * sizeof(int16_t), hipMemcpyHostToDevice, s->stream));
                        // ... and once more as bytes, with the largest |d| (decided behind the synchronisation below)
                        const size_t cells = (size_t)h.crows * h.cw;
                        const unsigned blocks = (unsigned)((cells / 8 + 255) / 256);
                        hipLaunchKernelGGL(k_narrow_coefficients, dim3(blocks < 2048 ? (blocks ? blocks : 1) : 2048), dim3(256), 0, s->stream,
                                           (const int16_t *)h.d, h.d8, cells, s->d_maxabs + c);
                }
                if(p.fdata) {
                        const float *src = p.fdata + (size_t)(h.frow0 - host_row0) * h.cw;
                        CREATE_TRY(hipMemcpyAsync(h.decoded, src, (size_t)h.frows * h.cw * sizeof(float), hipMemcpyHostToDevice, s->stream));
                } else {
                        // decode on the device (jpeg.c:83-92 + box.c:5-19): whole block rows [b0, b1) of the input
                        // window.  Band rows are block aligned, so when the window has no halo rows (band_local, or
                        // a whole canvas) the blocks are already in h.d; otherwise the coefficients of the
                        // window go up once more into the (not yet initialised) gradient plane as scratch.
                        const unsigned b0 = h.frow0 / 8, b1 = (h.frow0 + h.frows + 7) / 8;
                        const unsigned nb_rows = b1 - b0;
                        const unsigned groups = ((h.cw / 8 + 7) / 8) * nb_rows;
                        const int16_t *dsrc = nullptr;
                        if(h.crows && b0 * 8 >= h.crow0 && b1 * 8 <= h.crow0 + h.crows) {
                                dsrc = h.d + (size_t)(b0 * 8 - h.crow0) * h.cw;
                        } else {
                                // scratch: the first x buffer holds (rows + 4) * W floats >= the window's int16 data
                                if(!h.scratch_d && (size_t)nb_rows * 8 * h.cw * sizeof(int16_t) > plane_floats * sizeof(float)) {
                                        rc = fail(J2P_EINVAL, "channel %u: decode window does not fit the scratch plane", c);
                                        j2p_solver_destroy(s);
                                        return rc;
                                }
                                int16_t *dtmp = h.scratch_d ? h.scratch_d : reinterpret_cast<int16_t *>(h.xbuf[0]);
                                const int16_t *src = p.data + (size_t)(b0 * 8 - host_row0) * h.cw;
                                CREATE_TRY(hipMemcpyAsync(dtmp, src, (size_t)nb_rows * 8 * h.cw * sizeof(int16_t), hipMemcpyHostToDevice, s->stream));
                                dsrc = dtmp;
                        }
                        if(h.frow0 == b0 * 8 && h.frows == nb_rows * 8) {
                                hipLaunchKernelGGL(k_decode, dim3((groups + 3) / 4), dim3(256), 0, s->stream, dsrc,
                                                   (const float *)h.q, h.decoded, h.cw, nb_rows);
                        } else {
                                // window not block aligned (halo rows of a band): decode into the second x buffer, copy the rows
                                if(!h.scratch_f && (size_t)nb_rows * 8 * h.cw > plane_floats) {
                                        rc = fail(J2P_EINVAL, "channel %u: decode window does not fit the scratch plane", c);
                                        j2p_solver_destroy(s);
                                        return rc;
                                }
                                float *ftmp = h.scratch_f ? h.scratch_f : h.xbuf[1];
                                hipLaunchKernelGGL(k_decode, dim3((groups + 3) / 4), dim3(256), 0, s->stream, dsrc,
                                                   (const float *)h.q, ftmp, h.cw, nb_rows);
                                CREATE_TRY(hipMemcpyAsync(h.decoded, ftmp + (size_t)(h.frow0 - b0 * 8) * h.cw,
                                                          (size_t)h.frows * h.cw * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
                        }
                        CREATE_TRY(hipGetLastError());
                }
        }
        // (the largest |d| per channel comes back on the solver's own stream: a synchronous copy would wait for every
        // blocking stream of the device)
        unsigned maxabs[kMaxCh] = {0};
        CREATE_TRY(hipMemcpyAsync(maxabs, s->d_maxabs, sizeof(maxabs), hipMemcpyDeviceToHost, s->stream));
        // the host arrays (and the stack tables above) may go away as soon as this returns
        CREATE_TRY(hipStreamSynchronize(s->stream));
        {
                // one byte per coefficient where the channel's values allow it: the projection then reads d8 (ChanDev::d8)
                const char *env = j2p_exp_env("J2P_NARROW_COEFFICIENTS");
                for(unsigned c = 0; c < nchannel; c++) {
                        ChanHost &h = s->ch[c];
                        h.narrow_fits = h.crows != 0 && maxabs[c] <= 127;
                        h.narrow = h.narrow_fits && !(env && atoi(env) == 0);
                }
                account_coefficient_bytes(s);
        }
#undef CREATE_TRY
        rc = launch_init(s);
        if(rc != J2P_OK) { j2p_solver_destroy(s); return rc; }
        *out = s;
        return J2P_OK;
}

int j2p_solver_debug_option(j2p_solver *s, int option, int value)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(s->grad_done || s->interior_done) { return fail(J2P_ESTATE, "options change between iterations only"); }
        switch(option) {
        case J2P_OPT_NORM_FOLD:
                if((s->rowsum_alternate || s->linked) && !value) { return fail(J2P_ESTATE, "alternating / pushed row sums need the folded norm reduction"); }
                if(value && !s->fold) {
                        // (the slots must not carry the coming iteration's parity yet, see launch_init)
                        DeviceGuard guard(s->device);
                        HIP_TRY(hipMemsetAsync(s->part_g2, (s->iter & 1) ? 0x00 : 0xff, (size_t)s->ntx * s->ntr_local * s->nch * sizeof(double), s->stream));
                }
                // 0: reduction launch between the phases; 1: folded into k_gradient by tickets
                if(value != 0 && value != 1) { return fail(J2P_EINVAL, "J2P_OPT_NORM_FOLD is 0 or 1"); }
                s->fold = value == 1;
                break;
        case J2P_OPT_JOINT_INWAVE:
#ifndef J2P_EXPERIMENTS
                if(value) { return fail(J2P_ESTATE, "J2P_OPT_JOINT_INWAVE 1 (all channels in one wavefront) exists in the experiments build only"); }
#endif
                if(value && s->px == 1) { return fail(J2P_ESTATE, "the in-wavefront joint kernel has no one-column-per-lane form (set J2P_JOINT_INWAVE=1 before the solver is created)"); }
                s->joint_inwave = value != 0;
                break;
        case J2P_OPT_NORM_IN_PROJECT:
                // 0: off; 1: the per-wavefront tree; 2: the per-workgroup tree; (needs NORM_FOLD)
                s->norm_in_project = value != 0;
                s->nip_form = value == 2 ? 2 : 1;
                break;
        case J2P_OPT_NT_GRADIENT:
                s->nt_forced = value >= 0;                 // negative: back to the policy
                s->nt = value < 0 ? nt_policy(s) : (value > 3 ? 3 : value);
                break;
        case J2P_OPT_MIXED_PROJECT: s->mixed_project = value != 0; break;
        case J2P_OPT_NARROW_COEFFICIENTS:
                for(unsigned c = 0; c < s->nch; c++) { s->ch[c].narrow = value != 0 && s->ch[c].narrow_fits; }
                account_coefficient_bytes(s);
                break;
        default: return fail(J2P_EINVAL, "unknown option %d", option);
        }
        return J2P_OK;
}

int j2p_solver_coefficient_bytes(const j2p_solver *s, unsigned c, unsigned *bytes)
{
        if(!s || !bytes || c >= s->nch) { return fail(J2P_EINVAL, "j2p_solver_coefficient_bytes: bad argument"); }
        *bytes = s->ch[c].narrow ? 1u : 2u;
        return J2P_OK;
}

int j2p_solver_trace(j2p_solver *s, int on, unsigned long long *host_out, unsigned max_records, unsigned *n)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
#ifdef J2P_TRACE
        DeviceGuard guard(s->device);
        constexpr unsigned kCap = 1u << 19;                     // records (16 MiB)
        HIP_TRY(hipStreamSynchronize(s->stream));
        if(!s->trace) {
                HIP_TRY(dev_malloc((void **)&s->trace, (size_t)kCap * 32));
                HIP_TRY(hipMemset(s->trace, 0, (size_t)kCap * 32));
                s->trace_cap = kCap;
        }
        if(host_out && n) {
                unsigned long long count = s->trace_used;
                if(count > s->trace_cap - 1) { count = s->trace_cap - 1; }
                if(count > max_records) { count = max_records; }
                HIP_TRY(hipMemcpy(host_out, s->trace + 4, (size_t)count * 32, hipMemcpyDeviceToHost));
                *n = (unsigned)count;
                HIP_TRY(hipMemset(s->trace, 0, (size_t)(count + 1) * 32));     // start over
                s->trace_used = 0;
        }
        s->trace_on = on != 0;
        return J2P_OK;
#else
        (void)on; (void)host_out; (void)max_records; (void)n;
        return fail(J2P_ESTATE, "not a J2P_TRACE build");
#endif
}

int j2p_experiments_build(void)
{
#ifdef J2P_EXPERIMENTS
        return 1;
#else
        return 0;
#endif
}

int j2p_debug_grad_items(unsigned W, unsigned rows, unsigned rows_per_tile, unsigned channel_wavefronts, unsigned zone_d, unsigned zone_b,
                         unsigned zone_c, int reverse, unsigned *items /* [max][5]: strip, first row, rows, tile row, kind */, unsigned max_items,
                         unsigned *n_items, unsigned *workgroups)
{
        // the map of k_gradient's launch (grad_item) evaluated on the HOST, wavefront by wavefront: no device needed
        if(!items || !n_items || !workgroups || W < 8 || rows == 0 || rows_per_tile == 0 || channel_wavefronts < 1 || channel_wavefronts > 3) {
                return fail(J2P_EINVAL, "bad argument");
        }
        Geo g;
        memset(&g, 0, sizeof(g));
        g.W = W; g.H = rows; g.row0 = 0; g.rows = rows; g.rpw = rows_per_tile;
        g.ntx = W <= 4 ? 1u : (W - 4 + 123) / 124;
        g.seg_off = 0; g.seg_mul = 1;
        g.ntr_launch = (rows + rows_per_tile - 1) / rows_per_tile;
        const unsigned positions = g.ntx * ((g.ntr_launch + 1) / 2);
        const bool joint = channel_wavefronts > 1;
        g.units = joint ? positions : (positions + 3) / 4;
        g.zone_d = zone_d; g.zone_b = zone_b; g.zone_c = zone_c;
        g.reverse = reverse ? 1u : 0u;
        const unsigned nwg = grad_grid(g.units, zone_shares(g));
        *workgroups = nwg;
        unsigned n = 0;
        for(unsigned b = 0; b < nwg; b++) {
                for(int wave = 0; wave < (joint ? 1 : 4); wave++) {
                        StripItem it;
                        const bool ok = joint ? grad_item<3>(g, b, wave, it) : grad_item<1>(g, b, wave, it);
                        if(!ok || !it.active) { continue; }
                        if(n < max_items) {
                                unsigned *o = items + 5 * (size_t)n;
                                const unsigned t1 = (unsigned)(it.t0 + it.nrows) < rows ? (unsigned)(it.t0 + it.nrows) : rows;
                                o[0] = (unsigned)it.wcol; o[1] = (unsigned)it.t0; o[2] = t1 - (unsigned)it.t0; o[3] = it.tr; o[4] = (unsigned)it.kind;
                        }
                        n++;
                }
        }
        *n_items = n;
        return J2P_OK;
}

int j2p_debug_build(void)
{
#ifdef J2P_DEBUG
        return 1;
#else
        return 0;
#endif
}

int j2p_solver_debug_violations(j2p_solver *s, unsigned long long *count, unsigned *site, unsigned long long *offset)
{
        if(!s || !count) { return fail(J2P_EINVAL, "NULL argument"); }
#ifdef J2P_DEBUG
        unsigned long long h[3] = {0, 0, 0};
        DeviceGuard guard(s->device);
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemcpy(h, s->dbg_counters, sizeof(h), hipMemcpyDeviceToHost));
        *count = h[0];
        if(site) { *site = (unsigned)h[1]; }
        if(offset) { *offset = h[2]; }
        return J2P_OK;
#else
        (void)site;
        (void)offset;
        return fail(J2P_ESTATE, "not a J2P_DEBUG build: the address checks are compiled out");
#endif
}

int j2p_solver_canvas(const j2p_solver *s, unsigned *W, unsigned *H)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(W) { *W = s->W; }
        if(H) { *H = s->H; }
        return J2P_OK;
}

int j2p_solver_band(const j2p_solver *s, unsigned *row_begin, unsigned *row_end)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(row_begin) { *row_begin = s->row0; }
        if(row_end) { *row_end = s->row0 + s->rows; }
        return J2P_OK;
}

int j2p_solver_launches_per_iteration(const j2p_solver *s, unsigned *n)
{
        if(!s || !n) { return fail(J2P_EINVAL, "NULL argument"); }
        // (unlogged runs of a whole-canvas solver; logging adds the log kernels and takes the two-launch form)
        if(!s->whole) { *n = 2 + (s->band_nip ? 0u : 1u); }
        else { *n = s->fold ? 2 : 3; }
        return J2P_OK;
}

int j2p_solver_reset(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        int rc = flush_timing(s);
        if(rc != J2P_OK) { return rc; }
        if(!s->nt_forced) { s->nt = nt_policy(s); }       // other solvers may have come or gone on this device
        return launch_init(s);
}

int j2p_solver_phase_gradient(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        return do_phase_gradient(s, s->log_phases);
}

int j2p_solver_phase_gradient_part(j2p_solver *s, int part, void *stream)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        (void)stream;
#ifndef J2P_EXPERIMENTS
        // (measured slower than whole phases wherever tried, DESIGN.md section 10: the release build keeps the entry points, not the schedule)
        (void)part;
        return fail(J2P_ESTATE, "split phases exist in the experiments build only (buildlib.build_experiments)");
#endif
        if(part != J2P_GRADIENT_INTERIOR && part != J2P_GRADIENT_EDGES) { return fail(J2P_EINVAL, "part must be J2P_GRADIENT_INTERIOR or J2P_GRADIENT_EDGES"); }
        DeviceGuard guard(s->device);
        return do_phase_gradient(s, s->log_phases, part, (hipStream_t)stream);
}

int j2p_solver_phase_rowsums(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        return do_rowsums(s);
}

int j2p_solver_phase_project(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        return do_phase_project(s, s->log_phases);
}

int j2p_solver_phase_project_part(j2p_solver *s, int part)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
#ifndef J2P_EXPERIMENTS
        // (measured slower than whole phases wherever tried, DESIGN.md section 10: the release build keeps the entry points, not the schedule)
        (void)part;
        return fail(J2P_ESTATE, "split phases exist in the experiments build only (buildlib.build_experiments)");
#endif
        if(part != J2P_PROJECT_BOUNDARY && part != J2P_PROJECT_INTERIOR) { return fail(J2P_EINVAL, "part must be J2P_PROJECT_BOUNDARY or J2P_PROJECT_INTERIOR"); }
        DeviceGuard guard(s->device);
        return do_phase_project(s, s->log_phases, part);
}

// log rows from per-iteration sums {tv, tv2, prob distance per channel of the state LEFT by the iteration}
// (compute.c:226-272: total_alpha in float, objective in double).  carried[] is the prob distance of the
// state entering the first of the n iterations (0 at iteration 0: cos = d*q) and is updated.
static void rows_from_sums(unsigned nch, float weight, const float *pweight, unsigned n, const double *sums,
                           double *carried, bool carried_valid, j2p_log_row *rows)
{
        constexpr unsigned kRow = 2 + kMaxCh;
        float total_alpha = 0.f;
        for(unsigned c = 0; c < nch; c++) {
                if(pweight[c] != 0.f) { total_alpha += pweight[c] * 2 * 255 * sqrtf(2); }
        }
        total_alpha += nch;
        if(weight != 0.f) { total_alpha += (weight / sqrtf((float)(4 / 2))) * nch; }
        for(unsigned i = 0; i < n; i++) {
                const double *h = &sums[(size_t)i * kRow];
                double prob = 0.;
                for(unsigned c = 0; c < nch; c++) {
                        if(pweight[c] != 0.f) { prob += 0.5 * carried[c]; }   // compute_simd_step.c:61
                }
                if(!carried_valid) { prob = NAN; }
                rows[i].tv = h[0];
                rows[i].tv2 = weight != 0.f ? h[1] : 0.;
                rows[i].prob_dist = prob;
                rows[i].objective = (rows[i].tv + rows[i].tv2 + prob) / total_alpha;
                for(unsigned c = 0; c < nch; c++) { carried[c] = h[2 + c]; }
                carried_valid = true;
        }
}

}  // extern "C"

static bool countdown(std::atomic<int> &a)
{
        int v = a.load(std::memory_order_relaxed);
        while(v > 0) {
                if(a.compare_exchange_weak(v, v - 1)) { return v == 1; }
        }
        return false;
}
bool j2p_injected_failure() { return countdown(g_fail_run); }
bool j2p_injected_band_failure() { return countdown(g_fail_band); }

void j2p_rows_from_sums_carry(unsigned nch, float weight, const float *pweight, unsigned n, const double *sums,
                              double *carried, bool carried_valid, j2p_log_row *rows)
{
        rows_from_sums(nch, weight, pweight, n, sums, carried, carried_valid, rows);
}

extern "C" {

int j2p_log_rows_from_sums(unsigned nchannel, float weight, const float pweight[], unsigned n, const double *sums,
                           j2p_log_row *rows)
{
        if(!pweight || !sums || !rows) { return fail(J2P_EINVAL, "NULL argument"); }
        if(nchannel == 0 || nchannel > kMaxCh) { return fail(J2P_EINVAL, "nchannel must be 1..3"); }
        double carried[kMaxCh] = {0., 0., 0.};
        rows_from_sums(nchannel, weight, pweight, n, sums, carried, true, rows);
        return J2P_OK;
}

int j2p_solver_set_logging(j2p_solver *s, int on)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        // like the schedule switches: the two phases of an iteration have to agree on it (where the norm is reduced
        // depends on it)
        if(s->grad_done || s->interior_done) { return fail(J2P_ESTATE, "logging changes between iterations only"); }
        if(on && !s->log_band) {
                HIP_TRY(dev_malloc((void **)&s->log_band, (2 + kMaxCh) * sizeof(double)));
                HIP_TRY(hipMemsetAsync(s->log_band, 0, (2 + kMaxCh) * sizeof(double), s->stream));
        }
        s->log_phases = on != 0;
        return J2P_OK;
}

int j2p_solver_run(j2p_solver *s, unsigned n, j2p_log_row *rows)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(!s->whole) { return fail(J2P_ESTATE, "j2p_solver_run needs a whole-canvas solver; drive bands with the phase calls"); }
        if(j2p_injected_failure()) { return fail(J2P_EDEVICE, "injected failure (j2p_debug_fail_run_after)"); }
        DeviceGuard guard(s->device);
        const bool log = rows != nullptr;
        constexpr unsigned kRow = 2 + kMaxCh;
        if(log && s->logsums_cap < n) {
                (void)hipFree(s->logsums);
                s->logsums = nullptr;
                s->logsums_cap = 0;
                HIP_TRY(dev_malloc((void **)&s->logsums, (size_t)n * kRow * sizeof(double)));
                s->logsums_cap = n;
        }
        for(unsigned i = 0; i < n; i++) {
                int rc = do_phase_gradient(s, log);
                if(rc != J2P_OK) { return rc; }
                if(log) {
                        hipLaunchKernelGGL(k_log_sums, dim3(1), dim3(256), 0, s->stream, (const double *)s->part_tv,
                                           s->ntx * s->nseg, (const double *)s->part_prob, 0u, s->strips_stride, s->nch,
                                           s->logsums + (size_t)i * kRow, 0);
                }
                rc = do_phase_project(s, log);
                if(rc != J2P_OK) { return rc; }
                if(log) {
                        hipLaunchKernelGGL(k_log_sums, dim3(1), dim3(256), 0, s->stream, (const double *)s->part_tv, 0u,
                                           (const double *)s->part_prob, s->strips_stride, s->strips_stride, s->nch,
                                           s->logsums + (size_t)i * kRow, 1);
                }
                if(s->timing && s->ev_used >= 4096) {
                        rc = flush_timing(s);
                        if(rc != J2P_OK) { return rc; }
                }
        }
        HIP_TRY(hipGetLastError());
        if(log) {
                std::vector<double> host((size_t)n * kRow);
                HIP_TRY(hipMemcpyAsync(host.data(), s->logsums, host.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
                HIP_TRY(hipStreamSynchronize(s->stream));
                float pw[kMaxCh] = {0.f, 0.f, 0.f};
                for(unsigned c = 0; c < s->nch; c++) { pw[c] = s->ch[c].pweight; }
                rows_from_sums(s->nch, s->weight, pw, n, host.data(), s->carried_prob, s->carried_valid, rows);
                s->carried_valid = true;
        } else if(n) {
                s->carried_valid = false;
        }
        return J2P_OK;
}

int j2p_solver_exchange_info(j2p_solver *s, j2p_exchange *info)
{
        if(!s || !info) { return fail(J2P_EINVAL, "NULL argument"); }
        memset(info, 0, sizeof(*info));
        info->partials_local = s->rowsum_local;
        info->local_tile_rows = s->ntr_local;
        info->partials_all = s->rowsum_all;
        info->global_tile_rows = s->ntr_global;
        info->first_tile_row = s->first_tr;
        info->halo_floats = (size_t)kHalo * s->W;
        info->log_local = s->log_band;
        for(unsigned c = 0; c < s->nch; c++) {
                // between the two parts of a split projection phase the rows to exchange are those of the
                // iterate being written (the buffers swap when the interior part is issued)
                float *base = s->ch[c].xbuf[s->proj_boundary_done ? s->cur ^ 1 : s->cur];
                info->recv_top[c] = base;
                info->send_top[c] = base + (size_t)kHalo * s->W;
                info->send_bottom[c] = base + (size_t)s->rows * s->W;
                info->recv_bottom[c] = base + (size_t)(s->rows + kHalo) * s->W;
        }
        return J2P_OK;
}

int j2p_solver_stream(j2p_solver *s, void **stream)
{
        if(!s || !stream) { return fail(J2P_EINVAL, "NULL argument"); }
        *stream = (void *)s->stream;
        return J2P_OK;
}

int j2p_solver_halo_rows(j2p_solver *s, int buffer, j2p_exchange *info)
{
        if(!s || !info || (buffer != 0 && buffer != 1)) { return fail(J2P_EINVAL, "bad argument"); }
        memset(info, 0, sizeof(*info));
        info->halo_floats = (size_t)kHalo * s->W;
        for(unsigned c = 0; c < s->nch; c++) {
                float *base = s->ch[c].xbuf[buffer];
                info->recv_top[c] = base;
                info->send_top[c] = base + (size_t)kHalo * s->W;
                info->send_bottom[c] = base + (size_t)s->rows * s->W;
                info->recv_bottom[c] = base + (size_t)(s->rows + kHalo) * s->W;
        }
        return J2P_OK;
}

int j2p_solver_alternate_rowsums(j2p_solver *s, const double *buffers[2])
{
        if(!s || !buffers) { return fail(J2P_EINVAL, "NULL argument"); }
        if(s->whole || !s->rowsum_odd) { return fail(J2P_ESTATE, "alternating row sums are for band solvers"); }
        if(!s->fold) { return fail(J2P_ESTATE, "alternating row sums need the folded norm reduction"); }
        if(s->linked) { return fail(J2P_ESTATE, "alternating row sums: this solver's bands are linked (its sums go to the global arrays)"); }
        s->rowsum_alternate = true;
        buffers[0] = s->rowsum_local;
        buffers[1] = s->rowsum_odd;
        return J2P_OK;
}

int j2p_solver_global_rowsums(j2p_solver *s, double *arrays[2])
{
        if(!s || !arrays) { return fail(J2P_EINVAL, "NULL argument"); }
        if(s->whole || !s->rowsum_all_odd) { return fail(J2P_ESTATE, "global row sums: band solvers only"); }
        arrays[0] = s->rowsum_all;
        arrays[1] = s->rowsum_all_odd;
        return J2P_OK;
}

int j2p_solver_link_bands(j2p_solver *s, const j2p_band_links *links)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(s->grad_done || s->interior_done) { return fail(J2P_ESTATE, "bands are linked between iterations only"); }
        if(!links) {
                s->linked = false;
                return J2P_OK;
        }
        if(s->whole) { return fail(J2P_ESTATE, "link_bands: band solvers only"); }
        if(!s->fold) { return fail(J2P_ESTATE, "link_bands needs the folded norm reduction (the row sums leave from inside k_gradient)"); }
        if(s->rowsum_alternate) { return fail(J2P_ESTATE, "link_bands: this solver's row sums already alternate for norm_from_bands"); }
        if(links->npush == 0 || links->npush > (unsigned)kMaxBands) { return fail(J2P_EINVAL, "link_bands: 1..%d bands to push to", kMaxBands); }
        if(links->ncount > (unsigned)kMaxBands) { return fail(J2P_EINVAL, "link_bands: at most %d counters", kMaxBands); }
        for(unsigned b = 0; b < links->ncount; b++) {
                if(!links->count[b]) { return fail(J2P_EINVAL, "link_bands: counter %u is NULL", b); }
        }
        bool own[2] = {false, false};
        for(int par = 0; par < 2; par++) {
                for(unsigned b = 0; b < links->npush; b++) {
                        if(!links->push[par][b]) { return fail(J2P_EINVAL, "link_bands: push target %u is NULL", b); }
                        own[par] = own[par] || links->push[par][b] == (par ? s->rowsum_all_odd : s->rowsum_all);
                }
        }
        if(!own[0] || !own[1]) { return fail(J2P_EINVAL, "link_bands: the push lists must contain this solver's own arrays"); }
        for(unsigned c = 0; c < s->nch; c++) {
                // a neighbour is given for both buffers or for neither; the band at the top / bottom of the canvas has none
                const bool up = links->up_halo[0][c] != nullptr, down = links->down_halo[0][c] != nullptr;
                if(up != (links->up_halo[1][c] != nullptr) || down != (links->down_halo[1][c] != nullptr)) {
                        return fail(J2P_EINVAL, "link_bands: channel %u: a neighbour's rows are needed for both x buffers", c);
                }
                if(up != (s->row0 > 0) || down != (s->row0 + s->rows < s->H)) {
                        return fail(J2P_EINVAL, "link_bands: channel %u: neighbours do not match the band's place in the canvas", c);
                }
        }
        // the push lists live in device memory (see GradArgs::push)
        RowsumPush host[2];
        for(int par = 0; par < 2; par++) {
                memset(&host[par], 0, sizeof(host[par]));
                host[par].n = links->npush;
                host[par].first_tr = s->first_tr;
                for(unsigned b = 0; b < links->npush; b++) { host[par].dst[b] = links->push[par][b]; }
                host[par].ncount = links->ncount;
                for(unsigned b = 0; b < links->ncount; b++) { host[par].count[b] = links->count[b]; }
        }
        {
                DeviceGuard guard(s->device);
                HIP_TRY(hipMemcpyAsync(s->push_dev, host, sizeof(host), hipMemcpyHostToDevice, s->stream));
                HIP_TRY(hipStreamSynchronize(s->stream));       // `host` is on the stack
        }
        s->links = *links;
        s->linked = true;
        return J2P_OK;
}

int j2p_solver_norm_from_bands(j2p_solver *s, unsigned nband, const double *const rowsums[], const unsigned first_tile_row[],
                               const unsigned tile_rows[], unsigned nout, float *const norm_out[])
{
        if(!s || !rowsums || !first_tile_row || !tile_rows) { return fail(J2P_EINVAL, "NULL argument"); }
        if(nband == 0 || nband > (unsigned)kMaxBands) { return fail(J2P_EINVAL, "1..%d bands", kMaxBands); }
        if(nout > (unsigned)kMaxBands || (nout && !norm_out)) { return fail(J2P_EINVAL, "norm_from_bands: bad output list"); }
        if(!s->grad_done || s->rowsums_pending) { return fail(J2P_ESTATE, "norm_from_bands needs a finished gradient phase"); }
        // a whole-canvas solver above kNormInProjectPixels reduces its partials in one kernel and never forms the
        // level-1 row sums this call reads
        if(s->whole && !s->fold) { return fail(J2P_ESTATE, "norm_from_bands: this solver leaves no per-tile-row sums (whole canvas, norm folding off)"); }
        DeviceGuard guard(s->device);
        BandRowsums t;
        unsigned covered = 0;
        for(unsigned b = 0; b < nband; b++) {
                if(first_tile_row[b] + tile_rows[b] > s->ntr_global) { return fail(J2P_EINVAL, "band %u: tile rows out of range", b); }
                t.rowsum[b] = rowsums[b];
                t.first[b] = first_tile_row[b];
                t.count[b] = tile_rows[b];
                covered += tile_rows[b];
        }
        if(covered != s->ntr_global) { return fail(J2P_EINVAL, "the bands cover %u of %u tile rows", covered, s->ntr_global); }
        t.nband = nband;
        // where the float norm goes: this solver's own word(s), or the list given (every band's, this one included)
        if(nout == 0) {
                t.out[0] = s->norm;
                t.nout = 1;
        } else {
                bool own = false;
                for(unsigned b = 0; b < nout; b++) {
                        if(!norm_out[b]) { return fail(J2P_EINVAL, "norm_from_bands: output %u is NULL", b); }
                        t.out[b] = norm_out[b];
                        own = own || norm_out[b] == s->norm;
                }
                if(!own) { return fail(J2P_EINVAL, "norm_from_bands: the output list must contain the solver's own norm"); }
                t.nout = nout;
        }
        unsigned P = 1;
        while(P < s->ntr_global) { P <<= 1; }
        hipLaunchKernelGGL(k_norm_bands, dim3(s->nch), dim3(256), P * sizeof(double), s->stream, t, s->ntr_global, s->nch);
        HIP_TRY(hipGetLastError());
        s->norm_ready = true;
        return J2P_OK;
}

int j2p_solver_norm_ptr(j2p_solver *s, float **norm)
{
        if(!s || !norm) { return fail(J2P_EINVAL, "NULL argument"); }
        *norm = s->norm;
        return J2P_OK;
}

int j2p_solver_norm_external(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(!s->grad_done || s->rowsums_pending) { return fail(J2P_ESTATE, "norm_external needs a finished gradient phase"); }
        if(s->norm_by_project) { return fail(J2P_ESTATE, "norm_external: this solver reduces the norm inside its projection kernel"); }
        s->norm_ready = true;
        return J2P_OK;
}

int j2p_solver_copy_rows(j2p_solver *s, unsigned n, float *const dst[], const float *const src[], size_t floats)
{
        if(!s || !dst || !src) { return fail(J2P_EINVAL, "NULL argument"); }
        if(n == 0) { return J2P_OK; }
        if(n > 2u * kMaxCh || (floats & 1) || floats > 0xfffffffeu) { return fail(J2P_EINVAL, "copy_rows: bad count / size"); }
        DeviceGuard guard(s->device);
        RowCopies t;
        for(unsigned k = 0; k < n; k++) { t.dst[k] = dst[k]; t.src[k] = src[k]; }
        t.n = n;
        t.floats = (unsigned)floats;
        unsigned blocks = (unsigned)((floats / 2 + 255) / 256);
        if(blocks > 64) { blocks = 64; }
        hipLaunchKernelGGL(k_copy_rows, dim3(blocks ? blocks : 1), dim3(256), 0, s->stream, t);
        HIP_TRY(hipGetLastError());
        return J2P_OK;
}

int j2p_solver_commit_initial_halo(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        if(s->iter != 0 || s->grad_done) { return fail(J2P_ESTATE, "initial halo can only be committed at iteration 0"); }
        DeviceGuard guard(s->device);
        const size_t hb = (size_t)kHalo * s->W * sizeof(float);
        for(unsigned c = 0; c < s->nch; c++) {
                float *cur = s->ch[c].xbuf[s->cur], *prev = s->ch[c].xbuf[s->cur ^ 1];
                HIP_TRY(hipMemcpyAsync(prev, cur, hb, hipMemcpyDeviceToDevice, s->stream));
                const size_t off = (size_t)(s->rows + kHalo) * s->W;
                HIP_TRY(hipMemcpyAsync(prev + off, cur + off, hb, hipMemcpyDeviceToDevice, s->stream));
        }
        return J2P_OK;
}

int j2p_solver_download(j2p_solver *s, unsigned c, float *out)
{
        if(!s || !out) { return fail(J2P_EINVAL, "NULL argument"); }
        if(c >= s->nch) { return fail(J2P_EINVAL, "channel %u out of range", c); }
        if(s->grad_done) { return fail(J2P_ESTATE, "download between the two phases of an iteration"); }
        DeviceGuard guard(s->device);
        const float *src = s->ch[c].xbuf[s->cur] + (size_t)kHalo * s->W;
        HIP_TRY(hipMemcpyAsync(out, src, (size_t)s->rows * s->W * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        return J2P_OK;
}

int j2p_solver_download_gradient(j2p_solver *s, unsigned c, float *out)
{
        if(!s || !out) { return fail(J2P_EINVAL, "NULL argument"); }
        if(c >= s->nch) { return fail(J2P_EINVAL, "channel %u out of range", c); }
        DeviceGuard guard(s->device);
        HIP_TRY(hipMemcpyAsync(out, s->ch[c].grad, (size_t)s->rows * s->W * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        return J2P_OK;
}

int j2p_solver_plane_ptr(j2p_solver *s, unsigned c, float **dev_ptr)
{
        if(!s || !dev_ptr) { return fail(J2P_EINVAL, "NULL argument"); }
        if(c >= s->nch) { return fail(J2P_EINVAL, "channel %u out of range", c); }
        *dev_ptr = s->ch[c].xbuf[s->cur] + (size_t)kHalo * s->W;
        return J2P_OK;
}

int j2p_solver_sync(j2p_solver *s)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        HIP_TRY(hipStreamSynchronize(s->stream));
        return J2P_OK;
}

int j2p_solver_enable_timing(j2p_solver *s, int on)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        int rc = flush_timing(s);
        s->timing = on > 0 ? (unsigned)on : 0u;
        s->acc_grad_ms = s->acc_proj_ms = 0.;
        s->acc_samples = 0;
        if(rc == J2P_OK && s->timing && s->ev_pair_ms == 0.) {
                // calibration: 33 records back to back on the (idle) stream; the median of the 32 intervals is what a bracket
                // of two records costs by itself
                hipEvent_t e[33];
                int made = 0;
                for(; made < 33; made++) {
                        if(hipEventCreate(&e[made]) != hipSuccess) { break; }
                }
                // (no early return in here: the events are destroyed below whatever happens)
                if(made == 33 && hipStreamSynchronize(s->stream) == hipSuccess) {
                        for(int i = 0; i < 33; i++) { (void)hipEventRecord(e[i], s->stream); }
                        if(hipStreamSynchronize(s->stream) == hipSuccess) {
                                float d[32];
                                int n = 0;
                                for(int i = 0; i < 32; i++) {
                                        if(hipEventElapsedTime(&d[n], e[i], e[i + 1]) == hipSuccess) { n++; }
                                }
                                for(int i = 1; i < n; i++) {                     // insertion sort
                                        const float v = d[i];
                                        int k = i - 1;
                                        for(; k >= 0 && d[k] > v; k--) { d[k + 1] = d[k]; }
                                        d[k + 1] = v;
                                }
                                if(n) { s->ev_pair_ms = d[n / 2]; }
                        }
                }
                for(int i = 0; i < made; i++) { (void)hipEventDestroy(e[i]); }
                (void)hipGetLastError();
        }
        return rc;
}

int j2p_solver_timing_overhead(j2p_solver *s, double *event_pair_ms)
{
        if(!s || !event_pair_ms) { return fail(J2P_EINVAL, "NULL argument"); }
        *event_pair_ms = s->ev_pair_ms;
        return J2P_OK;
}

int j2p_solver_kernel_times(j2p_solver *s, double *gradient_ms, double *project_ms, unsigned *samples)
{
        if(!s) { return fail(J2P_EINVAL, "solver is NULL"); }
        DeviceGuard guard(s->device);
        int rc = flush_timing(s);
        if(rc != J2P_OK) { return rc; }
        const double n = s->acc_samples ? (double)s->acc_samples : 1.;
        if(gradient_ms) { *gradient_ms = s->acc_grad_ms / n; }
        if(project_ms) { *project_ms = s->acc_proj_ms / n; }
        if(samples) { *samples = s->acc_samples; }
        return J2P_OK;
}

int j2p_decode_plane(int device, unsigned w, unsigned h, const int16_t *data, const uint16_t *quant_table, float *out)
{
        if(!data || !quant_table || !out) { return fail(J2P_EINVAL, "NULL argument"); }
        if(w == 0 || h == 0 || (w & 7) || (h & 7)) { return fail(J2P_EINVAL, "plane %ux%u is not a positive multiple of 8", w, h); }
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return fail(J2P_EDEVICE, "no HIP device available"); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }
        const size_t n = (size_t)w * h;
        int16_t *dd = nullptr;
        float *df = nullptr, *dq = nullptr;
        float qf[64];
        for(int j = 0; j < 64; j++) { qf[j] = (float)quant_table[j]; }
        int rc = J2P_OK;
        hipError_t e = dev_malloc((void **)&dd, n * sizeof(int16_t));
        if(e == hipSuccess) { e = dev_malloc((void **)&df, n * sizeof(float)); }
        if(e == hipSuccess) { e = dev_malloc((void **)&dq, sizeof(qf)); }
        if(e == hipSuccess) { e = hipMemcpy(dd, data, n * sizeof(int16_t), hipMemcpyHostToDevice); }
        if(e == hipSuccess) { e = hipMemcpy(dq, qf, sizeof(qf), hipMemcpyHostToDevice); }
        if(e == hipSuccess) {
                const unsigned groups = ((w / 8 + 7) / 8) * (h / 8);
                hipLaunchKernelGGL(k_decode, dim3((groups + 3) / 4), dim3(256), 0, nullptr, (const int16_t *)dd,
                                   (const float *)dq, df, w, h / 8);
                e = hipGetLastError();
        }
        if(e == hipSuccess) { e = hipMemcpy(out, df, n * sizeof(float), hipMemcpyDeviceToHost); }
        if(e != hipSuccess) { rc = fail(e == hipErrorOutOfMemory ? J2P_ENOMEM : J2P_EDEVICE, "decode_plane: %s", hipGetErrorString(e)); }
        (void)hipFree(dd);
        (void)hipFree(df);
        (void)hipFree(dq);
        return rc;
}

int j2p_dct8x8_blocks(int device, float *blocks, size_t n, int inverse)
{
        if(!blocks) { return fail(J2P_EINVAL, "NULL argument"); }
        if(n == 0) { return J2P_OK; }
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return fail(J2P_EDEVICE, "no HIP device available"); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }
        float *db = nullptr;
        int rc = J2P_OK;
        hipError_t e = dev_malloc((void **)&db, n * 64 * sizeof(float));
        if(e == hipSuccess) { e = hipMemcpy(db, blocks, n * 64 * sizeof(float), hipMemcpyHostToDevice); }
        if(e == hipSuccess) {
                hipLaunchKernelGGL(k_dct_blocks, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, nullptr, db, n, inverse);
                e = hipGetLastError();
        }
        if(e == hipSuccess) { e = hipMemcpy(blocks, db, n * 64 * sizeof(float), hipMemcpyDeviceToHost); }
        if(e != hipSuccess) { rc = fail(e == hipErrorOutOfMemory ? J2P_ENOMEM : J2P_EDEVICE, "dct8x8_blocks: %s", hipGetErrorString(e)); }
        (void)hipFree(db);
        return rc;
}

// rows [y0, y1) of the image from three (solver, channel) pairs on one device that all hold those canvas rows
static int rgb_rows(const j2p_plane_ref planes[3], unsigned w, unsigned y0, unsigned y1, unsigned bits, uint8_t *out_host)
{
        const float *ptr[3];
        unsigned stride[3];
        for(int i = 0; i < 3; i++) {
                j2p_solver *s = planes[i].solver;
                if(!s || planes[i].channel >= s->nch) { return fail(J2P_EINVAL, "plane %d: bad solver/channel", i); }
                if(s->device != planes[0].solver->device) { return fail(J2P_EINVAL, "planes live on different devices"); }
                if(s->W < w || y0 < s->row0 || y1 > s->row0 + s->rows) {
                        return fail(J2P_EINVAL, "plane %d: rows [%u,%u) x %u columns are not inside the solver's [%u,%u) x %u", i, y0, y1, w,
                                    s->row0, s->row0 + s->rows, s->W);
                }
                if(s->grad_done) { return fail(J2P_ESTATE, "to_rgb between the two phases of an iteration"); }
                ptr[i] = s->ch[planes[i].channel].xbuf[s->cur] + (size_t)(kHalo + (y0 - s->row0)) * s->W;
                stride[i] = s->W;
        }
        const unsigned h = y1 - y0;
        j2p_solver *s0 = planes[0].solver;
        DeviceGuard guard(s0->device);
        for(int i = 1; i < 3; i++) {
                if(planes[i].solver != s0) { HIP_TRY(hipStreamSynchronize(planes[i].solver->stream)); }
        }
        const size_t bytes = (size_t)w * h * (bits == 8 ? 3 : 6);
        void *dout = nullptr;
        size_t dout_bytes = 0;
        HIP_TRY(pool_take(s0->device, bytes, &dout, &dout_bytes));       // pooled like the solvers' arenas: no hipFree per image
        hipLaunchKernelGGL(k_to_rgb, dim3(2048), dim3(256), 0, s0->stream, ptr[0], stride[0], ptr[1], stride[1], ptr[2], stride[2],
                           w, h, bits, static_cast<uint8_t *>(dout));
        hipError_t e = hipMemcpyAsync(out_host, dout, bytes, hipMemcpyDeviceToHost, s0->stream);
        if(e == hipSuccess) { e = hipStreamSynchronize(s0->stream); }
        pool_give(s0->device, dout, dout_bytes);
        if(e != hipSuccess) { return fail(J2P_EDEVICE, "planes_to_rgb: %s", hipGetErrorString(e)); }
        return J2P_OK;
}

int j2p_planes_to_rgb(const j2p_plane_ref planes[3], unsigned w, unsigned h, unsigned bits, uint8_t *out_host)
{
        if(!planes || !out_host) { return fail(J2P_EINVAL, "NULL argument"); }
        if(bits != 8 && bits != 16) { return fail(J2P_EINVAL, "bits must be 8 or 16 (png.c:22)"); }
        if(w == 0 || h == 0) { return fail(J2P_EINVAL, "empty image"); }
        for(int i = 0; i < 3; i++) {
                if(planes[i].solver && !planes[i].solver->whole) { return fail(J2P_ESTATE, "to_rgb needs whole-canvas solvers (bands: j2p_planes_rows_to_rgb)"); }
        }
        return rgb_rows(planes, w, 0, h, bits, out_host);
}

int j2p_planes_rows_to_rgb(const j2p_plane_ref planes[3], unsigned w, unsigned row_begin, unsigned row_end, unsigned bits,
                           uint8_t *out_host)
{
        if(!planes || !out_host) { return fail(J2P_EINVAL, "NULL argument"); }
        if(bits != 8 && bits != 16) { return fail(J2P_EINVAL, "bits must be 8 or 16 (png.c:22)"); }
        if(w == 0 || row_begin >= row_end) { return fail(J2P_EINVAL, "empty row range"); }
        return rgb_rows(planes, w, row_begin, row_end, bits, out_host);
}

int j2p_math_selftest(int device, size_t n, unsigned seed, unsigned long long *div_mismatches,
                      unsigned long long *sqrt_mismatches)
{
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return fail(J2P_EDEVICE, "no HIP device available"); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }
        unsigned long long *dm = nullptr, hm[2] = {0, 0};
        HIP_TRY(hipMalloc(&dm, sizeof(hm)));
        hipError_t e = hipMemset(dm, 0, sizeof(hm));
        if(e == hipSuccess) {
                hipLaunchKernelGGL(k_math_selftest, dim3(4096), dim3(256), 0, nullptr, n, seed, dm);
                e = hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
        }
        (void)hipFree(dm);
        if(e != hipSuccess) { return fail(J2P_EDEVICE, "math_selftest: %s", hipGetErrorString(e)); }
        if(div_mismatches) { *div_mismatches = hm[0]; }
        if(sqrt_mismatches) { *sqrt_mismatches = hm[1]; }
        return J2P_OK;
}

int j2p_sqrt_exhaustive(int device, unsigned long long *rsq_mismatches, unsigned long long *fast_mismatches)
{
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return fail(J2P_EDEVICE, "no HIP device available"); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }
        unsigned long long *dm = nullptr, hm[2] = {0, 0};
        HIP_TRY(hipMalloc(&dm, sizeof(hm)));
        hipError_t e = hipMemset(dm, 0, sizeof(hm));
        if(e == hipSuccess) {
                hipLaunchKernelGGL(k_sqrt_exhaustive, dim3(8192), dim3(256), 0, nullptr, dm);
                e = hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
        }
        (void)hipFree(dm);
        if(e != hipSuccess) { return fail(J2P_EDEVICE, "sqrt_exhaustive: %s", hipGetErrorString(e)); }
        if(rsq_mismatches) { *rsq_mismatches = hm[0]; }
        if(fast_mismatches) { *fast_mismatches = hm[1]; }
        return J2P_OK;
}

// one pass (or slice of a pass) of the short division's exhaustive checks (see k_recip_exhaustive);
// report[0] = mismatches, report[1..8] = the first offenders
int j2p_division_exhaustive(int device, int pass, unsigned first, unsigned count, unsigned long long report[9])
{
        if(pass < 1 || pass > 3) { return fail(J2P_EINVAL, "pass must be 1, 2 or 3"); }
        if(!report) { return fail(J2P_EINVAL, "NULL argument"); }
        int ndev = 0;
        if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { return fail(J2P_EDEVICE, "no HIP device available"); }
        DeviceGuard guard(device);
        if(!guard.ok) { return fail(J2P_EDEVICE, "hipSetDevice(%d) failed", device); }
        unsigned long long *dm = nullptr, hm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(dev_malloc((void **)&dm, sizeof(hm)));
        hipError_t e = hipMemset(dm, 0, sizeof(hm));
        if(e == hipSuccess) {
                if(pass == 1) { hipLaunchKernelGGL(k_recip_exhaustive, dim3(8192), dim3(256), 0, nullptr, dm); }
                else if(pass == 2) { hipLaunchKernelGGL(k_div_exhaustive<false>, dim3((count + 3) / 4), dim3(256), 0, nullptr, first, count, dm); }
                else { hipLaunchKernelGGL(k_div_exhaustive<true>, dim3((count + 3) / 4), dim3(256), 0, nullptr, first, count, dm); }
                e = hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
        }
        (void)hipFree(dm);
        if(e != hipSuccess) { return fail(J2P_EDEVICE, "division_exhaustive: %s", hipGetErrorString(e)); }
        for(int i = 0; i < 9; i++) { report[i] = hm[i]; }
        return J2P_OK;
}

}  // extern "C"
