// jpeg2png_amd — large host <-> device copies of the drop-in boundary.
//
// compute() (compute.h:8) is handed ordinary malloc'ed planes and hands one back (compute.c:304-305, 455-461); for a
// 4096 x 4096 channel that is 96 MiB up (int16 coefficients + the decoded float plane, which aux_init reads,
// compute.c:278-310) and 64 MiB down.  hipMemcpy from / to pageable memory stages through the runtime's own pinned
// buffer on ONE host thread: 13.7 ms of a 72.6 ms call at that size (profiles/r04_host_to_host.jsonl).  These two
// calls do the staging themselves with a small team of host threads per GPU — created once, each with a stream, two
// pinned slabs and two events of its own: a thread claims the next 2 MiB chunk of the copy, memcpy()s it between the
// caller's memory and a slab, and lets the DMA engine move it while it works on its other slab — the memcpy bandwidth
// of several cores on one side, PCIe on the other, overlapped.  Small copies keep the plain path.
// (A first version that created its threads and streams per call was SLOWER than hipMemcpy: 80 vs 72.6 ms.)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "jpeg2png_amd.h"
#include "j2p_internal.h"

namespace {

constexpr size_t kChunk = (size_t)2 << 20;        // bytes per DMA
constexpr size_t kMinBytes = (size_t)8 << 20;     // below this a copy is not worth waking the team
constexpr unsigned kMaxThreads = 8;
constexpr int kMaxDevices = 64;

unsigned team_threads()
{
        static const unsigned want = [] {
                const char *env = getenv("J2P_XFER_THREADS");          // 0 = always the plain hipMemcpy path
                if(env && *env) {
                        const int v = atoi(env);
                        return v < 0 ? 0u : ((unsigned)v > kMaxThreads ? kMaxThreads : (unsigned)v);
                }
                const unsigned hw = std::thread::hardware_concurrency();
                return hw >= 8 ? 4u : (hw >= 4 ? 2u : 0u);
        }();
        return want;
}

struct Team {
        int device = 0;
        unsigned n = 0;
        std::vector<std::thread> threads;
        std::mutex lock;                       // guards the fields below
        std::condition_variable wake, done;
        std::mutex serial;                     // one copy at a time per GPU: PCIe is the shared resource
        uint64_t generation = 0;
        bool quit = false;
        // the copy in progress
        bool up = false;
        char *dev = nullptr, *host = nullptr;
        size_t bytes = 0, chunks = 0;
        std::atomic<size_t> next{0};
        unsigned finished = 0;
        hipError_t rc = hipSuccess;
        bool broken = false;                   // a worker could not set itself up: the team is never used
        unsigned ready = 0;
};

struct Slot {
        void *slab = nullptr;
        hipEvent_t ev = nullptr;
        bool busy = false;                     // up: its DMA has not been waited for yet; down: holds a chunk to copy out
        size_t off = 0, len = 0;
};

void worker_main(Team *t)
{
        hipStream_t st = nullptr;
        Slot s[2];
        bool ok = hipSetDevice(t->device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for(int i = 0; i < 2 && ok; i++) {
                ok = hipHostMalloc(&s[i].slab, kChunk, hipHostMallocPortable) == hipSuccess &&
                     hipEventCreateWithFlags(&s[i].ev, hipEventDisableTiming) == hipSuccess;
        }
        {
                std::lock_guard<std::mutex> g(t->lock);
                if(!ok) { t->broken = true; }
                t->ready++;
        }
        t->done.notify_all();
        uint64_t seen = 0;
        for(;;) {
                {
                        std::unique_lock<std::mutex> g(t->lock);
                        t->wake.wait(g, [&] { return t->quit || t->generation != seen; });
                        if(t->quit) { break; }
                        seen = t->generation;
                }
                hipError_t e = ok ? hipSuccess : hipErrorUnknown;
                int cur = 0;
                for(;;) {
                        const size_t c = e == hipSuccess ? t->next.fetch_add(1, std::memory_order_relaxed) : t->chunks;
                        if(c >= t->chunks) { break; }
                        const size_t off = c * kChunk, len = off + kChunk <= t->bytes ? kChunk : t->bytes - off;
                        Slot &a = s[cur], &b = s[cur ^ 1];
                        if(t->up) {
                                if(a.busy) { e = hipEventSynchronize(a.ev); a.busy = false; }     // the slab's previous DMA has read it
                                if(e != hipSuccess) { break; }
                                memcpy(a.slab, t->host + off, len);
                                e = hipMemcpyAsync(t->dev + off, a.slab, len, hipMemcpyHostToDevice, st);
                                if(e == hipSuccess) { e = hipEventRecord(a.ev, st); }
                                a.busy = true;
                        } else {
                                // the DMA of this chunk is issued before the previous one is copied out of its slab
                                e = hipMemcpyAsync(a.slab, t->dev + off, len, hipMemcpyDeviceToHost, st);
                                if(e == hipSuccess) { e = hipEventRecord(a.ev, st); }
                                a.busy = true;
                                a.off = off;
                                a.len = len;
                                if(e == hipSuccess && b.busy) {
                                        e = hipEventSynchronize(b.ev);
                                        if(e == hipSuccess) { memcpy(t->host + b.off, b.slab, b.len); }
                                        b.busy = false;
                                }
                        }
                        cur ^= 1;
                }
                for(int i = 0; i < 2; i++) {
                        // (down: the chunk issued last but one is slot cur, the last one slot cur ^ 1: in that order)
                        Slot &a = s[(cur + i) & 1];
                        if(!a.busy) { continue; }
                        const hipError_t w = hipEventSynchronize(a.ev);
                        if(e == hipSuccess) { e = w; }
                        if(!t->up && e == hipSuccess) { memcpy(t->host + a.off, a.slab, a.len); }
                        a.busy = false;
                }
                {
                        std::lock_guard<std::mutex> g(t->lock);
                        if(e != hipSuccess && t->rc == hipSuccess) { t->rc = e; }
                        t->finished++;
                }
                t->done.notify_all();
        }
        for(int i = 0; i < 2; i++) {
                if(s[i].ev) { (void)hipEventDestroy(s[i].ev); }
                if(s[i].slab) { (void)hipHostFree(s[i].slab); }
        }
        if(st) { (void)hipStreamDestroy(st); }
}

std::mutex g_teams_lock;
Team *g_teams[kMaxDevices];

Team *team_of(int device)
{
        if(device < 0 || device >= kMaxDevices) { return nullptr; }
        std::lock_guard<std::mutex> g(g_teams_lock);
        Team *t = g_teams[device];
        if(!t) {
                const unsigned n = team_threads();
                if(n == 0) { return nullptr; }
                t = new(std::nothrow) Team();
                if(!t) { return nullptr; }
                t->device = device;
                t->n = n;
                for(unsigned k = 0; k < n; k++) { t->threads.emplace_back(worker_main, t); }
                {
                        std::unique_lock<std::mutex> l(t->lock);
                        t->done.wait(l, [&] { return t->ready == n; });
                }
                g_teams[device] = t;
        }
        return t->broken ? nullptr : t;
}

hipError_t team_copy(Team *t, bool up, void *dev, void *host, size_t bytes)
{
        std::lock_guard<std::mutex> one(t->serial);
        {
                std::lock_guard<std::mutex> g(t->lock);
                t->up = up;
                t->dev = static_cast<char *>(dev);
                t->host = static_cast<char *>(host);
                t->bytes = bytes;
                t->chunks = (bytes + kChunk - 1) / kChunk;
                t->next.store(0);
                t->finished = 0;
                t->rc = hipSuccess;
                t->generation++;
        }
        t->wake.notify_all();
        std::unique_lock<std::mutex> g(t->lock);
        t->done.wait(g, [&] { return t->finished == t->n; });
        return t->rc;
}

}  // namespace

// host -> device.  Returns when the host array has been consumed AND — team path — the bytes are on the device; the
// plain path is hipMemcpyAsync on `stream` (the caller synchronises before the host array goes away).
hipError_t j2p_upload_plane(int device, void *dev, const void *host, size_t bytes, hipStream_t stream)
{
        Team *t = bytes >= kMinBytes ? team_of(device) : nullptr;
        if(!t) { return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream); }
        return team_copy(t, true, dev, const_cast<void *>(host), bytes);
}

// device -> host, after everything queued on `stream`; complete on return
hipError_t j2p_download_plane(int device, void *host, const void *dev, size_t bytes, hipStream_t stream)
{
        Team *t = bytes >= kMinBytes ? team_of(device) : nullptr;
        if(!t) {
                hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream);
                if(e == hipSuccess) { e = hipStreamSynchronize(stream); }
                return e;
        }
        hipError_t e = hipStreamSynchronize(stream);
        if(e != hipSuccess) { return e; }
        return team_copy(t, false, const_cast<void *>(dev), host, bytes);
}

// (the teams live as long as the process: a handful of sleeping threads and 16 MiB of pinned memory per GPU in use)
void j2p_xfer_trim(void) {}
