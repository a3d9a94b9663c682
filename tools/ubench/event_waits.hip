// What a stream pays for waiting on n events of OTHER streams between two of its kernels — the row tiling's
// "projection(k) behind every band's gradient(k)" with n = N - 1 — one GPU.  Per iteration: n side streams run a tiny
// kernel and record an event; the main stream runs a kernel, records an event of its own (what the bands do after
// each phase), waits for the n side events, runs the kernel again, records.  Variants:
//   direct     the main stream waits for the n side events itself (n barrier packets in its queue)
//   collector  a helper stream waits for the n side events and records ONE event; the main stream waits for that one
//   stale      direct, but the side events were recorded once, before the loop (complete for ever)
// each with a kernel that streams 512 MiB (dirty caches at every boundary) and one that touches 4 MiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/event_waits tools/ubench/event_waits.hip && tools/ubench/event_waits
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void burn(float *p, size_t n, int rounds)
{
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if(i >= n) { return; }
        float v = p[i];
        for(int r = 0; r < rounds; r++) { v = v * 1.0001f + 0.5f; }
        p[i] = v;
}
__global__ void tiny(float *p) { if(threadIdx.x == 0 && blockIdx.x == 0) { p[0] += 1.f; } }

#define CHECK(e) do { hipError_t e_ = (e); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while(0)

int main()
{
        const size_t nbig = (size_t)64 << 20, nsmall = (size_t)1 << 20;
        float *buf = nullptr, *side = nullptr;
        CHECK(hipMalloc(&buf, nbig * sizeof(float)));
        CHECK(hipMalloc(&side, 64 * sizeof(float)));
        CHECK(hipMemset(buf, 0, nbig * sizeof(float)));
        CHECK(hipMemset(side, 0, 64 * sizeof(float)));
        hipStream_t main_s, coll_s;
        CHECK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&coll_s, hipStreamNonBlocking));
        const int iters = 300;
        const unsigned evflags = hipEventDisableTiming;
        printf("{\"what\": \"us per iteration of [kernel, record, wait for n side events, kernel, record] on one stream, one GPU\"");
        const char *modes[3] = {"direct", "collector", "stale"};
        for(int footprint = 0; footprint < 2; footprint++) {
                const size_t n = footprint ? nsmall : nbig;
                const int rounds = footprint ? 1500 : 8;
                for(int mode = 0; mode < 3; mode++) {
                        printf(", \"%s_%s\": {", modes[mode], footprint ? "4MiB_kernel" : "512MiB_kernel");
                        const int ns[] = {0, 1, 2, 3, 7};
                        for(int k = 0; k < 5; k++) {
                                const int nw = ns[k];
                                std::vector<hipStream_t> ss(nw);
                                std::vector<hipEvent_t> ev(nw);
                                hipEvent_t own[2], all;
                                for(int i = 0; i < nw; i++) {
                                        CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
                                        CHECK(hipEventCreateWithFlags(&ev[i], evflags));
                                }
                                CHECK(hipEventCreateWithFlags(&own[0], evflags));
                                CHECK(hipEventCreateWithFlags(&own[1], evflags));
                                CHECK(hipEventCreateWithFlags(&all, evflags));
                                if(mode == 2) {
                                        for(int i = 0; i < nw; i++) {
                                                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, ss[i], side + i);
                                                CHECK(hipEventRecord(ev[i], ss[i]));
                                        }
                                        CHECK(hipDeviceSynchronize());
                                }
                                auto run = [&](int count) -> int {
                                        for(int it = 0; it < count; it++) {
                                                for(int i = 0; i < nw && mode != 2; i++) {
                                                        // the side streams follow the main stream's previous phase (like bands do)
                                                        if(it > 0) { CHECK(hipStreamWaitEvent(ss[i], own[1], 0)); }
                                                        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, ss[i], side + i);
                                                        CHECK(hipEventRecord(ev[i], ss[i]));
                                                }
                                                hipLaunchKernelGGL(burn, dim3((unsigned)(n / 256)), dim3(256), 0, main_s, buf, n, rounds);
                                                CHECK(hipEventRecord(own[0], main_s));
                                                if(mode == 1 && nw) {
                                                        for(int i = 0; i < nw; i++) { CHECK(hipStreamWaitEvent(coll_s, ev[i], 0)); }
                                                        CHECK(hipEventRecord(all, coll_s));
                                                        CHECK(hipStreamWaitEvent(main_s, all, 0));
                                                } else {
                                                        for(int i = 0; i < nw; i++) { CHECK(hipStreamWaitEvent(main_s, ev[i], 0)); }
                                                }
                                                hipLaunchKernelGGL(burn, dim3((unsigned)(n / 256)), dim3(256), 0, main_s, buf, n, rounds);
                                                CHECK(hipEventRecord(own[1], main_s));
                                        }
                                        return 0;
                                };
                                if(run(20)) { return 1; }
                                CHECK(hipDeviceSynchronize());
                                const auto t0 = std::chrono::steady_clock::now();
                                if(run(iters)) { return 1; }
                                CHECK(hipDeviceSynchronize());
                                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
                                printf("%s\"n%d\": %.2f", k ? ", " : "", nw, us);
                                for(int i = 0; i < nw; i++) { (void)hipStreamDestroy(ss[i]); (void)hipEventDestroy(ev[i]); }
                                (void)hipEventDestroy(own[0]);
                                (void)hipEventDestroy(own[1]);
                                (void)hipEventDestroy(all);
                        }
                        printf("}");
                }
        }
        printf("}\n");
        return 0;
}
