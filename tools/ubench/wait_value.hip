// hipStreamWaitValue / hipStreamWriteValue (stream memory operations on signal memory) against HIP events as the
// cross-stream dependency between two kernels of a stream, one GPU.  Per iteration: n side streams run a tiny kernel
// and signal; the main stream runs a kernel, signals, waits for the n side signals, runs the kernel again, signals.
//   events    hipEventRecord / hipStreamWaitEvent (what j2p_tiled uses)
//   values    every side stream hipStreamWriteValue64(flag_i, it + 1); the main stream hipStreamWaitValue64(flag_i >= it + 1) x n
//   counter   the side streams' kernels atomicAdd (system scope) ONE counter in signal memory; the main stream waits once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void burn(float *p, size_t n, int rounds)
{
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if(i >= n) { return; }
        float v = p[i];
        for(int r = 0; r < rounds; r++) { v = v * 1.0001f + 0.5f; }
        p[i] = v;
}
__global__ void tiny(float *p, unsigned long long *counter)
{
        if(threadIdx.x == 0 && blockIdx.x == 0) {
                p[0] += 1.f;
                if(counter) { __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
}

#define CHECK(e) do { hipError_t e_ = (e); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while(0)

int main()
{
        const size_t n = (size_t)64 << 20;
        float *buf = nullptr, *side = nullptr;
        CHECK(hipMalloc(&buf, n * sizeof(float)));
        CHECK(hipMalloc(&side, 64 * sizeof(float)));
        CHECK(hipMemset(buf, 0, n * sizeof(float)));
        CHECK(hipMemset(side, 0, 64 * sizeof(float)));
        // (signal memory comes in single 8-byte allocations; J2P_WAIT_VALUE_HOSTMEM=1: ordinary coherent pinned host memory instead)
        unsigned long long *flag[64];
        const bool hostmem = getenv("J2P_WAIT_VALUE_HOSTMEM") != nullptr;
        if(hostmem) {
                unsigned long long *all = nullptr;
                CHECK(hipHostMalloc((void **)&all, 64 * 64, hipHostMallocPortable | hipHostMallocCoherent));
                for(int i = 0; i < 64; i++) { flag[i] = all + 8 * i; }
        } else {
                for(int i = 0; i < 64; i++) { CHECK(hipExtMallocWithFlags((void **)&flag[i], 8, hipMallocSignalMemory)); }
        }
        hipStream_t main_s;
        CHECK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
        const int iters = 300;
        printf("{\"what\": \"us per iteration of [kernel, signal, wait for n side signals, kernel, signal] on one stream, one GPU\"");
        const char *modes[3] = {"events", "values", "counter"};
        for(int mode = 0; mode < 3; mode++) {
                printf(", \"%s\": {", modes[mode]);
                const int ns[] = {0, 1, 2, 3, 7};
                for(int k = 0; k < 5; k++) {
                        const int nw = ns[k];
                        for(int i = 0; i < 64; i++) { *flag[i] = 0; }
                        std::vector<hipStream_t> ss(nw);
                        std::vector<hipEvent_t> ev(nw);
                        hipEvent_t own;
                        for(int i = 0; i < nw; i++) {
                                CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
                                CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
                        }
                        CHECK(hipEventCreateWithFlags(&own, hipEventDisableTiming));
                        unsigned long long done = 0;        // iterations issued so far (values / counter are cumulative)
                        auto run = [&](int count) -> int {
                                for(int it = 0; it < count; it++, done++) {
                                        for(int i = 0; i < nw; i++) {
                                                // the side streams follow the main stream's previous phase (like bands do)
                                                if(done > 0) {
                                                        if(mode == 0) { CHECK(hipStreamWaitEvent(ss[i], own, 0)); }
                                                        else { CHECK(hipStreamWaitValue64(ss[i], flag[32], done, hipStreamWaitValueGte, ~0ull)); }
                                                }
                                                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, ss[i], side + i, mode == 2 ? flag[40] : nullptr);
                                                if(mode == 0) { CHECK(hipEventRecord(ev[i], ss[i])); }
                                                else if(mode == 1) { CHECK(hipStreamWriteValue64(ss[i], flag[i], done + 1, 0)); }
                                        }
                                        hipLaunchKernelGGL(burn, dim3((unsigned)(n / 256)), dim3(256), 0, main_s, buf, n, 8);
                                        if(mode == 0) { for(int i = 0; i < nw; i++) { CHECK(hipStreamWaitEvent(main_s, ev[i], 0)); } }
                                        else if(mode == 1) { for(int i = 0; i < nw; i++) { CHECK(hipStreamWaitValue64(main_s, flag[i], done + 1, hipStreamWaitValueGte, ~0ull)); } }
                                        else if(nw) { CHECK(hipStreamWaitValue64(main_s, flag[40], (done + 1) * nw, hipStreamWaitValueGte, ~0ull)); }
                                        hipLaunchKernelGGL(burn, dim3((unsigned)(n / 256)), dim3(256), 0, main_s, buf, n, 8);
                                        if(mode == 0) { CHECK(hipEventRecord(own, main_s)); }
                                        else { CHECK(hipStreamWriteValue64(main_s, flag[32], done + 1, 0)); }
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
                        (void)hipEventDestroy(own);
                }
                printf("}");
        }
        printf("}\n");
        return 0;
}
