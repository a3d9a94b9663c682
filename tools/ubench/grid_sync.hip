// What a device-wide barrier inside ONE persistent launch costs on the GPU it runs on, against what it would replace:
// the boundary between two dependent kernel launches on one stream.  The question behind it ("persistent
// solver for small images"): a FISTA iteration is two phases whose data crosses workgroups (g, x_{k+1}, the prob state, the
// norm), so a persistent solver pays two barriers per iteration INCLUDING the cache maintenance that makes plain stores of
// one XCD visible to the others (release = write back that XCD's L2, acquire = invalidate it) — the same maintenance a
// kernel boundary performs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/grid_sync tools/ubench/grid_sync.hip
// Prints one JSON object: microseconds per boundary for
//   launch_chain      N dependent launches of a kernel of `blocks` workgroups that reads its left neighbour's slot and writes its own
//   barrier_flat      one counter, one agent-scope atomic per workgroup, spin on a generation word; release/acquire fences
//   barrier_tree      16-workgroup groups, the last arrival of a group arrives at the root; release/acquire fences
//   barrier_*_nofence the same without the fences (data crossing would have to use device-coherent accesses instead)
// and the number of stale reads each variant saw (must be 0 where fences are on).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                                   \
        do {                                                                                       \
                hipError_t e_ = (x);                                                               \
                if(e_ != hipSuccess) {                                                             \
                        fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
                        exit(1);                                                                   \
                }                                                                                  \
        } while(0)

struct Sync {
        unsigned *count;     // [1 + groups] arrival counters (root first), each on its own 128-byte line
        unsigned *gen;       // generation word
};
constexpr unsigned kLine = 32;       // unsigneds per 128-byte line

__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

template <bool TREE, bool FENCE>
__device__ __forceinline__ void grid_barrier(const Sync &s, unsigned nblocks, unsigned target)
{
        __syncthreads();
        if(threadIdx.x == 0) {
                if(FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
                wait_stores();
                bool last;
                if(TREE) {
                        const unsigned g = blockIdx.x >> 4, ngroups = (nblocks + 15) >> 4;
                        const unsigned in_group = g + 1 == ngroups ? nblocks - (g << 4) : 16u;
                        unsigned *c = s.count + (size_t)(1 + g) * kLine;
                        last = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == in_group;
                        if(last) {
                                __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                last = __hip_atomic_fetch_add(s.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == ngroups;
                        }
                } else {
                        last = __hip_atomic_fetch_add(s.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nblocks;
                }
                if(last) {
                        __hip_atomic_store(s.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wait_stores();
                        __hip_atomic_store(s.gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                        while(__hip_atomic_load(s.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) { __builtin_amdgcn_s_sleep(1); }
                }
                if(FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
        }
        __syncthreads();
}

// every round: write my slot (plain store), barrier, read the left neighbour's slot (plain load: a workgroup on another XCD)
template <bool TREE, bool FENCE>
__global__ __launch_bounds__(256) void k_persistent(Sync s, unsigned *slots, unsigned rounds, unsigned gen0, unsigned long long *stale)
{
        const unsigned nb = gridDim.x, me = blockIdx.x, left = (me + nb - 1) % nb;
        unsigned bad = 0;
        for(unsigned r = 1; r <= rounds; r++) {
                if(threadIdx.x == 0) { slots[(size_t)me * kLine] = gen0 + r; }
                grid_barrier<TREE, FENCE>(s, nb, gen0 + 2 * r - 1);
                if(threadIdx.x == 0 && slots[(size_t)left * kLine] != gen0 + r) { bad++; }
                grid_barrier<TREE, FENCE>(s, nb, gen0 + 2 * r);     // (nobody overwrites a slot before it was read)
        }
        if(threadIdx.x == 0 && bad) { atomicAdd(stale, (unsigned long long)bad); }
}

__global__ __launch_bounds__(256) void k_step(unsigned *slots_in, unsigned *slots_out, unsigned expect, unsigned long long *stale)
{
        const unsigned nb = gridDim.x, me = blockIdx.x, left = (me + nb - 1) % nb;
        if(threadIdx.x == 0) {
                if(slots_in[(size_t)left * kLine] != expect) { atomicAdd(stale, 1ull); }
                slots_out[(size_t)me * kLine] = expect + 1;
        }
}

static float elapsed_us(hipEvent_t a, hipEvent_t b)
{
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        return ms * 1e3f;
}

int main(int argc, char **argv)
{
        const unsigned rounds = argc > 1 ? (unsigned)atoi(argv[1]) : 2000;
        hipStream_t st;
        CHECK(hipStreamCreate(&st));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        unsigned *count, *gen, *slots_a, *slots_b;
        unsigned long long *stale;
        const unsigned max_blocks = 2048;
        CHECK(hipMalloc(&count, (size_t)(2 + max_blocks / 16) * kLine * 4));
        CHECK(hipMalloc(&gen, 512));
        CHECK(hipMalloc(&slots_a, (size_t)max_blocks * kLine * 4));
        CHECK(hipMalloc(&slots_b, (size_t)max_blocks * kLine * 4));
        CHECK(hipMalloc(&stale, 8));
        printf("{\"rounds\": %u", rounds);
        for(unsigned blocks : {256u, 512u, 1024u}) {
                // --- dependent launches
                CHECK(hipMemsetAsync(slots_a, 0, (size_t)max_blocks * kLine * 4, st));
                CHECK(hipMemsetAsync(slots_b, 0, (size_t)max_blocks * kLine * 4, st));
                CHECK(hipMemsetAsync(stale, 0, 8, st));
                for(int rep = 0; rep < 2; rep++) {       // (first repetition: warm-up)
                        if(rep) {
                                CHECK(hipMemsetAsync(slots_a, 0, (size_t)max_blocks * kLine * 4, st));
                                CHECK(hipMemsetAsync(stale, 0, 8, st));
                        }
                        CHECK(hipEventRecord(e0, st));
                        for(unsigned r = 0; r < 2 * rounds; r++) {
                                hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, r & 1 ? slots_b : slots_a, r & 1 ? slots_a : slots_b, r, stale);
                        }
                        CHECK(hipEventRecord(e1, st));
                        CHECK(hipStreamSynchronize(st));
                }
                unsigned long long bad;
                CHECK(hipMemcpy(&bad, stale, 8, hipMemcpyDeviceToHost));
                printf(",\n \"launch_chain_%u\": {\"us_per_boundary\": %.3f, \"stale\": %llu}", blocks, elapsed_us(e0, e1) / (2 * rounds), bad);
                // --- barriers inside one launch
                for(int variant = 0; variant < 4; variant++) {
                        const bool tree = variant & 1, fence = !(variant & 2);
                        Sync s{count, gen};
                        float us = 0;
                        for(int rep = 0; rep < 2; rep++) {
                                CHECK(hipMemsetAsync(count, 0, (size_t)(2 + max_blocks / 16) * kLine * 4, st));
                                CHECK(hipMemsetAsync(gen, 0, 512, st));
                                CHECK(hipMemsetAsync(slots_a, 0, (size_t)max_blocks * kLine * 4, st));
                                CHECK(hipMemsetAsync(stale, 0, 8, st));
                                CHECK(hipEventRecord(e0, st));
                                if(tree && fence) { hipLaunchKernelGGL((k_persistent<true, true>), dim3(blocks), dim3(256), 0, st, s, slots_a, rounds, 0u, stale); }
                                else if(fence) { hipLaunchKernelGGL((k_persistent<false, true>), dim3(blocks), dim3(256), 0, st, s, slots_a, rounds, 0u, stale); }
                                else if(tree) { hipLaunchKernelGGL((k_persistent<true, false>), dim3(blocks), dim3(256), 0, st, s, slots_a, rounds, 0u, stale); }
                                else { hipLaunchKernelGGL((k_persistent<false, false>), dim3(blocks), dim3(256), 0, st, s, slots_a, rounds, 0u, stale); }
                                CHECK(hipEventRecord(e1, st));
                                CHECK(hipStreamSynchronize(st));
                                us = elapsed_us(e0, e1);
                        }
                        CHECK(hipMemcpy(&bad, stale, 8, hipMemcpyDeviceToHost));
                        printf(",\n \"barrier_%s%s_%u\": {\"us_per_boundary\": %.3f, \"stale\": %llu}", tree ? "tree" : "flat", fence ? "" : "_nofence", blocks,
                               us / (2 * rounds), bad);
                }
        }
        printf("}\n");
        return 0;
}
