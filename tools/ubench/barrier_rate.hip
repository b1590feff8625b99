// Cost of a grid barrier on MI355X by how it is made visible (tools/ubench/barrier_rate.py): 256 workgroups x 1024 threads, `iters` barriers in a row,
// each workgroup writing one cache line before and reading its neighbour's after (so that a variant that is not coherent shows as errors).
//   mode 0: __threadfence() by every thread on both sides (the first version of csrc/heads_coop.hip)
//   mode 1: thread 0 only: release fence (agent) -> arrive -> spin -> acquire fence (agent)
//   mode 2: no fence at all; the exchanged line written / read with agent-scope relaxed atomics (sc1 accesses: coherent at the memory side)
//   mode 3: no fence, plain accesses (timing floor; NOT coherent across XCDs)
//   mode 4: as mode 2 with a TWO-LEVEL arrival: eight group counters (workgroup index mod 8), the last arriver of a group bumps the global one
//           (same-address atomics serialise at ~15 ns each: 256 arrivals on one word are the 3.9 us floor of modes 1-3)
#include <hip/hip_runtime.h>
__device__ __forceinline__ void arrive_wait(int mode, unsigned* ctr, unsigned nwg, unsigned k /* barriers so far, this one included */) {
    if (mode != 4) {
        atomicAdd(ctr, 1u);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k * nwg) __builtin_amdgcn_s_sleep(2);
        return;
    }
    const unsigned r = blockIdx.x & 7u, ng = nwg < 8u ? nwg : 8u, gsize = (nwg - r + 7u) / 8u;
    if ((atomicAdd(ctr + 16 + 16 * r, 1u) + 1u) == k * gsize) atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k * ng) __builtin_amdgcn_s_sleep(1);
}
extern "C" __global__ __launch_bounds__(1024) void barrier_probe(int mode, int iters, unsigned* ctr, float* lines, int* errors) {
    const unsigned nwg = gridDim.x;
    unsigned epoch = 0;
    const int wg = blockIdx.x, nb = (wg + 37) % nwg;
    for (int it = 0; it < iters; ++it) {
        const float val = (float)(it * 1000 + wg);
        if (threadIdx.x < 32) {
            if (mode == 2 || mode == 4) __hip_atomic_store(lines + wg * 32 + threadIdx.x, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else lines[wg * 32 + threadIdx.x] = val;
        }
        epoch += nwg;
        if (mode == 0) __threadfence();
        if (mode == 2 || mode == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (__syncthreads() alone does not wait for global stores)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (mode == 0) __threadfence();
            if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            arrive_wait(mode, ctr, nwg, 2 * it + 1);
            if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (mode == 0) __threadfence();
        if (threadIdx.x < 32) {
            const float want = (float)(it * 1000 + nb);
            const float got = (mode == 2 || mode == 4) ? __hip_atomic_load(lines + nb * 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lines[nb * 32 + threadIdx.x];
            if (got != want) atomicAdd(errors, 1);
        }
        // (a second barrier so that nobody overwrites a line before its reader has read it)
        epoch += nwg;
        __syncthreads();
        if (threadIdx.x == 0) {
            arrive_wait(mode, ctr, nwg, 2 * it + 2);
        }
        __syncthreads();
    }
}
extern "C" int barrier_probe_launch(int mode, int iters, int blocks, void* ctr, void* lines, void* errors, hipStream_t s) {
    hipLaunchKernelGGL(barrier_probe, dim3(blocks), dim3(1024), 0, s, mode, iters, (unsigned*)ctr, (float*)lines, (int*)errors);
    return (int)hipGetLastError();
}
