// Correctness probe for the phase-exchange scheme of csrc/heads_coop.hip: agent-scope stores, two-level grid barrier with s_waitcnt vmcnt(0), then
// reads of ANOTHER workgroup's data by (a) agent-scope scalar loads, (b) sc1 buffer loads of 16 bytes.  Repeated `iters` times on the same addresses.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
__device__ __forceinline__ void barrier2(unsigned* ctr, unsigned nwg, unsigned k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned r = blockIdx.x & 7u, ng = nwg < 8u ? nwg : 8u, gsize = (nwg - r + 7u) / 8u;
        if (atomicAdd(ctr + 16 + 16 * r, 1u) + 1u == k * gsize) atomicAdd(ctr, 1u);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k * ng) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
extern "C" __global__ __launch_bounds__(1024) void exchange_probe(int iters, unsigned* ctr, float* data, int per_wg, int* errors) {
    const unsigned nwg = gridDim.x;
    const int wg = blockIdx.x, nb = (wg + 37) % nwg;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)data, 0, (int)(nwg * per_wg * 4), 0x00020000);
    unsigned k = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < per_wg; i += 1024)
            __hip_atomic_store(data + (size_t)wg * per_wg + i, (float)(it * 7919 + wg * 31 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        barrier2(ctr, nwg, ++k);
        for (int i = threadIdx.x; i < per_wg; i += 1024) {
            const float want = (float)(it * 7919 + nb * 31 + i);
            const float got = __hip_atomic_load(data + (size_t)nb * per_wg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != want) atomicAdd(errors, 1);
        }
        for (int i = 4 * threadIdx.x; i < per_wg; i += 4096) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)nb * per_wg + i) * 4), 0, 0x10);
            for (int j = 0; j < 4; ++j) if (__builtin_bit_cast(float, v[j]) != (float)(it * 7919 + nb * 31 + i + j)) atomicAdd(errors + 1, 1);
        }
        // (c) 8-byte agent-scope atomic loads; (d) sc0 sc1 (system-scope) 16-byte buffer loads; (e) 16-byte buffer loads after an agent-scope acquire fence by this wave
        for (int i = 2 * threadIdx.x; i < per_wg; i += 2048) {
            const unsigned long long v = __hip_atomic_load((const unsigned long long*)(data + (size_t)nb * per_wg + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float lo = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffu)), hi = __builtin_bit_cast(float, (unsigned)(v >> 32));
            if (lo != (float)(it * 7919 + nb * 31 + i)) atomicAdd(errors + 2, 1);
            if (hi != (float)(it * 7919 + nb * 31 + i + 1)) atomicAdd(errors + 2, 1);
        }
        for (int i = 4 * threadIdx.x; i < per_wg; i += 4096) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)nb * per_wg + i) * 4), 0, 0x11);
            for (int j = 0; j < 4; ++j) if (__builtin_bit_cast(float, v[j]) != (float)(it * 7919 + nb * 31 + i + j)) atomicAdd(errors + 3, 1);
        }
        for (int i = 4 * threadIdx.x; i < per_wg; i += 4096) {
            typedef __attribute__((ext_vector_type(4))) float f4;
            const f4 v = __builtin_nontemporal_load((const f4*)(data + (size_t)nb * per_wg + i));
            const float f[4] = {v[0], v[1], v[2], v[3]};
            for (int j = 0; j < 4; ++j) if (f[j] != (float)(it * 7919 + nb * 31 + i + j)) atomicAdd(errors + 4, 1);
        }
        barrier2(ctr, nwg, ++k);
    }
}
extern "C" int exchange_probe_launch(int iters, int blocks, void* ctr, void* data, int per_wg, void* errors, hipStream_t s) {
    hipLaunchKernelGGL(exchange_probe, dim3(blocks), dim3(1024), 0, s, iters, (unsigned*)ctr, (float*)data, per_wg, (int*)errors);
    return (int)hipGetLastError();
}
