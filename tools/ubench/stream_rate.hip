// What a plain streaming kernel reaches at the LayerNorm launches' sizes (tools/ubench/stream_rate.py): NR input streams of
// n 16-byte words added (as bf16 pairs they would need unpacking: here as packed uint32 adds, the arithmetic is free either way)
// into one output stream, grid-stride, 16 B per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int NR>
__global__ __launch_bounds__(256) void stream_kernel(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u4 v = __builtin_nontemporal_load(a + i);
        if (NR > 1) v += __builtin_nontemporal_load(b + i);
        if (NR > 2) v += __builtin_nontemporal_load(c + i);
        o[i] = v;
    }
}
extern "C" void stream_launch(int nr, int blocks, const void* a, const void* b, const void* c, void* o, size_t n, hipStream_t s) {
    if (nr == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, s, (const u4*)a, (const u4*)b, (const u4*)c, (u4*)o, n);
    else if (nr == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(256), 0, s, (const u4*)a, (const u4*)b, (const u4*)c, (u4*)o, n);
    else hipLaunchKernelGGL(stream_kernel<3>, dim3(blocks), dim3(256), 0, s, (const u4*)a, (const u4*)b, (const u4*)c, (u4*)o, n);
}

// ---- what does each ingredient of a LayerNorm forward cost on top of a row-structured copy?  One wave per 768-column bf16 row
// (3 x 8 B per lane), next row requested before the current one is processed (as ln_fwd_kernel), LEVEL:
//   0 copy   1 + unpack / repack through fp32   2 + row sum (wave reduction) before the store   3 + second reduction (variance)
//   4 + the dropout hash on the output
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x2C1B3C6Du; x ^= x >> 16; return x; }
template <int LEVEL>
__global__ __launch_bounds__(256) void row_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int M, const float* __restrict__ gamma) {
    const int lane = threadIdx.x & 63;
    const int step = gridDim.x * 4;
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    bf16x4 nx[3];
    if (i < M) for (int c = 0; c < 3; ++c) nx[c] = *(const bf16x4*)(x + (size_t)i * 768 + c * 256 + lane * 4);
    float g[3][4];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 4; ++r) g[c][r] = LEVEL >= 1 ? gamma[c * 256 + lane * 4 + r] : 1.f;
    for (; i < M; i += step) {
        bf16x4 cur[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) cur[c] = nx[c];
        if (i + step < M) {
#pragma unroll
            for (int c = 0; c < 3; ++c) nx[c] = *(const bf16x4*)(x + (size_t)(i + step) * 768 + c * 256 + lane * 4);
        }
        if (LEVEL == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *(bf16x4*)(y + (size_t)i * 768 + c * 256 + lane * 4) = cur[c];
            continue;
        }
        float v[3][4], s = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[c][r] = (float)cur[c][r]; s += v[c][r]; }
        float mean = 0.f, rstd = 1.f;
        if (LEVEL >= 2) mean = wsum(s) * (1.f / 768);
        if (LEVEL >= 3) {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = v[c][r] - mean; q += d * d; }
            rstd = rsqrtf(wsum(q) * (1.f / 768) + 1e-12f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[c][r] - mean) * rstd * g[c][r];
            if (LEVEL >= 4) {
                const uint32_t p = (uint32_t)(((size_t)i * 768 + c * 256 + lane * 4) >> 1);
                const uint32_t h0 = mix(p * 0x9E3779B1u + 12345u), h1 = mix((p + 1) * 0x9E3779B1u + 12345u);
                o[0] = (int16_t)(h0 & 0xFFFF) >= -26214 ? o[0] * 1.1111f : 0.f; o[1] = (int16_t)(h0 >> 16) >= -26214 ? o[1] * 1.1111f : 0.f;
                o[2] = (int16_t)(h1 & 0xFFFF) >= -26214 ? o[2] * 1.1111f : 0.f; o[3] = (int16_t)(h1 >> 16) >= -26214 ? o[3] * 1.1111f : 0.f;
            }
            bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *(bf16x4*)(y + (size_t)i * 768 + c * 256 + lane * 4) = ob;
        }
    }
}
extern "C" void row_launch(int level, int blocks, const void* x, void* y, int M, const float* gamma, hipStream_t s) {
#define RL(L) hipLaunchKernelGGL(row_kernel<L>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, M, gamma)
    if (level == 0) RL(0); else if (level == 1) RL(1); else if (level == 2) RL(2); else if (level == 3) RL(3); else RL(4);
#undef RL
}
