// Microbenchmark: issue cost of integer multiplies vs adds on one wave per SIMD (s_memtime around an unrolled chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(uint32_t* out, unsigned long long* clk, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x + a, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { x0 += b; x1 += b; x2 += b; x3 += b; }
            if (MODE == 1) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; }
            if (MODE == 2) { x0 = __umul24(x0, b); x1 = __umul24(x1, b); x2 = __umul24(x2, b); x3 = __umul24(x3, b); }
            if (MODE == 3) { x0 = __umulhi(x0, b); x1 = __umulhi(x1, b); x2 = __umulhi(x2, b); x3 = __umulhi(x3, b); }
            if (MODE == 4) { x0 ^= x0 >> 15; x1 ^= x1 >> 15; x2 ^= x2 >> 15; x3 ^= x3 >> 15; }
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    uint32_t* out; unsigned long long* clk;
    hipMalloc(&out, 1 << 20); hipMalloc(&clk, 4096);
    const char* names[5] = {"v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mul_hi_u32", "xorshift (lshr+xor)"};
    for (int waves = 1; waves <= 8; ++waves)
    for (int m = 0; m < 5; m += 4) {
        // waves per SIMD: up to 4 from one workgroup (1024 threads); 5-8 by two workgroups per CU (grid = 2 x 256 CUs)
        dim3 g(waves <= 4 ? 1 : 512), b(waves <= 4 ? 256 * waves : 128 * waves);
        if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, clk, 1u, 3u);
        if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, clk, 1u, 3u);
        if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, clk, 1u, 3u);
        if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, clk, 1u, 3u);
        if (m == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, clk, 1u, 3u);
        unsigned long long cs[512]; hipMemcpy(cs, clk, 8 * g.x, hipMemcpyDeviceToHost);
        unsigned long long c = 0; for (unsigned i = 0; i < g.x; ++i) c += cs[i]; c /= g.x;
        const double n = 256.0 * 16 * 4 * (m == 4 ? 2 : 1);
        printf("%d wave(s)/SIMD  %-22s %6.2f clk per instruction (per wave)\n", waves, names[m], (double)c / n);
    }
    return 0;
}
