// Probe: cost of a 1-KiB wave store (16 B per lane) as a function of how the 64 lanes tile memory.
//   pattern 0: 16 rows x 64 B  (lane & 15 = row, lane >> 4 = 16-byte chunk)      -- the NT epilogue's shape
//   pattern 1:  8 rows x 128 B (lane >> 3 = row, lane & 7 = chunk)               -- whole cache lines
//   pattern 2:  4 rows x 256 B
//   pattern 3:  1 row  x 1 KiB (fully contiguous)
// Every workgroup (512 threads, one per CU) owns a [rows x pitch] region and walks it top to bottom like an epilogue does.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(512) void store_probe(int pattern, int pitch, int iters, char* base, size_t region, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int row, chunk, rows_per;
    if (pattern == 0) { row = lane & 15; chunk = lane >> 4; rows_per = 16; }
    else if (pattern == 1) { row = lane >> 3; chunk = lane & 7; rows_per = 8; }
    else if (pattern == 2) { row = lane >> 4; chunk = lane & 15; rows_per = 4; }
    else { row = 0; chunk = lane; rows_per = 1; }
    char* p = base + (size_t)blockIdx.x * region + (size_t)(wave * rows_per + row) * pitch + chunk * 16;
    const size_t step = (size_t)8 * rows_per * pitch;            // 8 waves advance together
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        *(u32x4*)(p + (size_t)(i % 64) * step) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

extern "C" int store_probe_launch(int pattern, int pitch, int iters, int blocks, void* base, size_t region, void* clk, hipStream_t s) {
    hipLaunchKernelGGL(store_probe, dim3(blocks), dim3(512), 0, s, pattern, pitch, iters, (char*)base, region, (unsigned long long*)clk);
    return (int)hipGetLastError();
}
