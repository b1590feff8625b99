// VALU issue-rate probe (gfx950): 8 independent chains per lane of one instruction kind, long unrolled loop; one wave per SIMD
// (256-thread workgroups, 1 per CU) and 2 waves per SIMD (512).  Reported: clk per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int KIND>
__global__ void probe(uint32_t* out, int iters, uint32_t c) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i * 40503u + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (KIND == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 7) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x[i]));
                if (KIND == 8) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(c));
                if (KIND == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (KIND == 11) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 12) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x[i]));
                if (KIND == 13) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 14) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
                if (KIND == 15) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345u) out[0] = s;
}
extern "C" int valu_probe(int kind, int threads, int blocks, int iters, uint32_t* out, hipStream_t s) {
#define L(K) case K: hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(threads), 0, s, out, iters, 0x9E3779B1u); break;
    switch (kind) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) default: return -1; }
    return (int)hipGetLastError();
}
