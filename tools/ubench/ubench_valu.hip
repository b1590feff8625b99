// VALU issue-cost microbenchmark (gfx950): shader cycles per instruction of the opcodes the GEMM epilogues and the attention kernels lean on.
// 1, 2 and 4 waves per SIMD, 8 independent dependency chains per wave, so the figure is issue cost, not latency.  Cycles are s_memtime
// (clock64) deltas of one wave, checked against the event time.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define DEF_KERNEL(NAME, L)                                                                                     \
__global__ void __launch_bounds__(1024) NAME(uint32_t* out, int iters, uint32_t c) {                           \
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    asm volatile("v_cmp_lt_u32 vcc, 7, %0\ns_mov_b64 s[10:11], vcc" :: "v"(r0) : "vcc", "s10", "s11");        \
    const long long t0 = clock64();                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                          \
            asm volatile(L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)                                                \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c) : "vcc", "s10", "s11"); \
    }                                                                                                           \
    const long long t1 = clock64();                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) *(long long*)(out + 256 * 1024) = t1 - t0;                       \
}

#define L_ADD(n)     "v_add_u32 %" #n ", %" #n ", %8\n"
#define L_ADDF(n)    "v_add_f32 %" #n ", %" #n ", %8\n"
#define L_MULF(n)    "v_mul_f32 %" #n ", %" #n ", %8\n"
#define L_FMA(n)     "v_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define L_FMAC(n)    "v_fmac_f32 %" #n ", %" #n ", %8\n"
#define L_AND(n)     "v_and_b32 %" #n ", %" #n ", %8\n"
#define L_XOR(n)     "v_xor_b32 %" #n ", %" #n ", %8\n"
#define L_LSHL(n)    "v_lshlrev_b32 %" #n ", 16, %" #n "\n"
#define L_MOV(n)     "v_mov_b32 %" #n ", %8\n"
#define L_MULLO(n)   "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define L_MULHI(n)   "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define L_MUL24(n)   "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define L_MAD24(n)   "v_mad_u32_u24 %" #n ", %" #n ", %8, %" #n "\n"
#define L_XORSD(n)   "v_xor_b32_sdwa %" #n ", %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define L_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define L_ADD3(n)    "v_add3_u32 %" #n ", %" #n ", %8, %" #n "\n"
#define L_XAD(n)     "v_xad_u32 %" #n ", %" #n ", %8, %" #n "\n"
#define L_ANDOR(n)   "v_and_or_b32 %" #n ", %" #n ", %8, %" #n "\n"
#define L_BFI(n)     "v_bfi_b32 %" #n ", %8, %" #n ", %" #n "\n"
#define L_PERM(n)    "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define L_BFE(n)     "v_bfe_i32 %" #n ", %" #n ", 15, 1\n"
#define L_EXP(n)     "v_exp_f32 %" #n ", %" #n "\n"
#define L_RCP(n)     "v_rcp_f32 %" #n ", %" #n "\n"
#define L_MED3(n)    "v_med3_f32 %" #n ", %" #n ", 0, %8\n"
#define L_MAX(n)     "v_max_f32 %" #n ", %" #n ", %8\n"
#define L_MAXE64(n)  "v_max_f32_e64 %" #n ", |%" #n "|, %8\n"
#define L_CVTPK(n)   "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %8\n"
#define L_PKSUB(n)   "v_pk_sub_i16 %" #n ", %" #n ", %8 clamp\n"
#define L_PKASHR(n)  "v_pk_ashrrev_i16 %" #n ", 15, %" #n "\n"
#define L_CNDVCC(n)  "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_CNDSGPR(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[10:11]\n"
#define L_CMP(n)     "v_cmp_ge_i32 vcc, %" #n ", %8\n"
#define L_CMPSD(n)   "v_cmp_ge_i32_sdwa vcc, sext(%" #n "), sext(%8) src0_sel:WORD_0 src1_sel:WORD_0\n"
#define L_CMPCND(n)  "v_cmp_ge_i32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_CMPSCND(n) "v_cmp_ge_i32_e64 s[10:11], %" #n ", %8\ns_nop 1\nv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[10:11]\n"


#define L_OR(n)      "v_or_b32 %" #n ", %" #n ", %8\n"
#define L_SUBU(n)    "v_sub_u32 %" #n ", %" #n ", %8\n"
#define L_SUBF(n)    "v_sub_f32 %" #n ", %" #n ", %8\n"
#define L_LSHR(n)    "v_lshrrev_b32 %" #n ", 16, %" #n "\n"
#define L_ASHR(n)    "v_ashrrev_i32 %" #n ", 16, %" #n "\n"
#define L_FMAMK(n)   "v_fmamk_f32 %" #n ", %" #n ", 0x3fb8aa3b, %8\n"
#define L_FMAAK(n)   "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3fb8aa3b\n"
#define L_MULLIT(n)  "v_mul_f32 %" #n ", 0x3fb8aa3b, %" #n "\n"
#define L_ADDLIT(n)  "v_add_u32 %" #n ", 0x9E3779B1, %" #n "\n"
#define L_MAX3(n)    "v_max3_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define L_BITOP3(n)  "v_bitop3_b32 %" #n ", %" #n ", %8, %" #n " bitop3:0x96\n"
#define L_MINI(n)    "v_min_i32 %" #n ", %" #n ", %8\n"
#define L_LDEXP(n)   "v_ldexp_f32 %" #n ", %" #n ", %8\n"
#define L_MOVDPP(n)  "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define L_ADDDPP(n)  "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define L_ACCW(n)    "v_accvgpr_write_b32 a" #n ", %" #n "\n"
#define L_ACCR(n)    "v_accvgpr_read_b32 %" #n ", a" #n "\n"
#define L_CMP2CND(n) "v_cmp_ge_i32 vcc, %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_CMP4CND(n) "v_cmp_ge_i32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_CMPS2CND(n) "v_cmp_ge_i32_e64 s[10:11], %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %8\nv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[10:11]\n"
#define L_CNDVCC2(n) "v_add_u32 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define L_ADDCO(n)   "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define L_ADDCCO(n)  "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"

#define ALL(X) X(ADD) X(ADDF) X(MULF) X(FMA) X(FMAC) X(AND) X(XOR) X(LSHL) X(MOV) X(MULLO) X(MULHI) X(MUL24) X(MAD24) X(XORSD) X(LSHLADD) X(ADD3) \
    X(XAD) X(ANDOR) X(BFI) X(PERM) X(BFE) X(EXP) X(RCP) X(MED3) X(MAX) X(MAXE64) X(CVTPK) X(PKSUB) X(PKASHR) X(CNDVCC) X(CNDSGPR) X(CMP) X(CMPSD) \
    X(CMPCND) X(CMPSCND) X(OR) X(SUBU) X(SUBF) X(LSHR) X(ASHR) X(FMAMK) X(FMAAK) X(MULLIT) X(ADDLIT) X(MAX3) X(BITOP3) X(MINI) X(LDEXP) \
    X(MOVDPP) X(ADDDPP) X(CMP2CND) X(CMP4CND) X(CMPS2CND) X(CNDVCC2) X(ADDCO) X(ADDCCO)
#define MK(N) DEF_KERNEL(k_##N, L_##N)
ALL(MK)

// packed fp32 operations take register pairs
#define DEF_PK(NAME, OPSTR)                                                                                     \
__global__ void __launch_bounds__(1024) NAME(uint32_t* out, int iters, uint32_t ci) {                          \
    typedef __attribute__((ext_vector_type(2))) float f2;                                                       \
    f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
    f2 cc = {1.0001f, 0.9999f};                                                                                 \
    const long long t0 = clock64();                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                          \
            asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(cc)); \
    }                                                                                                           \
    const long long t1 = clock64();                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y); \
    if (blockIdx.x == 0 && threadIdx.x == 0) *(long long*)(out + 256 * 1024) = t1 - t0;                       \
}
#define P_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define P_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define P_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
DEF_PK(k_PKFMA, P_FMA)
DEF_PK(k_PKMUL, P_MUL)
DEF_PK(k_PKADD, P_ADD)

// one MFMA (16x16x32 bf16, 16 clk of matrix pipe) followed by NV VALU instructions of one class: how much VALU work hides beside the matrix pipe
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int NV, int SLOW>
__global__ void __launch_bounds__(1024) k_mfma_mix(uint32_t* out, int iters, uint32_t c) {
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    f32x4_t acc[4] = {};
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.0f; }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
            asm volatile("" : "+v"(acc[u & 3]));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (SLOW) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r0) : "v"(c)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r0) : "v"(c));
                uint32_t t = r0; r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = r7; r7 = t;      // rotate the 8 chains (renaming only)
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ __float_as_uint(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) *(long long*)(out + 256 * 1024) = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, int, uint32_t);
// best-of-5 event time [ms]; *cyc = s_memtime span of wave 0 of workgroup 0 (the OLDEST wave of its SIMD: it is served first, so its span is the
// kernel's only when it runs alone -- the one-wave run gives the clock, the event time gives the throughput)
static double run(kern_t kern, int threads, uint32_t* out, int iters, long long* cyc) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kern<<<256, threads>>>(out, iters, 0x2C1B3Du);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a); kern<<<256, threads>>>(out, iters, 0x2C1B3Du); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) { best = ms; (void)hipMemcpy(cyc, out + 256 * 1024, 8, hipMemcpyDeviceToHost); }
    }
    return best;
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 1024 * 4 + 64);
    const int iters = 2000;
#define ROW(N) {#N, k_##N, 1},
    struct { const char* name; kern_t k; int per_line; } ks[] = { ALL(ROW) {"PKFMA", k_PKFMA, 1}, {"PKMUL", k_PKMUL, 1}, {"PKADD", k_PKADD, 1} };
    printf("# SIMD clocks per wave64 instruction (256 workgroups = one per CU; 1, 2, 4 waves per SIMD; 8 independent chains per wave).\n"
           "# clock = s_memtime span of the one-wave run / its event time; the 2- and 4-wave columns are event time x that clock / instructions per SIMD.\n"
           "# %-10s %10s %10s %10s\n", "opcode", "1 wave", "2 waves", "4 waves");
    for (auto& e : ks) {
        int per = 1;
        if (!strcmp(e.name, "CMPCND") || !strcmp(e.name, "CMPSCND")) per = 2;
        if (!strcmp(e.name, "CMP2CND") || !strcmp(e.name, "CMPS2CND")) per = 4;
        if (!strcmp(e.name, "CMP4CND")) per = 5;
        if (!strcmp(e.name, "CNDVCC2")) per = 3;
        const double n = (double)iters * 64 * per;                     // instructions per wave
        long long c1 = 0, c = 0;
        const double t1 = run(e.k, 256, out, iters, &c1), t2 = run(e.k, 512, out, iters, &c), t4 = run(e.k, 1024, out, iters, &c);
        const double mhz = c1 / (t1 * 1e3);
        printf("  %-10s %10.2f %10.2f %10.2f   (%4.0f MHz%s)\n", e.name, c1 / n, t2 * 1e3 * mhz / (2 * n), t4 * 1e3 * mhz / (4 * n), mhz, per > 1 ? ", average over the group's instructions" : "");
    }
    struct { const char* name; kern_t k; int nv; } ms[] = {
        {"MFMA+0", k_mfma_mix<0, 0>, 0}, {"MFMA+2fast", k_mfma_mix<2, 0>, 2}, {"MFMA+4fast", k_mfma_mix<4, 0>, 4}, {"MFMA+8fast", k_mfma_mix<8, 0>, 8},
        {"MFMA+2slow", k_mfma_mix<2, 1>, 2}, {"MFMA+4slow", k_mfma_mix<4, 1>, 4}, {"MFMA+8slow", k_mfma_mix<8, 1>, 8},
    };
    printf("# MFMA mixes: clk per group (one MFMA + its VALU instructions), per SIMD\n");
    for (auto& e : ms)
        for (int threads : {256, 512, 1024}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            e.k<<<256, threads>>>(out, 500, 3u); (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(a); e.k<<<256, threads>>>(out, 500, 3u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms_; (void)hipEventElapsedTime(&ms_, a, b); if (ms_ < best) best = ms_;
            }
            const double groups = 500.0 * 16 * (threads / 256);
            printf("%-12s %d w/SIMD  %8.3f ms  %7.2f ns/group/SIMD  (= %.1f clk at 2.38 GHz)\n", e.name, threads / 256, best, best * 1e6 / groups, best * 1e6 / groups * 2.38);
        }
    return 0;
}
