// Shared device helpers for the MMBert gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The only diagnostic switch the product sources know is MMB_STAMPS (tools/stamp_*.py: s_memtime stamps, never the shipped library).
// The round-3 timing-only ablation switches of the NT GEMM's K step computed WRONG outputs by construction; they live on the git tag
// r3-gemm-ablations and are refused here so that a stray -D can never produce a silently wrong product build.
#if defined(MMB_EXP_NOMFMA) || defined(MMB_EXP_ALOADS) || defined(MMB_EXP_NOBLOADS) || defined(MMB_EXP_EXEC0) || defined(MMB_EXP_WAVEA) || \
    defined(MMB_EXP_NOFRAGS) || defined(MMB_AB_FREE_HASH)
#error "MMB_EXP_* / MMB_AB_* ablation switches are not part of the product sources (see git tag r3-gemm-ablations)"
#endif

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MMB_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

#include <atomic>
// One-time opt-in of a kernel to more than 64 KiB of dynamic LDS, per device.  `done` is the call site's bit mask of devices
// already set (thread-safe; the attribute call itself is idempotent).  Returns 0 or a hipError_t.
static inline int mmb_allow_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// CU count of the current device (cached per device: a process may drive several)
static inline int mmb_device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (!c) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        c = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

// mmbert_set_deterministic (rowwise.hip defines the flag; one .so): 1 = every fp32 sum of the library is formed in an order that does not
// depend on how workgroups are scheduled -- slabs + ordered reduces (or a single adder per address) instead of fp32 atomics whose arrival
// order varies: the CE loss sums, the heads' skinny products, the weight-gradient kernel's bias sums (no token split), the LayerNorm
// partial-sum fold, column sums; the embedding scatter and the data-parallel row block go through mmbert_id_runs_sum_rows
// (no sort: one workgroup per row, the first row of an id collects the id's rows in ascending order and adds them once).
extern std::atomic<int> g_mmb_deterministic;
static inline bool mmb_deterministic() { return g_mmb_deterministic.load(std::memory_order_relaxed) != 0; }

// address-space casts for the LDS-DMA builtin
#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// Counter-based dropout RNG.  One 32-bit hash serves TWO consecutive elements (16 bits each), so
// the keep probability is quantised to 1/65536.  stream = mix(seed, site) is computed on the host
// (mmbert_rng_stream) and passed as a kernel argument; idx is the element's flat index in the
// tensor the dropout acts on.  Forward and backward kernels regenerate identical masks.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mmb_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// dropout bits of an element pair: Weyl sequence (pair_idx * golden ratio + stream) through ONE xorshift-multiply-
// xorshift round.  The full two-multiply finaliser (mmb_hash32, still used to derive streams) costs ~9 VALU ops per
// pair and was 38 % of the attention kernels' VALU work; this is 5, and Bernoulli sampling at 16-bit granularity
// does not need full avalanche (tests check keep rates and forward/backward mask identity).
// (A quad variant -- one 64-bit product per 4 elements -- was costed in round 1: it saves 0.75 VALU operations per score
// where 4 consecutive keys sit in one lane (forward, dQ) and costs 1 more in the dK/dV kernel, where the keys of a quad sit
// on 4 lanes: no net gain, not adopted.)
#define MMB_WEYL 0x9E3779B1u
// seed = pair_idx * MMB_WEYL + stream is linear in the index: callers that walk an index grid keep a per-lane seed and add
// (wave-uniform) multiples of MMB_WEYL instead of multiplying per pair
__host__ __device__ __forceinline__ uint32_t mmb_pair_mix(uint32_t x) {
    // (round 2 timed a "free hash" build of this function -- return x -- as an upper bound for any cheaper generator: attention forward
    // -10.6 %, backward -6.2 %; that switch is gone from the product source, see the #error at the top)
    // both folds shift by 16: hipcc emits each as ONE v_xor_b32_sdwa (src1_sel:WORD_1) -- round 2's first fold (>> 15) was a shift
    // plus an xor, one VALU instruction more per element pair in every dropout site (round 3: attention forward 330 -> 314 VALU
    // instructions per 64-key tile and wave)
    x ^= x >> 16; x *= 0x2C1B3C6Du; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t mmb_pair_bits(uint32_t stream, uint32_t pair_idx) {
    return mmb_pair_mix(pair_idx * MMB_WEYL + stream);
}
// THE keep rule: a 16-bit half v of the hash word keeps its element iff, read as SIGNED, it is >= thr16 - 32768 (drop
// probability thr16 / 65536 exactly as for an unsigned compare).  Signed so that two elements can be decided at once on the
// packed halves: saturating v_pk_sub_i16 against the packed threshold, then v_pk_ashrrev_i16 15 -> 0xFFFF in dropped halves.
__host__ __device__ __forceinline__ bool mmb_keep16(uint32_t v16, uint32_t thr16) {
    return (int16_t)(uint16_t)v16 >= (int16_t)(uint16_t)(thr16 - 32768u);
}
__host__ __device__ __forceinline__ uint32_t mmb_thr_packed(uint32_t thr16) {
    const uint32_t t = (thr16 - 32768u) & 0xFFFFu;
    return t | (t << 16);
}
// 0xFFFF in every half of h whose element is DROPPED (thr_pk = mmb_thr_packed(thr16)): 2 VALU operations per pair
__device__ __forceinline__ uint32_t mmb_drop_mask2(uint32_t h, uint32_t thr_pk) {
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    s16x2_t d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, h), __builtin_bit_cast(s16x2_t, thr_pk));
    d = d >> (s16x2_t){15, 15};
    return __builtin_bit_cast(uint32_t, d);
}
// keep flag of element idx (idx = 2*pair + sub)
__host__ __device__ __forceinline__ bool mmb_keep(uint32_t stream, uint64_t idx, uint32_t thr16) {
    uint32_t h = mmb_pair_bits(stream, (uint32_t)(idx >> 1));
    return mmb_keep16((idx & 1) ? (h >> 16) : (h & 0xFFFFu), thr16);
}
// keep flags of 4 consecutive elements starting at an idx that is a multiple of 2
__device__ __forceinline__ void mmb_keep4(uint32_t stream, uint64_t idx0, uint32_t thr16, bool k[4]) {
    uint32_t p = (uint32_t)(idx0 >> 1);
    uint32_t h0 = mmb_pair_bits(stream, p), h1 = mmb_pair_bits(stream, p + 1);
    k[0] = mmb_keep16(h0 & 0xFFFFu, thr16); k[1] = mmb_keep16(h0 >> 16, thr16);
    k[2] = mmb_keep16(h1 & 0xFFFFu, thr16); k[3] = mmb_keep16(h1 >> 16, thr16);
}

typedef __attribute__((ext_vector_type(2))) float mmb_f2;

// GELU, erf form (HF ACT2FN["gelu"]), with ONE transcendental per element.  The GEMM epilogues that apply it are VALU-bound
// (112 elements per lane and tile), and v_exp_f32 / v_rcp_f32 issue at quarter rate, so the textbook erf (Abramowitz-Stegun
// 7.1.26: rcp + exp + 6 FMA, used here until round 1) cost 14 VALU instructions per element.  Instead, with a = |x|:
//     gelu(x)  = max(x, 0) - a * T(a),        T(a) = 0.5 erfc(a / sqrt 2) = 2^(-a Q(a) - 1)
//     gelu'(x) = x > 0 ? 1 - g(a) : g(a),     g(a) = gelu'(-a) = T(a) - a phi(a) = 2^(-a^2 / (2 ln 2)) * W(a)
// Q (degree 4) and W (degree 7) are weighted min-max fits (scipy, float64; checked in float32 Horner form on 2e5 points of
// [0, 12]): |gelu error| <= 5.7e-7 absolute, |gelu' error| <= 4.1e-6 absolute -- three to four orders below the bf16 rounding
// of the values they produce; the negative tail keeps its RELATIVE accuracy (no 1 - erf cancellation).  a is clamped to 12
// (T, g < 1e-32 there), which also keeps the polynomials finite for any input.
// (round 5) min(|x|, 12) and max(x, 0) through v_med3_f32: as fminf / fmaxf they each came with a canonicalising `v_max_f32 x, x` in front
// (the library is built with -fno-finite-math-only, so LLVM keeps IEEE sNaN quieting) -- 3 of the 13 VALU instructions per element of the
// FFN-up epilogue, which is VALU-issue bound (1 488 VALU + 115 exp per wave and 224 x 256 tile).  Same values for every finite x.
__device__ __forceinline__ float mmb_clamp_abs12(float x) { return __builtin_amdgcn_fmed3f(fabsf(x), 0.0f, 12.0f); }
// (max(x, 0) as inline asm: hipcc folds med3(x, 0, inf) back into a max WITH the canonicalising v_max_f32 x, x in front)
__device__ __forceinline__ float mmb_relu(float x) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = mmb_clamp_abs12(x);
    float q = 0.000488118665642307f;
    q = fmaf(q, a, -0.007198809871008205f);
    q = fmaf(q, a, 0.05214680078704519f);
    q = fmaf(q, a, 0.4595957249475095f);
    q = fmaf(q, a, 1.1510005681479196f);
    const float t = __builtin_amdgcn_exp2f(fmaf(-a, q, -1.0f));
    return fmaf(-a, t, mmb_relu(x));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float a = mmb_clamp_abs12(x);
    float w = -0.00016849001371319273f;
    w = fmaf(w, a, 0.002318365387269111f);
    w = fmaf(w, a, -0.013920757956130972f);
    w = fmaf(w, a, 0.04993439894451681f);
    w = fmaf(w, a, -0.1258212077485634f);
    w = fmaf(w, a, 0.2479567789223165f);
    w = fmaf(w, a, -0.7976617799265314f);
    w = fmaf(w, a, 0.49999600551278855f);
    const float g = __builtin_amdgcn_exp2f(-0.72134752044448170f * a * a) * w;
    return x > 0.0f ? 1.0f - g : g;
}

// Two elements at a time (round 5): the same Horner steps as v_pk_fma_f32 / v_pk_mul_f32 -- one instruction per step and PAIR where the
// scalar forms spend one per element (hipcc's SLP pass packs the derivative's polynomial on its own, but not the forward's, whose steps
// carry literal constants -- v_fmaak_f32 -- that the packed encoding cannot hold).  Every operation is the IEEE fma / mul of the scalar
// form on each half: the results are BIT-identical to gelu_erf / gelu_erf_grad.  In the GEMM epilogues that apply them (224 x 256 tile:
// 112 elements per lane, VALU-issue bound) the forward form goes from 8 + 1 exp to 5 + 1 exp instructions per element.
__device__ __forceinline__ mmb_f2 gelu_erf2(mmb_f2 x) {
    const mmb_f2 a = {mmb_clamp_abs12(x.x), mmb_clamp_abs12(x.y)};
    mmb_f2 q = (mmb_f2){0.000488118665642307f, 0.000488118665642307f};
    q = q * a + (mmb_f2){-0.007198809871008205f, -0.007198809871008205f};
    q = q * a + (mmb_f2){0.05214680078704519f, 0.05214680078704519f};
    q = q * a + (mmb_f2){0.4595957249475095f, 0.4595957249475095f};
    q = q * a + (mmb_f2){1.1510005681479196f, 1.1510005681479196f};
    const mmb_f2 e = -a * q + (mmb_f2){-1.0f, -1.0f};
    const mmb_f2 t = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const mmb_f2 r = {mmb_relu(x.x), mmb_relu(x.y)};
    return -a * t + r;
}
__device__ __forceinline__ mmb_f2 gelu_erf_grad2(mmb_f2 x) {
    const mmb_f2 a = {mmb_clamp_abs12(x.x), mmb_clamp_abs12(x.y)};
    mmb_f2 w = (mmb_f2){-0.00016849001371319273f, -0.00016849001371319273f};
    w = w * a + (mmb_f2){0.002318365387269111f, 0.002318365387269111f};
    w = w * a + (mmb_f2){-0.013920757956130972f, -0.013920757956130972f};
    w = w * a + (mmb_f2){0.04993439894451681f, 0.04993439894451681f};
    w = w * a + (mmb_f2){-0.1258212077485634f, -0.1258212077485634f};
    w = w * a + (mmb_f2){0.2479567789223165f, 0.2479567789223165f};
    w = w * a + (mmb_f2){-0.7976617799265314f, -0.7976617799265314f};
    w = w * a + (mmb_f2){0.49999600551278855f, 0.49999600551278855f};
    const mmb_f2 e = ((mmb_f2){-0.72134752044448170f, -0.72134752044448170f} * a) * a;
    const mmb_f2 g = (mmb_f2){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} * w;
    const mmb_f2 og = (mmb_f2){1.0f, 1.0f} - g;
    return (mmb_f2){x.x > 0.0f ? og.x : g.x, x.y > 0.0f ? og.y : g.y};
}

// Two pairs in lockstep: hipcc schedules one pair's chain after the other (register-pressure heuristics) and pads the v_exp_f32 -> use
// and pk_fma -> v_exp_f32 wait states with s_nop; written side by side, each pair's waits are filled by the other pair's instructions.
__device__ __forceinline__ void gelu_erf4(mmb_f2& x0, mmb_f2& x1) {
    const mmb_f2 a0 = {mmb_clamp_abs12(x0.x), mmb_clamp_abs12(x0.y)}, a1 = {mmb_clamp_abs12(x1.x), mmb_clamp_abs12(x1.y)};
    mmb_f2 q0 = (mmb_f2){0.000488118665642307f, 0.000488118665642307f}, q1 = q0;
    q0 = q0 * a0 + (mmb_f2){-0.007198809871008205f, -0.007198809871008205f}; q1 = q1 * a1 + (mmb_f2){-0.007198809871008205f, -0.007198809871008205f};
    q0 = q0 * a0 + (mmb_f2){0.05214680078704519f, 0.05214680078704519f};    q1 = q1 * a1 + (mmb_f2){0.05214680078704519f, 0.05214680078704519f};
    q0 = q0 * a0 + (mmb_f2){0.4595957249475095f, 0.4595957249475095f};      q1 = q1 * a1 + (mmb_f2){0.4595957249475095f, 0.4595957249475095f};
    q0 = q0 * a0 + (mmb_f2){1.1510005681479196f, 1.1510005681479196f};      q1 = q1 * a1 + (mmb_f2){1.1510005681479196f, 1.1510005681479196f};
    const mmb_f2 e0 = -a0 * q0 + (mmb_f2){-1.0f, -1.0f}, e1 = -a1 * q1 + (mmb_f2){-1.0f, -1.0f};
    const mmb_f2 r0 = {mmb_relu(x0.x), mmb_relu(x0.y)}, r1 = {mmb_relu(x1.x), mmb_relu(x1.y)};
    const mmb_f2 t0 = {__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)}, t1 = {__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)};
    x0 = -a0 * t0 + r0;
    x1 = -a1 * t1 + r1;
}
__device__ __forceinline__ void gelu_erf_grad4(mmb_f2 x0, mmb_f2 x1, mmb_f2& d0, mmb_f2& d1) {
    const mmb_f2 a0 = {mmb_clamp_abs12(x0.x), mmb_clamp_abs12(x0.y)}, a1 = {mmb_clamp_abs12(x1.x), mmb_clamp_abs12(x1.y)};
    mmb_f2 w0 = (mmb_f2){-0.00016849001371319273f, -0.00016849001371319273f}, w1 = w0;
#define MMB_W_STEP(C) w0 = w0 * a0 + (mmb_f2){C, C}; w1 = w1 * a1 + (mmb_f2){C, C};
    MMB_W_STEP(0.002318365387269111f) MMB_W_STEP(-0.013920757956130972f) MMB_W_STEP(0.04993439894451681f) MMB_W_STEP(-0.1258212077485634f)
    MMB_W_STEP(0.2479567789223165f) MMB_W_STEP(-0.7976617799265314f) MMB_W_STEP(0.49999600551278855f)
#undef MMB_W_STEP
    const mmb_f2 k = {-0.72134752044448170f, -0.72134752044448170f};
    const mmb_f2 e0 = (k * a0) * a0, e1 = (k * a1) * a1;
    const mmb_f2 t0 = {__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)}, t1 = {__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)};
    const mmb_f2 g0 = t0 * w0, g1 = t1 * w1;
    const mmb_f2 o0 = (mmb_f2){1.0f, 1.0f} - g0, o1 = (mmb_f2){1.0f, 1.0f} - g1;
    d0 = (mmb_f2){x0.x > 0.0f ? o0.x : g0.x, x0.y > 0.0f ? o0.y : g0.y};
    d1 = (mmb_f2){x1.x > 0.0f ? o1.x : g1.x, x1.y > 0.0f ? o1.y : g1.y};
}
