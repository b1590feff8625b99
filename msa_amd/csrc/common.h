// Shared device helpers for the MMBert gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MMB_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

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
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 16;
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

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7: far below bf16 resolution): 1 rcp + 1 exp + 6 FMA,
// about half the issue slots of libm erff in the GEMM epilogues (GELU is the erf form, HF ACT2FN["gelu"]).
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// d/dx gelu = Phi(x) + x phi(x); erf(x/sqrt2) and phi share exp(-x^2/2): one exp and one rcp per element
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(1.0f + 0.3275911f * 0.70710678118654752f * fabsf(x));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half_erf = copysignf(0.5f - 0.5f * poly * e, x);
    return 0.5f + half_erf + x * (0.3989422804014327f * e);
}
