// Multi-head self-attention over the packed (variable-length) MMBert token matrix, head dim 64,
// bf16 MFMA (v_mfma_f32_16x16x32_bf16), flash-style: no S x S tensor ever reaches HBM.
//
//   scores = q.k^T / sqrt(64) + key_bias[key]      (key_bias = (1-mask)*-10000, REF get_extended_attention_mask)
//   P      = softmax(scores);  Pd = dropout(P, p);  ctx = Pd . V        (HF eager_attention_forward)
//
// Layout: qkv [tokens, 3H] bf16 (q | k | v, head h at columns h*64), ctx [tokens, H] bf16,
// lse [tokens, heads] fp32 (natural log).  A sequence is (start, len); a work item is a 64-row tile
// of one sequence x one head.  Workgroup = 4 waves, each wave owns 16 rows.
//
// All products are computed "key/query on the lane": S^T = K.Q^T puts the query on lane&15 and the
// keys in the accumulator registers, so (a) row statistics are per-lane scalars plus two cross-group
// shuffles and (b) the accumulator, packed to bf16, IS the B operand of the next product
// (O^T = V^T.P^T, dQ^T = K^T.dS^T, dV^T = dO^T.P, dK^T = Q^T.dS) with V^T/K^T/dO^T/Q^T fetched by
// ds_read_b64_tr_b16 from row-major LDS tiles.  Backward is two kernels (dQ per query tile, dK/dV
// per key tile): 7 products instead of 5, but no fp32 atomics and bitwise reproducible.
//
// Dropout index of P[i][j] in sequence s, head h: elem_base[s] + (h*S + i)*Spad + j, Spad = S
// rounded up to 4 (elem_base multiples of 4), so forward and both backward kernels agree.
#include "common.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

struct AttnArgs {
    const bf16_t* qkv; int ld_qkv;
    bf16_t* ctx;             // fwd out   [tokens, H]
    const bf16_t* dctx;      // bwd in    [tokens, H]
    bf16_t* dqkv;            // bwd out   [tokens, 3H]
    float* lse;              // [tokens, heads]
    float* delta;            // [tokens, heads]
    const float* key_bias;   // [tokens]
    const int* seq_start; const int* seq_len; const unsigned* elem_base;
    const int* tile_seq; const int* tile_r0;
    int H, heads;
    float scale;
    uint32_t dstream, dthr; float dscale;
};

// stage a [64 rows x 64 cols] bf16 tile (rows row0.., clamped to nrows-1) into LDS (8 KiB).
// MODE 0: chunk ^= row&7        (conflict-free ds_read_b128 row reads)
// MODE 1: chunk ^= ((row>>1)&3)<<1   (conflict-free ds_read_b64_tr_b16 transposed reads)
template <int MODE>
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* src, int ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = wave * 2 + i;
        const int r = chunk * 8 + (lane >> 3);
        const int c = lane & 7;
        const int sc = MODE == 0 ? (c ^ (r & 7)) : (c ^ (((r >> 1) & 3) << 1));
        const int gr = min(row0 + r, nrows - 1);
        __builtin_amdgcn_global_load_lds(GPTR(src + (size_t)gr * ld + sc * 8), LPTR(lds + chunk * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// transposed fragment for k-step ks (32 rows = row tiles 2ks, 2ks+1) and column tile ct (16 cols):
// lane (i = lane&15, g = lane>>4) receives tile[row(g,j)][ct*16 + i], row(g,j) = (2ks + (j>>2))*16 + 4g + (j&3)
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int ks, int ct, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = (2 * ks) * 16 + 4 * g + q, r1 = r0 + 16;
    const int o0 = r0 * 128 + ((ct ^ ((r0 >> 1) & 3)) << 5) + p * 8;
    const int o1 = r1 * 128 + ((ct ^ ((r1 >> 1) & 3)) << 5) + p * 8;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + o0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4 a, const f32x4 b) {
    bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return o;
}

__device__ __forceinline__ float group_sum(float v) {   // sum over the 4 lane groups that share lane&15
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v;
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v;
}

// =============================================================================================
// forward
// =============================================================================================
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[16384];       // K (mode 0) | V (mode 1)
    char* Ks = smem; char* Vs = smem + 8192;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seq = a.tile_seq[blockIdx.x], r0 = a.tile_r0[blockIdx.x], head = blockIdx.y;
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const int qi = r0 + wave * 16 + fr;                              // row inside the sequence
    const int qc = min(qi, S - 1);
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    bf16x8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const bf16x8*)(base + (size_t)qc * a.ld_qkv + kk * 32 + 8 * g);
    const bf16_t* kbase = base + a.H;
    const bf16_t* vbase = base + 2 * a.H;
    const float sl2 = a.scale * LOG2E;
    const unsigned rowbase = a.elem_base[seq] + (unsigned)((head * S + qc) * Spad);

    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m2 = -INFINITY, lpart = 0.f;

    for (int kv0 = 0; kv0 < S; kv0 += 64) {
        __syncthreads();                                             // previous tile fully consumed
        stage_tile<0>(Ks, kbase, a.ld_qkv, kv0, S, wave, lane);
        stage_tile<1>(Vs, vbase, a.ld_qkv, kv0, S, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, kt * 16 + fr, kk * 4 + g), qf[kk], s[kt], 0, 0, 0);
        }
        // scores (log2 domain), tile max
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = kv0 + kt * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = (key + r < S) ? a.key_bias[start + key + r] * LOG2E : -INFINITY;
                s[kt][r] = s[kt][r] * sl2 + b;
                tmax = fmaxf(tmax, s[kt][r]);
            }
        }
        tmax = group_max(tmax);
        const float mnew = fmaxf(m2, tmax);
        const float alpha = exp2f(m2 - mnew);                        // m2 = -inf on the first tile -> 0
        m2 = mnew;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bool keep[4] = {true, true, true, true};
            if (a.dthr) mmb_keep4(a.dstream, (uint64_t)rowbase + (kv0 + kt * 16 + 4 * g), a.dthr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[kt][r] - mnew);
                psum += p;
                s[kt][r] = a.dthr ? (keep[r] ? p * a.dscale : 0.f) : p;
            }
        }
        lpart = lpart * alpha + psum;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pf = pack8(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(Vs, ks, d, lane), pf, o[d], 0, 0, 0);
        }
    }
    const float l = group_sum(lpart);
    if (qi < S) {
        const float inv = 1.0f / l;
        bf16_t* orow = a.ctx + (size_t)(start + qi) * a.H + head * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 ov = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
            *(bf16x4*)(orow + d * 16 + 4 * g) = ov;
        }
        if (g == 0) a.lse[(size_t)(start + qi) * a.heads + head] = (m2 + log2f(l)) * LN2;
    }
}

// =============================================================================================
// backward, part 1: dQ (and delta = rowsum(dO*O)) per 64-query tile
// =============================================================================================
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[24576];       // K mode0 | K mode1 | V mode0
    char* Ks = smem; char* Kt = smem + 8192; char* Vs = smem + 16384;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seq = a.tile_seq[blockIdx.x], r0 = a.tile_r0[blockIdx.x], head = blockIdx.y;
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const int qi = r0 + wave * 16 + fr;
    const int qc = min(qi, S - 1);
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    const bf16_t* dob = a.dctx + (size_t)start * a.H + head * 64;
    const bf16_t* ob = a.ctx + (size_t)start * a.H + head * 64;
    bf16x8 qf[2], dof[2];
    float dl = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        qf[kk] = *(const bf16x8*)(base + (size_t)qc * a.ld_qkv + kk * 32 + 8 * g);
        dof[kk] = *(const bf16x8*)(dob + (size_t)qc * a.H + kk * 32 + 8 * g);
        const bf16x8 of = *(const bf16x8*)(ob + (size_t)qc * a.H + kk * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += bf2f(dof[kk][j]) * bf2f(of[j]);
    }
    const float delta = group_sum(dl);
    if (qi < S && g == 0) a.delta[(size_t)(start + qi) * a.heads + head] = delta;
    const float lse2 = a.lse[(size_t)(start + qc) * a.heads + head] * LOG2E;
    const bf16_t* kbase = base + a.H;
    const bf16_t* vbase = base + 2 * a.H;
    const float sl2 = a.scale * LOG2E;
    const unsigned rowbase = a.elem_base[seq] + (unsigned)((head * S + qc) * Spad);

    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kv0 = 0; kv0 < S; kv0 += 64) {
        __syncthreads();
        stage_tile<0>(Ks, kbase, a.ld_qkv, kv0, S, wave, lane);
        stage_tile<1>(Kt, kbase, a.ld_qkv, kv0, S, wave, lane);
        stage_tile<0>(Vs, vbase, a.ld_qkv, kv0, S, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ks, kt * 16 + fr, kk * 4 + g), qf[kk], s[kt], 0, 0, 0);
                dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Vs, kt * 16 + fr, kk * 4 + g), dof[kk], dp[kt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = kv0 + kt * 16 + 4 * g;
            bool keep[4] = {true, true, true, true};
            if (a.dthr) mmb_keep4(a.dstream, (uint64_t)rowbase + key, a.dthr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = 0.f;
                if (key + r < S) p = exp2f(s[kt][r] * sl2 + a.key_bias[start + key + r] * LOG2E - lse2);
                float dpm = dp[kt][r];
                if (a.dthr) dpm = keep[r] ? dpm * a.dscale : 0.f;
                s[kt][r] = p * (dpm - delta);                        // dS^T
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 dsf = pack8(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(Kt, ks, d, lane), dsf, dq[d], 0, 0, 0);
        }
    }
    if (qi < S) {
        bf16_t* drow = a.dqkv + (size_t)(start + qi) * a.ld_qkv + head * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 ov = {f2bf(dq[d][0] * a.scale), f2bf(dq[d][1] * a.scale), f2bf(dq[d][2] * a.scale), f2bf(dq[d][3] * a.scale)};
            *(bf16x4*)(drow + d * 16 + 4 * g) = ov;
        }
    }
}

// =============================================================================================
// backward, part 2: dK, dV per 64-key tile (wave owns 16 keys, sweeps all query tiles)
// =============================================================================================
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192 + 512];   // Q m0 | Q m1 | dO m0 | dO m1 | lse,delta
    char* Qs = smem; char* Qt = smem + 8192; char* Ds = smem + 16384; char* Dt = smem + 24576;
    float* stat = (float*)(smem + 32768);                              // [64] lse2, [64] delta
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seq = a.tile_seq[blockIdx.x], r0 = a.tile_r0[blockIdx.x], head = blockIdx.y;
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const int ki = r0 + wave * 16 + fr;                                // this lane's key
    const int kc = min(ki, S - 1);
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    const bf16_t* dob = a.dctx + (size_t)start * a.H + head * 64;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        kf[kk] = *(const bf16x8*)(base + a.H + (size_t)kc * a.ld_qkv + kk * 32 + 8 * g);
        vf[kk] = *(const bf16x8*)(base + 2 * a.H + (size_t)kc * a.ld_qkv + kk * 32 + 8 * g);
    }
    const float kb2 = a.key_bias[start + kc] * LOG2E;
    const float sl2 = a.scale * LOG2E;
    const unsigned ebase = a.elem_base[seq] + (unsigned)(head * S) * Spad;

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = 0; q0 < S; q0 += 64) {
        __syncthreads();
        stage_tile<0>(Qs, base, a.ld_qkv, q0, S, wave, lane);
        stage_tile<1>(Qt, base, a.ld_qkv, q0, S, wave, lane);
        stage_tile<0>(Ds, dob, a.H, q0, S, wave, lane);
        stage_tile<1>(Dt, dob, a.H, q0, S, wave, lane);
        if (threadIdx.x < 64) {
            const int q = min(q0 + (int)threadIdx.x, S - 1);
            stat[threadIdx.x] = a.lse[(size_t)(start + q) * a.heads + head] * LOG2E;
            stat[64 + threadIdx.x] = a.delta[(size_t)(start + q) * a.heads + head];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // D[row <-> query (A operand rows)][col <-> key (B operand = this lane's K / V row)]
                s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Qs, qt * 16 + fr, kk * 4 + g), kf[kk], s[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(Ds, qt * 16 + fr, kk * 4 + g), vf[kk], dp[qt], 0, 0, 0);
            }
        }
        f32x4 pm[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qt * 16 + 4 * g + r;                     // query row inside the tile
                const int q = q0 + ql;
                float p = 0.f;
                if (q < S && ki < S) p = exp2f(s[qt][r] * sl2 + kb2 - stat[ql]);
                bool keep = true;
                if (a.dthr) keep = mmb_keep(a.dstream, (uint64_t)ebase + (uint64_t)min(q, S - 1) * Spad + kc, a.dthr);
                const float kscale = a.dthr ? (keep ? a.dscale : 0.f) : 1.0f;
                pm[qt][r] = p * kscale;                                 // dropped P   -> dV
                s[qt][r] = p * (dp[qt][r] * kscale - stat[64 + ql]);    // dS          -> dK
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pf = pack8(pm[2 * ks], pm[2 * ks + 1]);
            const bf16x8 dsf = pack8(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(Dt, ks, d, lane), pf, dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr_frag(Qt, ks, d, lane), dsf, dk[d], 0, 0, 0);
            }
        }
    }
    if (ki < S) {
        bf16_t* drow = a.dqkv + (size_t)(start + ki) * a.ld_qkv + head * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 kvv = {f2bf(dk[d][0] * a.scale), f2bf(dk[d][1] * a.scale), f2bf(dk[d][2] * a.scale), f2bf(dk[d][3] * a.scale)};
            bf16x4 vvv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
            *(bf16x4*)(drow + a.H + d * 16 + 4 * g) = kvv;
            *(bf16x4*)(drow + 2 * a.H + d * 16 + 4 * g) = vvv;
        }
    }
}

// test/debug: the keep mask of one (sequence, head) as bytes [S, S]
__global__ void attn_mask_kernel(uint8_t* out, int S, unsigned elem_base, int head, uint32_t stream, uint32_t thr) {
    const int Spad = (S + 3) & ~3;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < S * S; e += gridDim.x * blockDim.x) {
        const int i = e / S, j = e % S;
        out[e] = mmb_keep(stream, (uint64_t)elem_base + (uint64_t)(head * S + i) * Spad + j, thr) ? 1 : 0;
    }
}

extern "C" {

static int fill_args(AttnArgs& a, const void* qkv, int H, int heads, const float* key_bias, const int* seq_start, const int* seq_len,
                     const unsigned* elem_base, const int* tile_seq, const int* tile_r0, float* lse,
                     uint32_t dstream, uint32_t dthr, float dscale) {
    if (heads <= 0 || H != heads * 64) return -1;      // head dim 64 only
    a.qkv = (const bf16_t*)qkv; a.ld_qkv = 3 * H; a.H = H; a.heads = heads; a.key_bias = key_bias;
    a.seq_start = seq_start; a.seq_len = seq_len; a.elem_base = elem_base; a.tile_seq = tile_seq; a.tile_r0 = tile_r0;
    a.lse = lse; a.scale = 0.125f; a.dstream = dstream; a.dthr = dthr; a.dscale = dscale;
    a.ctx = nullptr; a.dctx = nullptr; a.dqkv = nullptr; a.delta = nullptr;
    return 0;
}

int mmbert_attn_fwd(hipStream_t stream, const void* qkv, void* ctx, float* lse, const float* key_bias, int H, int heads,
                    const int* seq_start, const int* seq_len, const unsigned* elem_base, const int* tile_seq, const int* tile_r0, int ntiles,
                    uint32_t dstream, uint32_t dthr, float dscale) {
    if (ntiles <= 0) return 0;
    AttnArgs a;
    if (fill_args(a, qkv, H, heads, key_bias, seq_start, seq_len, elem_base, tile_seq, tile_r0, lse, dstream, dthr, dscale)) return -1;
    a.ctx = (bf16_t*)ctx;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(ntiles, heads), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_attn_bwd(hipStream_t stream, const void* qkv, const void* ctx, const void* dctx, void* dqkv, const float* lse, float* delta,
                    const float* key_bias, int H, int heads, const int* seq_start, const int* seq_len, const unsigned* elem_base,
                    const int* tile_seq, const int* tile_r0, int ntiles, uint32_t dstream, uint32_t dthr, float dscale) {
    if (ntiles <= 0) return 0;
    AttnArgs a;
    if (fill_args(a, qkv, H, heads, key_bias, seq_start, seq_len, elem_base, tile_seq, tile_r0, (float*)lse, dstream, dthr, dscale)) return -1;
    a.ctx = (bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv; a.delta = delta;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(ntiles, heads), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(ntiles, heads), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_attn_dropout_mask(hipStream_t stream, uint8_t* out, int S, unsigned elem_base, int head, uint32_t rng_stream, uint32_t thr16) {
    hipLaunchKernelGGL(attn_mask_kernel, dim3(256), dim3(256), 0, stream, out, S, elem_base, head, rng_stream, thr16);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
