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
#include <cstdlib>
#include "common.h"
#include <type_traits>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

struct AttnArgs {
    const bf16_t* qkv; int ld_qkv;
    bf16_t* ctx;             // fwd out   [tokens, H]
    const bf16_t* dctx;      // bwd in    [tokens, H]
    bf16_t* dqkv;            // bwd out   [tokens, 3H]
    float* lse;              // [tokens, heads]
    float* delta;            // [tokens, heads]
    const float* key_bias;   // per sequence, padded: bias_start[s] + key, entries past the sequence length <= -1e30 (ceil128(S) entries)
    const int* bias_start;
    const int* kv_len;       // optional, per sequence: the keys at and past kv_len[s] are all masked out (bias <= -10000: attn_kv_len_kernel)
    const int* seq_start; const int* seq_len; const unsigned* elem_base;
    const int* tile_seq; const int* tile_r0;
    // split (valid-first) layout, all optional: the QUERY rows of tile t sit at packed row tile_qshift[t] + (index in the sequence)
    // and end at index tile_qend[t]; split != 0: a sequence owns only its first kv_len[s] rows at seq_start[s] (dK/dV kernel)
    const int* tile_qshift; const int* tile_qend; int split;
    // backward, optional, per sequence: the query rows at index >= q_limit[s] have an exactly-zero output gradient (dO row == 0, hence
    // delta == 0 and dS == 0): dQ is zero there and they add nothing to dK / dV, so both kernels stop their query range at it
    // (round 4: the top encoder layer, whose output gradient lives on the MLM-labelled rows and the [CLS] rows only)
    const int* q_limit;
    int H, heads;
    float scale;
    uint32_t dstream, dthr; float dscale;
    // launch order: 1 = grid (heads, tiles), the heads of a tile are dispatched together and the tile list's order (longest work
    // first, ops.SplitLayout) holds for the WHOLE launch; 0 = grid (tiles, heads), every head walks the list on its own
    int head_fast;
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
// forward.  Work item = 128 query rows of one sequence x one head; 4 waves x 32 rows (two 16-row
// query blocks per wave share every K / V fragment read).  K, V and the key-bias slice of the next
// 64-key tile stream into the second LDS buffer by LDS-DMA while the current tile is consumed (one
// barrier per tile).  VALU work per probability (the binding resource at head dim 64) is kept minimal:
//   * the key bias enters as the INITIAL ACCUMULATOR of S^T = K.Q^T (bias/scale), so scaling and
//     max-subtraction are ONE fma feeding v_exp_f32:  p = exp2(acc*c - m*c),  c = scale*log2(e);
//   * the dropout scale 1/(1-p) is applied once to the output row, not to every probability;
//   * the transposed V reads go through inline asm (see tr_read in gemm.hip: the builtin makes hipcc
//     drain vmcnt(0) in front of them, which would serialise the prefetch).
// =============================================================================================
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& dst, unsigned lds_addr) {
    // "memory": the read must stay behind the s_waitcnt / s_barrier that publish the LDS-DMA data (without it hipcc
    // hoisted these reads above the barrier: stale LDS -> NaN)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}
// two transposed reads (rows r.. and r+16..) -> one 8-element MFMA operand; dword moves only
__device__ __forceinline__ bf16x8 frag_of(const u32x2 lo, const u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
// Row maximum of 16 MFMA results and the running maximum as ONE asm block of v_max3_f32 (hipcc otherwise adds a
// canonicalising v_max per fmaxf operand).  The inputs are MFMA result registers and hipcc pads NO hazards for
// inline asm (an XDL result needs 12 wait states before a VALU reads it): the block therefore opens with s_nop 15.
// Without it the asm read stale registers whenever it was scheduled right behind the MFMAs: rare wrong maxima ->
// exp2 overflow -> NaN rows, run-to-run different.
__device__ __forceinline__ float rowmax16(const f32x4 a, const f32x4 b, const f32x4 c, const f32x4 d, float m) {
    float r;
    asm("s_nop 15\n\t"
        "v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max3_f32 %0, %0, %16, %17"
        : "=&v"(r)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
          "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(m));
    return r;
}


template <int OFF>
__device__ __forceinline__ void lds_read16(f32x4& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}

#define FWD_BUF 16640          // K 8192 | V 8192 | bias 256

// Diagnostic build only (-DMMB_STAMPS, tools/stamp_attn.py): s_memtime stamps at the phase boundaries of a key tile, summed per
// wave in SGPRs and stored once at the end.  No stamp exists in the product build.
#ifdef MMB_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;
#define ATT_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define ATT_STAMP(var)
#endif

// Four waves per SIMD (round 5): 122 VGPRs.  The kernel sat at 156-164 (three waves) because all 16 transposed V reads of a tile were issued in front
// of the softmax block, 32 registers held across it beside the 32 score registers; read one 32-key slice at a time in front of its products
// (the score registers are dead by then) it fits 128 without a spill, and the fourth wave is worth more than the hidden LDS latency:
// forward -6 ... -7 % at the step's layout, bit-identical (profiles/r5_ab_attention_forward_4waves.log).  (Forcing the old body to 128 registers
// spilled 20 and cost 25 %: r5_ab_attention_occupancy.log.)
template <bool DROP>
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * FWD_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tix = a.head_fast ? blockIdx.y : blockIdx.x, head = a.head_fast ? blockIdx.x : blockIdx.y;       // see AttnArgs::head_fast
    const int seq = a.tile_seq[tix], r0 = a.tile_r0[tix];
    if (seq < 0) return;                                     // a padding entry of a device-built tile list (mmbert_split_layout)
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    const int qshift = a.tile_qshift ? a.tile_qshift[tix] : start;     // packed row of query index 0
    const int Sq = a.tile_qend ? a.tile_qend[tix] : S;                 // query indices of this tile end here
    const int Skv = a.kv_len ? min(S, a.kv_len[seq]) : S;                     // keys at and past Skv are all masked out
    const bool wave_active = r0 + wave * 32 < Sq;            // (128-row tiles: S = 550 leaves waves 2, 3 of the fifth tile without rows)
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    const bf16_t* qbase = a.qkv + (size_t)qshift * a.ld_qkv + head * 64;
    const bf16_t* kbase = base + a.H;
    const bf16_t* vbase = base + 2 * a.H;
    const float* bbase = a.key_bias + a.bias_start[seq];
    int qi[2], qc[2];
    unsigned rowbase[2];
    bf16x8 qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = r0 + wave * 32 + qb * 16 + fr;
        qc[qb] = min(qi[qb], Sq - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[qb][kk] = *(const bf16x8*)(qbase + (size_t)qc[qb] * a.ld_qkv + kk * 32 + 8 * g);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qb][kk][j] = f2bf(bf2f(qf[qb][kk][j]) * a.scale);   // 1/8: exact in bf16; the accumulator is q.k/8
        }
        rowbase[qb] = a.elem_base[seq] + (unsigned)((head * S + qc[qb]) * Spad);
    }
    const float c2 = LOG2E;                                  // scores leave the MFMA already scaled (Q carries the 1/8)
    const uint32_t thr_pk = mmb_thr_packed(a.dthr);

    auto stage = [&](int buf, int kv0) {
        char* B = smem + buf * FWD_BUF;
        stage_tile<0>(B, kbase, a.ld_qkv, kv0, Skv, wave, lane);
        stage_tile<1>(B + 8192, vbase, a.ld_qkv, kv0, Skv, wave, lane);
        // 64 bias floats (padded array: keys past the end read -1e30 -> p = 0, no per-tile range logic);
        // every wave writes the same 256 B (identical data) so that all waves keep the same vmcnt
        __builtin_amdgcn_global_load_lds(GPTR(bbase + kv0 + lane), LPTR(B + 16384), 4, 0, 0);
    };

    f32x4 o[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 4; ++d) o[qb][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mraw[2] = {-INFINITY, -INFINITY};
    // softmax denominators on the MFMA: ones(16 x keys) . P^T gives, in every accumulator row, the sum over the 32 keys of a
    // k-slice for this lane's query column -- of the bf16 probabilities the P.V product uses, before dropout.  Replaces one
    // v_add_f32 per score and the cross-lane sum at the end by 4 MFMAs per tile.
    f32x4 lacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (ones_s16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

    // transposed-read lane addressing (see lds_tr_frag): rows 4g+q (+16 per row tile), 4 columns at 4p
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(smem);
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    unsigned vaddr[4];                                       // one base per d-tile; row tiles by immediate offsets
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = 4 * g + tq;
        vaddr[d] = lds0 + 8192 + r * 128 + ((d ^ ((r >> 1) & 3)) << 5) + tp * 8;
    }

    const unsigned baddr = lds0 + 16 * g;                    // bias tile: 4 consecutive keys per lane group
    const int ntile = (Skv + 63) >> 6;                       // trailing masked-out keys contribute exact zeros: their tiles are skipped
#ifdef MMB_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, t_wait = 0, t_qk = 0, t_soft = 0, t_pv = 0, t_begin = 0, t_end = 0;
    ATT_STAMP(t_begin)
#endif
    stage(0, 0);
    // the buffer index must be a compile-time constant: with a runtime index hipcc cannot prove that the
    // fragment reads do not alias the LDS-DMA it has just issued and drains vmcnt(0) in front of them
    auto tile_body = [&](auto buf_c, int t) {
        constexpr int buf = decltype(buf_c)::value;
        const int kv0 = t << 6;
        ATT_STAMP(st0)
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): tile t landed (tile t+1 is issued below, after the barrier)
        __builtin_amdgcn_s_barrier();                        // ... for every wave; buffer buf^1 (tile t-1) is free
        ATT_STAMP(st1)
        if (t + 1 < ntile) stage(buf ^ 1, kv0 + 64);
        if (!wave_active) return;                            // a wave whose 32 rows lie past the sequence end only helps staging
        const char* Ks = smem + buf * FWD_BUF;

        // S^T = K.Q^T with the key bias (divided by the scale) as the initial accumulator
        f32x4 s[2][4], b4[4];
        lds_read16<buf * FWD_BUF + 16384>(b4[0], baddr); lds_read16<buf * FWD_BUF + 16384 + 64>(b4[1], baddr);
        lds_read16<buf * FWD_BUF + 16384 + 128>(b4[2], baddr); lds_read16<buf * FWD_BUF + 16384 + 192>(b4[3], baddr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[0][kt] = b4[kt];
            s[1][kt] = s[0][kt];
            const bf16x8 k0 = lds_row_frag(Ks, kt * 16 + fr, g), k1 = lds_row_frag(Ks, kt * 16 + fr, 4 + g);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                s[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], s[qb][kt], 0, 0, 0);
                s[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], s[qb][kt], 0, 0, 0);
            }
        }
        bf16x8 pf[2][2];
        ATT_STAMP(st2)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tmax = rowmax16(s[qb][0], s[qb][1], s[qb][2], s[qb][3], mraw[qb]);   // running maximum folded in
            tmax = group_max(tmax);
            const float mnew = tmax;                         // already >= mraw
            const bool grew = __ballot(mnew > mraw[qb]) != 0ull;                     // wave-uniform: did any row's maximum grow?
            const float alpha = grew ? __builtin_amdgcn_exp2f((mraw[qb] - mnew) * c2) : 1.0f;   // first tile: exp2(-inf) = 0
            mraw[qb] = mnew;
            const float mc = mnew * c2;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[qb][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], c2, -mc));
                }
            }
            if (grew) {                                      // steady state (maximum unchanged): no rescale pass
                lacc[qb] = lacc[qb] * alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qb][d][r] *= alpha;
            }
            pf[qb][0] = pack8(s[qb][0], s[qb][1]);
            pf[qb][1] = pack8(s[qb][2], s[qb][3]);
            lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qb][0], lacc[qb], 0, 0, 0);
            lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qb][1], lacc[qb], 0, 0, 0);
            if constexpr (DROP) {
                // dropout on the PACKED bf16 probabilities: dword (kt, pair) holds keys kv0 + 16 kt + 4 g + 2 pair (+1), the
                // element pair of one hash word; 2 VALU operations build the mask of both halves (see mmb_drop_mask2)
                const uint32_t pair0 = (rowbase[qb] + (uint32_t)(kv0 + 4 * g)) >> 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 w = __builtin_bit_cast(u32x4, pf[qb][ks]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t h = mmb_pair_bits(a.dstream, pair0 + (uint32_t)((2 * ks + (j >> 1)) * 8 + (j & 1)));
                        w[j] &= ~mmb_drop_mask2(h, thr_pk);
                    }
                    pf[qb][ks] = __builtin_bit_cast(bf16x8, w);
                }
            }
        }
        ATT_STAMP(st3)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // the transposed V reads of this 32-key slice (asm: see tr_read), then its products
            u32x2 vlo[4], vhi[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (ks == 0) { tr_read<buf * FWD_BUF>(vlo[d], vaddr[d]); tr_read<buf * FWD_BUF + 16 * 128>(vhi[d], vaddr[d]); }
                else { tr_read<buf * FWD_BUF + 32 * 128>(vlo[d], vaddr[d]); tr_read<buf * FWD_BUF + 48 * 128>(vhi[d], vaddr[d]); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 vf = frag_of(vlo[d], vhi[d]);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    o[qb][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][ks], o[qb][d], 0, 0, 0);
            }
        }
#ifdef MMB_STAMPS
        ATT_STAMP(st4)
        t_wait += st1 - st0; t_qk += st2 - st1; t_soft += st3 - st2; t_pv += st4 - st3;
#endif
    };
    for (int t = 0; t < ntile; t += 2) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = lacc[qb][0];
        if (qi[qb] < Sq) {
            const float inv = a.dscale / l;                  // dropout scale folded into the normalisation
            bf16_t* orow = a.ctx + (size_t)(qshift + qi[qb]) * a.H + head * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ov = {f2bf(o[qb][d][0] * inv), f2bf(o[qb][d][1] * inv), f2bf(o[qb][d][2] * inv), f2bf(o[qb][d][3] * inv)};
                *(bf16x4*)(orow + d * 16 + 4 * g) = ov;
            }
            if (g == 0) a.lse[(size_t)(qshift + qi[qb]) * a.heads + head] = (mraw[qb] * c2 + log2f(l)) * LN2;
        }
    }
#ifdef MMB_STAMPS
    ATT_STAMP(t_end)
    if (g_attn_stamps && lane == 0) {
        unsigned long long* o = g_attn_stamps + ((size_t)(head * (a.head_fast ? gridDim.y : gridDim.x) + tix) * 4 + wave) * 8;
        o[0] = t_wait; o[1] = t_qk; o[2] = t_soft; o[3] = t_pv; o[4] = t_end - t_begin; o[5] = wave_active ? ntile : 0; o[6] = t_begin; o[7] = t_end;
    }
#endif
}

// =============================================================================================
// backward, part 1: dQ (and delta = rowsum(dO*O)) per 128-query tile; same structure as forward:
// 4 waves x 32 query rows, double-buffered K (row image + transposed-read image), V and key-bias tiles.
//   S^T  = K.Q^T (+bias/scale as initial accumulator)      p  = exp2(S^T*c - lse*c)      (normalised)
//   dP^T = V.dO^T                                          dS^T = p * (keep*dscale*dP^T - delta)
//   dQ^T += K^T.dS^T   (K^T by transposed LDS reads, dS^T straight from the accumulator registers)
// =============================================================================================
#define DQ_BUF 24832           // K row image 8192 | K transposed-read image 8192 | V row image 8192 | bias 256

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * DQ_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tix = a.head_fast ? blockIdx.y : blockIdx.x, head = a.head_fast ? blockIdx.x : blockIdx.y;       // see AttnArgs::head_fast
    const int seq = a.tile_seq[tix], r0 = a.tile_r0[tix];
    if (seq < 0) return;                                     // a padding entry of a device-built tile list (mmbert_split_layout)
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    const int qshift = a.tile_qshift ? a.tile_qshift[tix] : start;
    const int Sq = a.tile_qend ? a.tile_qend[tix] : S;
    const int Skv = a.kv_len ? min(S, a.kv_len[seq]) : S;
    if (a.q_limit && r0 >= a.q_limit[seq]) {                 // every dO row of this tile is zero: dQ = 0 (q_limit, see AttnArgs)
        const int fr_ = lane & 15, g_ = lane >> 4;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q = r0 + wave * 32 + qb * 16 + fr_;
            if (q < Sq) {
                bf16_t* drow = a.dqkv + (size_t)(qshift + q) * a.ld_qkv + head * 64;
                const bf16x4 z = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
#pragma unroll
                for (int d = 0; d < 4; ++d) *(bf16x4*)(drow + d * 16 + 4 * g_) = z;
            }
        }
        return;
    }
    const bool wave_active = r0 + wave * 32 < Sq;            // (128-row tiles: S = 550 leaves waves 2, 3 of the fifth tile without rows)
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    const bf16_t* qbase = a.qkv + (size_t)qshift * a.ld_qkv + head * 64;
    const bf16_t* dob = a.dctx + (size_t)qshift * a.H + head * 64;
    const bf16_t* ob = a.ctx + (size_t)qshift * a.H + head * 64;
    const bf16_t* kbase = base + a.H;
    const bf16_t* vbase = base + 2 * a.H;
    const float* bbase = a.key_bias + a.bias_start[seq];
    const float c2 = LOG2E;                                  // scores leave the MFMA already scaled (Q carries the 1/8)
    int qi[2];
    uint32_t hseed[2];                                       // dropout hash seed of (row, keys 4g..) -- linear in the key offset
    bf16x8 qf[2][2], dof[2][2];
    float delta[2], nlse[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = r0 + wave * 32 + qb * 16 + fr;
        const int qc = min(qi[qb], Sq - 1);
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[qb][kk] = *(const bf16x8*)(qbase + (size_t)qc * a.ld_qkv + kk * 32 + 8 * g);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qb][kk][j] = f2bf(bf2f(qf[qb][kk][j]) * a.scale);   // 1/8: exact in bf16
            dof[qb][kk] = *(const bf16x8*)(dob + (size_t)qc * a.H + kk * 32 + 8 * g);
            const bf16x8 of = *(const bf16x8*)(ob + (size_t)qc * a.H + kk * 32 + 8 * g);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dof[qb][kk][j]) * bf2f(of[j]);
        }
        delta[qb] = group_sum(dl);
        if (qi[qb] < Sq && g == 0) a.delta[(size_t)(qshift + qi[qb]) * a.heads + head] = delta[qb];
        nlse[qb] = -a.lse[(size_t)(qshift + qc) * a.heads + head] * LOG2E;
        hseed[qb] = ((a.elem_base[seq] + (unsigned)((head * S + qc) * Spad) + 4u * (unsigned)g) >> 1) * MMB_WEYL + a.dstream;
    }

    auto stage = [&](int buf, int kv0) {
        char* B = smem + buf * DQ_BUF;
        stage_tile<0>(B, kbase, a.ld_qkv, kv0, Skv, wave, lane);
        stage_tile<1>(B + 8192, kbase, a.ld_qkv, kv0, Skv, wave, lane);
        stage_tile<0>(B + 16384, vbase, a.ld_qkv, kv0, Skv, wave, lane);
        __builtin_amdgcn_global_load_lds(GPTR(bbase + kv0 + lane), LPTR(B + 24576), 4, 0, 0);
    };

    f32x4 dq[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[qb][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(smem);
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    unsigned kaddr[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = 4 * g + tq;
        kaddr[d] = lds0 + 8192 + r * 128 + ((d ^ ((r >> 1) & 3)) << 5) + tp * 8;
    }
    const unsigned baddr = lds0 + 16 * g;
    const int ntile = (Skv + 63) >> 6;
    stage(0, 0);
    auto tile_body = [&](auto buf_c, int t) {
        constexpr int buf = decltype(buf_c)::value;
        const int kv0 = t << 6;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntile) stage(buf ^ 1, kv0 + 64);
        if (!wave_active) return;                            // a wave whose 32 rows lie past the sequence end only helps staging
        const char* Ks = smem + buf * DQ_BUF;
        const char* Vs = Ks + 16384;
        f32x4 s[2][4], dp[2][4], b4[4];
        lds_read16<buf * DQ_BUF + 24576>(b4[0], baddr); lds_read16<buf * DQ_BUF + 24576 + 64>(b4[1], baddr);
        lds_read16<buf * DQ_BUF + 24576 + 128>(b4[2], baddr); lds_read16<buf * DQ_BUF + 24576 + 192>(b4[3], baddr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[0][kt] = b4[kt];
            s[1][kt] = s[0][kt];
            dp[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16x8 k0 = lds_row_frag(Ks, kt * 16 + fr, g), k1 = lds_row_frag(Ks, kt * 16 + fr, 4 + g);
            const bf16x8 v0 = lds_row_frag(Vs, kt * 16 + fr, g), v1 = lds_row_frag(Vs, kt * 16 + fr, 4 + g);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                s[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], s[qb][kt], 0, 0, 0);
                s[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], s[qb][kt], 0, 0, 0);
                dp[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, dof[qb][0], dp[qb][kt], 0, 0, 0);
                dp[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, dof[qb][1], dp[qb][kt], 0, 0, 0);
            }
        }
        // transposed K reads for dQ^T: issue now, consume after the VALU block
        u32x2 klo[2][4], khi[2][4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            tr_read<buf * DQ_BUF>(klo[0][d], kaddr[d]); tr_read<buf * DQ_BUF + 16 * 128>(khi[0][d], kaddr[d]);
            tr_read<buf * DQ_BUF + 32 * 128>(klo[1][d], kaddr[d]); tr_read<buf * DQ_BUF + 48 * 128>(khi[1][d], kaddr[d]);
        }
        bf16x8 dsf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float lc = nlse[qb];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bool keep[4] = {true, true, true, true};
                if constexpr (DROP) {
                    const uint32_t sv = (uint32_t)((kv0 >> 1) + kt * 8) * MMB_WEYL;          // wave-uniform
                    const uint32_t h0 = mmb_pair_mix(hseed[qb] + sv), h1 = mmb_pair_mix(hseed[qb] + sv + MMB_WEYL);
                    keep[0] = mmb_keep16(h0 & 0xFFFFu, a.dthr); keep[1] = mmb_keep16(h0 >> 16, a.dthr);
                    keep[2] = mmb_keep16(h1 & 0xFFFFu, a.dthr); keep[3] = mmb_keep16(h1 >> 16, a.dthr);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], c2, lc));   // keys past the end: bias -1e30 -> 0
                    const float dpm = keep[r] ? dp[qb][kt][r] * a.dscale : 0.f;
                    s[qb][kt][r] = p * (dpm - delta[qb]);                       // dS^T
                }
            }
            dsf[qb][0] = pack8(s[qb][0], s[qb][1]);
            dsf[qb][1] = pack8(s[qb][2], s[qb][3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 kf = frag_of(klo[ks][d], khi[ks][d]);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    dq[qb][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf[qb][ks], dq[qb][d], 0, 0, 0);
            }
    };
    for (int t = 0; t < ntile; t += 2) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        if (qi[qb] < Sq) {
            bf16_t* drow = a.dqkv + (size_t)(qshift + qi[qb]) * a.ld_qkv + head * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ov = {f2bf(dq[qb][d][0] * a.scale), f2bf(dq[qb][d][1] * a.scale), f2bf(dq[qb][d][2] * a.scale), f2bf(dq[qb][d][3] * a.scale)};
                *(bf16x4*)(drow + d * 16 + 4 * g) = ov;
            }
        }
    }
}

// =============================================================================================
// backward, part 2: dK, dV per 128-key tile.  4 waves x 32 keys (two 16-key blocks per wave); the
// 64-query tiles (Q and dO, each as a row image and a transposed-read image, plus the LSE / delta
// slices) stream through two LDS buffers.  Products, all with the key on the lane:
//   S  = Q.K^T  (+ (bias[key] - lse[q]) / scale as initial accumulator)     p = exp2(S*c)
//   dP = dO.V^T            Pd = keep*dscale*p            dS = p*(keep*dscale*dP - delta[q])
//   dV^T += dO^T.Pd        dK^T += Q^T.dS       (dO^T / Q^T by transposed LDS reads)
// The dropout hash is per (query row, key PAIR): the two lanes that hold the keys of a pair split the
// four query rows of an accumulator register group between them and swap results with one DPP move.
// =============================================================================================
#define DKV_BUF 33280          // Q row 8192 | Q tr 8192 | dO row 8192 | dO tr 8192 | lse 256 | delta 256

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * DKV_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tix = a.head_fast ? blockIdx.y : blockIdx.x, head = a.head_fast ? blockIdx.x : blockIdx.y;       // see AttnArgs::head_fast
    const int seq = a.tile_seq[tix], r0 = a.tile_r0[tix];
    if (seq < 0) return;                                     // a padding entry of a device-built tile list (mmbert_split_layout)
    const int start = a.seq_start[seq], S = a.seq_len[seq];
    // keys at and past Skv are all masked out: p = exp(s - 10000 - lse) underflows to exactly 0, so dK = dV = 0 for them.  A wave
    // whose 32 keys lie there computes nothing (its zero accumulators are stored at the end); a tile that lies there entirely
    // stores its zeros and leaves before any staging.
    const int Skv = a.kv_len ? min(S, a.kv_len[seq]) : S;
    // split layout: the sequence owns only its first Skv rows here, and only those queries can have a gradient
    const int Sq0 = a.split ? Skv : S, Skw = a.split ? Skv : S;
    const int Sq = a.q_limit ? min(Sq0, a.q_limit[seq]) : Sq0;      // queries past q_limit have dO == 0: nothing to add (see AttnArgs)
    if (r0 >= Skv) {
        const int fr_ = lane & 15, g_ = lane >> 4;
#pragma unroll
        for (int kb_ = 0; kb_ < 2; ++kb_) {
            const int k = r0 + wave * 32 + kb_ * 16 + fr_;
            if (k < Skw) {
                bf16_t* drow = a.dqkv + (size_t)(a.seq_start[seq] + k) * a.ld_qkv + head * 64;
                const bf16x4 z = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
#pragma unroll
                for (int d = 0; d < 4; ++d) { *(bf16x4*)(drow + a.H + d * 16 + 4 * g_) = z; *(bf16x4*)(drow + 2 * a.H + d * 16 + 4 * g_) = z; }
            }
        }
        return;
    }
    const bool wave_active = r0 + wave * 32 < Skv;           // (also: S = 550 leaves waves 2, 3 of the fifth 128-row tile without rows)
    const int Spad = (S + 3) & ~3;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)start * a.ld_qkv + head * 64;
    const bf16_t* dob = a.dctx + (size_t)start * a.H + head * 64;
    const float* lbase = a.lse + (size_t)start * a.heads + head;
    const float* dbase = a.delta + (size_t)start * a.heads + head;
    const float c2 = a.scale * LOG2E, inv_scale = 1.0f / a.scale;
    int ki[2], kc[2];
    bf16x8 kf[2][2], vf[2][2];
    float kb[2];
#pragma unroll
    for (int kb_ = 0; kb_ < 2; ++kb_) {
        ki[kb_] = r0 + wave * 32 + kb_ * 16 + fr;
        kc[kb_] = min(ki[kb_], Skv - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[kb_][kk] = *(const bf16x8*)(base + a.H + (size_t)kc[kb_] * a.ld_qkv + kk * 32 + 8 * g);
            vf[kb_][kk] = *(const bf16x8*)(base + 2 * a.H + (size_t)kc[kb_] * a.ld_qkv + kk * 32 + 8 * g);
        }
        kb[kb_] = a.key_bias[a.bias_start[seq] + ki[kb_]] * inv_scale;    // padded array: keys past the end read -1e30 -> p = 0
    }
    const unsigned ebase = a.elem_base[seq] + (unsigned)(head * S) * Spad;
    // dropout hash seed of (row 4g + (lane & 1), this lane's key pair of key block 0); the other rows / key block are
    // wave-uniform offsets away (keys past the end hash out of range: their p is 0 through the -1e30 bias)
    const uint32_t hseed = (((ebase + (uint32_t)(4 * g + (lane & 1)) * (uint32_t)Spad) >> 1) + ((uint32_t)ki[0] >> 1)) * MMB_WEYL + a.dstream;

    auto stage = [&](int buf, int q0) {
        char* B = smem + buf * DKV_BUF;
        stage_tile<0>(B, base, a.ld_qkv, q0, Sq, wave, lane);
        stage_tile<1>(B + 8192, base, a.ld_qkv, q0, Sq, wave, lane);
        stage_tile<0>(B + 16384, dob, a.H, q0, Sq, wave, lane);
        stage_tile<1>(B + 24576, dob, a.H, q0, Sq, wave, lane);
        const size_t qoff = (size_t)min(q0 + lane, Sq - 1) * a.heads;
        __builtin_amdgcn_global_load_lds(GPTR(lbase + qoff), LPTR(B + 32768), 4, 0, 0);
        __builtin_amdgcn_global_load_lds(GPTR(dbase + qoff), LPTR(B + 33024), 4, 0, 0);
    };

    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kb_ = 0; kb_ < 2; ++kb_)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dk[kb_][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[kb_][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(smem);
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    unsigned taddr2[2][4];                                   // per buffer (ds offsets are 16-bit: the second buffer needs its own base)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = 4 * g + tq;
        taddr2[0][d] = lds0 + r * 128 + ((d ^ ((r >> 1) & 3)) << 5) + tp * 8;
        taddr2[1][d] = taddr2[0][d] + DKV_BUF;
    }
    const unsigned saddr2[2] = {lds0 + 16 * g, lds0 + 16 * g + DKV_BUF};   // lse / delta: 4 consecutive query rows per lane group
    const bool odd = lane & 1;
    const int ntile = (Sq + 63) >> 6;
    if (ntile > 0) stage(0, 0);                              // (q_limit may leave a sequence without any query: zeros are stored below)
    auto tile_body = [&](auto buf_c, int t) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr int BO = 0;                                // buffer base lives in the address registers
        const unsigned saddr = saddr2[buf];
        const unsigned (&taddr)[4] = taddr2[buf];
        const int q0 = t << 6;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntile) stage(buf ^ 1, q0 + 64);
        if (!wave_active) return;                            // a wave whose 32 keys lie past the sequence end only helps staging
        const char* Qs = smem + buf * DKV_BUF;
        const char* Ds = Qs + 16384;
        f32x4 l4[4], d4[4];
        lds_read16<BO + 32768>(l4[0], saddr); lds_read16<BO + 32768 + 64>(l4[1], saddr);
        lds_read16<BO + 32768 + 128>(l4[2], saddr); lds_read16<BO + 32768 + 192>(l4[3], saddr);
        lds_read16<BO + 33024>(d4[0], saddr); lds_read16<BO + 33024 + 64>(d4[1], saddr);
        lds_read16<BO + 33024 + 128>(d4[2], saddr); lds_read16<BO + 33024 + 192>(d4[3], saddr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool qtail = q0 + 64 > Sq;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            l4[qt] = l4[qt] * (-inv_scale);                  // -lse[q]/scale  (natural-log lse: exp2(S*c) with c = scale*log2e)
            if (qtail) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (q0 + qt * 16 + 4 * g + r >= Sq) l4[qt][r] = -INFINITY;   // query rows past the end: p = 0
            }
        }
#pragma unroll
        for (int kb_ = 0; kb_ < 2; ++kb_) {
            f32x4 s[4], dp[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt] = l4[qt] + kb[kb_];
                dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bf16x8 q0f = lds_row_frag(Qs, qt * 16 + fr, g), q1f = lds_row_frag(Qs, qt * 16 + fr, 4 + g);
                const bf16x8 o0f = lds_row_frag(Ds, qt * 16 + fr, g), o1f = lds_row_frag(Ds, qt * 16 + fr, 4 + g);
                // D[row <-> query (A operand rows)][col <-> key (B operand = this lane's K / V row)]
                s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[kb_][0], s[qt], 0, 0, 0);
                s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[kb_][1], s[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0f, vf[kb_][0], dp[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1f, vf[kb_][1], dp[qt], 0, 0, 0);
            }
            f32x4 pm[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                float ksc[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (DROP) {
                    // pair index of (row q, keys {2j,2j+1}); even lane hashes rows r = 0,2, odd lane rows 1,3, then swap.  The
                    // seed is linear in the row: per-lane part hseed[][] (set up once), wave-uniform part by scalar arithmetic
                    // (rows past the sequence end hash whatever they hash: their contributions are zero anyway)
                    const uint32_t sv = ((uint32_t)(q0 + qt * 16) * ((uint32_t)Spad >> 1) + (uint32_t)kb_ * 8u) * MMB_WEYL;
                    const uint32_t ha = mmb_pair_mix(hseed + sv);
                    const uint32_t hb = mmb_pair_mix(hseed + sv + (uint32_t)Spad * MMB_WEYL);      // rows + 2
                    const uint32_t xa = __builtin_amdgcn_mov_dpp(ha, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]: neighbour lane's value
                    const uint32_t xb = __builtin_amdgcn_mov_dpp(hb, 0xB1, 0xF, 0xF, true);
                    const uint32_t h0 = odd ? xa : ha, h1 = odd ? ha : xa, h2 = odd ? xb : hb, h3 = odd ? hb : xb;
                    const int sh = (kc[kb_] & 1) * 16;
                    ksc[0] = mmb_keep16((h0 >> sh) & 0xFFFFu, a.dthr) ? a.dscale : 0.f;
                    ksc[1] = mmb_keep16((h1 >> sh) & 0xFFFFu, a.dthr) ? a.dscale : 0.f;
                    ksc[2] = mmb_keep16((h2 >> sh) & 0xFFFFu, a.dthr) ? a.dscale : 0.f;
                    ksc[3] = mmb_keep16((h3 >> sh) & 0xFFFFu, a.dthr) ? a.dscale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[qt][r] * c2);
                    pm[qt][r] = p * ksc[r];                                  // dropped P   -> dV
                    s[qt][r] = p * (dp[qt][r] * ksc[r] - d4[qt][r]);         // dS          -> dK
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = pack8(pm[2 * ks], pm[2 * ks + 1]);
                const bf16x8 dsf = pack8(s[2 * ks], s[2 * ks + 1]);
                u32x2 dlo[4], dhi[4], qlo[4], qhi[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (ks == 0) {
                        tr_read<BO + 24576>(dlo[d], taddr[d]); tr_read<BO + 24576 + 16 * 128>(dhi[d], taddr[d]);
                        tr_read<BO + 8192>(qlo[d], taddr[d]); tr_read<BO + 8192 + 16 * 128>(qhi[d], taddr[d]);
                    } else {
                        tr_read<BO + 24576 + 32 * 128>(dlo[d], taddr[d]); tr_read<BO + 24576 + 48 * 128>(dhi[d], taddr[d]);
                        tr_read<BO + 8192 + 32 * 128>(qlo[d], taddr[d]); tr_read<BO + 8192 + 48 * 128>(qhi[d], taddr[d]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv[kb_][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(dlo[d], dhi[d]), pf, dv[kb_][d], 0, 0, 0);
                    dk[kb_][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(qlo[d], qhi[d]), dsf, dk[kb_][d], 0, 0, 0);
                }
            }
        }
    };
    for (int t = 0; t < ntile; t += 2) {
        tile_body(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntile) tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int kb_ = 0; kb_ < 2; ++kb_) {
        if (ki[kb_] < Skw) {
            bf16_t* drow = a.dqkv + (size_t)(start + ki[kb_]) * a.ld_qkv + head * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 kvv = {f2bf(dk[kb_][d][0] * a.scale), f2bf(dk[kb_][d][1] * a.scale), f2bf(dk[kb_][d][2] * a.scale), f2bf(dk[kb_][d][3] * a.scale)};
                bf16x4 vvv = {f2bf(dv[kb_][d][0]), f2bf(dv[kb_][d][1]), f2bf(dv[kb_][d][2]), f2bf(dv[kb_][d][3])};
                *(bf16x4*)(drow + a.H + d * 16 + 4 * g) = kvv;
                *(bf16x4*)(drow + 2 * a.H + d * 16 + 4 * g) = vvv;
            }
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

// Effective key count per sequence: 1 + the last key whose additive bias is above -10000 (the reference's value for a masked
// key, REF:MMBertForPretraining.py:152-153: (1 - mask) * -10000).  Behind it every key is masked out, its probability
// exp(s - 10000 - max) underflows to exactly 0 in fp32 (as it does in the reference's fp32 softmax) as long as one key of the
// row is NOT masked, so the attention kernels skip those key tiles / key blocks and the result is unchanged bit for bit.
// A sequence with no unmasked key keeps its full length (softmax over equally biased keys is NOT zero).
__global__ void attn_kv_len_kernel(const float* __restrict__ key_bias, const int* __restrict__ bias_start, const int* __restrict__ seq_len, int nseq, int* __restrict__ out) {
    const int s = blockIdx.x;
    if (s >= nseq) return;
    const int S = seq_len[s];
    const float* b = key_bias + bias_start[s];
    int last = -1;
    for (int k = threadIdx.x; k < S; k += 64)
        if (b[k] > -10000.0f) last = k;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (threadIdx.x == 0) out[s] = last < 0 ? S : last + 1;
}

// q_limit[s] = 1 + the largest query index (packed row - seq_start[s]) among `rows` that falls into sequence s; 0 for a sequence
// without any (out is zeroed by the launcher).  seq_start ascending.
__global__ void attn_q_limit_kernel(const int* __restrict__ rows, int n, const int* __restrict__ seq_start, int nseq, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    int lo = 0, hi = nseq - 1;
    while (lo < hi) {                                        // last sequence whose start is <= r
        const int mid = (lo + hi + 1) >> 1;
        if (seq_start[mid] <= r) lo = mid; else hi = mid - 1;
    }
    atomicMax(out + lo, r - seq_start[lo] + 1);
}
// the same in ONE workgroup for up to 1024 sequences: the limits are built in LDS and every entry of out is written (no memset launch
// in front; the top layer's row list is a few hundred rows)
__global__ __launch_bounds__(256) void attn_q_limit_wg_kernel(const int* __restrict__ rows, int n, const int* __restrict__ seq_start, int nseq,
                                                              int* __restrict__ out) {
    __shared__ int lim[1024], st[1024];
    for (int s = threadIdx.x; s < nseq; s += 256) { lim[s] = 0; st[s] = seq_start[s]; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = rows[i];
        int lo = 0, hi = nseq - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (st[mid] <= r) lo = mid; else hi = mid - 1;
        }
        atomicMax(&lim[lo], r - st[lo] + 1);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nseq; s += 256) out[s] = lim[s];
}

extern "C" {

static int fill_args(AttnArgs& a, const void* qkv, int H, int heads, const float* key_bias, const int* bias_start, const int* seq_start, const int* seq_len,
                     const unsigned* elem_base, const int* tile_seq, const int* tile_r0, float* lse,
                     uint32_t dstream, uint32_t dthr, float dscale, const int* kv_len) {
    if (heads <= 0 || H != heads * 64) return -1;      // head dim 64 only
    a.qkv = (const bf16_t*)qkv; a.ld_qkv = 3 * H; a.H = H; a.heads = heads; a.key_bias = key_bias; a.bias_start = bias_start;
    a.seq_start = seq_start; a.seq_len = seq_len; a.elem_base = elem_base; a.tile_seq = tile_seq; a.tile_r0 = tile_r0;
    a.lse = lse; a.scale = 0.125f; a.dstream = dstream; a.dthr = dthr; a.dscale = dscale;
    a.ctx = nullptr; a.dctx = nullptr; a.dqkv = nullptr; a.delta = nullptr; a.kv_len = kv_len;
    a.tile_qshift = nullptr; a.tile_qend = nullptr; a.split = 0; a.q_limit = nullptr;
    a.head_fast = 1;                                   // grid (heads, tiles): the tile list's order holds for the whole launch (round 2: -1.3 % of the step)
    return 0;
}

int mmbert_attn_fwd(hipStream_t stream, const void* qkv, void* ctx, float* lse, const float* key_bias, const int* bias_start, int H, int heads,
                    const int* seq_start, const int* seq_len, const unsigned* elem_base, const int* tile_seq, const int* tile_r0, int ntiles,
                    uint32_t dstream, uint32_t dthr, float dscale, const int* kv_len, const int* tile_qshift, const int* tile_qend) {
    if (ntiles <= 0) return 0;
    if ((tile_qshift == nullptr) != (tile_qend == nullptr)) return -1;
    AttnArgs a;
    if (fill_args(a, qkv, H, heads, key_bias, bias_start, seq_start, seq_len, elem_base, tile_seq, tile_r0, lse, dstream, dthr, dscale, kv_len)) return -1;
    a.ctx = (bf16_t*)ctx; a.tile_qshift = tile_qshift; a.tile_qend = tile_qend;
    if (ntiles > 65535) a.head_fast = 0;               // gridDim.y is a 16-bit field: very long tile lists go back to grid (tiles, heads)
    constexpr int extra_lds = 0;
    if (dthr) hipLaunchKernelGGL(attn_fwd_kernel<true>, a.head_fast ? dim3(heads, ntiles) : dim3(ntiles, heads), dim3(256), extra_lds, stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, a.head_fast ? dim3(heads, ntiles) : dim3(ntiles, heads), dim3(256), extra_lds, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_attn_bwd(hipStream_t stream, const void* qkv, const void* ctx, const void* dctx, void* dqkv, const float* lse, float* delta,
                    const float* key_bias, const int* bias_start, int H, int heads, const int* seq_start, const int* seq_len, const unsigned* elem_base,
                    const int* qtile_seq, const int* qtile_r0, int nqtiles, const int* tile_seq, const int* tile_r0, int ntiles,
                    uint32_t dstream, uint32_t dthr, float dscale, const int* kv_len, const int* qtile_qshift, const int* qtile_qend, int split,
                    const int* q_limit) {
    if (ntiles <= 0 || nqtiles <= 0) return 0;
    if ((qtile_qshift == nullptr) != (qtile_qend == nullptr) || (split && !kv_len)) return -1;
    AttnArgs a;
    if (fill_args(a, qkv, H, heads, key_bias, bias_start, seq_start, seq_len, elem_base, tile_seq, tile_r0, (float*)lse, dstream, dthr, dscale, kv_len)) return -1;
    a.ctx = (bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv; a.delta = delta; a.q_limit = q_limit;
    {
        AttnArgs q = a;
        q.tile_seq = qtile_seq; q.tile_r0 = qtile_r0; q.tile_qshift = qtile_qshift; q.tile_qend = qtile_qend;
        if (nqtiles > 65535) q.head_fast = 0;          // (gridDim.y limit, see mmbert_attn_fwd)
        if (dthr) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, q.head_fast ? dim3(heads, nqtiles) : dim3(nqtiles, heads), dim3(256), 0, stream, q);
        else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, q.head_fast ? dim3(heads, nqtiles) : dim3(nqtiles, heads), dim3(256), 0, stream, q);
        MMB_CHECK_LAUNCH();
    }
    a.split = split;
    if (ntiles > 65535) a.head_fast = 0;
    if (dthr) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, a.head_fast ? dim3(heads, ntiles) : dim3(ntiles, heads), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, a.head_fast ? dim3(heads, ntiles) : dim3(ntiles, heads), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

#ifdef MMB_STAMPS
int mmbert_debug_set_attn_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &buf, sizeof(buf));
}
#endif

int mmbert_attn_q_limit(hipStream_t stream, const int* rows, int n, const int* seq_start, int nseq, int* q_limit) {
    if (nseq <= 0) return 0;
    if (nseq <= 1024 && n <= 16384) {
        hipLaunchKernelGGL(attn_q_limit_wg_kernel, dim3(1), dim3(256), 0, stream, rows, n < 0 ? 0 : n, seq_start, nseq, q_limit);
        MMB_CHECK_LAUNCH();
        return 0;
    }
    if (hipMemsetAsync(q_limit, 0, (size_t)nseq * sizeof(int), stream) != hipSuccess) return (int)hipGetLastError();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(attn_q_limit_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rows, n, seq_start, nseq, q_limit);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_attn_kv_len(hipStream_t stream, const float* key_bias, const int* bias_start, const int* seq_len, int nseq, int* kv_len) {
    if (nseq <= 0) return 0;
    hipLaunchKernelGGL(attn_kv_len_kernel, dim3(nseq), dim3(64), 0, stream, key_bias, bias_start, seq_len, nseq, kv_len);
    MMB_CHECK_LAUNCH();
    return 0;
}

// rows per work item of the tile lists: which = 0 forward (128), 1 backward (64)
int mmbert_attn_tile_rows(int which) { (void)which; return 128; }

int mmbert_attn_dropout_mask(hipStream_t stream, uint8_t* out, int S, unsigned elem_base, int head, uint32_t rng_stream, uint32_t thr16) {
    hipLaunchKernelGGL(attn_mask_kernel, dim3(256), dim3(256), 0, stream, out, S, elem_base, head, rng_stream, thr16);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
