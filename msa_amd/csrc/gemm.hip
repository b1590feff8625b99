// bf16 MFMA GEMMs for the MMBert encoder stack on gfx950 (MI355X).
//
//   gemm_nt : C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)       forward projections and input gradients
//             (A and B both K-contiguous: activations x PyTorch Linear weights [out,in], or
//              gradients x the pre-transposed bf16 weight copy).
//   gemm_tn : W[N,K] (+)= A[M,N]^T . B[M,K]  in fp32          weight gradients (reduction over tokens)
//
// Kernels (round 5: one per job; the 4-slot-ring kernels of rounds 1-3 -- gemm_nt256 / gemm_ntp / gemm_tn -- were retired once the 8-phase
// forms covered every shape they ran, bit-identically where the K order per element is the same: git history):
//   gemm_nt_kernel  : 128 x 128 tile, 4 waves, two LDS buffers -- small shapes, K % 128 != 0, and the split-K form (mmbert_gemm_nt_splitk);
//   gemm_nt8_kernel : 256-column tiles of 128 / 192 / 224 / 256 rows, 8 waves, 64-deep K tiles in 8 phases, LDS-DMA half-tiles 3 ahead;
//                     one tile per workgroup or a stream of tiles per workgroup (static walk or the caller's device tile queue);
//   gemm_tn8_kernel : the same K loop over tokens with transposed fragment reads, up to 48 problems per call.
// Operands are staged global -> LDS with 16-byte LDS-DMA (lane-linear writes: the XOR swizzle is applied on the per-lane SOURCE address
// and again on the LDS read); tile ids are remapped so that each XCD (private L2) works on a contiguous band of the tile walk.
#include "common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define EPI_BIAS 1
#define EPI_GELU 2       // out = gelu(v); optional aux = v (pre-activation, bf16)
#define EPI_RESID 4      // out = dropout(v) + R
#define EPI_GELU_BWD 8   // out = v * gelu'(U)
#define EPI_OUT_F32 16

struct GemmNT {
    const bf16_t* A; const bf16_t* B; void* C;
    const float* bias; const bf16_t* R; bf16_t* aux; const bf16_t* U; const float* alpha_dev;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, ldu;
    float alpha;
    uint32_t drop_stream, drop_thr16; float drop_scale;
    int kt_per_split; long long split_stride;     // gemm_nt_kernel only: split-K over blockIdx.z into fp32 slabs (0 = no split)
    int* tile_counter; int* tile_counter_next;    // gemm_nt8_kernel, multi-tile form: dynamic tile queue = {8 fetch counters (one per XCD), exit counter} (null = static b, b+G, ...)
    int queue_xcd;                                // 1: a workgroup draws from its XCD's counter (tile order stays v = x mod 8: the XCD's L2 keeps its panels); 0: one counter
    int group_m;                                  // gemm_nt8_kernel, multi-tile form: tile walk in groups of group_m row tiles (<= 1: row-major), see ntp_tile_mn
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "each XCD gets a contiguous chunk" remap (blocks b and b+8 share an XCD)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// Tile walk of the multi-tile form.  Linear tile index t (the XCD-contiguous order of xcd_remap) -> (row tile, column tile).
// group_m <= 1: row-major (all column tiles of a row tile, then the next row tile): right when the B operand (N x K weights) fits an
// XCD's 4 MiB L2 -- the 32 workgroups of an XCD then share a few A row panels and the resident B.  For the vocabulary projection
// (B = 47 MB) that order streams the WHOLE B from the Infinity Cache once per row tile (round 1, PMC: 3.6 GB fetched per launch for
// 75 MB of operands); with group_m = 4 the walk sweeps the column tiles with 4 row tiles at a time (index within the group fastest),
// so 32 consecutive tiles = 4 row tiles x 8 column tiles: the 4 A panels (1.6 MB) stay in L2 for the whole sweep and every B panel
// is fetched once per GROUP -- a quarter of the B traffic.
__device__ __forceinline__ void ntp_tile_mn(int t, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    if (gm <= 1) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
    const int per = gm * tiles_n, g = t / per, r = t - g * per;
    const int gs = min(gm, tiles_m - g * gm);                  // the last group may be short
    tn = r / gs;
    tm = g * gm + (r - tn * gs);
}

// shared epilogue: v[0..3] = alpha-scaled accumulators of C[m][n..n+3]
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmNT& p, int m, int n, float (&v)[4]) {
    if constexpr (EPI & EPI_BIAS) {
        const float4 b = *(const float4*)(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if constexpr (EPI & EPI_GELU) {
        if (p.aux) {
            bf16x4 u = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(p.aux + (size_t)m * p.ldaux + n) = u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if constexpr (EPI & EPI_GELU_BWD) {
        const bf16x4 u = *(const bf16x4*)(p.U + (size_t)m * p.ldu + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad(bf2f(u[r]));
    }
    if constexpr (EPI & EPI_RESID) {
        if (p.drop_thr16) {
            bool k[4];
            mmb_keep4(p.drop_stream, (uint64_t)m * p.N + n, p.drop_thr16, k);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = k[r] ? v[r] * p.drop_scale : 0.f;
        }
        const bf16x4 rr = *(const bf16x4*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(rr[r]);
    }
    if constexpr (EPI & EPI_OUT_F32) {
        *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
    }
}

// 8-wide epilogue on row-contiguous data: v[0..7] = alpha-scaled C[m][n..n+7] (n % 8 == 0), 16-byte accesses
template <int EPI>
__device__ __forceinline__ void epi_store8(const GemmNT& p, int m, int n, float (&v)[8]) {
    if constexpr (EPI & EPI_BIAS) {
        const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if constexpr (EPI & EPI_GELU) {
        if (p.aux) {
            bf16x8 u;
#pragma unroll
            for (int r = 0; r < 8; ++r) u[r] = f2bf(v[r]);
            *(bf16x8*)(p.aux + (size_t)m * p.ldaux + n) = u;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
    }
    if constexpr (EPI & EPI_GELU_BWD) {
        const bf16x8 u = *(const bf16x8*)(p.U + (size_t)m * p.ldu + n);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= gelu_erf_grad(bf2f(u[r]));
    }
    if constexpr (EPI & EPI_RESID) {
        if (p.drop_thr16) {
            bool k0[4], k1[4];
            const uint64_t idx = (uint64_t)m * p.N + n;
            mmb_keep4(p.drop_stream, idx, p.drop_thr16, k0);
            mmb_keep4(p.drop_stream, idx + 4, p.drop_thr16, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = k0[r] ? v[r] * p.drop_scale : 0.f; v[4 + r] = k1[r] ? v[4 + r] * p.drop_scale : 0.f; }
        }
        const bf16x8 rr = *(const bf16x8*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bf2f(rr[r]);
    }
    if constexpr (EPI & EPI_OUT_F32) {
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = f2bf(v[r]);
        *(bf16x8*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + 127) >> 7, tiles_m = (p.M + 127) >> 7;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) << 7, n0 = (tile % tiles_n) << 7;
    // split-K (long K, few output tiles): workgroup z takes K tiles [kt0, kt1) and writes fp32 partial sums to slab z
    const int kt0 = p.kt_per_split ? (int)blockIdx.z * p.kt_per_split : 0;
    const int kt1 = p.kt_per_split ? min(p.K >> 6, kt0 + p.kt_per_split) : (p.K >> 6);

    // staging: wave w issues chunks 4w..4w+3 of A and of B; a chunk = 8 rows x 128 B = 1 KiB
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;     // source chunk pre-swizzled
    const bf16_t* a_src[4]; const bf16_t* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        const int ra = min(m0 + r, p.M - 1), rb = min(n0 + r, p.N - 1);
        a_src[i] = p.A + (size_t)ra * p.lda + schunk * 8;
        b_src[i] = p.B + (size_t)rb * p.ldb + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GPTR(a_src[i] + kt * 64), LPTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GPTR(b_src[i] + kt * 64), LPTR(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const char* As = smem + buf * 32768;
        const char* Bs = As + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
                af[i] = *(const bf16x8*)(As + ra * 128 + (((kk * 4 + fq) ^ (ra & 7)) << 4));
                bfr[i] = *(const bf16x8*)(Bs + rb * 128 + (((kk * 4 + fq) ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)     // operands swapped: D[row<->n][col<->m] => 4 consecutive n per lane
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1 - 1; ++kt) {
        stage(cur ^ 1, kt + 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue ----
    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha;
            if constexpr (EPI == EPI_OUT_F32) {
                if (p.kt_per_split) {
                    *(float4*)((float*)p.C + (size_t)blockIdx.z * p.split_stride + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
            }
            epi_store<EPI>(p, m, n, v);
        }
    }
}

template <int EPI>
static int launch_nt(hipStream_t s, const GemmNT& p) {
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)gemm_nt_kernel<EPI>, 65536, attr_done)) return e;
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(tiles), dim3(256), 65536, s, p);
    MMB_CHECK_LAUNCH();
    return 0;
}

// Diagnostic builds only (-DMMB_STAMPS: tools/stamp_tn8.py, tools/stamp_attn.py): s_memtime stamps at phase boundaries, kept in SGPRs and
// stored once at the end to a buffer nothing else reads.  No stamp exists in the product build (msa_amd/build.py refuses -DMMB_*).
#ifdef MMB_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define MMB_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define MMB_STAMP(var)
#endif

// Test / A-B knob (mmbert_gemm_nt_force): process-global, relaxed atomic, default 0 = "by shape".  It selects between kernels / tile
// heights that compute the same product; nothing else in the library keeps state between calls, and NO environment variable is read.
static std::atomic<int> g_nt_force{0};   // 0 auto | 1 the 128 x 128 kernel | 8 the 8-phase kernel wherever eligible | 128 / 192 / 224 / 256: that tile height of it
static int device_cus() { return mmb_device_cus(); }    // (common.h: cached per device)

__host__ __device__ constexpr int mmb_waitcnt(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | 0x70 | (lgkm << 8); }

template <int OFF>
__device__ __forceinline__ void lds_read16f(f32x4& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}


// -------------------------------------------------------------------------------------------------
// NT, every large shape: the 256 x 256 "8-phase" structure of the CDNA4 guide (cdna_hip_programming.md S5).  Round 4 rebuilt the
// guide's template as an in-tree yardstick (tools/yardstick/, profiles/r4_nt_yardstick.log) -- on cold random operands it beat the
// 4-slot-ring kernel of rounds 1-3 by 11-23 % on every shape whose tiles fit the chip in one round and by 21-39 % on 4096^3 / 8192^3 --
// and then ported what differed:
//   * BK = 64: an LDS-DMA wave instruction moves 8 rows x 128 B -- whole cache lines (32-deep stages' 16 x 64-B pieces cost the
//     texture path 25 % more per instruction) -- and a K tile of 64 has HALF the barriers per FLOP;
//   * half-tiles are QUADRANT operands (A-half h = rows {wr*128 + h*64 ..}, B-half h = columns {wc*64 + h*32 ..}): every wave reads
//     b0, a0 in phase 1, b1 in phase 2, a1 in phase 3, nothing in phase 4, so an LDS half-tile is free again one to two phases after
//     its phase and the LDS-DMA stream runs 3 half-tiles ahead behind ONE counted vmcnt(6) per K tile;
//   * two wave groups (wr = 0 / 1: one wave of each per SIMD) one barrier apart: one group's 16 MFMAs of a phase run under the other
//     group's fragment reads and LDS-DMA issue; fragments are single-buffered (64 VGPRs), all 256 rows fit (210 VGPRs, no spills).
// LDS image of a half-tile: [128 rows][64 k] bf16; 16-byte chunk c of row r sits at chunk c ^ key(r), key = (r >> 1) & 7 for A and
// ((r >> 1) & 1) | (((r >> 3) & 3) << 1) for B (B rows are read in the permuted order that gives a lane 8 consecutive output columns):
// every ds_read_b128 lane group hits 16 distinct 16-byte slots; swizzle on the per-lane SOURCE address.
// Epilogue: straight from the accumulators.  The MFMA takes the B fragment as its first operand, so a lane holds 4 consecutive COLUMNS
// of one output row per 16 x 16 block, and with the permuted B rows a wave's four column blocks give each lane 2 x 8 consecutive columns
// per row block: 16-byte stores, 64 B contiguous per row and instruction, no LDS transposition.
// -------------------------------------------------------------------------------------------------
// MULTI = false: one tile per workgroup (launches of no more tiles than CUs -- the form the train step uses): no next-tile bookkeeping,
// 40 registers fewer.  MULTI = true: the workgroup walks tiles b, b + G, ... and the half-tile stream crosses the tile seams.
// MQ = 16-row blocks per quadrant along M: 4 (256-row tile), 3 (192 rows), 7 = 4 in A half 0 and 3 in A half 1 (224 rows) or 2 (128-row tile: A half-tiles of 64 rows, ONE LDS-DMA piece per wave;
// for launches whose 256-row tiles would leave more than half the chip idle -- the reference's default model, M = 6400 x N = 1024).
template <int EPI, bool MULTI, int MQ>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(const GemmNT p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];   // buffer d at d * 65536: A0h | A1h | B0h | B1h, 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // MQ encodes the 16-row blocks a wave owns in A half 0 / A half 1: 4 -> (4, 4) 256 rows, 3 -> (3, 3) 192, 2 -> (2, 2) 128, 7 -> (4, 3) 224
    constexpr int MQ0 = (MQ == 7) ? 4 : MQ, MQ1 = (MQ == 7) ? 3 : MQ, NB = MQ0 + MQ1;
    constexpr int BMT = 32 * NB;                                 // tile rows
    // LDS-DMA pieces per wave and A half-tile (B half-tiles: always 2): MQ / 2 -- for the 192-row tile (MQ = 3: an A half-tile is 96 rows =
    // 12 pieces over 8 waves) TWO for waves 0-3 and ONE for waves 4-7.  The counted vmcnt waits are per wave, so the two classes run two
    // compile-time copies of everything below the set-up (`body`): a run-time branch around one LDS-DMA inside the loop makes hipcc drain.
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + BMT - 1) / BMT;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int nt = p.K >> 6;                                     // K tiles (K % 128 == 0: an even count, >= 4)
    // alpha * alpha_dev[0], ONCE per workgroup and before anything is in flight: read in the epilogue (where this kernel first had it) the
    // optional device scalar is a vector load behind a branch, and hipcc waits vmcnt(0) at the join whether or not it was issued -- every
    // epilogue began by waiting for the next tile's in-flight LDS-DMAs (or the single-tile form's dead re-reads): one exposed memory latency per tile
    const float alpha_all = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f))));   // (an SGPR)

    // ---- staging: wave w issues pieces j = w and w + 8 of a half-tile (piece = local rows 8j .. 8j + 7, 1 KiB) ----
    // The half-tiles of ALL tiles of this workgroup form one stream (persistent form, launches of more tiles than CUs): the last two
    // K tiles of a tile issue K tiles 0 and 1 of the workgroup's next tile, which land under its epilogue -- the phase / wait / buffer
    // pattern does not change at the seam (nt is even, so K tile 0 always sits in buffer 0).
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((uint32_t)p.M * (uint32_t)p.lda * 2u), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((uint32_t)p.N * (uint32_t)p.ldb * 2u), 0x00020000);
    struct Src { uint32_t a[2][2], b[2][2]; };                   // [half][piece]: per-lane byte offsets into A / B (the K offset is scalar)
    auto set_src = [&](int v, Src& o) {
        int l = lane;
        asm volatile("" : "+v"(l));                               // recomputed per call, nothing kept alive across the K loop
        int tmi, tni;
        ntp_tile_mn(xcd_remap(v, ntiles), tiles_m, tiles_n, p.group_m, tmi, tni);
        const int tm0 = tmi * BMT, tn0 = tni << 8;
        const int pos = l & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave + 8 * i, r = 8 * j + (l >> 3);                 // local row of the half-tile
            const uint32_t chunk_a = (uint32_t)(pos ^ ((r >> 1) & 7));
            const uint32_t chunk_b = (uint32_t)(pos ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1)));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mqh = h ? MQ1 : MQ0;                  // (a piece past the half-tile's 32 * mqh rows is never issued)
                const int ga = tm0 + (r / (16 * mqh)) * (16 * NB) + (h ? 16 * MQ0 : 0) + (r % (16 * mqh));
                const int gb = tn0 + (r >> 5) * 64 + h * 32 + (r & 31);
                o.a[h][i] = ((uint32_t)min(ga, p.M - 1) * (uint32_t)p.lda + chunk_a * 8u) * 2u;
                o.b[h][i] = ((uint32_t)min(gb, p.N - 1) * (uint32_t)p.ldb + chunk_b * 8u) * 2u;
            }
        }
    };
    Src cur;
    set_src(blockIdx.x, cur);
    Src nxt_store;
    Src& nxt = MULTI ? nxt_store : cur;                          // single-tile form: the "next tile" is this one again (dead re-reads)
    if constexpr (MULTI) nxt_store = cur;
    // half-tile ids in the order of first use: 0 = B0h, 1 = A0h, 2 = B1h, 3 = A1h
    auto stage_c = [&](auto apw_c, int buf, int which, const Src& o, int kt) {
        constexpr int APW = decltype(apw_c)::value;
        const uint32_t kb = (uint32_t)kt * 128u;
        const int h = which >> 1;
        char* base = smem + buf * 65536 + ((which & 1) ? 0 : 32768) + h * 16384 + wave * 1024;
        if (which & 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base), 16, o.a[h][0], kb, 0, 0);
            // second piece of this A half-tile: every wave (128 rows), no wave (64 rows), or this wave's class (96 rows)
            const int mqh = h ? MQ1 : MQ0;
            if (mqh == 4 || (mqh == 3 && APW == 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base + 8192), 16, o.a[h][1], kb, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base), 16, o.b[h][0], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base + 8192), 16, o.b[h][1], kb, 0, 0);
        }
    };

    // ---- fragment reads ----
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
    const int fr = lane & 15, fq = lane >> 4;
    const int swa = ((fq ^ (fr >> 1)) & 7) << 4;                                  // k step 0: chunk fq; k step 1: the same ^ 64 bytes
    const int swb = ((fq ^ (((fr >> 1) & 1) | ((fr >> 2) << 1))) & 7) << 4;       // key of B row 8 (fr >> 2) + 4 j + (fr & 3)
    const lds_cptr a_rd = (lds_cptr)LPTR(smem) + (wr * (16 * MQ0) + fr) * 128;    // (half 1 of the 224-row tile: wave row 1 starts 16 * MQ1 rows in, see read_a)
    const lds_cptr b_rd = (lds_cptr)LPTR(smem) + 32768 + (wc * 32 + 8 * (fr >> 2) + (fr & 3)) * 128;
    lds_cptr a_rd1 = a_rd + 65536, b_rd1 = b_rd + 65536;        // second buffer: ds offsets are 16-bit
    asm volatile("" : "+v"(a_rd1), "+v"(b_rd1));

    bf16x8 af[2][MQ0], b0f[2][2], b1f[2][2];
    f32x4 acc[NB][4];

    auto read_a = [&](int buf, int h) {
        lds_cptr ab = buf ? a_rd1 : a_rd;
        if (MQ0 != MQ1 && h) ab += wr * (16 * (MQ1 - MQ0) * 128);     // (wave-uniform: half 1's wave rows are 16 * MQ1 apart, not 16 * MQ0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MQ0; ++i)
                if (i < (h ? MQ1 : MQ0)) af[ks][i] = *(lds_frag)(ab + h * 16384 + i * 2048 + (swa ^ (ks * 64)));
    };
    auto read_b = [&](int buf, int h, bf16x8 (&bf)[2][2]) {
        const lds_cptr bb = buf ? b_rd1 : b_rd;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[ks][j] = *(lds_frag)(bb + h * 16384 + j * 512 + (swb ^ (ks * 64)));
    };
    // (first: K tile 0 of an output tile -- its four phases touch the four accumulator quadrants once each -- starts from C = 0: no clearing pass)
    auto mma = [&](int qa, int qb, const bf16x8 (&bf)[2][2], bool first) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MQ0; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {   // operands swapped (B first): a lane holds 4 consecutive COLUMNS of one output row
                    if (i >= (qa ? MQ1 : MQ0)) continue;
                    f32x4& c = acc[qa * MQ0 + i][qb * 2 + j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], (first && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : c, 0, 0, 0);
                }
    };

    auto body = [&](auto apw_c) {
    // this wave's LDS-DMAs of the three half-tiles that stay in flight (B0h, A0h, B1h): 4 + its pieces of A half 0
    constexpr int INFL = 4 + (MQ0 == 4 ? 2 : (MQ0 == 2 ? 1 : decltype(apw_c)::value));
    auto stage = [&](int buf, int which, const Src& o, int kt) { stage_c(apw_c, buf, which, o, kt); };
    // ---- dynamic tile queue (multi-tile form, data-parallel runs: RCCL's channel kernels hold CUs, a static share would strand tiles) --
    // Protocol: per XCD the k-th draw of XCD x is tile G + 8 k + x (the static walk's residue class: the XCD's chunk of
    // the grouped walk holds); thread 0 draws with an inline-asm returning atomic that is only read behind counted waits (hipcc's own
    // atomicAdd drains vmcnt(0)) and parks the result in an LDS word behind the ring; every wave picks it up a K tile later.
    const int qx = (MULTI && p.queue_xcd) ? (int)(blockIdx.x & 7) : 0, qs = (MULTI && p.queue_xcd) ? 8 : 1;
    auto queue_fetch = [](int* counter) {
        int r;
        // (s_nop 4: hipcc may hand the counter's address over in SGPRs it has just restored with v_readlane -- a VALU write of an SGPR needs 5 wait
        // states before a vector-memory instruction reads it, and the hazard recognizer does not look inside inline asm: without the nops
        // one build of the GELU' instantiation drew from a stale address and faulted)
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(r) : "v"(0), "v"(1), "s"(counter) : "memory");
        return r;
    };
    auto vq_write = [&](int value) {
        const uint32_t vq_addr = (uint32_t)(size_t)LPTR(smem) + 131072u;
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(vq_addr), "v"(value) : "memory");
    };
    int first_fetch = 0;
    if constexpr (MULTI) { if (p.tile_counter && tid == 0) first_fetch = queue_fetch(p.tile_counter + qx); }     // older than every prologue load
    // ---- prologue: K tile 0 (4 half-tiles, even buffer) and the first 3 half-tiles of K tile 1 (odd buffer) of the first tile ----
    stage(0, 0, cur, 0); stage(0, 1, cur, 0); stage(0, 2, cur, 0); stage(0, 3, cur, 0);
    stage(1, 0, cur, 1); stage(1, 1, cur, 1); stage(1, 2, cur, 1);
    __builtin_amdgcn_s_waitcnt(mmb_waitcnt(INFL, 15));            // K tile 0 landed (this wave's pieces)
    if constexpr (MULTI) {
        if (p.tile_counter && tid == 0) {                           // (the atomic is older than the loads the wait above has retired)
            asm volatile("" : "+v"(first_fetch) :: "memory");
            vq_write(G + qs * first_fetch + qx);
        }
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // the stagger: group 1 runs one barrier behind group 0

    // one phase: { LDS reads of this phase's quadrant operands ; one half-tile of LDS-DMA ; [counted waits] ; barrier ; MFMAs ; barrier }
#define NT8_PHASE(READS, LGK_BEFORE_BARRIER, STAGE, VMWAIT, QA, QB, BF, FIRST)                                       \
    {                                                                                                               \
        READS;                                                                                                      \
        STAGE;                                                                                                      \
        if (LGK_BEFORE_BARRIER >= 0) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(63, LGK_BEFORE_BARRIER < 0 ? 0 : LGK_BEFORE_BARRIER)); \
        if (VMWAIT >= 0) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(VMWAIT < 0 ? 0 : VMWAIT, 15));                       \
        __builtin_amdgcn_s_barrier();                                                                               \
        __builtin_amdgcn_s_waitcnt(mmb_waitcnt(63, 0));                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                              \
        mma(QA, QB, BF, FIRST);                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    // K tile in buffer D; S1 / K1 = source and K tile of the half-tile that completes the NEXT K tile (A1h), S2 / K2 = those of the three
    // half-tiles of the K tile after it.  RAW: everything of the next K tile is issued by phase 1 and retired by the vmcnt(6) of phase 4
    // (3 half-tiles of the K tile after it stay in flight), one phase before its first read.  WAR: b0 (read FIRST in phase 1 and retired
    // by lgkmcnt(8) before that phase's first barrier) is restaged in phase 2; a0 (phase 1) in phase 3; b1 (phase 2) in phase 4; a1
    // (phase 3) in phase 1 of the next K tile -- two phases after their reads, which covers the group that runs a barrier behind.
#define NT8_KTILE(D, S1, K1, S2, K2, FIRST)                                                                                   \
    NT8_PHASE((read_b(D, 0, b0f), __builtin_amdgcn_sched_barrier(0), read_a(D, 0)), 2 * MQ0, stage(D ^ 1, 3, S1, K1), -1, 0, 0, b0f, FIRST) \
    NT8_PHASE(read_b(D, 1, b1f), -1, stage(D, 0, S2, K2), -1, 0, 1, b1f, FIRST)                                               \
    NT8_PHASE(read_a(D, 1), -1, stage(D, 1, S2, K2), -1, 1, 1, b1f, FIRST)                                                    \
    NT8_PHASE((void)0, -1, stage(D, 2, S2, K2), INFL, 1, 0, b0f, FIRST)

    for (int v = blockIdx.x, vn = ntiles; v < ntiles; v = vn) {
        int tmi, tni;
        ntp_tile_mn(xcd_remap(v, ntiles), tiles_m, tiles_n, p.group_m, tmi, tni);
        const int m0 = tmi * BMT, n0 = tni << 8;
        // Seam: K tile 0 of this tile was issued by the previous tile's last two K tiles (or by the prologue) and is followed, in
        // vector-memory issue order, by the 6 LDS-DMAs of K tile 1's first three half-tiles and by the previous epilogue's loads and
        // stores (any number of them) -- so "at most 6 outstanding" proves it landed on every path.  As a BUILTIN, so that hipcc's own
        // scoreboard sees it: without it hipcc drains vmcnt(0) in front of this tile's first fragment reads (they alias the pending
        // LDS-DMA destinations), i.e. waits for the previous epilogue's last store.
        __builtin_amdgcn_s_waitcnt(mmb_waitcnt(INFL, 15));
        NT8_KTILE(0, cur, 1, cur, 2, true)
        // Bias row of this wave's 64 columns: ONE 4-byte-per-lane LDS-DMA into a wave-private 256 B behind the ring,
        // issued here -- older than K tile 1's LDS-DMAs, so K tile 1's counted wait retires it -- and read with DS instructions in the
        // epilogue.  (As register loads issued IN the epilogue, the way this kernel first did it, the four bias loads return only behind
        // the 14 LDS-DMAs of the next tile's first two K tiles -- or, single-tile form, behind the dead re-reads --: vector memory retires
        // in issue order, and every bias epilogue began with one exposed memory latency, `s_waitcnt vmcnt(4) ... vmcnt(0)`.)
        if constexpr (EPI & EPI_BIAS) {
            int l = lane;
            asm volatile("" : "+v"(l));
            auto kpb = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kpb));
            const __attribute__((address_space(4))) GemmNT& qb = *(const __attribute__((address_space(4))) GemmNT*)kpb;
            const auto rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)qb.bias, 0, qb.N * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LPTR(smem + 131136 + wave * 256), 4, l * 4, (n0 + wc * 64) * 4, 0, 0);
        }
        // the workgroup's next tile: b + G (static) or the queue's word, written a K tile or more ago and 8 barriers behind us; its source
        // offsets are first used by the last two K tiles.  Past the last tile: dead re-reads of this tile's first K tiles.
        int fetched = 0;
        if constexpr (MULTI) {
            if (p.tile_counter) {
                int qv;
                const uint32_t vq_addr = (uint32_t)(size_t)LPTR(smem) + 131072u;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(qv) : "v"(vq_addr) : "memory");
                vn = __builtin_amdgcn_readfirstlane(qv);
                if (tid == 0 && vn < ntiles) fetched = queue_fetch(p.tile_counter + qx);    // the tile after the next one; complete behind K tile 1's vmcnt
            } else {
                vn = v + G;
            }
            if (vn < ntiles) set_src(vn, nxt_store);
        }
        NT8_KTILE(1, cur, 2, cur, 3, false)
        if constexpr (MULTI) {
            if (p.tile_counter && tid == 0 && vn < ntiles) {        // (every wave has read the word: that was 8 barriers ago)
                asm volatile("" : "+v"(fetched) :: "memory");
                vq_write(G + qs * fetched + qx);
            }
        }
        for (int t = 2; t < nt - 2; t += 2) {
            NT8_KTILE(0, cur, t + 1, cur, t + 2, false)
            NT8_KTILE(1, cur, t + 2, cur, t + 3, false)
        }
        // the last two K tiles bring in K tiles 0 and 1 of the workgroup's next tile
        NT8_KTILE(0, cur, nt - 1, nxt, 0, false)
        NT8_KTILE(1, nxt, 0, nxt, 1, false)
        if constexpr (MULTI) cur = nxt_store;

        // ---- epilogue, straight from the accumulators: acc[i][2 h + (r >> 2)][r & 3] = C[m0 + wr*128 + 16 i + fr][n0 + wc*64 + 32 h + 8 fq + r];
        // the next tile's first half-tiles are in flight / landed meanwhile.  Every per-lane address comes from an opaque copy of the lane
        // id and every parameter from an opaque copy of the kernel-argument pointer, so that nothing of it is kept alive across the K loop
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const __attribute__((address_space(4))) GemmNT& q = *(const __attribute__((address_space(4))) GemmNT*)kp;
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int efr = elane & 15, efq = elane >> 4;
        const float alpha = alpha_all;
        const bool interior = (m0 + BMT <= q.M) && (n0 + 256 <= q.N);
        const int mrow = m0 + wr * (16 * NB) + efr;                  // + 16 i
        const int ncol = n0 + wc * 64 + efq * 8;                     // 8 columns here (h = 0) and 8 at + 32 (h = 1)
        constexpr int PRE = 3;
        bf16x8 pre[NB][2];
        // Epilogue operands as raw buffers: the per-lane byte offset of (mrow, ncol) once per operand and tile, row block i by ONE 32-bit add
        // (16 i pitch: a scalar), the two column halves by the instruction's immediate -- where 64-bit pointers cost 3.5 VALU instructions
        // per access (v_mad_i64_i32 + v_lshl_add_u64 + moves: 98 of FFN-up's 985 epilogue instructions).  Rows past M are past the
        // descriptor's range: their loads return 0 and nothing of them is stored (`ok` below), so the row clamp is gone too.
        constexpr uint32_t CSZ = (EPI & EPI_OUT_F32) ? 4u : 2u;
        // (descriptor range = up to the LAST VALID element, (M - 1) ld + N: with a strided view (ld > N) nothing behind row M - 1's N columns
        // is readable -- a view that ends at its parent allocation's end must not be read past it)
        const uint32_t um = (uint32_t)q.M, um1 = um ? um - 1u : 0u, un = um ? (uint32_t)q.N : 0u;
        const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)q.C, 0, (int)((um1 * (uint32_t)q.ldc + un) * CSZ), 0x00020000);
        const uint32_t voffC = ((uint32_t)mrow * (uint32_t)q.ldc + (uint32_t)ncol) * CSZ, rowC = 16u * (uint32_t)q.ldc * CSZ;
        const bf16_t* presrc = (EPI & EPI_RESID) ? q.R : q.U;
        const uint32_t preld = (uint32_t)((EPI & EPI_RESID) ? q.ldr : q.ldu);
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)presrc, 0, (int)((EPI & (EPI_RESID | EPI_GELU_BWD)) ? (um1 * preld + un) * 2u : 0u), 0x00020000);
        const uint32_t voffP = ((uint32_t)mrow * preld + (uint32_t)ncol) * 2u, rowP = 32u * preld;
        auto load_pre = [&](int i) {
            if constexpr (EPI & (EPI_RESID | EPI_GELU_BWD)) {
                const uint32_t vo = voffP + (uint32_t)i * rowP;
                pre[i][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsP, vo, 0, 0));
                pre[i][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsP, vo + 64u, 0, 0));
            }
        };
#pragma unroll
        for (int i = 0; i < PRE; ++i) load_pre(i);
        float bias[16];
        if constexpr (EPI & EPI_BIAS) {      // columns efq*8 .. +7 and + 32 .. of the wave's 64: two 32-byte pieces of its LDS bias row (inline asm: invisible to hipcc's
            const uint32_t baddr = (uint32_t)(size_t)LPTR(smem) + 131136u + wave * 256u + efq * 32u;     // scoreboard, which would drain vmcnt(0) for the LDS-DMAs in flight)
            f32x4 b4[4];
            lds_read16f<0>(b4[0], baddr); lds_read16f<16>(b4[1], baddr); lds_read16f<128>(b4[2], baddr); lds_read16f<144>(b4[3], baddr);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) :: "memory");
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[4 * c + r] = b4[c][r];
        }
        // phase 1 (needs no residual data): scale, bias and the dropout decision, in place in the accumulators.  Dropout seeds are linear
        // in the element index (common.h): pair(m, n) = m * N/2 + n/2 (mod 2^32) -- one multiply per lane, wave-uniform increments after it
        const uint32_t dthr = (EPI & EPI_RESID) ? q.drop_thr16 : 0u;
        const uint32_t dthr_s = dthr - 32768u;                       // the signed-compare form of mmb_keep16
        const float dscale = (EPI & EPI_RESID) ? q.drop_scale : 1.0f;
        const uint32_t halfN = (uint32_t)q.N >> 1;
        const uint32_t seed0 = (EPI & EPI_RESID) ? ((uint32_t)mrow * halfN + ((uint32_t)ncol >> 1)) * MMB_WEYL + q.drop_stream : 0u;
        const uint32_t seed_row = 16u * halfN * MMB_WEYL;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float v_ = acc[i][2 * h + (r >> 2)][r & 3] * alpha;
                    if constexpr (EPI & EPI_BIAS) v_ += bias[8 * h + r];
                    acc[i][2 * h + (r >> 2)][r & 3] = v_;
                }
                if constexpr (EPI & EPI_RESID) {
                    if (dthr) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {                // element pair k of the 8 columns: elements 2k, 2k + 1
                            uint32_t hb = mmb_pair_mix(seed0 + (uint32_t)i * seed_row + (uint32_t)(16 * h + k) * MMB_WEYL);
                            asm("" : "+v"(hb));                      // the hash word as written (v_mul_lo + one SDWA fold): hipcc otherwise feeds the two 16-bit compares
                                                                     // from y and y ^ (y << 16), i.e. a SECOND v_mul_lo_u32 by C << 16 and a v_xor per pair
                            f32x4& a4 = acc[i][2 * h + (k >> 1)];
                            const bool keep0 = (int16_t)(uint16_t)(hb & 0xFFFFu) >= (int16_t)(uint16_t)dthr_s;
                            const bool keep1 = (int16_t)(uint16_t)(hb >> 16) >= (int16_t)(uint16_t)dthr_s;
                            a4[(2 * k) & 3] = keep0 ? a4[(2 * k) & 3] * dscale : 0.f;
                            a4[(2 * k + 1) & 3] = keep1 ? a4[(2 * k + 1) & 3] * dscale : 0.f;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i + PRE < NB) load_pre(i + PRE);
            const int m = mrow + 16 * i;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = ncol + 32 * h;
                float vv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) vv[r] = acc[i][2 * h + (r >> 2)][r & 3];
                const bool ok = interior || (m < q.M && n + 8 <= q.N);
                if constexpr (EPI & EPI_GELU) {
                    if (q.aux) {
                        bf16x8 u;
#pragma unroll
                        for (int r = 0; r < 8; ++r) u[r] = f2bf(vv[r]);
                        const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)q.aux, 0, (int)((um1 * (uint32_t)q.ldaux + un) * 2u), 0x00020000);
                        const uint32_t vo = ((uint32_t)mrow * (uint32_t)q.ldaux + (uint32_t)ncol) * 2u + (uint32_t)i * (32u * (uint32_t)q.ldaux) + (uint32_t)h * 64u;
                        if (ok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, u), rsX, vo, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 8; r += 4) {                 // packed forms (common.h): bit-identical to gelu_erf, 5 instead of 8 VALU per element
                        mmb_f2 g0 = {vv[r], vv[r + 1]}, g1 = {vv[r + 2], vv[r + 3]};
                        gelu_erf4(g0, g1);
                        vv[r] = g0.x; vv[r + 1] = g0.y; vv[r + 2] = g1.x; vv[r + 3] = g1.y;
                    }
                }
                if constexpr (EPI & EPI_GELU_BWD) {
#pragma unroll
                    for (int r = 0; r < 8; r += 4) {
                        mmb_f2 d0, d1;
                        gelu_erf_grad4((mmb_f2){bf2f(pre[i][h][r]), bf2f(pre[i][h][r + 1])}, (mmb_f2){bf2f(pre[i][h][r + 2]), bf2f(pre[i][h][r + 3])}, d0, d1);
                        vv[r] *= d0.x; vv[r + 1] *= d0.y; vv[r + 2] *= d1.x; vv[r + 3] *= d1.y;
                    }
                }
                if constexpr (EPI & EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] += bf2f(pre[i][h][r]);
                }
                // The value to store is pinned in registers BEFORE the (edge-tile) predicate: hipcc otherwise sinks the whole computation
                // -- the use of the prefetched residual row included -- into the predicated block, the prefetch loads stay "pending" on
                // the path around it, and the NEXT tile's first fragment reads (which reuse those registers) get an s_waitcnt vmcnt(0):
                // a drain of the previous epilogue's stores at every tile seam.
                const uint32_t voC = voffC + (uint32_t)i * rowC + (uint32_t)h * (32u * CSZ);
                if constexpr (EPI & EPI_OUT_F32) {
                    f32x4 lo = {vv[0], vv[1], vv[2], vv[3]}, hi = {vv[4], vv[5], vv[6], vv[7]};
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    if (ok) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), rsC, voC, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rsC, voC + 16u, 0, 0);
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = f2bf(vv[r]);
                    u32x4 ow = __builtin_bit_cast(u32x4, o);
                    asm volatile("" : "+v"(ow));
                    if (ok) __builtin_amdgcn_raw_buffer_store_b128(ow, rsC, voC, 0, 0);
                }
            }
        }
    }
#undef NT8_KTILE
#undef NT8_PHASE
    };
    if constexpr (MQ0 == 3 || MQ1 == 3) {                          // a 96-row A half-tile: waves 0-3 issue two of its pieces, waves 4-7 one
        if (wave < 4) body(std::integral_constant<int, 2>{}); else body(std::integral_constant<int, 1>{});
    } else {
        body(std::integral_constant<int, MQ0 / 2>{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead tail half-tiles
    if constexpr (MULTI) {    // the last workgroup to leave hands the queue back zeroed (its draws are complete: vmcnt(0) above) for the stream's next launch
        auto kpe = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kpe));
        const __attribute__((address_space(4))) GemmNT& qe = *(const __attribute__((address_space(4))) GemmNT*)kpe;
        if (qe.tile_counter && tid == 0) {
            if (atomicAdd(qe.tile_counter_next, 1) == (int)gridDim.x - 1) {
                for (int x = 0; x < 8; ++x) atomicExch(qe.tile_counter + x, 0);
                atomicExch(qe.tile_counter_next, 0);
            }
        }
    }
#endif
}

template <int EPI, bool MULTI, int MQ>
static int launch_nt8_form(hipStream_t s, const GemmNT& q, int workgroups) {
    static std::atomic<unsigned long long> attr_done{0};
    constexpr int LDS = 131072 + 64 + 8 * 256;                  // ring | tile-queue word of the multi-tile form (padded) | bias rows
    if (int e = mmb_allow_lds((const void*)gemm_nt8_kernel<EPI, MULTI, MQ>, LDS, attr_done)) return e;
    hipLaunchKernelGGL((gemm_nt8_kernel<EPI, MULTI, MQ>), dim3(workgroups), dim3(512), LDS, s, q);
    MMB_CHECK_LAUNCH();
    return 0;
}
template <int EPI>
static int launch_nt8(hipStream_t s, const GemmNT& p, int bm, int tiles, int workgroups, int group_m) {
    GemmNT q = p;
    q.group_m = group_m;
    // the device tile queue (data-parallel runs): only the multi-tile form draws from it, one counter per XCD
    const bool use_queue = p.tile_counter && tiles > workgroups && !(workgroups & 7);
    if (!use_queue) q.tile_counter = q.tile_counter_next = nullptr;
    q.queue_xcd = 1;
    const bool multi = tiles > workgroups;                       // more tiles than workgroups: the half-tile stream crosses tile seams
    if (bm == 128) return launch_nt8_form<EPI, false, 2>(s, q, workgroups);    // (128-row tiles: single-round launches only)
    if (bm == 192) return multi ? launch_nt8_form<EPI, true, 3>(s, q, workgroups) : launch_nt8_form<EPI, false, 3>(s, q, workgroups);
    if (bm == 224) return multi ? launch_nt8_form<EPI, true, 7>(s, q, workgroups) : launch_nt8_form<EPI, false, 7>(s, q, workgroups);
    return multi ? launch_nt8_form<EPI, true, 4>(s, q, workgroups) : launch_nt8_form<EPI, false, 4>(s, q, workgroups);
}

// ---- which kernel, which tile, which tile walk: ONE function of the shape (and of mmbert_gemm_nt_force), shared by the launch path
// and by mmbert_gemm_nt_describe() (bench.py reports the choice per shape; tests pin it).  Nothing here reads the environment: the
// rules below are the outcome of rounds 2-4's in-situ A/Bs (the switches they were measured with are gone) ----
enum { NTK_128 = 0, NTK_8PHASE = 3 };
struct NTChoice { int kernel, bm, tiles, workgroups, group_m, use_queue; };

// the 8-phase kernel: an even number (>= 4) of 64-deep K tiles, 32-bit buffer offsets
static bool nt8_eligible(const GemmNT& p, int epi) {
    // (the epilogue addresses C / R / U / aux as buffers too: M * pitch * element size below 2^32 bytes)
    const long long m = p.M, esz = (epi & EPI_OUT_F32) ? 4 : 2;
    return !(p.K & 127) && p.K >= 256 && m * p.lda < (1ll << 31) && (long long)p.N * p.ldb < (1ll << 31) && m * p.ldc * esz < (1ll << 32) &&
           m * p.ldr * 2 < (1ll << 32) && m * p.ldaux * 2 < (1ll << 32) && m * p.ldu * 2 < (1ll << 32);
}

// Tile walk of the multi-tile form (ntp_tile_mn).  An XCD's 32 workgroups own a contiguous chunk of the walk (xcd_remap), i.e.
// ceil(tiles_m / 8) row panels; with MORE tiles than CUs the order inside that chunk decides which panels its 32 concurrent tiles share:
//  * row-major (group_m = 1): ~3 row panels x ALL column panels at a time -- a weight panel is wanted by 3 workgroups at once and the
//    whole weight matrix (3.5-4.7 MB at N = 2304 / 3072) passes through the 4-MiB L2 once per 3 row panels;
//  * one group per XCD (group_m = ceil(tiles_m / 8), round 3): the XCD sweeps the column panels with ALL its row panels, 11 row panels
//    x ~3 column panels at a time -- a weight panel is wanted by 11 workgroups at once and then never again on this XCD.  Same-process
//    A/B of the train step over group_m = 1 / 6 / 8 / 11 / 16 / 32 / 100: 15.56 / 15.34 / 15.30 / 14.98 / 15.30 / 15.18 / 15.16 ms (round 3),
//    4 / 6 / 8 / 16 / 1 against 11 again at the end of round 4: +2.6 / +1.8 / +1.4 / +0.2 / +3.0 %.  (Fetched bytes barely move: DESIGN 3.1.)
//  * the vocabulary projection (B = 47 MB, 9 960 tiles) keeps its groups of 4 row panels (round 2: 1 / 2 / 4 / 8 -> 871 / 855 / 844 /
//    875 us): its column sweep is 120 panels long.
static int nt_group_m(int M, int N, int K, int bm, int tiles, int cus) {
    if (tiles <= cus) return 1;
    if ((long long)N * K * 2 > (8ll << 20)) return tiles > 4 * cus ? 4 : 1;
    return ((M + bm - 1) / bm + 7) / 8;
}

static NTChoice nt_choose(const GemmNT& p, int epi) {
    NTChoice c = {NTK_128, 128, ((p.M + 127) / 128) * ((p.N + 127) / 128), 0, 1, 0};
    c.workgroups = c.tiles;
    const int force = g_nt_force.load();
    // the 256-wide pipeline needs enough rows and columns to fill its tiles, 16-byte epilogue accesses and an eligible K
    const bool big = (p.M >= 512) && (p.N >= 256) && !(p.N & 7) && !(p.ldc & 7) && !(p.ldr & 7) && !(p.ldaux & 7) && !(p.ldu & 7);
    if (force == 1 || !nt8_eligible(p, epi) || (p.N & 7) || (force == 0 && !big)) return c;
    const int cus = device_cus(), tn = (p.N + 255) / 256;
    auto tiles_of = [&](int h) { return ((p.M + h - 1) / h) * tn; };
    const int t128 = tiles_of(128), t192 = tiles_of(192), t224 = tiles_of(224), t256 = tiles_of(256);
    auto take = [&](int h) {
        const int t = tiles_of(h);
        c.kernel = NTK_8PHASE; c.bm = h; c.tiles = t;
        c.workgroups = (h == 128 || t <= cus) ? t : cus;          // (128-row tiles have no multi-tile form: forced on a multi-round shape they launch every tile)
        c.group_m = nt_group_m(p.M, p.N, p.K, h, t, cus);
        c.use_queue = p.tile_counter != nullptr && t > c.workgroups && !(cus & 7);
        return c;
    };
    if (force == 128 || force == 192 || force == 224 || force == 256) return take(force);
    // ---- launches whose 256-row tiles fit the chip in ONE round: the smallest tile height that still fits one round ----
    if (t256 <= cus) {
        if (2 * t256 <= cus) {
            // ... that would leave more than half the chip idle (the reference's default model: M = 6400, N = 1024 is 100 tiles): 128-row
            // tiles when those fit one round and fill at least half of it (22-24 % over the 128 x 128 kernel there,
            // profiles/r4_bert_large_gemm_modes_bm128.log); fewer still: the 128 x 128 kernel's 4 x as many tiles on 2 workgroups per CU
            if (t128 <= cus && 2 * t128 >= cus) return take(128);
            if (force == 0 && c.tiles <= 2 * cus) return c;
            return take(256);
        }
        // the input gradients at ~13 850 packed rows: 165 tiles of 256 rows on 256 CUs, 219 of 192 (-0.9 % of the step, round 4)
        if (t192 <= cus) return take(192);
        // the forward N = 768 shapes at 18 400 rows: 216 tiles of 256 rows, 249 of 224 (-0.4 ... -0.9 %, six A/Bs: profiles/r4_ab_8phase_bm224.log)
        if (t224 <= cus && t224 > t256) return take(224);
        return take(256);
    }
    // ---- more tiles than CUs: the multi-tile form.  Tile height by started rounds x time per K tile (192 rows 1 680 clk -- LDS-DMA bound --,
    // 224 rows 1 800, 256 rows 2 048 -- MFMA bound): 224 rows unless another height is >= 3 % cheaper.  At the headline shapes 224 wins
    // everywhere (QKV 3 rounds, FFN-up 4, GELU' input gradient 3, vocabulary 39); not with a few hundred rows more in backward (14 400
    // rows x N = 3072: 780 tiles of 224 rows = 4 rounds, 684 of 256 = 3) or at bert-large's QKV (6 400 x 3072: 192-row tiles).
    // Round 4, on top of the 8-phase weight-gradient kernel: -2.3 / -2.3 / -2.5 % of the step against the former ring-persistent kernel,
    // whose bits it reproduces (same K order per element; profiles/r4_ab_8phase_m224.log).
    auto rounds = [&](int t) { return (long long)((t + cus - 1) / cus); };
    const long long c192 = rounds(t192) * 1680, c224 = rounds(t224) * 1800, c256 = rounds(t256) * 2048;
    int h = 224;
    if (100 * c256 < 97 * c224 && c256 <= c192) h = 256;
    else if (100 * c192 < 97 * c224 && c192 < c256) h = 192;
    return take(h);
}

template <int EPI>
static int dispatch_nt(hipStream_t s, const GemmNT& p) {
    const NTChoice c = nt_choose(p, EPI);
    if (c.kernel == NTK_128) return launch_nt<EPI>(s, p);
    return launch_nt8<EPI>(s, p, c.bm, c.tiles, c.workgroups, c.group_m);
}

// -------------------------------------------------------------------------------------------------
// TN (weight gradients): W_p[N_p,K_p] (+)= alpha * A_p[M,N_p]^T . B_p[M,K_p]  for up to TN_MAXP problems that share the token
// axis M (the four dense layers of one, two or -- deferred to the end of backward -- all encoder layers go out as ONE call), optional
// split over M into fp32 slabs reduced deterministically by tn_reduce_kernel, and optionally bias_p[N_p] += alpha * colsum(A_p).
// Workgroup = 512 threads, output tile 256(n) x 256(k), one workgroup per CU: a CU issues a 16-byte-per-lane LDS-DMA every ~30-37 clk at
// best, and a 256 x 128 tile (rounds 1-2) needed 48 of them per 1024 MFMA clocks -- load-issue bound at 58 % of the MFMA rate.
// -------------------------------------------------------------------------------------------------
// (M: the problem's own token rows -- round 6: problems of FEW rows ride in the last launch of the deferred call, see mmbert_gemm_tn_grouped_rows;
//  slab_off / bias_off in elements: int, so that 52 records stay inside the 4 KiB of kernel arguments; acc: this problem adds to W instead of
//  overwriting it)
struct TNProb { const bf16_t* A; const bf16_t* B; float* W; float* bias; int N, K, lda, ldb, tiles_k, tile0, M, slab_off, bias_off, acc; };
// up to TN_MAXP problems per launch: the four dense layers of an encoder layer -- or of TWO layers (model._EncoderFn pairs them: 216
// tiles fill the chip in one round without splitting the token axis, so no fp32 slabs and no reduce launch) -- or, round 4, of up to
// TWELVE layers at once: without a gradient hook (one GPU) nothing needs a layer's weight gradients before the optimizer, so the model
// defers them all to ONE call at the end of backward, which goes out as whole rounds of CUs-many tiles (11 layers = 1188 tiles = 4 full
// launches + one of 164, against 5 paired launches at 216 of 256 CUs plus a split single layer).  52 x 72 B = 3.7 KiB of kernel arguments.
#define TN_MAXP 52
struct GemmTNG {
    TNProb pr[TN_MAXP];
    float* slab; const float* alpha_dev;
    float* bias_slab; long long bias_stride;   // deterministic mode with a token split: the splits >= 1 STORE their bias sums here (tn_reduce folds them in split order)
    long long slab_stride;
    int nprob, total_tiles, M, splits, rows_per_split, accumulate;
    float alpha;
    int tile_base;
    int remap_n;                                // the launch's first remap_n workgroups take their tiles in the XCD-contiguous order; the rest (few-row tiles) in index order
};

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

// ds_read_b64_tr_b16 through inline asm: behind the builtin hipcc cannot prove that the read does not
// alias the LDS-DMA still in flight and drains vmcnt(0) in front of every stage's reads.  The asm form is
// invisible to its scoreboard: the caller waits lgkmcnt(0) itself and fences with sched_barrier(0).
template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& dst, unsigned lds_addr) {
    // "memory": the read must stay behind the s_waitcnt / s_barrier that publish the LDS-DMA data (without it hipcc
    // hoisted these reads above the barrier: stale LDS -> NaN)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}

// -------------------------------------------------------------------------------------------------
// TN in the 8-phase structure: the weight gradients are the deepest products of the step (the reduction runs over ~14 000 tokens =
// 216 K tiles of 64).  The K loop of gemm_nt8_kernel (1 363 clk per 32 tokens where the 4-slot-ring form of rounds 1-3 spent 1 476 and the
// MFMAs need 1 024; that form's bits are reproduced: same 32-token summation blocks -- round 4's test of it is in git history):
//   * K tile = 64 tokens; half-tiles of [64 tokens][128 columns] (256-byte rows: an LDS-DMA wave instruction moves 4 token rows x
//     two whole cache lines) are the unit of staging, waiting and re-use; order of first use Y0h, X0h, Y1h, X1h;
//   * waves 2 (wr) x 4 (wc): a wave owns k columns {ha*128 + wr*64 ..+63} (X, 4 blocks of 16 per half) x n columns
//     {hb*128 + wc*32 ..+31} (dY, 2 blocks per half) -- the halves are CONTIGUOUS 128-column panels, the wave interleave sits inside;
//   * phases: (X0h,Y0h) (X0h,Y1h) (X1h,Y1h) (X1h,Y0h), 16 MFMAs each; two wave groups one barrier apart; the stream runs three
//     half-tiles ahead behind one counted vmcnt(6) per K tile;
//   * fragments by ds_read_b64_tr_b16 (48 per K tile and wave), conflict-free with the chunk XOR tn_swz on
//     256-byte rows: the 8 token rows a 32-lane half touches fall on 8 distinct 32-byte bank groups;
//   * bias gradients on the VALU instead of an all-ones MFMA: a bias tile (k0 == 0) sums the dY fragments it reads anyway, wave
//     (wr, wc) the k-step ks = wr of its own 32 columns: 20 VALU operations per phase beside 16 MFMAs; lanes, then the two wr waves
//     (through LDS) are added in a fixed order: one atomic per column, reproducible.
// Token rows past the split's end read as zeros through the buffer range check (num_records = mend rows), so ragged ends need no
// masking in registers.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(const GemmTNG g) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];   // buffer d at d * 65536: X0h | X1h | Y0h | Y1h, 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int t_lin = g.tile_base + ((int)blockIdx.x < g.remap_n ? xcd_remap(blockIdx.x, g.remap_n) : (int)blockIdx.x);
    const __attribute__((address_space(4))) GemmTNG& gq = *(const __attribute__((address_space(4))) GemmTNG*)__builtin_amdgcn_kernarg_segment_ptr();
    int pi = 0;
    {
        int lo = 0, hi = g.nprob - 1;                              // tile0 ascending: the last problem that starts at or before t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (gq.pr[mid].tile0 <= t_lin) lo = mid; else hi = mid - 1; }
        pi = lo;
    }
    const __attribute__((address_space(4))) TNProb& pr = gq.pr[pi];
    const bf16_t* Ap = pr.A; const bf16_t* Bp = pr.B; float* Wp = pr.W; float* biasp = pr.bias;
    const int N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, tiles_k = pr.tiles_k;
    const int slab_off = pr.slab_off, bias_off = pr.bias_off;
    t_lin -= pr.tile0;
    const int n0 = (t_lin / tiles_k) << 8, k0 = (t_lin % tiles_k) << 8;
    const int split = blockIdx.y;
    const int mbeg = split * g.rows_per_split;
    const int mend = min(pr.M, mbeg + g.rows_per_split);
    const int nt = (mend - mbeg + 63) >> 6;                        // K tiles of 64 tokens; <= 0 for a trailing empty split
    const bool do_bias = (biasp != nullptr) && (k0 == 0);

    // ---- staging: piece j (token rows 4j .. 4j+3 of the half-tile, 1 KiB) by wave j & 7; lane -> (row lane >> 4, chunk lane & 15) ----
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)((uint32_t)max(mend, 0) * (uint32_t)lda * 2u), 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)((uint32_t)max(mend, 0) * (uint32_t)ldb * 2u), 0x00020000);
    uint32_t y_off[2], x_off[2];
    {
        const int row4 = lane >> 4, pos = lane & 15;
        const int key = (row4 | (((wave >> 1) & 1) << 2)) << 1;    // tn_swz(4 j + row4), the same for both pieces of a wave (j = wave, wave + 8)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cy = min(n0 + h * 128 + ((pos ^ key) << 3), N - 8);
            const int cx = min(k0 + h * 128 + ((pos ^ key) << 3), K - 8);
            y_off[h] = ((uint32_t)row4 * (uint32_t)lda + (uint32_t)cy) * 2u;
            x_off[h] = ((uint32_t)row4 * (uint32_t)ldb + (uint32_t)cx) * 2u;
        }
    }
    // half-tile ids in the order of first use: 0 = Y0h, 1 = X0h, 2 = Y1h, 3 = X1h
    auto stage = [&](int buf, int which, int kt) {
        const int h = which >> 1;
        const uint32_t row = (uint32_t)(mbeg + kt * 64 + 4 * wave);
        char* base = smem + buf * 65536 + ((which & 1) ? 0 : 32768) + h * 16384 + wave * 1024;
        if (which & 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LPTR(base), 16, x_off[h], row * (uint32_t)ldb * 2u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LPTR(base + 8192), 16, x_off[h], (row + 32u) * (uint32_t)ldb * 2u, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LPTR(base), 16, y_off[h], row * (uint32_t)lda * 2u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, LPTR(base + 8192), 16, y_off[h], (row + 32u) * (uint32_t)lda * 2u, 0, 0);
        }
    };

    // ---- transposed-read addresses: lane (g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3) supplies token row 8 g4 + q (+ 4), 4 columns at 4 pp ----
    const int g4 = lane >> 4, r0 = 8 * g4 + ((lane >> 2) & 3), pp = lane & 3;
    const int keyr = tn_swz(r0);
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(smem);
    unsigned adX[2][4], adY[2][2];                                 // [buffer][16-column block]; ds offsets (half, k step, + 4 rows) are immediates
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chunk = wr * 8 + 2 * i + (pp >> 1);
        adX[0][i] = lds0 + r0 * 256 + ((chunk ^ keyr) << 4) + ((pp & 1) << 3);
        adX[1][i] = adX[0][i] + 65536;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int chunk = wc * 4 + 2 * j + (pp >> 1);
        adY[0][j] = lds0 + 32768 + r0 * 256 + ((chunk ^ keyr) << 4) + ((pp & 1) << 3);
        adY[1][j] = adY[0][j] + 65536;
    }

    u32x2 xlo[2][4], xhi[2][4], y0lo[2][2], y0hi[2][2], y1lo[2][2], y1hi[2][2];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};

    auto read_x_ks = [&](auto buf_c, auto h_c, auto ks_c) {
        constexpr int D = decltype(buf_c)::value, OFF = decltype(h_c)::value * 16384 + decltype(ks_c)::value * 8192, KS = decltype(ks_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) { tr_read<OFF>(xlo[KS][i], adX[D][i]); tr_read<OFF + 1024>(xhi[KS][i], adX[D][i]); }
    };
    auto read_y = [&](auto buf_c, auto h_c, u32x2 (&lo)[2][2], u32x2 (&hi)[2][2]) {
        constexpr int D = decltype(buf_c)::value, OFF = decltype(h_c)::value * 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ks == 0) { tr_read<OFF>(lo[0][j], adY[D][j]); tr_read<OFF + 1024>(hi[0][j], adY[D][j]); }
                else         { tr_read<OFF + 8192>(lo[1][j], adY[D][j]); tr_read<OFF + 8192 + 1024>(hi[1][j], adY[D][j]); }
            }
    };
    auto mma = [&](int qa, int qb, const u32x2 (&ylo)[2][2], const u32x2 (&yhi)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 yf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { const u32x4 v = {ylo[ks][j][0], ylo[ks][j][1], yhi[ks][j][0], yhi[ks][j][1]}; yf[j] = __builtin_bit_cast(bf16x8, v); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = {xlo[ks][i][0], xlo[ks][i][1], xhi[ks][i][0], xhi[ks][i][1]};
                const bf16x8 xf = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int j = 0; j < 2; ++j)     // D[row <-> k_out (X^T as the first operand)][col <-> n_out (dY as the second)]
                    acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[j], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);
            }
        }
    };
    // bias tiles: this wave's k step (ks = wr) of the dY half just read, summed per lane (8 tokens of one column) in a fixed order
    auto bias_add = [&](int hb, const u32x2 (&ylo)[2][2], const u32x2 (&yhi)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t w0 = wr ? ylo[1][j][0] : ylo[0][j][0], w1 = wr ? ylo[1][j][1] : ylo[0][j][1];
            const uint32_t w2 = wr ? yhi[1][j][0] : yhi[0][j][0], w3 = wr ? yhi[1][j][1] : yhi[0][j][1];
            float s = bsum[hb][j];
            s += __builtin_bit_cast(float, w0 << 16); s += __builtin_bit_cast(float, w0 & 0xFFFF0000u);
            s += __builtin_bit_cast(float, w1 << 16); s += __builtin_bit_cast(float, w1 & 0xFFFF0000u);
            s += __builtin_bit_cast(float, w2 << 16); s += __builtin_bit_cast(float, w2 & 0xFFFF0000u);
            s += __builtin_bit_cast(float, w3 << 16); s += __builtin_bit_cast(float, w3 & 0xFFFF0000u);
            bsum[hb][j] = s;
        }
    };

#ifdef MMB_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr0) :: "memory");
    MMB_STAMP(ts0)
#endif
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
    // one phase: { transposed reads of this phase's quadrant operands ; one half-tile of LDS-DMA ; [counted vmcnt] ; barrier ; lgkmcnt(0) ;
    //              MFMAs (+ the bias sums) ; barrier } -- RAW / WAR argument as in gemm_nt8_kernel (the half-tile stream and the phase in
    // which a half-tile is read are the same); the Y0h reads are retired (lgkmcnt(8): only the second k step of X0h behind them) before
    // phase 1's first barrier, because Y0h is restaged in phase 2.
#define TN8_PHASE(READS, STAGE, VMWAIT, QA, QB, YLO, YHI, BIAS_HB)                                                    \
    {                                                                                                                \
        READS;                                                                                                       \
        STAGE;                                                                                                       \
        if (VMWAIT >= 0) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(VMWAIT < 0 ? 0 : VMWAIT, 15));                        \
        __builtin_amdgcn_s_barrier();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        mma(QA, QB, YLO, YHI);                                                                                       \
        if (BIAS_HB >= 0) { if (do_bias) bias_add(BIAS_HB < 0 ? 0 : BIAS_HB, YLO, YHI); }                             \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
#define TN8_KTILE(DC, D, T)                                                                                                          \
    TN8_PHASE((read_y(DC{}, I0{}, y0lo, y0hi), read_x_ks(DC{}, I0{}, I0{}), ({ asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }), \
               read_x_ks(DC{}, I0{}, I1{})), stage(D ^ 1, 3, (T) + 1), -1, 0, 0, y0lo, y0hi, 0)                                       \
    TN8_PHASE(read_y(DC{}, I1{}, y1lo, y1hi), stage(D, 0, (T) + 2), -1, 0, 1, y1lo, y1hi, 1)                                          \
    TN8_PHASE((read_x_ks(DC{}, I1{}, I0{}), read_x_ks(DC{}, I1{}, I1{})), stage(D, 1, (T) + 2), -1, 1, 1, y1lo, y1hi, -1)             \
    TN8_PHASE((void)0, stage(D, 2, (T) + 2), 6, 1, 0, y0lo, y0hi, -1)

    if (nt > 0) {
        stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
        stage(1, 0, 1); stage(1, 1, 1); stage(1, 2, 1);
        __builtin_amdgcn_s_waitcnt(mmb_waitcnt(6, 15));            // K tile 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();                 // the stagger: group 1 runs one barrier behind group 0
        int t = 0;
        while (true) {
            TN8_KTILE(I0, 0, t) if (++t >= nt) break;
            TN8_KTILE(I1, 1, t) if (++t >= nt) break;
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();                 // balances the stagger
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the dead tail half-tiles (token rows past the end: zeros, no traffic)
    }
#undef TN8_KTILE
#undef TN8_PHASE
#ifdef MMB_STAMPS
    MMB_STAMP(ts1)
#endif

    // ---- epilogue: acc[ha*4 + i][hb*2 + j][r] = W[n0 + hb*128 + wc*32 + 16 j + fr][k0 + ha*128 + wr*64 + 16 i + 4 fq + r] ----
    const float alpha = g.alpha * (g.alpha_dev ? *g.alpha_dev : 1.0f);
    const int fr = lane & 15, fq = lane >> 4;
    float* out = split > 0 ? g.slab + (size_t)(split - 1) * g.slab_stride + slab_off : Wp;
    const bool accum = (split == 0) && pr.acc;
#pragma unroll
    for (int ib = 0; ib < 8; ib += 2) {                            // two k blocks at a time: their eight accumulate reads are issued together
        float4 old[2][4];
        if (accum) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kc = min(k0 + ((ib + u) >> 2) * 128 + wr * 64 + ((ib + u) & 3) * 16 + fq * 4, K - 4);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int nc = min(n0 + (jb >> 1) * 128 + wc * 32 + (jb & 1) * 16 + fr, N - 1);
                    old[u][jb] = *(const float4*)(out + (size_t)nc * K + kc);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + ((ib + u) >> 2) * 128 + wr * 64 + ((ib + u) & 3) * 16 + fq * 4;
            if (k >= K) continue;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int n = n0 + (jb >> 1) * 128 + wc * 32 + (jb & 1) * 16 + fr;
                if (n >= N) continue;
                const f32x4 a = acc[ib + u][jb];
                float4 v = make_float4(a[0] * alpha, a[1] * alpha, a[2] * alpha, a[3] * alpha);
                if (accum) { v.x += old[u][jb].x; v.y += old[u][jb].y; v.z += old[u][jb].z; v.w += old[u][jb].w; }
                *(float4*)(out + (size_t)n * K + k) = v;
            }
        }
    }
    if (do_bias) {                                                 // workgroup-uniform
        float* red = (float*)smem;                                 // [wc][hb][j][fr]: the wr = 1 waves' sums (the ring is dead: vmcnt(0) above + this barrier)
        float v[2][2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s = bsum[hb][j];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                v[hb][j] = s;
            }
        __syncthreads();
        if (wr == 1 && fq == 0) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < 2; ++j) red[((wc * 2 + hb) * 2 + j) * 16 + fr] = v[hb][j];
        }
        __syncthreads();
        if (wr == 0 && fq == 0) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + hb * 128 + wc * 32 + j * 16 + fr;
                    if (n >= N) continue;
                    const float bv = (v[hb][j] + red[((wc * 2 + hb) * 2 + j) * 16 + fr]) * alpha;
                    // one adder per column: split 0 (the only split of an unsplit launch).  The other splits add in arrival order -- or, in
                    // deterministic mode, store their sums for tn_reduce_kernel to fold in split order
                    if (split > 0 && gq.bias_slab) gq.bias_slab[(size_t)(split - 1) * gq.bias_stride + bias_off + n] = bv;
                    else atomicAdd(biasp + n, bv);
                }
        }
    }
#ifdef MMB_STAMPS
    if (g_stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr1) :: "memory");
        unsigned long long te;
        MMB_STAMP(te)
        unsigned long long* o = g_stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 6;
        o[0] = ts1 - ts0; o[1] = (unsigned long long)(nt > 0 ? 2 * nt : 0); o[2] = te - ts1; o[3] = 1; o[4] = tr0; o[5] = tr1;
    }
#endif
#endif
}

// W[i] += sum_{s >= 1} slab[s - 1][i] over the concatenated outputs of all problems of a launch (split 0 went to W directly)
struct TNReduce { float* W[TN_MAXP]; long long off[TN_MAXP + 1]; int nprob, splits, accumulate; long long slab_stride;
                  float* bias[TN_MAXP]; long long boff[TN_MAXP + 1]; const float* bias_slab; long long bias_stride; };
__global__ void tn_reduce_kernel(const TNReduce r, const float* __restrict__ slab) {
    if (r.bias_slab) {                                      // deterministic mode: bias[n] += the bias sums of splits 1, 2, ... in that order
        const long long nb = r.boff[r.nprob];
        for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nb; e += (long long)gridDim.x * blockDim.x) {
            int lo = 0, hi = r.nprob - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (r.boff[mid] <= e) lo = mid; else hi = mid - 1; }
            if (!r.bias[lo]) continue;
            float v = r.bias[lo][e - r.boff[lo]];
            for (int s = 0; s + 1 < r.splits; ++s) v += r.bias_slab[(size_t)s * r.bias_stride + e];
            r.bias[lo][e - r.boff[lo]] = v;
        }
    }
    const long long total4 = r.off[r.nprob] >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i << 2;
        int lo = 0, hi = r.nprob - 1;                      // off ascending: the last problem that starts at or before e (per lane)
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (r.off[mid] <= e) lo = mid; else hi = mid - 1; }
        float* W = r.W[lo];
        const long long o = r.off[lo];
        float4* dst = (float4*)(W + (e - o));
        float4 v = *dst;                                   // split 0 of the GEMM has stored its part here
        for (int s = 0; s + 1 < r.splits; ++s) {
            const float4 t = *(const float4*)(slab + (size_t)s * r.slab_stride + e);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        *dst = v;
    }
}

// column sums: out[n] += alpha * sum_m X[m][n]   (bias gradients).  grid (ceil(N/512), row chunks);
// a lane owns 8 columns (16-byte loads), the 4 waves take interleaved rows, one atomic per column per workgroup.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int M, int N, int ldx, float* __restrict__ out, float alpha, const float* alpha_dev, int rows_per_block) {
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + lane) * 8;
    const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float s[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) s[r] = 0.f;
    if (n < N) {
        int m = mbeg + sub;
        for (; m + 12 < mend; m += 16) {                              // 4 independent 16-byte loads in flight
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(X + (size_t)(m + 4 * u) * ldx + n);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) s[r] += bf2f(v[u][r]);
        }
        for (; m < mend; m += 4) {
            const bf16x8 v = *(const bf16x8*)(X + (size_t)m * ldx + n);
#pragma unroll
            for (int r = 0; r < 8; ++r) s[r] += bf2f(v[r]);
        }
    }
    __shared__ float red[4][64][9];
#pragma unroll
    for (int r = 0; r < 8; ++r) red[sub][lane][r] = s[r];
    __syncthreads();
    if (sub == 0 && n < N) {
        const float a = alpha * (alpha_dev ? *alpha_dev : 1.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            atomicAdd(out + n + r, (red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r]) * a);
    }
}

extern "C" {

// C (bf16) = sum over the split-K slabs (+ R)
__global__ void nt_splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long stride, int M, int N, bf16_t* __restrict__ C, int ldc,
                                        const bf16_t* __restrict__ R, int ldr) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long long)M * N) return;
    float4 a = *(const float4*)(slabs + i);
    for (int z = 1; z < splits; ++z) { const float4 b = *(const float4*)(slabs + z * stride + i); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    if (R) { const bf16x4 r = *(const bf16x4*)(R + (size_t)m * ldr + n); a.x += bf2f(r[0]); a.y += bf2f(r[1]); a.z += bf2f(r[2]); a.w += bf2f(r[3]); }
    bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
    *(bf16x4*)(C + (size_t)m * ldc + n) = o;
}

// Split-K form for long-K products with few output tiles (the MLM head's compact dlogits . E^T: M ~ 360, K = 30592):
// C[M,N] (bf16) = A[M,K] . B[N,K]^T (+ R[M,N] bf16, optional: the residual form of the top layer's few-row input gradients)
// through fp32 slabs (deterministic).  workspace >= mmbert_gemm_nt_splitk_workspace() bytes.
static int nt_splitk_plan(int M, int N, int K) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128), kt = K >> 6;
    int splits = (2 * device_cus() + tiles - 1) / tiles;          // ~2 workgroups per CU
    if (splits > kt / 4) splits = kt / 4;                          // at least 4 K tiles per workgroup
    if (splits < 1) splits = 1;
    return splits;
}
size_t mmbert_gemm_nt_splitk_workspace(int M, int N, int K) { return (size_t)nt_splitk_plan(M, N, K) * M * N * sizeof(float); }

int mmbert_gemm_nt_splitk(hipStream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                          int M, int N, int K, void* workspace, const void* R, int ldr) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 63) || (N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3) || !workspace || (R && (ldr & 3))) return -1;
    GemmNT p = {};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = workspace; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = N;
    p.alpha = 1.0f;
    const int splits = nt_splitk_plan(M, N, K), kt = K >> 6;
    p.kt_per_split = (kt + splits - 1) / splits;
    p.split_stride = (long long)M * N;
    const int zs = (kt + p.kt_per_split - 1) / p.kt_per_split;
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)gemm_nt_kernel<EPI_OUT_F32>, 65536, attr_done)) return e;
    hipLaunchKernelGGL(gemm_nt_kernel<EPI_OUT_F32>, dim3(((M + 127) / 128) * ((N + 127) / 128), 1, zs), dim3(256), 65536, stream, p);
    MMB_CHECK_LAUNCH();
    const long long n4 = ((long long)M * N + 3) / 4;
    hipLaunchKernelGGL(nt_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, zs, p.split_stride, M, N, (bf16_t*)C, ldc,
                       (const bf16_t*)R, ldr);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_gemm_nt(hipStream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                   int M, int N, int K, int epi, const float* bias, const void* R, int ldr, void* aux, int ldaux,
                   const void* U, int ldu, float alpha, const float* alpha_dev,
                   uint32_t drop_stream, uint32_t drop_thr16, float drop_scale, int* tile_queue) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 63) || (N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3)) return -1;
    GemmNT p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.R = (const bf16_t*)R;
    p.aux = (bf16_t*)aux; p.U = (const bf16_t*)U; p.alpha_dev = alpha_dev;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.ldu = ldu;
    p.alpha = alpha; p.drop_stream = drop_stream; p.drop_thr16 = drop_thr16; p.drop_scale = drop_scale;
    p.kt_per_split = 0; p.split_stride = 0; p.group_m = 1;
    p.tile_counter = tile_queue; p.tile_counter_next = tile_queue ? tile_queue + 8 : nullptr;
    switch (epi) {
        case 0: return dispatch_nt<0>(stream, p);
        case EPI_BIAS: return dispatch_nt<EPI_BIAS>(stream, p);
        case EPI_BIAS | EPI_GELU: return dispatch_nt<EPI_BIAS | EPI_GELU>(stream, p);
        case EPI_BIAS | EPI_RESID: return dispatch_nt<EPI_BIAS | EPI_RESID>(stream, p);
        case EPI_RESID: return dispatch_nt<EPI_RESID>(stream, p);
        case EPI_GELU_BWD: return dispatch_nt<EPI_GELU_BWD>(stream, p);
        case EPI_OUT_F32: return dispatch_nt<EPI_OUT_F32>(stream, p);
        case EPI_BIAS | EPI_OUT_F32: return dispatch_nt<EPI_BIAS | EPI_OUT_F32>(stream, p);
        default: return -2;
    }
}

// Which kernel mmbert_gemm_nt would launch for this shape on the current device (contiguous operands assumed: ld = K / N), without
// launching anything: out[0] = kernel (0: the 128 x 128 kernel, 3: the 8-phase kernel), out[1] = tile rows (128 / 192 / 224 / 256),
// out[2] = tile columns, out[3] = output tiles, out[4] = workgroups launched (fewer than tiles: the multi-tile form),
// out[5] = tile rounds x 100 over the device's CUs, out[6] = group_m of the tile walk, out[7] = CUs.  Host-only.
int mmbert_gemm_nt_describe(int M, int N, int K, int epi, int with_queue, int* out) {
    if (!out || M <= 0 || N <= 0 || K <= 0) return -1;
    GemmNT p = {};
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ldr = (epi & (EPI_RESID)) ? N : 0; p.ldaux = (epi & EPI_GELU) ? N : 0;
    p.ldu = (epi & EPI_GELU_BWD) ? N : 0;
    static int dummy_queue[16];
    p.tile_counter = with_queue ? dummy_queue : nullptr;
    const NTChoice c = nt_choose(p, epi);
    const int cus = device_cus();
    out[0] = c.kernel; out[1] = c.bm; out[2] = c.kernel == NTK_128 ? 128 : 256; out[3] = c.tiles; out[4] = c.workgroups;
    out[5] = (int)(100.0 * c.tiles / cus + 0.5); out[6] = c.group_m; out[7] = cus;
    return 0;
}

#ifdef MMB_STAMPS
int mmbert_debug_set_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf));
}
#endif
// test / benchmarking hook: 0 = by shape | 1 = always the 128 x 128 kernel | 8 = the 8-phase kernel wherever it is eligible (K % 128 == 0,
// K >= 256, N % 8 == 0), tile height by the shape rules | 128 / 192 / 224 / 256 = the 8-phase kernel on that tile height (single- or
// multi-tile form by the tile count; 128-row tiles always one tile per workgroup).  Anything else is read as 0.
void mmbert_gemm_nt_force(int mode) {
    g_nt_force.store((mode == 1 || mode == 8 || mode == 128 || mode == 192 || mode == 224 || mode == 256) ? mode : 0);
}

static std::atomic<int> g_tn_splits{0};   // 0 = by shape; > 0 forces the split count of the token axis (A/B benchmarking)
void mmbert_gemm_tn_force_splits(int splits) { g_tn_splits.store(splits); }
static std::atomic<int> g_tn_one_launch{0};   // 1: a call of more long tiles than CUs goes out as ONE launch instead of one per round (A/B benchmarking)
void mmbert_gemm_tn_force_one_launch(int on) { g_tn_one_launch.store(on ? 1 : 0); }
static int tn_plan(int nprob, const int* N, const int* K, int M, int* splits_out, int* tiles_out) {
    int tiles = 0;
    double elems = 0;
    for (int i = 0; i < nprob; ++i) { tiles += ((N[i] + 255) / 256) * ((K[i] + 255) / 256); elems += (double)N[i] * K[i]; }
    // Split count of the token axis by a cost model (one 256x256-tile workgroup per CU): a launch of W workgroups costs
    // ceil(W / CUs) workgroup times, a workgroup streams a token row in ~T_ROW and spends ~T_FIX outside its loop; every extra
    // slab costs a write and a read of the weights' fp32 image (mostly L2 / MALL hits).  Constants fitted with
    // tools/bench_tn.py at M = 18400.  The smallest split count within 3 % of the best modelled cost wins.
    constexpr double T_ROW = 21e-9, T_FIX = 8e-6;
    const int slots = device_cus();
    const int max_splits = (M + 511) / 512;
    double cost[9];
    double best_cost = 1e300;
    int top = 1;
    for (int sp = 1; sp <= 8 && sp <= (max_splits < 1 ? 1 : max_splits); ++sp) {
        const long long wgs = (long long)tiles * sp;
        const long long rounds = (wgs + slots - 1) / slots;
        const double rows = (double)((M + sp - 1) / sp);
        cost[sp] = (rows * T_ROW + T_FIX) * (double)rounds + (sp - 1) * elems * 8.0 / 9.4e12;
        if (cost[sp] < best_cost) best_cost = cost[sp];
        top = sp;
    }
    int best = 1;
    for (int sp = 1; sp <= top; ++sp)
        if (cost[sp] <= 1.03 * best_cost) { best = sp; break; }
    int splits = best;
    // more than 8 problems = the deferred multi-layer call (model._auto_defer_wgrads): whole rounds of CUs-many tiles, never split.  Up to
    // 8 problems keep the cost model's answer (the dense tied-decoder gradient, 360 tiles at M ~ 18 k: 2 splits)
    if (tiles >= slots && nprob > 8) splits = 1;
    if (g_tn_splits.load() > 0) splits = g_tn_splits.load();       // the forced count wins (tests, A/B runs)
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    *splits_out = splits; *tiles_out = tiles;
    return 0;
}

// slab bytes the caller must provide for a grouped launch (0 when the token axis is not split)
size_t mmbert_gemm_tn_grouped_workspace(int nprob, const int* N, const int* K, int M, int* splits_out) {
    int splits, tiles;
    tn_plan(nprob, N, K, M, &splits, &tiles);
    if (splits_out) *splits_out = splits;
    if (splits == 1) return 0;
    size_t elems = 0, cols = 0;
    for (int i = 0; i < nprob; ++i) { elems += (size_t)N[i] * K[i]; cols += (size_t)N[i]; }
    // (+ one row of bias sums per problem and extra split: deterministic mode stores them instead of adding them with atomics)
    return (size_t)(splits - 1) * (elems + cols) * sizeof(float);
}

size_t mmbert_gemm_tn_workspace(int M, int N, int K, int* splits_out) {
    return mmbert_gemm_tn_grouped_workspace(1, &N, &K, M, splits_out);
}

// up to TN_MAXP (52) problems:  W_i[N_i,K_i] (+)= alpha * A_i^T . B_i over M_i token rows;  bias_i[N_i] += alpha * colsum(A_i) (bias_i may be null).
// The problems of M_0 rows ("long": M_i == M_0) come first; problems of FEWER rows may follow (round 6: the tied-decoder, MLM-transform and
// top-layer weight gradients of the few hundred rows that carry a loss).  With more long tiles than CUs -- the deferred multi-layer call --
// the long tiles go out in whole rounds of CUs-many, and the few-row tiles ride behind the LAST round's long tiles in the same launch: its
// idle CUs (65 of 256 at the headline shape) work through them while the long tiles run, where they took four launches of their own on the
// serial tail of backward.  The token axis is never split when few-row problems are present.
int mmbert_gemm_tn_grouped_rows(hipStream_t stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* W, float* const* bias, const int* N, const int* K, const int* Mrows,
                                int accumulate, const int* accumulate_each, float alpha, const float* alpha_dev, void* slab) {
    if (nprob <= 0) return 0;
    if (nprob > TN_MAXP) return -1;
    const int M = Mrows[0];
    if (M <= 0) return 0;
    int nlong = 0;
    while (nlong < nprob && Mrows[nlong] == M) ++nlong;
    for (int i = nlong; i < nprob; ++i)
        if (Mrows[i] >= M || Mrows[i] <= 0) return -1;             // long problems first, then the shorter ones
    GemmTNG g;
    int splits, tiles_long, tiles_all;
    tn_plan(nlong, N, K, M, &splits, &tiles_long);
    if (nlong < nprob) splits = 1;
    long long off = 0, boff = 0;
    int tile0 = 0;
    TNReduce r;
    for (int i = 0; i < nprob; ++i) {
        if ((N[i] & 7) || (K[i] & 7) || (lda[i] & 7) || (ldb[i] & 7) || N[i] < 8 || K[i] < 8) return -1;
        TNProb& q = g.pr[i];
        q.A = (const bf16_t*)A[i]; q.B = (const bf16_t*)B[i]; q.W = W[i]; q.bias = bias ? bias[i] : nullptr;
        q.N = N[i]; q.K = K[i]; q.lda = lda[i]; q.ldb = ldb[i]; q.M = Mrows[i]; q.acc = accumulate_each ? accumulate_each[i] : accumulate;
        q.tiles_k = (K[i] + 255) / 256; q.tile0 = tile0; q.slab_off = (int)off; q.bias_off = (int)boff;
        r.W[i] = W[i]; r.off[i] = off; r.bias[i] = q.bias; r.boff[i] = boff;
        boff += N[i];
        tile0 += ((N[i] + 255) / 256) * q.tiles_k;
        off += (long long)N[i] * K[i];
    }
    tiles_all = tile0;
    if (splits > 1 && off > 0x7fffffffLL) return -1;               // (slab offsets are ints; an unsplit launch does not read them)
    for (int i = nprob; i < TN_MAXP; ++i) { g.pr[i] = g.pr[0]; g.pr[i].tile0 = 0x7fffffff; r.W[i] = nullptr; r.bias[i] = nullptr; }
    for (int i = nprob; i <= TN_MAXP; ++i) { r.off[i] = off; r.boff[i] = boff; }
    if (splits > 1 && !slab) return -3;
    g.slab = (float*)slab; g.alpha_dev = alpha_dev; g.slab_stride = off;
    // deterministic mode, token axis split: the bias sums of splits >= 1 go through the tail of the slab
    const bool det_bias = splits > 1 && mmb_deterministic();
    g.bias_slab = det_bias ? (float*)slab + (size_t)(splits - 1) * off : nullptr;
    g.bias_stride = boff;
    r.bias_slab = g.bias_slab; r.bias_stride = boff; g.nprob = nprob; g.total_tiles = tiles_all; g.M = M;
    g.splits = splits; g.rows_per_split = (((M + splits - 1) / splits) + 31) / 32 * 32; g.accumulate = accumulate; g.alpha = alpha;
    static std::atomic<unsigned long long> attr_done8{0};
    if (int e = mmb_allow_lds((const void*)gemm_tn8_kernel, 131072, attr_done8)) return e;
    auto kern = gemm_tn8_kernel;
    const int cus_ = device_cus();
    const int tiles_short = tiles_all - tiles_long;
    if (splits == 1 && (tiles_long > cus_ || tiles_short > 0)) {
        // whole rounds of CUs-many long tiles, one launch per round (only the deferred multi-layer launches have more tiles than CUs; never
        // split); the few-row tiles behind the last round's
        const int step_ = g_tn_one_launch.load() ? (tiles_long > 0 ? tiles_long : 1) : cus_;
        for (int base = 0; base < tiles_long; base += step_) {
            const int n = tiles_long - base < step_ ? tiles_long - base : step_;
            const bool last = base + step_ >= tiles_long;
            g.tile_base = base; g.remap_n = n;
            hipLaunchKernelGGL(kern, dim3(n + (last ? tiles_short : 0), 1), dim3(512), 131072, stream, g);
            MMB_CHECK_LAUNCH();
        }
    } else {
        g.tile_base = 0; g.remap_n = tiles_long;
        hipLaunchKernelGGL(kern, dim3(tiles_long, splits), dim3(512), 131072, stream, g);
    }
    MMB_CHECK_LAUNCH();
    if (splits > 1) {
        r.nprob = nprob; r.splits = splits; r.accumulate = accumulate; r.slab_stride = off;
        const long long total4 = off / 4;
        const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, r, (const float*)slab);
        MMB_CHECK_LAUNCH();
    }
    return 0;
}

// ... all problems over the same M token rows
int mmbert_gemm_tn_grouped(hipStream_t stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                           float* const* W, float* const* bias, const int* N, const int* K, int M,
                           int accumulate, float alpha, const float* alpha_dev, void* slab) {
    if (nprob <= 0 || M <= 0) return 0;
    if (nprob > TN_MAXP) return -1;
    int Ms[TN_MAXP];
    for (int i = 0; i < nprob; ++i) Ms[i] = M;
    return mmbert_gemm_tn_grouped_rows(stream, nprob, A, lda, B, ldb, W, bias, N, K, Ms, accumulate, nullptr, alpha, alpha_dev, slab);
}

int mmbert_gemm_tn(hipStream_t stream, const void* A, int lda, const void* B, int ldb, float* W, int ldw,
                   int M, int N, int K, int accumulate, float alpha, const float* alpha_dev, void* slab, float* bias_out) {
    if (ldw != K) return -1;                 // weight-gradient tensors are contiguous
    return mmbert_gemm_tn_grouped(stream, 1, &A, &lda, &B, &ldb, &W, &bias_out, &N, &K, M, accumulate, alpha, alpha_dev, slab);
}

int mmbert_colsum(hipStream_t stream, const void* X, int ldx, int M, int N, float* out, float alpha, const float* alpha_dev) {
    if (M <= 0 || N <= 0) return 0;
    if ((N & 7) || (ldx & 7)) return -1;
    const int gx = (N / 8 + 63) / 64;
    int gy = (2048 + gx - 1) / gx;
    int rows = (M + gy - 1) / gy; if (rows < 32) rows = 32;
    if (mmb_deterministic()) rows = M;                            // one workgroup per column block: a single adder per address
    gy = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16_t*)X, M, N, ldx, out, alpha, alpha_dev, rows);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
