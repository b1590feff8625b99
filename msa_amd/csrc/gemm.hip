// bf16 MFMA GEMMs for the MMBert encoder stack on gfx950 (MI355X).
//
//   gemm_nt : C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)       forward projections and dgrads
//             (A and B both K-contiguous: activations x PyTorch Linear weights [out,in], or
//              gradients x the pre-transposed bf16 weight copy).
//   gemm_tn : W[N,K] (+)= A[M,N]^T . B[M,K]  in fp32          weight gradients (reduction over tokens)
//
// Structure (both): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as 4x4
// v_mfma_f32_16x16x32_bf16 tiles), BK=64, operands staged global->LDS with 16-byte
// global_load_lds into two LDS buffers (64 KiB -> 2 workgroups per CU), one barrier per K tile,
// XOR swizzle applied on the per-lane SOURCE address and on the LDS read (LDS-DMA writes are
// lane-linear).  Tile ids are remapped so that each XCD (private L2) works on a contiguous band.
//
// NT reads fragments with ds_read_b128 (rows are K-contiguous).  TN needs 8 consecutive m for one
// column, i.e. a column of the row-major LDS tile: ds_read_b64_tr_b16 (hardware transpose read).
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
    int* tile_counter; int* tile_counter_next;    // gemm_ntp_kernel only: dynamic tile queue = {8 fetch counters (one per XCD), exit counter} (null = static b, b+G, ...)
    int queue_xcd;                                // 1: a workgroup draws from its XCD's counter (tile order stays v = x mod 8: the XCD's L2 keeps its panels); 0: one counter
    int group_m;                                  // gemm_ntp_kernel only: tile walk in groups of group_m row tiles (<= 1: row-major), see ntp_tile_mn
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "each XCD gets a contiguous chunk" remap (blocks b and b+8 share an XCD)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// Tile walk of the persistent kernel.  Linear tile index t (the XCD-contiguous order of xcd_remap) -> (row tile, column tile).
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

// -------------------------------------------------------------------------------------------------
// NT, large shapes: 256x256 output tile, 512 threads (8 waves as 2(M) x 4(N), 128x64 per wave =
// 8x4 MFMA tiles, 128 accumulator VGPRs), K consumed in 32-deep stages through a 4-slot LDS ring
// (4 x 32 KiB = 128 KiB, one workgroup per CU).  Per stage and wave: 4 global_load_lds (16 B),
// 12 ds_read_b128, 32 MFMAs.  The LDS-DMA of stages s+2..s+4 stays in flight across the (single,
// raw) barrier of stage s behind a COUNTED s_waitcnt vmcnt(8); fragments of stage s+1 are read into
// a second register set while the MFMAs of stage s issue.  LDS rows are 64 B (4 chunks of 16 B);
// chunk' = chunk ^ G[(row>>2)&3], G = {0,2,3,1}, makes every ds_read_b128 lane group hit 16
// distinct 16-byte slots (the groups mix rows {0-3,12-15} of one chunk with rows {4-11} of chunk^1).
// Loads past the last stage re-read the last stage (clamped) into a dead slot, so the in-flight
// count is the same in every iteration and the waits need no tail variants.
// Epilogue: accumulators -> LDS (fp32, wave-private 16 KiB, 64 rows at a time) -> 8 consecutive columns
// per lane -> fused math -> 16-byte global accesses (8 full 128-B lines per store instruction; the direct
// 8-byte fragment-shaped stores were store-issue bound: 27 us vs 19 us per tile round).
// Tried and rejected (round 1, same-process A/B, tools/bench_gemm.py): a 256x128-tile / 256-thread /
// 3-slot-ring variant with two workgroups per CU (to overlap one's epilogue with the other's K loop):
// deep-K shapes fell from ~1080 to ~850 TF/s and the K=768 shapes did not improve.
// -------------------------------------------------------------------------------------------------
// Diagnostic build only (-DMMB_STAMPS, tools/stamp_gemm.py): s_memtime stamps at the phase boundaries of the ring
// kernel, kept in SGPRs and stored once at the end to a buffer nothing else reads.  No stamp exists in the product build.
#ifdef MMB_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
// timing-only experiments (bit mask; outputs are wrong by construction): 1 zero-record descriptors (LDS-DMA loads dropped by
// the range check: instruction stream and waits stay, memory traffic goes), 32 whole-cache-line source pattern (8 rows x 128 B
// per load instruction instead of 16 x 64 B).  Round-1 findings with these and with per-part ablations of the K step (MFMAs,
// fragment reads, LDS-DMA issue switched off one at a time): the step costs ~1180-1300 clk whether or not the MFMAs run and
// whether or not the loads touch memory -- ISSUING 32 sixteen-byte-per-lane vector loads per CU and step is the floor
// (~37 clk per wave instruction, LDS-DMA and register loads alike, = 27-32 B/clk/CU), just above the 1024 clk of the MFMAs.
__device__ int g_nt_dbg = 0;
#define MMB_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define MMB_STAMP(var)
#endif

// MI = 16-row MFMA blocks per wave along M: 8 (256-row tile) or 7 (224-row tile).  The 224-row variant exists for
// tile-round quantisation only: at M = 18400 the 256-row tiling leaves every launch at 84 % of a whole number of
// rounds over 256 CUs (216 / 648 / 864 tiles), the 224-row tiling at 97 % (249 / 747 / 996).  Staging is identical
// (256 A rows are loaded; rows past the tile are never read), only the MFMA count and the epilogue shrink.
template <int EPI, int MI>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(const GemmNT p) {
    constexpr int BM = 32 * MI;
#ifdef MMB_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, rt0 = 0, rt3 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
    MMB_STAMP(st0)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 slots x [A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) << 8;
    const int ns = p.K >> 5;
    constexpr unsigned GT = 0x78;                                  // G = {0,2,3,1} packed 2 bits each: 0b01111000

    // staging: a wave-instruction writes 16 rows x 64 B; wave w owns rows [32w, 32w+32) of A and of B
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((GT >> (2 * ((srow >> 2) & 3))) & 3);
    const bf16_t* a_src[2]; const bf16_t* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 32 + i * 16 + srow;
        a_src[i] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + schunk * 8;
        b_src[i] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + schunk * 8;
    }
    auto stage = [&](int slot, int st) {
        const int ko = min(st, ns - 1) * 32;
        char* base = smem + slot * 32768 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(GPTR(a_src[i] + ko), LPTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GPTR(b_src[i] + ko), LPTR(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // lane-constant part of every fragment address: row fr of a 16-row block, swizzled chunk fq
    const int lane_off = fr * 64 + ((fq ^ ((GT >> (2 * ((fr >> 2) & 3))) & 3)) << 4);
    const char* a_rd = smem + (wr * MI * 16) * 64 + lane_off;
    const char* b_rd = smem + 16384 + (wc * 64) * 64 + lane_off;
    auto load_frags = [&](int slot, bf16x8 (&af)[MI], bf16x8 (&bfr)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(b_rd + slot * 32768 + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(a_rd + slot * 32768 + i * 1024);
    };
    auto mma = [&](const bf16x8 (&af)[MI], const bf16x8 (&bfr)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };

    bf16x8 a0[MI], b0[4], a1[MI], b1[4];
    stage(0, 0); stage(1, 1); stage(2, 2); stage(3, 3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MMB_STAMP(st1)
    load_frags(0, a0, b0);

#define NT256_STEP(SLOT, CUR_A, CUR_B, NXT_A, NXT_B)                                  \
    {                                                                                 \
        /* stage s+1 landed (my loads); my ds_reads of the slot about to be restaged are complete */ \
        __builtin_amdgcn_s_waitcnt(0x0078);   /* vmcnt(8) lgkmcnt(0): a builtin, so hipcc's own scoreboard sees it */ \
        __builtin_amdgcn_s_barrier();                      /* ... and everyone's; slot of stage s is free */ \
        stage(SLOT, s + 4);                                                           \
        load_frags((SLOT + 1) & 3, NXT_A, NXT_B);                                     \
        __builtin_amdgcn_s_setprio(1);                                                \
        mma(CUR_A, CUR_B);                                                            \
        __builtin_amdgcn_s_setprio(0);                                                \
        ++s;                                                                          \
    }
    int s = 0;
    while (true) {
        NT256_STEP(0, a0, b0, a1, b1) if (s >= ns) break;
        NT256_STEP(1, a1, b1, a0, b0) if (s >= ns) break;
        NT256_STEP(2, a0, b0, a1, b1) if (s >= ns) break;
        NT256_STEP(3, a1, b1, a0, b0) if (s >= ns) break;
    }
#undef NT256_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the dummy tail stages before LDS is reused
    __builtin_amdgcn_s_barrier();                            // every wave is done with the ring
    MMB_STAMP(st2)

    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);
    char* wl = smem + wave * 16384;                          // [64 rows][64 fp32] = 256-B rows, 16-B chunks XOR (row & 15)
    const int er = lane >> 3, ec = lane & 7;                 // read-back: 8 rows per pass, lane owns columns 8*ec .. 8*ec+7
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int nblk = half == 0 ? 4 : MI - 4;             // 16-row blocks in this half (compile-time after unrolling)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nblk) continue;
            const int row = i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = acc[half * 4 + i][j];
                *(float4*)(wl + row * 256 + (((j * 4 + fq) ^ (row & 15)) << 4)) = make_float4(a4[0] * alpha, a4[1] * alpha, a4[2] * alpha, a4[3] * alpha);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: no barrier needed
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it >= 2 * nblk) continue;
            const int row = it * 8 + er;
            const float4 lo = *(const float4*)(wl + row * 256 + (((2 * ec) ^ (row & 15)) << 4));
            const float4 hi = *(const float4*)(wl + row * 256 + (((2 * ec + 1) ^ (row & 15)) << 4));
            const int m = m0 + wr * (MI * 16) + half * 64 + row;
            const int n = n0 + wc * 64 + ec * 8;
            if (m < p.M && n < p.N) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                epi_store8<EPI>(p, m, n, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next half overwrites the region
    }
#ifdef MMB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // stores acknowledged
    MMB_STAMP(st3)
    if (g_stamps && lane == 0) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt3) :: "memory");
        unsigned long long* o = g_stamps + ((size_t)blockIdx.x * 8 + wave) * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = rt0; o[5] = rt3;
    }
#endif
}

// Test / A-B knobs (mmbert_gemm_nt_force, mmbert_gemm_tn_force_splits): process-global, relaxed atomics, default 0 = "by shape".
// They select between kernels that compute the same product; nothing else in the library keeps state between calls.
static std::atomic<int> g_nt_force{0};   // 0 auto, 1 force 128^2, 2 force the 256-wide kernels
static std::atomic<int> g_nt_bm{0};      // 0 auto, 256 / 224 forced
static std::atomic<int> g_nt_persist{1}; // persistent stream kernel where eligible (0: launch-per-tile ring kernel)
static int device_cus() { return mmb_device_cus(); }    // (common.h: cached per device)

// -------------------------------------------------------------------------------------------------
// NT, persistent form of the ring kernel (the default for large shapes with K % 128 == 0).
// Stamps of the launch-per-tile kernel above (tools/stamp_gemm.py, M = 18400) showed a workgroup outside its
// K loop for 26-49 % of its life at K = 768: ~5 k clk from start to the first stage landing, 5-25 k clk of
// epilogue with nothing in flight.  Here one workgroup per CU walks tiles v = b, b + G, b + 2G, ...; the
// stages of ALL its tiles form one stream through the 4-slot ring (stream stage g sits in slot g & 3, is
// issued at step g - 4), so the first four stages of the next tile are issued by the last four K steps of the
// current one and land under its epilogue.
// Epilogue: straight from the accumulators.  The MFMA runs with the operands swapped (B fragment first), so a lane holds
// 4 consecutive COLUMNS of one output row per 16x16 block, and the B fragment rows are read in a permuted order (b_rd) so
// that the four column blocks of a wave give each lane 2 x 8 consecutive columns: 16-byte stores, 64 B contiguous per row
// and instruction, no LDS transposition (the round-1 form went through a fifth 32 KiB LDS region: +2 k clk per tile).
// What the epilogue costs is vector-memory ISSUE, like the K loop: ~47 clk per 1 KiB store instruction per CU (stores
// switched off in the stamped build: vocabulary GEMM 1083 -> 886 us), plus, for GELU / GELU' / dropout, VALU work
// (common.h: single-transcendental GELU forms).  Anything the epilogue LOADS into registers returns only after the next
// tile's 16 stage loads issued ahead of it (in-order retirement, ~5 k clk): the bias row therefore arrives by a small
// LDS-DMA issued in K step 3 and is read with DS instructions; scale, bias and the dropout decision are applied in place
// to all accumulators (phase 1) before the first residual / GELU-input row is consumed (phase 2).
// vmcnt bookkeeping across a tile boundary: vector memory operations retire in issue order, and the
// epilogue's loads and stores are issued BEHIND the next tile's stages 0-3.  Waiting for stage 0 therefore
// allows 12 + E outstanding operations and the first three K steps allow 8 + E, E = a LOWER bound of the
// epilogue's operation count (its stores; interior tiles issue them unconditionally, edge tiles use E = 0):
// a bound that is too low only waits longer.  Steady state is the ring kernel's vmcnt(8).
// -------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int mmb_waitcnt(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | 0x70 | (lgkm << 8); }

template <int OFF>
__device__ __forceinline__ void lds_read16f(f32x4& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF) : "memory");
}

// STAG selects the K-step schedule:
//  false: one barrier per step, fragments of step s+1 read into a second register set under the MFMAs of step s (96 fragment
//         VGPRs; with 8 row blocks per wave that spills, so this form runs the 224-row tile);
//  true : two wave groups (the two M halves: one wave of each per SIMD) half a step apart -- the second group takes one extra
//         barrier up front, so its k-th barrier meets the first group's (k+1)-th -- and a step = two phases of <= 16 MFMAs,
//         phase = { LDS reads of its operands (phase b: + wait for this wave's part of stage s+1, issue stage s+3) ; barrier ;
//         MFMAs ; barrier }: one group's MFMAs run under the other's reads, fragments are single-buffered (32 VGPRs) and the
//         256-row tile fits.  Measured per step: 1286 clk (false, 224 rows) vs ~1300 (true, 256 rows = 14 % more work).
//         RAW: a wave waits for ITS part of stage s+1 in phase b of step s, before that phase's first barrier; every reader
//              of stage s+1 (phase a of step s+1, either group) has passed a barrier that pairs with or follows it.
//         WAR: stage s+3 goes into the slot of stage s-1, whose last reads (phase b of step s-1) completed before their
//              MFMAs, i.e. before that phase's second barrier in BOTH groups; phase b of step s lies behind it for both.
// (The timing-only ablations of this K step -- MFMAs / fragment reads / A or B loads switched off at compile time, EXEC = 0 loads, a
// wave-uniform run-time branch around the A loads: DESIGN 3.1, profiles/r3_stamp_nt_ablation.log -- live on the git tag
// r3-gemm-ablations, not in the product source; common.h refuses their -D switches.)
template <int EPI, int MI, bool STAG>
__global__ __launch_bounds__(512, 2) void gemm_ntp_kernel(const GemmNT p) {
#if __HIP_DEVICE_COMPILE__   // the host pass only needs the launch stub (the body uses device-only builtins)
    constexpr int BM = 32 * MI;
    constexpr int EST = MI * ((EPI & EPI_OUT_F32) ? 4 : 2) * ((EPI & EPI_GELU) ? 2 : 1);   // stores per wave per interior tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 slots x [A 16K | B 16K] | tile-queue word | 8 x 256 B bias rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int ns = p.K >> 5;                                       // multiple of 4 (dispatch)
    constexpr unsigned GT = 0x78;

    // ---- load cursor: runs 4 stages ahead of the MFMAs, crosses tile boundaries ----
    struct Src { uint32_t a[2], b[2]; };
    auto set_src = [&](int v, Src& o) {
        int l = lane;
        asm volatile("" : "+v"(l));                               // recomputed per call, nothing kept alive across the K loop
        const int srow = l >> 2;
        const uint32_t schunk = (l & 3) ^ ((GT >> (2 * ((srow >> 2) & 3))) & 3);
        int tmi, tni;
        ntp_tile_mn(xcd_remap(v, ntiles), tiles_m, tiles_n, p.group_m, tmi, tni);
        const int tm0 = tmi * BM, tn0 = tni << 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 32 + i * 16 + srow;
            // B rows are read in the permuted order of b_rd below (a fragment's 16 lanes read rows 8(fr>>2) + (fr&3) + const),
            // so their chunk swizzle is keyed on row bits 3-4 instead of 2-3
            const uint32_t schunk_b = (l & 3) ^ ((GT >> (2 * ((2 * i + (srow >> 3)) & 3))) & 3);
            o.a[i] = ((uint32_t)min(tm0 + r, p.M - 1) * (uint32_t)p.lda + schunk * 8u) * 2u;
            o.b[i] = ((uint32_t)min(tn0 + r, p.N - 1) * (uint32_t)p.ldb + schunk_b * 8u) * 2u;
#ifdef MMB_STAMPS
            if (g_nt_dbg & 32) {
                const int r8 = wave * 32 + i * 8 + (l >> 3);
                o.a[i] = ((uint32_t)min(tm0 + r8, p.M - 1) * (uint32_t)p.lda + (l & 7) * 8u) * 2u;
                o.b[i] = ((uint32_t)min(tn0 + r8, p.N - 1) * (uint32_t)p.ldb + (l & 7) * 8u) * 2u;
            }
#endif
        }
    };
    Src cur, nxt;
    set_src(blockIdx.x, cur);
    nxt = cur;
    // buffer addressing: descriptor (SGPRs) + 32-bit per-lane offset + scalar K offset -- no 64-bit VALU address
    // arithmetic and no hoisted 64-bit per-lane pointers (the global_load_lds form spilled registers here)
#ifdef MMB_STAMPS
    const int dbg_bits = __builtin_amdgcn_readfirstlane(g_nt_dbg);
    const uint32_t recA = (dbg_bits & 1) ? 0u : (uint32_t)p.M * (uint32_t)p.lda * 2u, recB = (dbg_bits & 1) ? 0u : (uint32_t)p.N * (uint32_t)p.ldb * 2u;
#else
    const uint32_t recA = (uint32_t)p.M * (uint32_t)p.lda * 2u, recB = (uint32_t)p.N * (uint32_t)p.ldb * 2u;
#endif
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)recA, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)recB, 0x00020000);
    auto issue = [&](int slot, const Src& o, uint32_t kb) {      // kb: byte offset along K (wave-uniform)
        char* base = smem + slot * 32768 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base + i * 1024), 16, o.a[i], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base + 16384 + i * 1024), 16, o.b[i], kb, 0, 0);
        }
    };
    // s_waitcnt vmcnt(k stages x loads per stage and wave + e)
    constexpr int NTP_LOADS_PER_STAGE = 4;                         // issue(): 2 A-row + 2 B-row LDS-DMA instructions per wave and stage
#define NTP_WAIT(K_, E_, LG_) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(NTP_LOADS_PER_STAGE * (K_) + (E_), LG_));

    const int fr = lane & 15, fq = lane >> 4;
    const int lane_off = fr * 64 + ((fq ^ ((GT >> (2 * ((fr >> 2) & 3))) & 3)) << 4);
    // 32-bit LDS addresses; ds offsets are 16-bit, so slots 2 and 3 read through a second pair of bases (opaque, or
    // hipcc materialises one address register per fragment read of those slots and keeps them all alive)
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
    const lds_cptr a_rd = (lds_cptr)LPTR(smem) + (wr * MI * 16) * 64 + lane_off;
    // Output columns are assigned to MFMA rows so that a lane ends up with 2 x 8 CONSECUTIVE columns of one output row and the
    // epilogue stores straight from the accumulators (no LDS transposition): MFMA row i of column block j (acc[.][j], this lane
    // holds i = 4 fq .. 4 fq + 3) is column 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) + (i & 3) of the wave's 64, i.e. the lane holds
    // columns 8 fq .. 8 fq + 7 in acc[.][0..1] and 32 + 8 fq .. + 7 in acc[.][2..3].  Only the B fragment rows change:
    const int lane_off_b = (8 * (fr >> 2) + (fr & 3)) * 64 + ((fq ^ ((GT >> (2 * (fr >> 2))) & 3)) << 4);
    const lds_cptr b_rd = (lds_cptr)LPTR(smem) + 16384 + (wc * 64) * 64 + lane_off_b;
#define NTP_BOFF(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 64)
    lds_cptr a_rd_hi = a_rd + 65536, b_rd_hi = b_rd + 65536;
    asm volatile("" : "+v"(a_rd_hi), "+v"(b_rd_hi));
    auto load_frags = [&](int slot, bf16x8 (&af)[MI], bf16x8 (&bfr)[4]) {
        const lds_cptr ab = slot < 2 ? a_rd : a_rd_hi;
        const lds_cptr bb = slot < 2 ? b_rd : b_rd_hi;
        const int so = (slot & 1) * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(lds_frag)(bb + so + NTP_BOFF(j));
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(lds_frag)(ab + so + i * 1024);
    };
    f32x4 acc[MI][4];
    auto mma = [&](const bf16x8 (&af)[MI], const bf16x8 (&bfr)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };
    auto mma_first = [&](const bf16x8 (&af)[MI], const bf16x8 (&bfr)[4]) {     // C = 0: no accumulator clearing pass
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };

    bf16x8 sf_a[4], sf_b[4];                                       // STAG: single-buffered fragments
    auto load_b1 = [&](int slot) {
        const lds_cptr bb = slot < 2 ? b_rd : b_rd_hi;
        const int so = (slot & 1) * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) sf_b[j] = *(lds_frag)(bb + so + NTP_BOFF(j));
    };
    auto load_a1 = [&](int slot, int i0, int n) {                  // row blocks i0 .. i0 + n - 1 -> sf_a[0 .. n - 1]
        const lds_cptr ab = slot < 2 ? a_rd : a_rd_hi;
        const int so = (slot & 1) * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) sf_a[i] = *(lds_frag)(ab + so + (i0 + i) * 1024);
    };

    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);

    int first_fetch = 0;
    // Queue: per XCD (workgroups are dealt round-robin: XCD = blockIdx & 7; G is a multiple of 8 whenever the queue is on), the k-th
    // draw of XCD x is tile number G + 8 k + x -- the same residue class the static walk gives that XCD, so xcd_remap()'s contiguous
    // chunk per XCD (and the grouped walk inside it) holds and only the order WITHIN an XCD's 32 CUs is dynamic.
    const int qx = p.queue_xcd ? (int)(blockIdx.x & 7) : 0, qs = p.queue_xcd ? 8 : 1;
    // The fetch is an inline-asm atomic whose result is read only behind a COUNTED wait: hipcc's own atomicAdd goes through the
    // wave-aggregation pass (s_bcnt + v_readfirstlane) and needs its result at once -- an s_waitcnt vmcnt(0) right behind the atomic,
    // i.e. a drain of the stage loads in flight at the start of every epilogue (the dynamic queue cost 1.8 % of the step that way).
    auto queue_fetch = [](int* counter) {
        int r;
        // (s_nop 4: hipcc may hand the counter's address over in SGPRs it has just restored with v_readlane -- a VALU write of an SGPR needs 5 wait
        // states before a vector-memory instruction reads it, and the hazard recognizer does not look inside inline asm: without the nops
        // one build of the GELU' instantiation drew from a stale address and faulted)
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(r) : "v"(0), "v"(1), "s"(counter) : "memory");
        return r;
    };
    if (p.tile_counter && tid == 0)
        first_fetch = queue_fetch(p.tile_counter + qx);                  // issued ahead of the prologue loads; read behind them (below)
    bf16x8 a0[MI], b0[4], a1[MI], b1[4];
    if constexpr (STAG) {
        issue(0, cur, 0); issue(1, cur, 64); issue(2, cur, 128);
        __builtin_amdgcn_s_waitcnt(mmb_waitcnt(8, 15));            // this wave's part of stage 0
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();                 // the stagger; balanced by the other group at the end
    } else {
        issue(0, cur, 0); issue(1, cur, 64); issue(2, cur, 128); issue(3, cur, 192);
    }
    bool early = false;          // the previous tile's epilogue stores sit behind this tile's first stages in vmcnt order
#ifdef MMB_STAMPS
    unsigned long long sa = 0, sb = 0, sc_ = 0, sd = 0, t_wait = 0, t_loop = 0, t_epi = 0, rt0 = 0, rt3 = 0, ntile = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
#endif

#define NTP_STEP(SLOT, CUR_A, CUR_B, NXT_A, NXT_B, MMA, WEARLY, LOADNEXT, SRC, KB)        \
    {                                                                                     \
        if (WEARLY) NTP_WAIT(2, EST, 0)                                                   \
        else NTP_WAIT(2, 0, 0)                                                            \
        __builtin_amdgcn_s_barrier();                                                     \
        issue(SLOT, SRC, KB);                                                             \
        if (LOADNEXT) load_frags((SLOT + 1) & 3, NXT_A, NXT_B);                           \
        __builtin_amdgcn_s_setprio(1);                                                    \
        MMA(CUR_A, CUR_B);                                                                \
        __builtin_amdgcn_s_setprio(0);                                                    \
    }

#define NT3_STEP(SLOT, FIRST, WEARLY, SRC, KB)                                                \
    {                                                                                         \
        load_b1(SLOT);                                                                        \
        load_a1(SLOT, 0, 4);                                                                  \
        __builtin_amdgcn_s_barrier();                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        __builtin_amdgcn_s_setprio(1);                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf_b[j], sf_a[i], (FIRST) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        load_a1(SLOT, 4, MI - 4);                                                             \
        if (WEARLY) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(4 + EST, 15));                     \
        else __builtin_amdgcn_s_waitcnt(mmb_waitcnt(4, 15));                                  \
        issue((SLOT + 3) & 3, SRC, KB);                                                       \
        __builtin_amdgcn_s_barrier();                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        __builtin_amdgcn_s_setprio(1);                                                        \
        _Pragma("unroll") for (int i = 4; i < MI; ++i)                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf_b[j], sf_a[i - 4], (FIRST) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }

    // Tile queue.  A workgroup's first tile is its block index; further tiles come from a device counter (G + fetch-and-add)
    // when the launch has more tiles than workgroups -- so that workgroups which start late (CUs held by another stream's
    // kernels, e.g. RCCL channels during the gradient all-reduce) do not leave their whole static share for a second wave.
    // One lane fetches the tile AFTER the next one at the start of an epilogue (at kernel start for the second tile), when
    // registers are free and the latency has the whole epilogue to hide in, and parks it in an LDS word behind
    // the ring; every wave picks it up after K step 3 of the next tile, several barriers later.
    auto vq_write = [&](int value) {
        const uint32_t vq_addr = (uint32_t)(size_t)LPTR(smem) + 131072u;
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(vq_addr), "v"(value) : "memory");
    };
    // Bias row of a wave's 64 columns: one 4-byte-per-lane LDS-DMA per tile into a wave-private 256 B of LDS, issued in K
    // step 3 -- ahead of the next tile's stages in vmcnt order, so the epilogue reads it (DS, lgkmcnt) without having to
    // wait for those stages the way a register load issued in the epilogue would (vector memory retires in issue order)
    auto fetch_bias = [&](int n0) {
        if constexpr (EPI & EPI_BIAS) {
            int l = lane;
            asm volatile("" : "+v"(l));
            auto kpb = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kpb));
            const __attribute__((address_space(4))) GemmNT& qb = *(const __attribute__((address_space(4))) GemmNT*)kpb;
            const auto rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)qb.bias, 0, qb.N * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LPTR(smem + 132096 + wave * 256), 4, l * 4, (n0 + wc * 64) * 4, 0, 0);
        }
    };
    if (p.tile_counter && tid == 0) {
        // the atomic is older than the prologue's stage loads (16 per wave, 12 in the staggered form): done once no more than those are outstanding
        __builtin_amdgcn_s_waitcnt(mmb_waitcnt(STAG ? 12 : 16, 15));
        asm volatile("" : "+v"(first_fetch) :: "memory");          // (the register holds the counter only behind the wait: no use may move above it)
        vq_write(G + qs * first_fetch + qx);
    }
    for (int v = blockIdx.x, vn = 0; v < ntiles; v = vn) {
        int tmi, tni;
        ntp_tile_mn(xcd_remap(v, ntiles), tiles_m, tiles_n, p.group_m, tmi, tni);
        const int m0 = tmi * BM, n0 = tni << 8;

        // K steps 0-3 issue stages 4-7 of this tile (ns >= 8).  `early` only selects the wait immediate (a scalar branch
        // around one s_waitcnt); two full copies of the steps made hipcc spill accumulator tuples at the join
        const uint32_t kbytes = (uint32_t)p.K * 2u;
        auto read_queue = [&]() {
            if (p.tile_counter) {
                int q;
                const uint32_t vq_addr = (uint32_t)(size_t)LPTR(smem) + 131072u;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(vq_addr) : "memory");
                vn = __builtin_amdgcn_readfirstlane(q);
            } else {
                vn = v + G;
            }
        };
        if constexpr (STAG) {
            MMB_STAMP(sa)
            MMB_STAMP(sb)
            // K steps 0-3 bring in stages 3-6 of this tile (ns >= 8); the epilogue's stores sit behind the next tile's stages 0-2
            NT3_STEP(0, true, early, cur, 192)
            NT3_STEP(1, false, early, cur, 256)
            NT3_STEP(2, false, false, cur, 320)
            NT3_STEP(3, false, false, cur, 384)
            read_queue();
            fetch_bias(n0);
            for (int s = 4; s < ns - 4; s += 4) {
                const uint32_t kb = (uint32_t)(s + 3) * 64u;
                NT3_STEP(0, false, false, cur, kb)
                NT3_STEP(1, false, false, cur, kb + 64)
                NT3_STEP(2, false, false, cur, kb + 128)
                NT3_STEP(3, false, false, cur, kb + 192)
            }
            // the last four K steps bring in the last stage of this tile and stages 0-2 of the workgroup's next tile
            if (vn < ntiles) set_src(vn, nxt);                    // past the last tile: dead re-reads of the same stages
            NT3_STEP(0, false, false, cur, kbytes - 64)
            NT3_STEP(1, false, false, nxt, 0)
            NT3_STEP(2, false, false, nxt, 64)
            NT3_STEP(3, false, false, nxt, 128)
        } else {
            MMB_STAMP(sa)
            if (early) NTP_WAIT(3, EST, 0)
            else NTP_WAIT(3, 0, 0)
            __builtin_amdgcn_s_barrier();
            MMB_STAMP(sb)
            load_frags(0, a0, b0);
            NTP_STEP(0, a0, b0, a1, b1, mma_first, early, true, cur, 256)
            NTP_STEP(1, a1, b1, a0, b0, mma, early, true, cur, 320)
            NTP_STEP(2, a0, b0, a1, b1, mma, early, true, cur, 384)
            NTP_STEP(3, a1, b1, a0, b0, mma, false, true, cur, 448)
            read_queue();
            fetch_bias(n0);
            for (int s = 4; s < ns - 4; s += 4) {
                const uint32_t kb = (uint32_t)(s + 4) * 64u;
                NTP_STEP(0, a0, b0, a1, b1, mma, false, true, cur, kb)
                NTP_STEP(1, a1, b1, a0, b0, mma, false, true, cur, kb + 64)
                NTP_STEP(2, a0, b0, a1, b1, mma, false, true, cur, kb + 128)
                NTP_STEP(3, a1, b1, a0, b0, mma, false, true, cur, kb + 192)
            }
            // the last four K steps issue stages 0-3 of the workgroup's next tile
            if (vn < ntiles) set_src(vn, nxt);                        // past the last tile: dead re-reads of the same stages
            NTP_STEP(0, a0, b0, a1, b1, mma, false, true, nxt, 0)
            NTP_STEP(1, a1, b1, a0, b0, mma, false, true, nxt, 64)
            NTP_STEP(2, a0, b0, a1, b1, mma, false, true, nxt, 128)
            NTP_STEP(3, a1, b1, a0, b0, mma, false, false, nxt, 192)
        }
        cur = nxt;
        MMB_STAMP(sc_)

        // ---- epilogue of this tile; the next tile's stages 0-3 are in flight / landed in the ring ----
        // every per-lane epilogue address is derived from an opaque copy of the lane id, so that hipcc recomputes them
        // here (a dozen VALU operations) instead of keeping ~20 loop-invariant registers alive across the K loop: there
        // they spill, and a scratch reload is a vmcnt(0) drain of the stages in flight
        // ... and every epilogue parameter is re-read (s_load) from an opaque copy of the kernel-argument pointer: kept in
        // SGPRs across the K loop they overflow the scalar file into VGPR lanes, and from there into scratch
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const __attribute__((address_space(4))) GemmNT& q = *(const __attribute__((address_space(4))) GemmNT*)kp;
        int fetched = 0;
        const bool fetcher = q.tile_counter && tid == 0 && vn < ntiles;     // the tile after the next one (if there is a next one)
        if (fetcher) fetched = queue_fetch(q.tile_counter + (q.queue_xcd ? (int)(blockIdx.x & 7) : 0));   // raw counter value, read at the end of the epilogue
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int efr = elane & 15, efq = elane >> 4;
        const bool interior = (m0 + BM <= q.M) && (n0 + 256 <= q.N);
        const int mrow = m0 + wr * (MI * 16) + efr;                  // + 16 i
        const int ncol = n0 + wc * 64 + efq * 8;                     // 8 columns here (acc[.][0..1]) and 8 at + 32 (acc[.][2..3])
        // residual / GELU-input rows: prefetched PRE row blocks ahead of their use
        constexpr int PRE = 3;
        bf16x8 pre[MI][2];
        auto load_pre = [&](int i) {
            if constexpr (EPI & (EPI_RESID | EPI_GELU_BWD)) {
                const bf16_t* src = (EPI & EPI_RESID) ? q.R : q.U;
                const int ld = (EPI & EPI_RESID) ? q.ldr : q.ldu;
                const int m = mrow + 16 * i;
                // unconditional, clamped in-bounds (a half that is out of range is never stored): no branches, and
                // nothing conditionally defined that hipcc would carry around the tile loop
                const bf16_t* rp = src + (size_t)min(m, q.M - 1) * ld;
                pre[i][0] = *(const bf16x8*)(rp + min(ncol, q.N - 8));
                pre[i][1] = *(const bf16x8*)(rp + min(ncol + 32, q.N - 8));
            }
        };
#pragma unroll
        for (int i = 0; i < PRE; ++i) load_pre(i);
        float bias[16];
        if constexpr (EPI & EPI_BIAS) {
            const uint32_t baddr = (uint32_t)(size_t)LPTR(smem) + 132096u + wave * 256u + efq * 32u;
            f32x4 b4[4];
            lds_read16f<0>(b4[0], baddr); lds_read16f<16>(b4[1], baddr); lds_read16f<128>(b4[2], baddr); lds_read16f<144>(b4[3], baddr);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) :: "memory");
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[4 * c + r] = b4[c][r];
        }
        // phase 1, needs no global data (the first residual / GELU-input rows are still queued behind the next tile's stages):
        // scale, bias and the dropout decision, in place in the accumulators
        // (epilogue scalars are copied out of the kernel-argument segment ONCE: read through `q` inside a select, hipcc turns
        // every select into a divergent branch around an s_load)
        const uint32_t dthr = (EPI & EPI_RESID) ? q.drop_thr16 : 0u;
        const uint32_t dthr_s = dthr - 32768u;                       // the signed-compare form of mmb_keep16
        const float dscale = (EPI & EPI_RESID) ? q.drop_scale : 1.0f;
        // dropout seeds are linear in the element index (common.h): pair(m, n) = m * N/2 + n/2 (mod 2^32), so one multiply per
        // lane and wave-uniform increments per row block / column pair
        const uint32_t halfN = (uint32_t)q.N >> 1;
        const uint32_t seed0 = (EPI & EPI_RESID) ? ((uint32_t)mrow * halfN + ((uint32_t)ncol >> 1)) * MMB_WEYL + q.drop_stream : 0u;
        const uint32_t seed_row = 16u * halfN * MMB_WEYL;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float v = acc[i][2 * h + (r >> 2)][r & 3] * alpha;
                    if constexpr (EPI & EPI_BIAS) v += bias[8 * h + r];
                    acc[i][2 * h + (r >> 2)][r & 3] = v;
                }
                if constexpr (EPI & EPI_RESID) {
                    if (dthr) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {                // element pair k of the 8 columns: elements 2k, 2k + 1
                            const uint32_t hb = mmb_pair_mix(seed0 + (uint32_t)i * seed_row + (uint32_t)(16 * h + k) * MMB_WEYL);
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
        for (int i = 0; i < MI; ++i) {
            if (i + PRE < MI) load_pre(i + PRE);
            const int m = mrow + 16 * i;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = ncol + 32 * h;
                float vv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) vv[r] = acc[i][2 * h + (r >> 2)][r & 3];
#ifdef MMB_STAMPS
                const bool ok = (interior || (m < q.M && n + 8 <= q.N)) && !(g_nt_dbg & 64);     // 64: timing without the stores
#else
                const bool ok = interior || (m < q.M && n + 8 <= q.N);
#endif
                if constexpr (EPI & EPI_GELU) {
                    if (q.aux) {
                        bf16x8 u;
#pragma unroll
                        for (int r = 0; r < 8; ++r) u[r] = f2bf(vv[r]);
                        if (ok) *(bf16x8*)(q.aux + (size_t)m * q.ldaux + n) = u;
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] = gelu_erf(vv[r]);
                }
                if constexpr (EPI & EPI_GELU_BWD) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] *= gelu_erf_grad(bf2f(pre[i][h][r]));
                }
                if constexpr (EPI & EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] += bf2f(pre[i][h][r]);
                }
                // (round 4) The value to store is pinned in registers BEFORE the edge-tile predicate: hipcc otherwise sinks its computation --
                // the use of the prefetched residual / GELU-input row included -- into the predicated block, the prefetch loads stay
                // "pending" on the path around it, and the NEXT tile's first fragment reads (which reuse those registers) got an
                // s_waitcnt vmcnt(3..0): a drain of this epilogue's stores at every tile seam of the residual and GELU' epilogues
                // (rounds 1-3 shipped that: the GELU' input gradient, three rounds per launch, paid it twice per launch).
                if constexpr (EPI & EPI_OUT_F32) {
                    float* c = (float*)q.C + (size_t)m * q.ldc + n;
                    f32x4 lo = {vv[0], vv[1], vv[2], vv[3]}, hi = {vv[4], vv[5], vv[6], vv[7]};
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    if (ok) {
                        *(f32x4*)c = lo;
                        *(f32x4*)(c + 4) = hi;
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = f2bf(vv[r]);
                    u32x4 ow = __builtin_bit_cast(u32x4, o);
                    asm volatile("" : "+v"(ow));
                    if (ok) *(u32x4*)((bf16_t*)q.C + (size_t)m * q.ldc + n) = ow;
                }
            }
        }
        if (fetcher) {
            // the atomic is older than everything this epilogue issued: at least EST stores on an interior tile
            if (interior) __builtin_amdgcn_s_waitcnt(mmb_waitcnt(EST, 15));
            else __builtin_amdgcn_s_waitcnt(mmb_waitcnt(0, 15));
            asm volatile("" : "+v"(fetched) :: "memory");         // (as above)
            const int fx = q.queue_xcd ? (int)(blockIdx.x & 7) : 0;
            vq_write(G + (q.queue_xcd ? 8 : 1) * fetched + fx);
        }
#ifdef MMB_STAMPS
        MMB_STAMP(sd)
        t_wait += sb - sa; t_loop += sc_ - sb; t_epi += sd - sc_; ++ntile;
#endif
        early = interior && !((EPI & EPI_GELU) && !q.aux);
    }
#undef NTP_STEP
#undef NT3_STEP
#undef NTP_WAIT
#undef NTP_BOFF
    if constexpr (STAG) { if (wr == 0) __builtin_amdgcn_s_barrier(); }   // balances the stagger barrier of the other group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the dead tail stages
    {   // the last workgroup to leave hands the queue back zeroed (its fetches are complete: vmcnt(0) above) for the stream's next launch
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
#ifdef MMB_STAMPS
    if (g_stamps && lane == 0) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt3) :: "memory");
        unsigned long long* o = g_stamps + ((size_t)blockIdx.x * 8 + wave) * 6;
        o[0] = t_wait; o[1] = t_loop; o[2] = t_epi; o[3] = ntile; o[4] = rt0; o[5] = rt3;
    }
#endif
#endif
}


// -------------------------------------------------------------------------------------------------
// NT, single-round shapes (round 4): the 256 x 256 "8-phase" structure of the CDNA4 guide (cdna_hip_programming.md S5), one tile per
// workgroup.  VERDICT r3 asked for the guide's template as an in-tree yardstick (tools/yardstick/, profiles/r4_nt_yardstick.log): on cold
// random operands it beat the persistent kernel below by 11-23 % on every shape whose tiles fit the chip in ONE round -- the N = 768
// launches of the step: out-proj, FFN-down and all input gradients, 45 % of the family's time -- (and by 21-39 % on 4096^3 / 8192^3),
// tied it on the multi-round K = 768 shapes (QKV, FFN-up, GELU' dgrad) and lost 3 % on the vocabulary projection, whose 34 rounds the
// persistent kernel's cross-tile stream serves better.  What differs from the ring kernels and why it is faster in the K loop:
//   * BK = 64: an LDS-DMA wave instruction moves 8 rows x 128 B -- whole cache lines (the 32-deep stages' 16 x 64-B pieces cost the
//     texture path 25 % more per instruction) -- and a K tile of 64 has HALF the barriers per FLOP;
//   * half-tiles are QUADRANT operands (A-half h = rows {wr*128 + h*64 ..}, B-half h = columns {wc*64 + h*32 ..}): every wave reads
//     b0, a0 in phase 1, b1 in phase 2, a1 in phase 3, nothing in phase 4, so an LDS half-tile is free again one to two phases after
//     its phase and the LDS-DMA stream runs 3 half-tiles ahead behind ONE counted vmcnt(6) per K tile;
//   * two wave groups (wr = 0 / 1: one wave of each per SIMD) one barrier apart: one group's 16 MFMAs of a phase run under the other
//     group's fragment reads and LDS-DMA issue; fragments are single-buffered (64 VGPRs), all 256 rows fit (210 VGPRs, no spills).
// LDS image of a half-tile: [128 rows][64 k] bf16; 16-byte chunk c of row r sits at chunk c ^ key(r), key = (r >> 1) & 7 for A and
// ((r >> 1) & 1) | (((r >> 3) & 3) << 1) for B (B rows are read in the permuted order that gives a lane 8 consecutive output columns,
// as in gemm_ntp_kernel): every ds_read_b128 lane group hits 16 distinct 16-byte slots; swizzle on the per-lane SOURCE address.
// Epilogue: straight from the accumulators, the persistent kernel's lane map (2 x 8 consecutive columns per row block and lane).
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
    // gemm_ntp_kernel's protocol: per XCD the k-th draw of XCD x is tile G + 8 k + x (the static walk's residue class: the XCD's chunk of
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
        // Bias row of this wave's 64 columns: ONE 4-byte-per-lane LDS-DMA into a wave-private 256 B behind the ring (gemm_ntp_kernel's way),
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
        auto load_pre = [&](int i) {
            if constexpr (EPI & (EPI_RESID | EPI_GELU_BWD)) {
                const bf16_t* src = (EPI & EPI_RESID) ? q.R : q.U;
                const int ld = (EPI & EPI_RESID) ? q.ldr : q.ldu;
                const bf16_t* rp = src + (size_t)min(mrow + 16 * i, q.M - 1) * ld;   // clamped in range: a half that is out of range is never stored
                pre[i][0] = *(const bf16x8*)(rp + min(ncol, q.N - 8));
                pre[i][1] = *(const bf16x8*)(rp + min(ncol + 32, q.N - 8));
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
                            const uint32_t hb = mmb_pair_mix(seed0 + (uint32_t)i * seed_row + (uint32_t)(16 * h + k) * MMB_WEYL);
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
                        if (ok) *(bf16x8*)(q.aux + (size_t)m * q.ldaux + n) = u;
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] = gelu_erf(vv[r]);
                }
                if constexpr (EPI & EPI_GELU_BWD) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] *= gelu_erf_grad(bf2f(pre[i][h][r]));
                }
                if constexpr (EPI & EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) vv[r] += bf2f(pre[i][h][r]);
                }
                // The value to store is pinned in registers BEFORE the (edge-tile) predicate: hipcc otherwise sinks the whole computation
                // -- the use of the prefetched residual row included -- into the predicated block, the prefetch loads stay "pending" on
                // the path around it, and the NEXT tile's first fragment reads (which reuse those registers) get an s_waitcnt vmcnt(0):
                // a drain of the previous epilogue's stores at every tile seam.
                if constexpr (EPI & EPI_OUT_F32) {
                    float* c = (float*)q.C + (size_t)m * q.ldc + n;
                    f32x4 lo = {vv[0], vv[1], vv[2], vv[3]}, hi = {vv[4], vv[5], vv[6], vv[7]};
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    if (ok) {
                        *(f32x4*)c = lo;
                        *(f32x4*)(c + 4) = hi;
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = f2bf(vv[r]);
                    u32x4 ow = __builtin_bit_cast(u32x4, o);
                    asm volatile("" : "+v"(ow));
                    if (ok) *(u32x4*)((bf16_t*)q.C + (size_t)m * q.ldc + n) = ow;
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
    // the device tile queue: only the multi-tile 224-row form draws from it (data-parallel runs), one counter per XCD as in gemm_ntp_kernel
    const bool use_queue = p.tile_counter && tiles > workgroups && !(workgroups & 7);
    if (!use_queue) q.tile_counter = q.tile_counter_next = nullptr;
    q.queue_xcd = 1;
    if (bm == 128) return launch_nt8_form<EPI, false, 2>(s, q, workgroups);    // (128-row tiles: single-round launches only)
    if (bm == 192) {
        if (tiles > workgroups) return launch_nt8_form<EPI, true, 3>(s, q, workgroups);      // multi-tile form on 192-row tiles (where that height starts fewer row-rounds)
        return launch_nt8_form<EPI, false, 3>(s, q, workgroups);
    }
    if (bm == 224) {
        if (tiles > workgroups) return launch_nt8_form<EPI, true, 7>(s, q, workgroups);      // multi-tile form on 224-row tiles (A/B: MMBERT_NT_8PHASE_M224)
        return launch_nt8_form<EPI, false, 7>(s, q, workgroups);
    }
    const char* f = getenv("MMBERT_NT_8PHASE_FORM");              // A/B switch, read per call: "multi" runs the multi-tile form everywhere
    if (tiles > workgroups || (f && f[0] == 'm')) return launch_nt8_form<EPI, true, 4>(s, q, workgroups);
    return launch_nt8_form<EPI, false, 4>(s, q, workgroups);
}

constexpr int NTP_LDS_BYTES = 131072 + 1024 + 8 * 256;      // ring | tile-queue word (padded) | bias rows

// ---- which kernel, which tile, which tile walk: ONE function of the shape (and of the test / A-B knobs), shared by the launch path
// and by mmbert_gemm_nt_describe() (bench.py reports the choice per shape; tests pin it) ----
enum { NTK_128 = 0, NTK_RING = 1, NTK_PERSIST = 2, NTK_8PHASE = 3 };
constexpr int NT8_DEFAULT_LEVEL = 1;
struct NTChoice { int kernel, bm, tiles, workgroups, group_m, use_queue; };

static bool ntp_eligible(const GemmNT& p) {
    return !(p.K & 127) && p.K >= 256 && (long long)p.M * p.lda < (1ll << 31) && (long long)p.N * p.ldb < (1ll << 31);
}

// Tile walk of the persistent kernel (ntp_tile_mn).  An XCD's 32 workgroups own a contiguous chunk of the walk (xcd_remap), i.e.
// ceil(tiles_m / 8) row panels; with MORE tiles than CUs the order inside that chunk decides which panels its 32 concurrent tiles share:
//  * row-major (group_m = 1): ~3 row panels x ALL column panels at a time -- a weight panel is wanted by 3 workgroups at once and the
//    whole weight matrix (3.5-4.7 MB at N = 2304 / 3072) passes through the 4-MiB L2 once per 3 row panels;
//  * one group per XCD (group_m = ceil(tiles_m / 8), round 3): the XCD sweeps the column panels with ALL its row panels, 11 row panels
//    x ~3 column panels at a time -- a weight panel is wanted by 11 workgroups at once and then never again on this XCD.  Same-process
//    A/B of the train step (tools/ab_step.py, MMBERT_NT_GROUP_M): group_m = 1 / 6 / 8 / 11 / 16 / 32 / 100 -> 15.56 / 15.34 / 15.30 /
//    14.98 / 15.30 / 15.18 / 15.16 ms; the rule below = 11 at 18 400 rows.  (Fetched bytes barely move -- DESIGN 3.1.)
//  * the vocabulary projection (B = 47 MB, 9 960 tiles) keeps its groups of 4 row panels (round 2: 1 / 2 / 4 / 8 -> 871 / 855 / 844 /
//    875 us): its column sweep is 120 panels long.
// A/B switches, read per call: MMBERT_NT_GROUP_M=g (every non-huge multi-round shape) and, round 4, MMBERT_NT_GM_TABLE="N:K:E=g;N:K:E=g"
// (E = the epilogue flags: FFN-up and the GELU' input gradient share N and K; one shape at a time: the rule was decided on the step
// total in round 3 and QKV paid for it).
static int ntp_group_m(int M, int N, int K, int epi, int bm, int tiles, int cus) {
    if (tiles <= cus) return 1;
    const bool huge_b = (long long)N * K * 2 > (8ll << 20);
    if (huge_b) return tiles > 4 * cus ? 4 : 1;
    int gm = ((M + bm - 1) / bm + 7) / 8;
    if (const char* t = getenv("MMBERT_NT_GM_TABLE")) {
        for (const char* q = t; q && *q; ) {
            int n = 0, k = 0, e = 0, g = 0;
            if (sscanf(q, "%d:%d:%d=%d", &n, &k, &e, &g) == 4 && n == N && k == K && e == epi) return g;
            q = strchr(q, ';');
            if (q) ++q;
        }
    }
    if (const char* gs = getenv("MMBERT_NT_GROUP_M")) { const int g = atoi(gs); if (g >= 0) gm = g; }
    return gm;
}

static NTChoice nt_choose(const GemmNT& p, int epi) {
    NTChoice c = {NTK_128, 128, ((p.M + 127) / 128) * ((p.N + 127) / 128), 0, 1, 0};
    c.workgroups = c.tiles;
    // shape dispatch: the 256-wide pipeline needs >= 4 stages of K and enough rows to fill its tiles
    const bool big = (p.K >= 128) && (p.M >= 512) && (p.N >= 256) && !(p.N & 7) && !(p.ldc & 7) && !(p.ldr & 7) && !(p.ldaux & 7) && !(p.ldu & 7);
    if (!((g_nt_force >= 2 && !(p.N & 7)) || (g_nt_force == 0 && big))) return c;
    // Measured cost model (tools/stamp_gemm.py, tools/bench_gemm.py): a K step costs the same ~1270 clk for the 224- and
    // the 256-row tile (LDS-bound), so what counts is the number of tile rounds over the CUs; the shorter tile also has
    // the shorter epilogue.  224 rows unless that takes more rounds.  The persistent stream kernel (224-row form: the only
    // one that fits the register file without spills) hides every prologue but the first and has the leaner epilogue
    // (operand prefetch, 32-byte runs per lane): it won or tied on every shape of the step, single-round ones included.
    const int cus = device_cus(), tn = (p.N + 255) / 256;
    const int t256 = ((p.M + 255) / 256) * tn, t224 = ((p.M + 223) / 224) * tn;
    const int r256 = (t256 + cus - 1) / cus, r224 = (t224 + cus - 1) / cus;
    const bool can_persist = g_nt_persist && ntp_eligible(p);
    // The 8-phase kernel (round 4; see gemm_nt8_kernel).  Level by MMBERT_NT_8PHASE (A/B switch, read per call; default below):
    //   0 never | 1 launches whose 256-row tiles fit the chip in ONE round | 2 also multi-round launches (persistent form: the half-tile
    //   stream runs across tile seams) except the vocabulary-sized ones | 3 those too.  It has no device tile queue: launches that
    //   need one (data parallel, more tiles than CUs) keep the ring-persistent kernel.  mmbert_gemm_nt_force(8) forces it where eligible.
    {
        const char* e8 = getenv("MMBERT_NT_8PHASE");
        const int lvl = e8 ? atoi(e8) : NT8_DEFAULT_LEVEL;
        const bool huge_b = (long long)p.N * p.K * 2 > (8ll << 20);
        // ... single-round launches that leave more than half the chip idle (the reference's default model: M = 6400, N = 1024 is 100
        // tiles): the 128 x 128 kernel's 4 x as many tiles on 2 workgroups per CU are 6-14 % faster there (profiles/r4_bert_large_gemm_modes.log)
        if (g_nt_force == 0 && lvl >= 1 && 2 * t256 <= cus) {
            // (round 4, later) ... better still: the 8-phase kernel on 128-row tiles when those fit the chip in one round
            const int t128 = ((p.M + 127) / 128) * tn;
            static const bool bm128_ok = !(getenv("MMBERT_NT_8PHASE_BM128") && atoi(getenv("MMBERT_NT_8PHASE_BM128")) == 0);     // A/B switch
            if (bm128_ok && ntp_eligible(p) && t128 <= cus && 2 * t128 >= cus) {      // (fewer than half the CUs: the 128 x 128 kernel's finer tiles)
                c.kernel = NTK_8PHASE; c.bm = 128; c.tiles = t128; c.workgroups = t128; c.group_m = 1;
                return c;
            }
            if (c.tiles <= 2 * cus) return c;
        }
        bool shape_on = false;                                   // MMBERT_NT_8PHASE_MULTI="N:K:E;..." : the multi-tile form for single shapes (A/B)
        if (const char* t = getenv("MMBERT_NT_8PHASE_MULTI")) {
            for (const char* q = t; q && *q; ) {
                int n = 0, k = 0, e = 0;
                if (sscanf(q, "%d:%d:%d", &n, &k, &e) == 3 && n == p.N && k == p.K && e == epi) shape_on = true;
                q = strchr(q, ';');
                if (q) ++q;
            }
        }
        // (round 4, second half) single-round launches whose 192-row tiles ALSO fit the chip in one round -- the input gradients at ~13 850
        // packed rows: 165 tiles of 256 rows on 256 CUs, 219 of 192 -- take the 192-row tile: 3/4 of the MFMAs and 7/8 of the LDS-DMAs per
        // K tile on more of the chip (MMBERT_NT_8PHASE_BM192=0: A/B switch, read per call)
        {
            const int t192 = ((p.M + 191) / 192) * tn;
            const char* e192 = getenv("MMBERT_NT_8PHASE_BM192");
            if (g_nt_force == 0 && lvl >= 1 && ntp_eligible(p) && t192 <= cus && 2 * t256 > cus && !(e192 && atoi(e192) == 0)) {
                c.kernel = NTK_8PHASE; c.bm = 192; c.tiles = t192; c.workgroups = t192; c.group_m = ntp_group_m(p.M, p.N, p.K, epi, 192, t192, cus);
                return c;
            }
            // ... 224-row tiles (A half 0 = 128 rows, A half 1 = 96) where 192-row ones do not fit but these do -- the forward N = 768 shapes
            // at 18 400 rows: 216 tiles of 256 rows, 249 of 224.  Measured six times on several boxes (profiles/r4_ab_8phase_bm224.log): same
            // process with 8-step windows +0.6 %; with 40-step windows -0.4, -0.5, -0.6, -0.7, -0.8 %; alternating 600-step processes
            // -0.7 ... -0.9 %: on by default (MMBERT_NT_8PHASE_BM224=0: A/B switch, read per call).
            const char* e224 = getenv("MMBERT_NT_8PHASE_BM224");
            if (g_nt_force == 0 && lvl >= 1 && ntp_eligible(p) && t224 <= cus && t224 > t256 && 2 * t256 > cus && !(e224 && atoi(e224) == 0)) {
                c.kernel = NTK_8PHASE; c.bm = 224; c.tiles = t224; c.workgroups = t224; c.group_m = ntp_group_m(p.M, p.N, p.K, epi, 224, t224, cus);
                return c;
            }
        }
        // (round 4, last) The multi-round shapes -- QKV, FFN-up + GELU, the GELU' input gradient; not the vocabulary-sized ones -- on the
        // MULTI-TILE 8-phase form with 224-row tiles (A half 0 = 128 rows, A half 1 = 96: MQ = 7): the ring kernel's round count (2.92 /
        // 3.89 / 2.9 rounds at the headline shapes, where the 256-row multi-tile form started the same number of rounds of 14 % more work
        // each) with the 8-phase K loop.  Bit-identical to the ring kernel's results, 2-6 % faster per launch stand-alone.  In the step its
        // first A/Bs (same process, 8-step windows, on top of the ring weight-gradient kernel) read -1.95 ... +1.5 % by box; with 40-step
        // windows and with alternating 600-step processes, on top of the 8-phase weight-gradient kernel: -2.3 / -2.3 / -2.5 %
        // (profiles/r4_ab_8phase_m224.log); the vocabulary projection on it as well: -0.3 / -0.5 %.  Default (= 2); MMBERT_NT_8PHASE_M224=0
        // switches it off, =1 leaves the vocabulary-sized shapes on the ring kernel,
        // MMBERT_NT_8PHASE_M224_SKIP="N:K:E;..." leaves single shapes on the ring kernel (A/B switches, read per call).  With a caller's
        // tile queue (data-parallel runs) the same form draws its tiles from it (gemm_ntp_kernel's protocol; MMBERT_NT_8PHASE_QUEUE=0: the
        // ring kernel for those launches).
        {
            const char* em = getenv("MMBERT_NT_8PHASE_M224");
            bool skip = false;
            if (const char* t = getenv("MMBERT_NT_8PHASE_M224_SKIP")) {
                for (const char* q = t; q && *q; ) {
                    int n = 0, k = 0, e = 0;
                    if (sscanf(q, "%d:%d:%d", &n, &k, &e) == 3 && n == p.N && k == p.K && e == epi) skip = true;
                    q = strchr(q, ';');
                    if (q) ++q;
                }
            }
            const int m224 = em ? atoi(em) : 2;
            static const int q8_env = getenv("MMBERT_NT_8PHASE_QUEUE") ? atoi(getenv("MMBERT_NT_8PHASE_QUEUE")) : 1;   // 0: launches with a tile queue keep the ring kernel
            if (m224 >= 1 && !skip && lvl >= 1 && g_nt_force == 0 && ntp_eligible(p) && t224 > cus && t256 > cus && (!huge_b || m224 >= 2) &&
                (!p.tile_counter || (q8_env && !(cus & 7)))) {
                // Tile height of the multi-tile form: 224 rows unless another height is clearly cheaper in started rounds x time per tile
                // (K-tile clocks of the three forms: 192 rows 1 680 -- LDS-DMA bound --, 224 rows 1 800, 256 rows 2 048 -- MFMA bound).  At the
                // headline shapes 224 wins everywhere (QKV 3 rounds, FFN-up 4, GELU' input gradient 3); it would not with a few hundred rows
                // more in backward (14 400 valid rows x N = 3072: 780 tiles of 224 rows = 4 rounds, 684 of 256 = 3) or at other models'
                // shapes (bert-large QKV, 6 400 x 3072: 348 tiles of 224 rows = 2 rounds, 408 of 192 = 2 rounds of smaller tiles).
                // MMBERT_NT_8PHASE_MH=192|224|256 forces a height (A/B switch, read per call).
                const int t192 = ((p.M + 191) / 192) * tn;
                const long long c192 = (long long)((t192 + cus - 1) / cus) * 1680, c224 = (long long)r224 * 1800, c256 = (long long)r256 * 2048;
                int h = 224, th = t224;
                if (100 * c256 < 97 * c224 && c256 <= c192) { h = 256; th = t256; }
                else if (100 * c192 < 97 * c224 && c192 < c256) { h = 192; th = t192; }
                if (const char* mh = getenv("MMBERT_NT_8PHASE_MH")) {
                    const int v = atoi(mh);
                    if (v == 192 && t192 > cus) { h = 192; th = t192; } else if (v == 224) { h = 224; th = t224; } else if (v == 256) { h = 256; th = t256; }
                }
                c.use_queue = p.tile_counter != nullptr;
                c.kernel = NTK_8PHASE; c.bm = h; c.tiles = th; c.workgroups = cus;
                c.group_m = ntp_group_m(p.M, p.N, p.K, epi, h, th, cus);
                return c;
            }
        }
        const bool multi_ok = (t256 > cus) && !p.tile_counter && (lvl >= 3 || (lvl >= 2 && !huge_b) || shape_on);
        if (ntp_eligible(p) && ((g_nt_force == 0 && ((lvl >= 1 && t256 <= cus) || multi_ok)) || g_nt_force == 3)) {
            c.kernel = NTK_8PHASE; c.bm = 256; c.tiles = t256; c.workgroups = t256 < cus ? t256 : cus;
            c.group_m = ntp_group_m(p.M, p.N, p.K, epi, 256, t256, cus);
            return c;
        }
    }
    bool persist, tall;
    if (g_nt_bm == 0) {
        // default: the 224-row form.  The 256-row staggered form (mode 6) wins 1-7 % on single-round and very wide shapes in
        // isolation (tools/bench_gemm.py MODES=6,7; after the epilogue rewrite: N = 768 shapes -4..-6 %, vocabulary -6 %, QKV
        // equal, GELU epilogues +8..9 %) but not in the train step, neither everywhere (797 vs 805 samples/s) nor chosen per
        // launch by rounds x relative tile time (815 vs 820, two alternating runs on one box): not the default.
        // ... except where the taller tile saves whole ROUNDS over the CUs (static tile shares: a launch costs ceil(tiles / CUs)
        // tile times): backward runs on a data-dependent row count (model.py, SplitLayout), e.g. 14 400 rows x N = 3072 is 780
        // tiles = 4 rounds at 224 rows but 684 = 3 rounds at 256.  A 256-row tile is priced at 1.1 of a 224-row one.
        static const bool tall_ok = !(getenv("MMBERT_NT_TALL") && atoi(getenv("MMBERT_NT_TALL")) == 0);     // A/B switch
        // Round 2 (profiles/r2_exp_nt_tile_heights.log): 192- and 160-row forms of this kernel were built and timed on the
        // single-round N = 768 input-gradient shapes (13-14.4 k rows): a tile takes the SAME time at 160 / 192 / 224 / 256 rows
        // (23.2-23.6, 52.7-53.6, 68.3-70.0 us) -- the K step issues its 32 stage loads whatever the tile height (all 256 A rows
        // are staged) and that, not the MFMA count, is its length -- so only the round count matters and the two forms below
        // stay.  A start stagger of the workgroups (to spread the epilogues' store bursts) was a loss on every shape, the
        // 34-round vocabulary projection included (profiles/r2_exp_nt_start_stagger.log).
        // (round 2, same-process A/B of the train step: taking the 256-row form whenever it needs no more rounds, for the
        // epilogues without GELU / GELU', is 0.5-0.7 % SLOWER in situ although it wins 3-5 % per shape in isolation)
        persist = can_persist;
        tall = can_persist ? (tall_ok && (float)r256 * 1.1f < (float)r224) : !(r224 <= r256);
    } else {
        persist = can_persist && g_nt_force == 2 && g_nt_persist == 2;
        tall = g_nt_bm != 224;
    }
    c.kernel = persist ? NTK_PERSIST : NTK_RING;
    c.bm = tall ? 256 : 224;
    c.tiles = tall ? t256 : t224;
    c.workgroups = persist ? (c.tiles < cus ? c.tiles : cus) : c.tiles;
    if (persist) {
        c.group_m = ntp_group_m(p.M, p.N, p.K, epi, c.bm, c.tiles, cus);
        c.use_queue = (p.tile_counter != nullptr) && c.tiles > cus;      // launches with no more tiles than workgroups need no queue
    }
    return c;
}

template <int EPI, int MI>
static int launch_ntp_mi(hipStream_t s, const GemmNT& p, const NTChoice& c) {
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)gemm_ntp_kernel<EPI, MI, MI == 8>, NTP_LDS_BYTES, attr_done)) return e;
    const int cus = device_cus();
    GemmNT q = p;
    // Dynamic tile queue: the CALLER's 16 zero-initialised ints (mmbert_gemm_nt(..., tile_queue): 8 per-XCD fetch counters, exit
    // counter, padding); the kernel leaves them zero again (the last workgroup to exit resets them), so one 64-byte buffer serves
    // every launch of a stream.
    static const int qg_env = getenv("MMBERT_NT_QUEUE_GLOBAL") ? atoi(getenv("MMBERT_NT_QUEUE_GLOBAL")) : 0;   // A/B switch: one counter
    q.queue_xcd = (qg_env || (cus & 7)) ? 0 : 1;
    if (!c.use_queue) q.tile_counter = q.tile_counter_next = nullptr;
    q.group_m = c.group_m;
    hipLaunchKernelGGL((gemm_ntp_kernel<EPI, MI, MI == 8>), dim3(c.workgroups), dim3(512), NTP_LDS_BYTES, s, q);
    MMB_CHECK_LAUNCH();
    return 0;
}

// Round 2, tried and dropped: a "big tile" form of this kernel -- 320 / 384 x 256 output tile, ONE 4-wave workgroup per CU (one wave
// per SIMD, a wave owns 160 / 192 x 128 = 320 / 384 accumulator registers of the SIMD's 512), 4-slot ring, the step's one barrier
// in the middle of its MFMA stream -- to lift the FLOPs per staged byte from 115 to 142 / 157 (S3.1: the K step is bound by its
// (A rows + B rows) / 16 one-KiB LDS-DMA instructions, so only a bigger tile helps).  It does not survive hipcc: the accumulator
// file (AGPRs) holds 256 registers, the remaining 64 / 128 accumulators live in VGPRs and the allocator shuttles tuples between
// the two files inside the K loop (MI2 = 10: 773 v_accvgpr moves and 72 scratch accesses per 320 MFMAs; MI2 = 12: 1044 and 252 per
// 384), with or without scheduling barriers and with single- or double-buffered B fragments.  At 256 x 256 (all accumulators in
// AGPRs, no spills in the loop) the tile is the one the staggered 8-wave form already has.
template <int EPI, int MI>
static int launch_nt256_mi(hipStream_t s, const GemmNT& p) {
    constexpr int BM = 32 * MI;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)gemm_nt256_kernel<EPI, MI>, 131072, attr_done)) return e;
    hipLaunchKernelGGL((gemm_nt256_kernel<EPI, MI>), dim3(tiles), dim3(512), 131072, s, p);
    MMB_CHECK_LAUNCH();
    return 0;
}

template <int EPI>
static int dispatch_nt(hipStream_t s, const GemmNT& p) {
    const NTChoice c = nt_choose(p, EPI);
    if (c.kernel == NTK_128) return launch_nt<EPI>(s, p);
    if (c.kernel == NTK_8PHASE) return launch_nt8<EPI>(s, p, c.bm, c.tiles, c.workgroups, c.group_m);
    if (c.kernel == NTK_PERSIST) return c.bm == 256 ? launch_ntp_mi<EPI, 8>(s, p, c) : launch_ntp_mi<EPI, 7>(s, p, c);
    return c.bm == 256 ? launch_nt256_mi<EPI, 8>(s, p) : launch_nt256_mi<EPI, 7>(s, p);
}

// -------------------------------------------------------------------------------------------------
// TN (weight gradients): W_p[N_p,K_p] (+)= alpha * A_p[M,N_p]^T . B_p[M,K_p]  for up to 4 problems that
// share the token axis M (the four dense layers of one encoder layer go out as ONE launch), optional
// split over M into fp32 slabs reduced deterministically by tn_reduce_kernel, and optionally
// bias_p[N_p] += alpha * colsum(A_p): the bias gradient rides on the MFMA with an all-ones A operand.
//
// Workgroup = 512 threads (8 waves as 4(k) x 2(n)), output tile 256(n) x 256(k), one workgroup per CU; a wave owns
// 64(k) x 128(n) = 4 x 8 MFMA tiles.  The token axis is consumed in 32-row stages through a 4-slot LDS ring
// (A slots at 0, B slots at 64 KiB, 16 KiB each): stages s+1, s+2 are in flight behind a counted vmcnt(8) while
// stage s is read with ds_read_b64_tr_b16 (both operands are needed "m-major", i.e. transposed) and multiplied.
// Why 256 x 256: a CU issues a 16-byte-per-lane vector load every ~37 clk at best (see g_nt_dbg above).  The former
// 256(n) x 128(k) tile (4 waves, two workgroups per CU) needed 48 loads for the MFMA work of 1024 clk -> ~1780 clk,
// load-issue bound at ~58 % of the MFMA rate (measured ~54 %); this tile needs 32 (~1180 clk), like the NT kernel.
// LDS rows are 512 B; 16-byte chunks are XOR-swizzled by
// ((row&3)|((row>>1)&4))<<1 on the source address and on the read, so the 8 rows a 32-lane half
// touches in one transposed read fall on 8 distinct 32-byte bank groups.
// -------------------------------------------------------------------------------------------------
struct TNProb { const bf16_t* A; const bf16_t* B; float* W; float* bias; int N, K, lda, ldb, tiles_k, tile0; long long slab_off; };
// up to TN_MAXP problems per launch: the four dense layers of an encoder layer -- or of TWO layers (model._EncoderFn pairs them: 216
// tiles fill the chip in one round without splitting the token axis, so no fp32 slabs and no reduce launch) -- or, round 4, of up to
// TWELVE layers at once: without a gradient hook (one GPU) nothing needs a layer's weight gradients before the optimizer, so the model
// defers them all to ONE call at the end of backward, which goes out as whole rounds of CUs-many tiles (11 layers = 1188 tiles = 4 full
// launches + one of 164, against 5 paired launches at 216 of 256 CUs plus a split single layer).  48 x 64 B = 3 KiB of kernel arguments.
#define TN_MAXP 48
struct GemmTNG {
    TNProb pr[TN_MAXP];
    float* slab; const float* alpha_dev;
    long long slab_stride;
    int nprob, total_tiles, M, splits, rows_per_split, accumulate;
    float alpha;
    int tile_base;
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

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(const GemmTNG g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A: 4 slots x 32x256 (16 KiB) at 0 | B: 4 slots x 32x256 at 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;                       // 8 waves as 4(k) x 2(n): a wave owns 64(k) x 128(n)
  {
    // (round 4) a launch covers the tiles [tile_base, tile_base + gridDim.x) of the problem list: the deferred multi-layer form goes out
    // as whole ROUNDS of CUs-many tiles, one launch per round (a tile loop inside the kernel cost 18-89 spilled VGPRs: the accumulators
    // and the epilogue's accumulate operands leave no room for anything carried across tiles)
    int t = g.tile_base + xcd_remap(blockIdx.x, gridDim.x);
    const int ln = lane;
    // the problem table is read from the kernel-argument segment through a (wave-uniform) computed index: scalar loads, no select
    // chain over up to 48 entries and no private copy of the 3-KiB struct
    const __attribute__((address_space(4))) GemmTNG& gq = *(const __attribute__((address_space(4))) GemmTNG*)__builtin_amdgcn_kernarg_segment_ptr();
    int pi = 0;
    {
        int lo = 0, hi = g.nprob - 1;                              // tile0 ascending: the last problem that starts at or before t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (gq.pr[mid].tile0 <= t) lo = mid; else hi = mid - 1; }
        pi = lo;
    }
    const __attribute__((address_space(4))) TNProb& pr = gq.pr[pi];
    const bf16_t* Ap = pr.A; const bf16_t* Bp = pr.B; float* Wp = pr.W; float* biasp = pr.bias;
    const int N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, tiles_k = pr.tiles_k, tile0 = pr.tile0;
    const long long slab_off = pr.slab_off;
    t -= tile0;
    const int n0 = (t / tiles_k) << 8, k0 = (t % tiles_k) << 8;
    const int split = blockIdx.y;
    const int mbeg = split * g.rows_per_split;
    const int mend = min(g.M, mbeg + g.rows_per_split);
    const int ns = (mend - mbeg + 31) >> 5;                        // may be <= 0 for a trailing empty split
    const bool do_bias = (biasp != nullptr) && (k0 == 0) && (wk == 0);

    // ---- staging (LDS-DMA, ln-linear destination, swizzle on the source chunk) ----
    // A and B stage tiles: 32 rows x 512 B = 16 wave-instructions each (2 rows per instruction); wave w issues 2w, 2w+1
    int s_row[2], a_col[2], b_col[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        s_row[i] = (wave * 2 + i) * 2 + (ln >> 5);
        a_col[i] = min(n0 + (((ln & 31) ^ tn_swz(s_row[i])) << 3), N - 8);
        b_col[i] = min(k0 + (((ln & 31) ^ tn_swz(s_row[i])) << 3), K - 8);
    }
    // buffer addressing (descriptor + constant per-ln offset + scalar row-block offset): no per-stage 64-bit VALU address
    // arithmetic; token rows past M read as zeros through the range check
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)((uint32_t)g.M * (uint32_t)lda * 2u), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)((uint32_t)g.M * (uint32_t)ldb * 2u), 0x00020000);
    uint32_t a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((uint32_t)s_row[i] * (uint32_t)lda + (uint32_t)a_col[i]) * 2u;
        b_off[i] = ((uint32_t)s_row[i] * (uint32_t)ldb + (uint32_t)b_col[i]) * 2u;
    }
    // one of the wave's four loads of a stage (q = 0..3: A rows 0, B rows 0, A rows 1, B rows 1)
    auto stage_one = [&](int slot, int st, int q) {
        const uint32_t mb = (uint32_t)(mbeg + min(st, max(ns - 1, 0)) * 32);
        char* base = smem + slot * 16384 + wave * 2048 + (q >> 1) * 1024;
        if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base + 65536), 16, b_off[q >> 1], mb * (uint32_t)ldb * 2u, 0, 0);
        else       __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base), 16, a_off[q >> 1], mb * (uint32_t)lda * 2u, 0, 0);
    };
    auto stage = [&](int slot, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_one(slot, st, q);
    };

    f32x4 acc[4][8], accb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- transposed-read addresses: ln (g4 = ln>>4, q = (ln>>2)&3, pp = ln&3) supplies row 8*g4+q (+4), 4 columns at 4*pp ----
    const int g4 = ln >> 4, r0 = 8 * g4 + ((ln >> 2) & 3), pp = ln & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)LPTR(smem);
    unsigned offA[8], offB[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = wn * 128 + j * 16 + 4 * pp;
        offA[j] = lds0 + r0 * 512 + ((((col >> 3) ^ tn_swz(r0)) << 4) | ((col & 4) << 1));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = wk * 64 + i * 16 + 4 * pp;
        offB[i] = lds0 + 65536 + r0 * 512 + ((((col >> 3) ^ tn_swz(r0)) << 4) | ((col & 4) << 1));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

    u32x2 ylo[8], yhi[8], xlo[4], xhi[4];
    auto reads = [&](auto slot_c) {
        constexpr int SB = decltype(slot_c)::value * 16384;        // both regions: slot pitch 16 KiB, every ds offset < 64 KiB
#pragma unroll
        for (int i = 0; i < 4; ++i) { tr_read<SB>(xlo[i], offB[i]); tr_read<SB + 4 * 512>(xhi[i], offB[i]); }   // rows r0 and r0+4 (same swizzle: bit 2 unused)
#pragma unroll
        for (int j = 0; j < 8; ++j) { tr_read<SB>(ylo[j], offA[j]); tr_read<SB + 4 * 512>(yhi[j], offA[j]); }
    };
    auto mma = [&](int st, int nslot) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ay[8], bx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u32x4 v = {xlo[i][0], xlo[i][1], xhi[i][0], xhi[i][1]}; bx[i] = __builtin_bit_cast(bf16x8, v); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u32x4 v = {ylo[j][0], ylo[j][1], yhi[j][0], yhi[j][1]}; ay[j] = __builtin_bit_cast(bf16x8, v); }
        const int mrem = (mend - mbeg) - st * 32;                  // valid rows in this stage
        if (mrem < 32) {                                           // wave-uniform: only the last stage of a split
            const int mb8 = 8 * g4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (mb8 + e >= mrem) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ay[j][e] = (bf16_t)0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bx[i][e] = (bf16_t)0.0f;
                }
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            stage_one(nslot, st + 3, i);    // the four LDS-DMA loads of stage s + 3, spread between the MFMA groups: a burst
                                            // of loads in front of the MFMAs stalls the wave on the (saturated) load issue queue
#pragma unroll
            for (int j = 0; j < 8; ++j)     // D[row <-> k_out (X^T as the A operand)][col <-> n_out (dY as the B operand)]
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[i], ay[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ay[j], accb[j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // Two wave groups (waves 0-3 and 4-7: one wave of each per SIMD) run half a stage apart: the second group takes one extra
    // barrier up front, so its k-th barrier meets the first group's (k + 1)-th, and each stage is { transposed reads, wait
    // for this wave's loads of stage s + 1 ; barrier X ; LDS-DMA of stage s + 3, MFMAs ; barrier Y }: one group's MFMAs run
    // under the other's reads (in lockstep the fragment reads and their latency sat in front of every wave's MFMAs: 1680 clk
    // per stage against ~1200 of load issue).
    //  RAW: every wave has waited for its part of stage s + 1 before its X(s); group 0 reads stage s + 1 behind its Y(s),
    //       which pairs with group 1's X(s); group 1 behind its Y(s), which pairs with group 0's X(s + 1).
    //  WAR: stage s + 3 overwrites stage s - 1, whose reads a wave completes (lgkmcnt(0)) before its Y(s - 1); the issue
    //       sits behind X(s), which pairs with the other group's Y(s - 1) (group 0) or Y(s) (group 1).
#ifdef MMB_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr0) :: "memory");
    MMB_STAMP(ts0)
#endif
    if (ns > 0) {
        stage(0, 0); stage(1, 1); stage(2, 2);
        __builtin_amdgcn_s_waitcnt(0x0F78);                        // vmcnt(8): stage 0 (this wave's part)
        __builtin_amdgcn_s_barrier();
        if (wave >= 4) __builtin_amdgcn_s_barrier();               // stagger
#define TN_STEP(SLOT)                                                                       \
        {                                                                                   \
            reads(std::integral_constant<int, SLOT>{});                                     \
            __builtin_amdgcn_s_waitcnt(0x0F74);   /* vmcnt(4): this wave's part of stage s + 1 has landed */ \
            __builtin_amdgcn_s_barrier();                                                   \
            mma(s, (SLOT + 3) & 3);                                                         \
            __builtin_amdgcn_s_barrier();                                                   \
            ++s;                                                                            \
        }
        int s = 0;
        while (true) {
            TN_STEP(0) if (s >= ns) break;
            TN_STEP(1) if (s >= ns) break;
            TN_STEP(2) if (s >= ns) break;
            TN_STEP(3) if (s >= ns) break;
        }
#undef TN_STEP
        if (wave < 4) __builtin_amdgcn_s_barrier();                // balances the stagger barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef MMB_STAMPS
    MMB_STAMP(ts1)
#endif

    const float alpha = g.alpha * (g.alpha_dev ? *g.alpha_dev : 1.0f);
    const int fr = ln & 15, fq = ln >> 4;
    // split 0 writes (or accumulates into) the weight gradient itself, splits 1.. write fp32 slabs that tn_reduce_kernel adds to it
    // afterwards: one slab write, one slab read and one launch-wide pass less than "all splits to slabs" (order stays fixed)
    float* out = split > 0 ? g.slab + (size_t)(split - 1) * g.slab_stride + slab_off : Wp;
    const bool accum = (split == 0) && g.accumulate;
    // (accumulating form: the eight reads of a k block are issued TOGETHER, clamped in range, then added and stored -- as first written
    // every float4 was read, waited for with vmcnt(0) -- which also waits for the previous store --, added and stored: 32 dependent
    // round trips per ln at the end of every tile; same-process A/B of the step: -0.15 %)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + i * 16 + fq * 4;
        const int kc = min(k, K - 4);
        float4 old[8];
        if (accum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nc = min(n0 + wn * 128 + j * 16 + fr, N - 1);
                old[j] = *(const float4*)(out + (size_t)nc * K + kc);
            }
        }
        if (k >= K) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + fr;
            if (n >= N) continue;
            float* dst = out + (size_t)n * K + k;
            float4 v = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
            if (accum) { v.x += old[j].x; v.y += old[j].y; v.z += old[j].z; v.w += old[j].w; }
            *(float4*)dst = v;
        }
    }
    if (do_bias && fq == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + fr;
            if (n < N) atomicAdd(biasp + n, accb[j][0] * alpha);
        }
    }
#ifdef MMB_STAMPS
    if (g_stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr1) :: "memory");
        unsigned long long te;
        MMB_STAMP(te)
        unsigned long long* o = g_stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 6;
        o[0] = ts1 - ts0; o[1] = (unsigned long long)ns; o[2] = te - ts1; o[3] = 1; o[4] = tr0; o[5] = tr1;
    }
#endif
  }
}

// -------------------------------------------------------------------------------------------------
// TN in the 8-phase structure (round 4, second half): the weight gradients are the deepest products of the step (the reduction runs
// over ~14 000 tokens = 216 K tiles of 64) and the ring kernel above spends 1 473 clk per 32-token stage where its MFMAs need 1 024.
// Same tile (256 x 256, 8 waves, one workgroup per CU, same problem table / splits / slabs / accumulate semantics, same 32-token
// summation blocks: the weight gradients are BIT-IDENTICAL to gemm_tn_kernel's), the K loop of gemm_nt8_kernel:
//   * K tile = 64 tokens; half-tiles of [64 tokens][128 columns] (256-byte rows: an LDS-DMA wave instruction moves 4 token rows x
//     two whole cache lines) are the unit of staging, waiting and re-use; order of first use Y0h, X0h, Y1h, X1h;
//   * waves 2 (wr) x 4 (wc): a wave owns k columns {ha*128 + wr*64 ..+63} (X, 4 blocks of 16 per half) x n columns
//     {hb*128 + wc*32 ..+31} (dY, 2 blocks per half) -- the halves are CONTIGUOUS 128-column panels, the wave interleave sits inside;
//   * phases: (X0h,Y0h) (X0h,Y1h) (X1h,Y1h) (X1h,Y0h), 16 MFMAs each; two wave groups one barrier apart; the stream runs three
//     half-tiles ahead behind one counted vmcnt(6) per K tile;
//   * fragments by ds_read_b64_tr_b16 (48 per K tile and wave), conflict-free with the chunk XOR of the ring kernel (tn_swz) on
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
    int t_lin = g.tile_base + xcd_remap(blockIdx.x, gridDim.x);
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
    const long long slab_off = pr.slab_off;
    t_lin -= pr.tile0;
    const int n0 = (t_lin / tiles_k) << 8, k0 = (t_lin % tiles_k) << 8;
    const int split = blockIdx.y;
    const int mbeg = split * g.rows_per_split;
    const int mend = min(g.M, mbeg + g.rows_per_split);
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
    const bool accum = (split == 0) && g.accumulate;
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
                    if (n < N) atomicAdd(biasp + n, (v[hb][j] + red[((wc * 2 + hb) * 2 + j) * 16 + fr]) * alpha);
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
struct TNReduce { float* W[TN_MAXP]; long long off[TN_MAXP + 1]; int nprob, splits, accumulate; long long slab_stride; };
__global__ void tn_reduce_kernel(const TNReduce r, const float* __restrict__ slab) {
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
// launching anything: out[0] = kernel (0: 128x128-tile kernel, 1: 4-slot-ring kernel with one launch slot per tile, 2: persistent stream
// kernel), out[1] = tile rows (128 / 224 / 256), out[2] = tile columns, out[3] = output tiles, out[4] = workgroups launched,
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

// test / benchmarking hook: 0 = automatic shape dispatch, 1 = always the 128^2 kernel, 2 = always the 256^2 kernel
#ifdef MMB_STAMPS
int mmbert_debug_set_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf));
}
int mmbert_debug_set_nt_dbg(int v) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nt_dbg), &v, sizeof(v));
}
#endif
void mmbert_gemm_nt_force(int mode) {
    // 0 auto | 1 128^2 | ring kernel, one launch slot per tile: 2 (tile height auto), 3 (256x256), 4 (224x256)
    // | persistent stream kernel: 5 (tile height auto), 6 (256x256), 7 (224x256) | 8: the 8-phase kernel for every eligible shape
    if (mode == 8) { g_nt_force.store(3); g_nt_bm.store(0); g_nt_persist.store(1); return; }
    g_nt_force.store(mode >= 2 ? 2 : mode);
    g_nt_bm.store((mode == 3 || mode == 6) ? 256 : (mode == 4 || mode == 7) ? 224 : 0);
    g_nt_persist.store((mode >= 2 && mode <= 4) ? 0 : (mode >= 6 ? 2 : 1));      // 2: forced persistent tile height
}

static std::atomic<int> g_tn_splits{0};   // 0 = by shape; > 0 forces the split count of the token axis (A/B benchmarking)
void mmbert_gemm_tn_force_splits(int splits) { g_tn_splits.store(splits); }
// 1 = the 8-phase K loop (gemm_tn8_kernel, default), 0 = the 4-slot ring of 32-token stages (gemm_tn_kernel).  7.7 % fewer cycles per token,
// -4.5 % stand-alone; in the train step its first A/Bs (same process, 8-step windows) read +-0 -- the chip answers the denser MFMA stream with a
// lower clock, and a short window right behind a switch still carries the other variant's clock state --; with 40-step windows -0.6 / -0.8 %,
// with alternating 600-step processes -0.84 % (profiles/r4_stamp_tn8.log, r4_ab_tn8.log)
static std::atomic<int> g_tn_form{1};
void mmbert_gemm_tn_force_form(int form) { g_tn_form.store(form != 0); }

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
    if (g_tn_splits.load() > 0) splits = g_tn_splits.load();
    if (tiles >= slots) splits = 1;                                // (enough tiles to fill the chip: the persistent walk, no slabs)
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
    size_t elems = 0;
    for (int i = 0; i < nprob; ++i) elems += (size_t)N[i] * K[i];
    return (size_t)(splits - 1) * elems * sizeof(float);
}

size_t mmbert_gemm_tn_workspace(int M, int N, int K, int* splits_out) {
    return mmbert_gemm_tn_grouped_workspace(1, &N, &K, M, splits_out);
}

// up to TN_MAXP (8) problems sharing M:  W_i[N_i,K_i] (+)= alpha * A_i^T . B_i ;  bias_i[N_i] += alpha * colsum(A_i) (bias_i may be null)
int mmbert_gemm_tn_grouped(hipStream_t stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                           float* const* W, float* const* bias, const int* N, const int* K, int M,
                           int accumulate, float alpha, const float* alpha_dev, void* slab) {
    if (nprob <= 0 || M <= 0) return 0;
    if (nprob > TN_MAXP) return -1;
    GemmTNG g;
    int splits, tiles;
    tn_plan(nprob, N, K, M, &splits, &tiles);
    long long off = 0;
    int tile0 = 0;
    TNReduce r;
    for (int i = 0; i < nprob; ++i) {
        if ((N[i] & 7) || (K[i] & 7) || (lda[i] & 7) || (ldb[i] & 7) || N[i] < 8 || K[i] < 8) return -1;
        TNProb& q = g.pr[i];
        q.A = (const bf16_t*)A[i]; q.B = (const bf16_t*)B[i]; q.W = W[i]; q.bias = bias ? bias[i] : nullptr;
        q.N = N[i]; q.K = K[i]; q.lda = lda[i]; q.ldb = ldb[i];
        q.tiles_k = (K[i] + 255) / 256; q.tile0 = tile0; q.slab_off = off;
        r.W[i] = W[i]; r.off[i] = off;
        tile0 += ((N[i] + 255) / 256) * q.tiles_k;
        off += (long long)N[i] * K[i];
    }
    for (int i = nprob; i < TN_MAXP; ++i) { g.pr[i] = g.pr[0]; g.pr[i].tile0 = 0x7fffffff; r.W[i] = nullptr; }
    for (int i = nprob; i <= TN_MAXP; ++i) r.off[i] = off;
    if (splits > 1 && !slab) return -3;
    g.slab = (float*)slab; g.alpha_dev = alpha_dev; g.slab_stride = off; g.nprob = nprob; g.total_tiles = tiles; g.M = M;
    g.splits = splits; g.rows_per_split = (((M + splits - 1) / splits) + 31) / 32 * 32; g.accumulate = accumulate; g.alpha = alpha;
    static std::atomic<unsigned long long> attr_done{0}, attr_done8{0};
    if (int e = mmb_allow_lds((const void*)gemm_tn_kernel, 131072, attr_done)) return e;
    if (int e = mmb_allow_lds((const void*)gemm_tn8_kernel, 131072, attr_done8)) return e;
    // which K loop: the 8-phase form (gemm_tn8_kernel) unless the ring form is asked for (mmbert_gemm_tn_force_form / MMBERT_TN_8PHASE=0, A/B
    // switch read per call); both give the same bits in the weight gradients
    int form = g_tn_form.load();
    if (const char* e8 = getenv("MMBERT_TN_8PHASE")) form = atoi(e8) != 0;
    auto kern = form ? gemm_tn8_kernel : gemm_tn_kernel;
    // more tiles than CUs (only the deferred multi-layer launches; never split): whole rounds of CUs-many tiles, one launch per round
    const int cus_ = device_cus();
    if (splits == 1 && tiles > cus_) {
        for (int base = 0; base < tiles; base += cus_) {
            g.tile_base = base;
            hipLaunchKernelGGL(kern, dim3(tiles - base < cus_ ? tiles - base : cus_, 1), dim3(512), 131072, stream, g);
            MMB_CHECK_LAUNCH();
        }
    } else {
        g.tile_base = 0;
        hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(512), 131072, stream, g);
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
    gy = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16_t*)X, M, N, ldx, out, alpha, alpha_dev, rows);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
