// YARDSTICK, never linked into the product library: the 256 x 256 "8-phase" bf16 GEMM structure that the CDNA4 programming guide
// describes (cdna_hip_programming.md S5 "The 256^2 8-phase template", S5.5 T2-T5), re-implemented here from that description (the
// guide's example file is not in this image) to answer one question of VERDICT r3: is the product's persistent NT kernel
// (msa_amd/csrc/gemm.hip: 224 x 256 tile, 32-deep stages through a 4-slot ring, 16 x 64-B LDS-DMA pieces, one barrier per step) slow
// because of its STRUCTURE or because of the SHAPES (K = 768, M = 18 400)?  Same product C[M,N] = A[M,K] . B[N,K]^T, bf16 in, fp32
// accumulate, bf16 out, plain epilogue; run by tools/yardstick/run_yardstick.py against mmbert_gemm_nt on the headline shapes.
//
// Geometry (the guide's table): tile 256 x 256, BK = 64, 8 waves as 2(M) x 4(N), 128 x 64 per wave = 8 x 4 MFMA 16x16x32 tiles
// (128 accumulator registers), LDS = 2 K-tile buffers x 4 half-tiles x 16 KiB = 128 KiB, 2 LDS-DMA instructions per thread and
// half-tile, each wave instruction = 8 rows x 128 B (whole cache lines: the source pattern the product's 32-deep stages cannot have).
// A half-tile is a QUADRANT operand, not a wave's share: A-half h = rows {wr*128 + h*64 + 0..63, wr = 0, 1}, B-half h = columns
// {wc*64 + h*32 + 0..31, wc = 0..3} -- every wave reads b0 / a0 in phase 1, b1 in phase 2, a1 in phase 3, nothing in phase 4, so a
// half-tile's LDS is free again one to two phases after its phase and the stream of half-tiles runs 3 ahead (counted vmcnt(6) once
// per K tile, never 0 in the loop).  Two wave groups (wr = 0 / 1: one wave of each per SIMD) run one barrier apart: one group's 16
// MFMAs of a phase run under the other's LDS reads and LDS-DMA issue.
// LDS image of a half-tile: [128 rows][64 k] bf16 = 128-B rows, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7):
// every ds_read_b128 lane group {rows 0-3, 12-15 at chunk q | rows 4-11 at chunk q^1} then hits 16 distinct 16-byte slots of the
// 256-B bank row (conflict-free); the swizzle sits on the per-lane SOURCE address (LDS-DMA writes lane-linear) and on the read.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

struct YArgs { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K, lda, ldb, ldc; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {     // bijective: each XCD gets a contiguous chunk (guide S5)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
__host__ __device__ constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | 0x70 | (lgkm << 8); }

__global__ __launch_bounds__(512, 2) void gemm_8phase_kernel(const YArgs p) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];   // buffer d at d * 65536: A0h | A1h | B0h | B1h, 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + 255) >> 8;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) << 8, n0 = (tile % tiles_n) << 8;
    const int nt = p.K >> 6;                                     // K tiles (K % 128 == 0: an even count)

    // ---- staging: wave w issues pieces j = w and w + 8 of a half-tile (piece = local rows 8j .. 8j + 7, 1 KiB) ----
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((uint32_t)p.M * (uint32_t)p.lda * 2u), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((uint32_t)p.N * (uint32_t)p.ldb * 2u), 0x00020000);
    uint32_t va[2][2], vb[2][2];                                 // [half][piece]: per-lane byte offsets into A / B (K offset is scalar)
    {
        const int pos = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave + 8 * i, r = 8 * j + (lane >> 3);              // local row of the half-tile
            const uint32_t chunk = (uint32_t)(pos ^ ((r >> 1) & 7));                                  // A image
            const uint32_t chunk_b = (uint32_t)(pos ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1)));     // B image (rows are read permuted, below)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ga = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
                const int gb = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                va[h][i] = ((uint32_t)min(ga, p.M - 1) * (uint32_t)p.lda + chunk * 8u) * 2u;
                vb[h][i] = ((uint32_t)min(gb, p.N - 1) * (uint32_t)p.ldb + chunk_b * 8u) * 2u;
            }
        }
    }
    // half-tile ids: 0 = B0h, 1 = A0h, 2 = B1h, 3 = A1h (the order of first use)
    auto stage = [&](int buf, int which, int kt) {
        const uint32_t kb = (uint32_t)min(kt, nt - 1) * 128u;    // past the last tile: dead re-reads, the in-flight count stays constant
        const int h = which >> 1;
        char* base = smem + buf * 65536 + ((which & 1) ? 0 : 32768) + h * 16384 + wave * 1024;
        if (which & 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base), 16, va[h][0], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LPTR(base + 8192), 16, va[h][1], kb, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base), 16, vb[h][0], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LPTR(base + 8192), 16, vb[h][1], kb, 0, 0);
        }
    };

    // ---- fragment reads ----
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (fr >> 1)) & 7) << 4;                // k step 0: chunk fq;  k step 1: chunk 4 + fq = the same ^ 64 bytes
    const lds_cptr a_rd = (lds_cptr)LPTR(smem) + (wr * 64 + fr) * 128;
    // B rows are read in a permuted order so that the two column blocks (j = 0, 1) of a quadrant give each lane 8 CONSECUTIVE output
    // columns (one 16-byte store instead of two 8-byte ones): MFMA row fr of block j = column 8 (fr >> 2) + 4 j + (fr & 3) of the
    // quadrant's 32; the B image's chunk swizzle is keyed on row bits 1, 3, 4 so that these reads stay conflict-free
    const int brow = 8 * (fr >> 2) + (fr & 3);                   // + 4 j
    const lds_cptr b_rd = (lds_cptr)LPTR(smem) + 32768 + (wc * 32 + brow) * 128;
    lds_cptr a_rd1 = a_rd + 65536, b_rd1 = b_rd + 65536;        // second buffer: ds offsets are 16-bit
    asm volatile("" : "+v"(a_rd1), "+v"(b_rd1));

    bf16x8 af[2][4], b0f[2][2], b1f[2][2];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto read_a = [&](int buf, int h) {
        const lds_cptr ab = buf ? a_rd1 : a_rd;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *(lds_frag)(ab + h * 16384 + i * 2048 + (sw0 ^ (ks * 64)));
    };
    auto read_b = [&](int buf, int h, bf16x8 (&bf)[2][2]) {
        const lds_cptr bb = buf ? b_rd1 : b_rd;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // key(row) for row = brow + 4 j: bit 0 = (row >> 1) & 1 = (fr >> 1) & 1, bits 1-2 = (row >> 3) & 3 = fr >> 2
                const int swb = ((fq ^ (((fr >> 1) & 1) | ((fr >> 2) << 1))) & 7) << 4;
                bf[ks][j] = *(lds_frag)(bb + h * 16384 + j * 512 + (swb ^ (ks * 64)));
            }
    };
    auto mma = [&](int qa, int qb, const bf16x8 (&bf)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)     // operands swapped (B first): a lane holds 4 consecutive COLUMNS of one output row
                    acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);
    };

    // ---- prologue: tile 0 (4 half-tiles, even buffer) and the first 3 half-tiles of tile 1 (odd buffer) ----
    stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
    stage(1, 0, 1); stage(1, 1, 1); stage(1, 2, 1);
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(6, 15));               // tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // the stagger: group 1 runs one barrier behind group 0

    // one phase: { LDS reads of this phase's quadrant operands ; one half-tile of LDS-DMA ; [counted waits] ; barrier ; MFMAs ; barrier }
#define PHASE(READS, LGK_BEFORE_BARRIER, STAGE, VMWAIT, QA, QB, BF)                        \
    {                                                                                     \
        READS;                                                                            \
        STAGE;                                                                            \
        if (LGK_BEFORE_BARRIER >= 0) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, LGK_BEFORE_BARRIER < 0 ? 0 : LGK_BEFORE_BARRIER)); \
        if (VMWAIT >= 0) __builtin_amdgcn_s_waitcnt(waitcnt_imm(VMWAIT < 0 ? 0 : VMWAIT, 15));               \
        __builtin_amdgcn_s_barrier();                                                     \
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));                                   \
        __builtin_amdgcn_sched_barrier(0);                                                \
        __builtin_amdgcn_s_setprio(1);                                                    \
        mma(QA, QB, BF);                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                    \
        __builtin_amdgcn_s_barrier();                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }
    // K tile t in buffer D (tile t + 1 in D ^ 1).  RAW: everything of tile t + 1 is issued by phase 1 of tile t and retired by the
    // vmcnt(6) of phase 4 (3 half-tiles of tile t + 2 stay in flight), one phase before its first read.  WAR: b0 (read first in phase 1,
    // retired by lgkmcnt(8) before that phase's first barrier) is restaged in phase 2; a0 (phase 1) in phase 3; b1 (phase 2) in phase 4;
    // a1 (phase 3) in phase 1 of the next tile -- two phases after their reads, which covers the group that runs a barrier behind.
#define KTILE(D, T)                                                                                                        \
    PHASE((read_b(D, 0, b0f), __builtin_amdgcn_sched_barrier(0), read_a(D, 0)), 8, stage(D ^ 1, 3, (T) + 1), -1, 0, 0, b0f)   \
    PHASE(read_b(D, 1, b1f), -1, stage(D, 0, (T) + 2), -1, 0, 1, b1f)                                                      \
    PHASE(read_a(D, 1), -1, stage(D, 1, (T) + 2), -1, 1, 1, b1f)                                                           \
    PHASE((void)0, -1, stage(D, 2, (T) + 2), 6, 1, 0, b0f)

    for (int t = 0; t < nt; t += 2) {
        KTILE(0, t)
        KTILE(1, t + 1)
    }
#undef KTILE
#undef PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead tail half-tiles

    // ---- epilogue: bf16, 8 consecutive columns per lane and quadrant (acc[i][2 qb], acc[i][2 qb + 1]) ----
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    const bool interior = (m0 + 256 <= p.M) && (n0 + 256 <= p.N);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 + fr;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int n = n0 + wc * 64 + qb * 32 + fq * 8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[i][2 * qb + (e >> 2)][e & 3];
            if (interior || (m < p.M && n + 8 <= p.N)) *(bf16x8*)(p.C + (size_t)m * p.ldc + n) = o;
        }
    }
#endif
}

extern "C" int yardstick_gemm_nt(hipStream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 128 || (K & 127) || (N & 7) || (lda & 7) || (ldb & 7) || (ldc & 7)) return -1;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_8phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess) return -2;
        attr = true;
    }
    YArgs p = {(const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, lda, ldb, ldc};
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(gemm_8phase_kernel, dim3(tiles), dim3(512), 131072, stream, p);
    return (int)hipGetLastError();
}
