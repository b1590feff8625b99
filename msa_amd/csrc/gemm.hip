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
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "each XCD gets a contiguous chunk" remap (blocks b and b+8 share an XCD)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
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
    const int nt = p.K >> 6;

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

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nt - 1; ++kt) {
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
    }
}

template <int EPI>
static int launch_nt(hipStream_t s, const GemmNT& p) {
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(tiles), dim3(256), 65536, s, p);
    MMB_CHECK_LAUNCH();
    return 0;
}

// -------------------------------------------------------------------------------------------------
// TN: W[N,K] (+)= A[M,N]^T . B[M,K], reduction over the token axis M, optional split over M.
// LDS tiles are [64 m][128 cols] bf16 (256-B rows), chunk-swizzled by f(row) = ((row&3)|((row>>1)&4))<<1
// so that the 8 rows a 32-lane half touches in one ds_read_b64_tr_b16 fall on distinct bank groups.
// -------------------------------------------------------------------------------------------------
struct GemmTN {
    const bf16_t* A; const bf16_t* B; float* W; float* slab;
    int M, N, K, lda, ldb, ldw, splits, rows_per_split, accumulate;
    float alpha; const float* alpha_dev;
};

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;
    const int tiles_k = (p.K + 127) >> 7, tiles_n = (p.N + 127) >> 7;
    const int tile = xcd_remap(blockIdx.x, tiles_k * tiles_n);
    const int n0 = (tile / tiles_k) << 7, k0 = (tile % tiles_k) << 7;
    const int split = blockIdx.y;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nt = (mend - mbeg + 63) >> 6;

    // staging: a wave-instruction covers 4 rows x 256 B; wave w issues row-groups 4w..4w+3 (16 rows)
    const int srow = lane >> 4, sc = lane & 15;
    auto stage = [&](int buf, int mt) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 4 + srow;            // row inside the 64-row tile
            const int m = mbeg + mt * 64 + r;
            const int mc = min(m, p.M - 1);                      // clamped; rows >= mend are zeroed by the row mask below
            const int c = sc ^ tn_swz(r);
            const int ca = min(n0 + c * 8, p.N - 8), cb = min(k0 + c * 8, p.K - 8);
            __builtin_amdgcn_global_load_lds(GPTR(p.A + (size_t)mc * p.lda + ca), LPTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GPTR(p.B + (size_t)mc * p.ldb + cb), LPTR(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto compute = [&](int buf, int mt) {
        const char* As = smem + buf * 32768;
        const char* Bs = As + 16384;
        const int mrem = (mend - mbeg) - mt * 64;                // valid rows in this tile (>0)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // lane supplies the address of row (kk*32 + 8g + q [+4]), columns col0 + 4*pp
            const int r0 = kk * 32 + 8 * g + q, r1 = r0 + 4;
            s16x4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int colA = wn * 64 + i * 16 + 4 * pp;       // n_out columns of the A (=dY) tile
                const int colB = wk * 64 + i * 16 + 4 * pp;       // k_out columns of the B (=X) tile
                const int offA0 = r0 * 256 + ((((colA >> 3) ^ tn_swz(r0)) << 4) | ((colA & 4) << 1));
                const int offA1 = r1 * 256 + ((((colA >> 3) ^ tn_swz(r1)) << 4) | ((colA & 4) << 1));
                const int offB0 = r0 * 256 + ((((colB >> 3) ^ tn_swz(r0)) << 4) | ((colB & 4) << 1));
                const int offB1 = r1 * 256 + ((((colB >> 3) ^ tn_swz(r1)) << 4) | ((colB & 4) << 1));
                a0[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + offA0));
                a1[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + offA1));
                b0[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + offB0));
                b1[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + offB1));
            }
            // zero the m rows past the end of this split (lane holds m = kk*32 + 8g + j, j = 0..7)
            const int mb = kk * 32 + 8 * g;
            if (mb + 8 > mrem) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (mb + j >= mrem) { a0[i][j] = 0; b0[i][j] = 0; }
                        if (mb + 4 + j >= mrem) { a1[i][j] = 0; b1[i][j] = 0; }
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 bx = {b0[i][0], b0[i][1], b0[i][2], b0[i][3], b1[i][0], b1[i][1], b1[i][2], b1[i][3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const s16x8 ay = {a0[j][0], a0[j][1], a0[j][2], a0[j][3], a1[j][0], a1[j][1], a1[j][2], a1[j][3]};
                    // D[row <-> k_out (X^T as the A operand)][col <-> n_out (dY as the B operand)]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bx), __builtin_bit_cast(bf16x8, ay), acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    if (nt > 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int mt = 0; mt < nt - 1; ++mt) {
            stage(cur ^ 1, mt + 1);
            compute(cur, mt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        compute(cur, nt - 1);
    }

    const float alpha = p.alpha * (p.alpha_dev ? *p.alpha_dev : 1.0f);
    const int fr = lane & 15, fq = lane >> 4;
    float* out = p.splits > 1 ? p.slab + (size_t)split * p.N * p.ldw : p.W;
    const bool accum = (p.splits == 1) && p.accumulate;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + i * 16 + fq * 4;
        if (k >= p.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fr;
            if (n >= p.N) continue;
            float* dst = out + (size_t)n * p.ldw + k;
            float4 v = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
            if (accum) { const float4 o = *(const float4*)dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *(float4*)dst = v;
        }
    }
}

// W[n][k] (+)= sum_s slab[s][n][k]
__global__ void tn_reduce_kernel(float* __restrict__ W, const float* __restrict__ slab, int N, int K4, int ldw4, int splits, int accumulate) {
    const size_t total = (size_t)N * K4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K4), k4 = (int)(i % K4);
        const size_t off = (size_t)n * ldw4 + k4;
        float4 v = accumulate ? ((const float4*)W)[off] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < splits; ++s) {
            const float4 t = ((const float4*)slab)[(size_t)s * total + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        ((float4*)W)[off] = v;
    }
}

// column sums: out[n] += alpha * sum_m X[m][n]   (bias gradients); grid (N/256 x row-splits)
__global__ void colsum_kernel(const bf16_t* __restrict__ X, int M, int N, int ldx, float* __restrict__ out, float alpha, const float* alpha_dev, int rows_per_block) {
    const int n = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int sub = threadIdx.x >> 6;                                  // 4 waves take interleaved rows
    const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (int m = mbeg + sub; m < mend; m += 4) {
            const bf16x4 v = *(const bf16x4*)(X + (size_t)m * ldx + n);
            s[0] += bf2f(v[0]); s[1] += bf2f(v[1]); s[2] += bf2f(v[2]); s[3] += bf2f(v[3]);
        }
    }
    __shared__ float red[4][64][4];
    for (int r = 0; r < 4; ++r) red[sub][threadIdx.x & 63][r] = s[r];
    __syncthreads();
    if (sub == 0 && n < N) {
        const float a = alpha * (alpha_dev ? *alpha_dev : 1.0f);
        for (int r = 0; r < 4; ++r) {
            const float t = red[0][threadIdx.x][r] + red[1][threadIdx.x][r] + red[2][threadIdx.x][r] + red[3][threadIdx.x][r];
            atomicAdd(out + n + r, t * a);
        }
    }
}

extern "C" {

int mmbert_gemm_nt(hipStream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                   int M, int N, int K, int epi, const float* bias, const void* R, int ldr, void* aux, int ldaux,
                   const void* U, int ldu, float alpha, const float* alpha_dev,
                   uint32_t drop_stream, uint32_t drop_thr16, float drop_scale) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 63) || (N & 3) || (lda & 7) || (ldb & 7) || (ldc & 3)) return -1;
    GemmNT p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.R = (const bf16_t*)R;
    p.aux = (bf16_t*)aux; p.U = (const bf16_t*)U; p.alpha_dev = alpha_dev;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.ldu = ldu;
    p.alpha = alpha; p.drop_stream = drop_stream; p.drop_thr16 = drop_thr16; p.drop_scale = drop_scale;
    switch (epi) {
        case 0: return launch_nt<0>(stream, p);
        case EPI_BIAS: return launch_nt<EPI_BIAS>(stream, p);
        case EPI_BIAS | EPI_GELU: return launch_nt<EPI_BIAS | EPI_GELU>(stream, p);
        case EPI_BIAS | EPI_RESID: return launch_nt<EPI_BIAS | EPI_RESID>(stream, p);
        case EPI_RESID: return launch_nt<EPI_RESID>(stream, p);
        case EPI_GELU_BWD: return launch_nt<EPI_GELU_BWD>(stream, p);
        case EPI_OUT_F32: return launch_nt<EPI_OUT_F32>(stream, p);
        case EPI_BIAS | EPI_OUT_F32: return launch_nt<EPI_BIAS | EPI_OUT_F32>(stream, p);
        default: return -2;
    }
}

// returns the slab size in bytes the caller must provide for (M,N,K); 0 when no split is used
size_t mmbert_gemm_tn_workspace(int M, int N, int K, int* splits_out) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    int splits = 1;
    if (tiles < 512) splits = (512 + tiles - 1) / tiles;
    const int max_splits = (M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits_out) *splits_out = splits;
    return splits > 1 ? (size_t)splits * N * K * sizeof(float) : 0;
}

int mmbert_gemm_tn(hipStream_t stream, const void* A, int lda, const void* B, int ldb, float* W, int ldw,
                   int M, int N, int K, int accumulate, float alpha, const float* alpha_dev, void* slab) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || (ldw & 3) || N < 8 || K < 8) return -1;
    int splits;
    const size_t need = mmbert_gemm_tn_workspace(M, N, K, &splits);
    if (need && !slab) return -3;
    GemmTN p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.W = W; p.slab = (float*)slab;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.splits = splits;
    p.ldw = splits > 1 ? K : ldw;
    p.rows_per_split = (((M + splits - 1) / splits) + 63) / 64 * 64;
    p.accumulate = accumulate; p.alpha = alpha; p.alpha_dev = alpha_dev;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 65536, stream, p);
    MMB_CHECK_LAUNCH();
    if (splits > 1) {
        const size_t total = (size_t)N * (K / 4);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, W, (const float*)slab, N, K / 4, ldw / 4, splits, accumulate);
        MMB_CHECK_LAUNCH();
    }
    return 0;
}

int mmbert_colsum(hipStream_t stream, const void* X, int ldx, int M, int N, float* out, float alpha, const float* alpha_dev) {
    if (M <= 0 || N <= 0) return 0;
    if ((N & 3) || (ldx & 3)) return -1;
    const int gx = (N / 4 + 63) / 64;
    int gy = (1024 + gx - 1) / gx;
    int rows = (M + gy - 1) / gy; if (rows < 16) rows = 16;
    gy = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16_t*)X, M, N, ldx, out, alpha, alpha_dev, rows);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
