// The pretraining heads, one launch per DEPENDENCY LEVEL (round 6): everything downstream of the [CLS] rows -- pooler, alignment / next-sentence
// scores, the three gates, the gated concatenation, classifier1_1 / 1_2, the three CPC projections and their in-batch InfoNCE, the 2-way CE, the
// label loss, the joint loss (REF:MMBertForPretraining.py:293-301, 399-443; CPC REF:MMBertEmbedding.py:21-32; pooler HF:457-463) -- in SEVEN forward
// launches, and the whole hand-derived backward (input gradient of the [CLS] rows + every weight / bias gradient of those layers) in SIX.
// Before (csrc/heads.hip): 19 launches plus ~8 ATen launches around them (gather + cast of the rows, zero fills, cat, copy), ~210 us of a 13-ms
// step; the dense layers there are lists of 64-deep chunk products added with fp32 atomics (11-13 us per launch, a zero fill in front, a second
// "ordered" form for deterministic mode).  Here every level is ONE launch of 1024-thread workgroups:
//   * dense products with a long inner dimension (K = H .. 3H) are 16 x 16 output tiles on v_mfma_f32_16x16x4_f32 (exact fp32 products):
//     one tile per workgroup, K split over its 16 waves, partial tiles summed through LDS in wave order; activations (tanh), biases, the gated
//     concatenation and cat(x, x) ride in the epilogues / operand loaders;
//   * products with a short inner dimension (the batch: weight gradients, the CPC seeds) are one 16 x 16 tile per WAVE;
//   * row-wise pieces (gates, norms, softmax over the batch) are one wave per row; column sums one thread per column;
//   * the weight gradients ride in the backward level at which their operands are final.
// Operands come straight from global memory / L2 in MFMA fragment order (float4 along k per lane; no LDS staging).  No atomics on data: every
// sum has ONE owner and a fixed order, so the results do not depend on scheduling -- there is no separate deterministic form.  Every parameter
// gradient is ACCUMULATED (+=) into the caller's buffers (views of the flat gradient buffer).
//
// MEASURED AND WITHDRAWN (same round, profiles/r6_heads_persistent_kernel.txt): the same levels inside ONE persistent kernel per direction with grid
// barriers between them.  An MI355X has one L2 per XCD, so data exchanged between levels inside a launch needs agent-scope ("sc1") stores and
// loads, or cache-wide fences: a fence pair by one thread per workgroup costs 8 us per barrier, by every thread 131 us (!), agent-scope accesses
// + a two-level arrival 2.8 us (tools/ubench/barrier_rate.py, profiles/r6_ubench_barrier.txt) -- and each level then pays an sc1 store completion
// and an sc1 load latency (both go past the L2): 84 us forward / 135 us backward by rocprofv3, no better than the 19 launches.  A launch boundary
// costs the same ~5 us as that barrier + exchange and makes plain cached accesses legal, without a spinning grid that has to stay resident.
// Two findings from that form are kept under tools/ubench: HIP's __syncthreads() does not wait for outstanding global stores (a workgroup-scope
// release on one CU needs no vmcnt wait), and hipcc 22 turns __builtin_amdgcn_raw_buffer_load_b128 with the sc1 cache-policy bit into a ONE-dword
// load when its lanes are extracted (tools/ubench/exchange_check.py).
#include "common.h"
#include "../../include/mmbert_hip.h"

namespace {

constexpr int HC_THREADS = 1024, HC_WAVES = 16;

struct HcCtx { unsigned nwg; float* red; int wg; };

// what one level writes the next one (the next launch) reads: plain cached accesses; hc_ldc / hc_stc / mem mark the arrays exchanged between levels
struct HcMem {
    __device__ __forceinline__ float4 ld4(const float* q) const { return *(const float4*)q; }
    __device__ __forceinline__ void st4(float* q, float4 x) const { *(float4*)q = x; }
};
__device__ __forceinline__ float hc_ldc(const float* q) { return *q; }
__device__ __forceinline__ void hc_stc(float* q, float v) { *q = v; }
// the part sums of the loss level travel between workgroups of ONE launch: agent-scope atomics (coherent across the XCDs' L2s)
__device__ __forceinline__ float hc_lda(const float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hc_sta(float* q, float v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// __syncthreads() with every wave's global stores COMPLETE first.  HIP's __syncthreads() is a workgroup-scope release + s_barrier, and at
// workgroup scope (all waves of a workgroup share the CU's L1) hipcc does not wait for outstanding vector-memory operations: a thread that
// signals ANOTHER workgroup behind a plain __syncthreads() can overtake its neighbours' stores still in flight.
__device__ __forceinline__ void hc_sync_stores() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ f32x4 hc_mfma4(const float4 a, const float4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// One 16 x 16 tile per WORKGROUP: D[m][n] = sum_k A(m, k) B(k, n), K % 16 == 0, split over the 16 waves in 16-deep granules.
// la(row, k) / lb(col, k) return the four operand values at k .. k + 3 (lane (c, g) asks for row / col = 16 t + c and k = granule + 4 g);
// ep(row, col, value) is called once per tile element (thread = (row = tid >> 4, col = tid & 15), 64-byte runs per row).
// NB = 16-deep granules whose operand loads are in flight together (2 NB 16-byte loads per lane): 4 by default, 12 for the K = 3 H products,
// whose waves hold 9 (H = 768) to 12 (H = 1024) granules -- one round trip instead of three.
template <int NB = 4, class LA, class LB, class EP>
__device__ __forceinline__ void hc_wg_tile(const HcCtx& c, int tm, int tn, int K, LA la, LB lb, EP ep) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cc = lane & 15, g = lane >> 4;
    const int gran = K >> 4, gpw = (gran + HC_WAVES - 1) / HC_WAVES;
    const int g0 = wave * gpw, g1 = min(g0 + gpw, gran);
    const int m = tm * 16 + cc, n = tn * 16 + cc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = g0; q < g1; q += NB) {
        float4 a[NB], b[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int k = (min(q + u, g1 - 1) << 4) + 4 * g;
            a[u] = la(m, k);
            b[u] = lb(n, k);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (q + u < g1) acc = hc_mfma4(a[u], b[u], acc);           // (wave-uniform)
    }
    *(f32x4*)(c.red + wave * 256 + lane * 4) = acc;                    // acc[r] = D[4 g + r][cc]
    __syncthreads();
    if (tid < 256) {
        const int mm = tid >> 4, nn = tid & 15;
        const float* r = c.red + ((mm >> 2) * 16 + nn) * 4 + (mm & 3);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < HC_WAVES; ++w) s += r[w * 256];           // wave order: the same bits every run
        ep(tm * 16 + mm, tn * 16 + nn, s);
    }
    __syncthreads();
}

// One 16 x 16 tile per WAVE, short inner dimension K (any K >= 1: the loaders return 0 past it)
template <class LA, class LB, class EP>
__device__ __forceinline__ void hc_wave_tile(int tm, int tn, int K, LA la, LB lb, EP ep) {
    const int lane = threadIdx.x & 63, cc = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) acc = hc_mfma4(la(tm * 16 + cc, k0 + 4 * g), lb(tn * 16 + cc, k0 + 4 * g), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) ep(tm * 16 + 4 * g + r, tn * 16 + cc, acc[r]);
}
// ... two tiles of one job side by side (tile t and, when has2, tile t2; tiles_n column tiles per row of tiles): the second tile's operand loads
// and read-modify-write traffic overlap the first one's (a wave's tiles are otherwise a chain of dependent round trips)
template <class LA, class LB, class EP>
__device__ __forceinline__ void hc_wave_tile_pair(int t, int t2, bool has2, int tiles_n, int K, LA la, LB lb, EP ep) {
    const int lane = threadIdx.x & 63, cc = lane & 15, g = lane >> 4;
    const int tm = t / tiles_n, tn = t - tm * tiles_n, tm2 = t2 / tiles_n, tn2 = t2 - tm2 * tiles_n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        const float4 a = la(tm * 16 + cc, k0 + 4 * g), b = lb(tn * 16 + cc, k0 + 4 * g);
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a2;
        if (has2) { a2 = la(tm2 * 16 + cc, k0 + 4 * g); b2 = lb(tn2 * 16 + cc, k0 + 4 * g); }
        acc = hc_mfma4(a, b, acc);
        if (has2) acc2 = hc_mfma4(a2, b2, acc2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ep(tm * 16 + 4 * g + r, tn * 16 + cc, acc[r]);
    if (has2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ep(tm2 * 16 + 4 * g + r, tn2 * 16 + cc, acc2[r]);
    }
}
// p.arr[m] with a run-time m as two selects (a run-time index into the kernel-argument struct would send the whole struct to scratch memory)
template <class T> __device__ __forceinline__ T hc_pick(T const (&a)[3], int m) { return m == 0 ? a[0] : (m == 1 ? a[1] : a[2]); }
__device__ __forceinline__ float4 hc_ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 hc_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 hc_scale4(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
__device__ __forceinline__ float4 hc_add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// four rows k .. k + 3 of a column: W[(k + j) * ld + n]   (C: agent-scope loads, for arrays written earlier in this launch)
template <bool C = false>
__device__ __forceinline__ float4 hc_col4(const float* W, int ld, int k, int n) {
    const float* p = W + (size_t)k * ld + n;
    return make_float4(p[0], p[ld], p[2 * (size_t)ld], p[3 * (size_t)ld]);
}
// ... of which only the rows below `lim` exist (short inner dimensions)
template <bool C = false>
__device__ __forceinline__ float4 hc_col4_lim(const float* W, int ld, int k, int n, int lim) {
    const float* p = W + (size_t)min(k, lim - 1) * ld + n;
    const float* p1 = p + (size_t)ld * (k + 1 < lim ? 1 : 0); const float* p2 = p + (size_t)ld * (k + 2 < lim ? 2 : 0); const float* p3 = p + (size_t)ld * (k + 3 < lim ? 3 : 0);
    float v0, v1, v2, v3;
    v0 = p[0]; v1 = p1[0]; v2 = p2[0]; v3 = p3[0];
    return make_float4(k < lim ? v0 : 0.f, k + 1 < lim ? v1 : 0.f, k + 2 < lim ? v2 : 0.f, k + 3 < lim ? v3 : 0.f);
}

// workspace layout (floats); R = 3 B
struct HcWs {
    float *first, *P, *Apre, *XP, *T, *g, *nx, *ny, *lo, *S, *dS, *rsum, *csum, *part, *drel, *dlo;       // forward (kept for backward)
    float *dPc, *dXP, *dT, *dC, *dP0, *dA, *E, *dpre, *dgv;                                               // backward scratch
};
__host__ __device__ __forceinline__ HcWs hc_ws(int B, int H, float* base, size_t* total) {
    const size_t R = 3 * (size_t)B, RH = R * H, BH = (size_t)B * H, BB = 3 * (size_t)B * B;
    size_t o = 0;
    HcWs t;
#define HC_TAKE(field, n) t.field = base + o; o += ((size_t)(n) + 3) & ~(size_t)3;
    HC_TAKE(first, RH) HC_TAKE(P, RH) HC_TAKE(Apre, RH) HC_TAKE(XP, RH) HC_TAKE(T, BH) HC_TAKE(g, R) HC_TAKE(nx, R) HC_TAKE(ny, R)
    HC_TAKE(lo, B) HC_TAKE(S, BB) HC_TAKE(dS, BB) HC_TAKE(rsum, R) HC_TAKE(csum, R) HC_TAKE(part, 8) HC_TAKE(drel, 4 * (size_t)B) HC_TAKE(dlo, B)
    HC_TAKE(dPc, RH) HC_TAKE(dXP, RH) HC_TAKE(dT, BH) HC_TAKE(dC, 3 * BH) HC_TAKE(dP0, RH) HC_TAKE(dA, RH) HC_TAKE(E, RH) HC_TAKE(dpre, RH) HC_TAKE(dgv, R)
#undef HC_TAKE
    if (total) *total = o;
    return t;
}

// the [CLS] rows as the pooler reads them: fp32 matrix, or rows of the bf16 encoder output
struct HcFirst {
    const float* f; const bf16_t* y; const long long* rows; int ldy, H, R;
    __device__ __forceinline__ float4 operator()(int row, int k) const {
        row = min(row, R - 1);
        if (f) return hc_ld4(f + (size_t)row * H + k);
        const bf16x4 v = *(const bf16x4*)(y + (size_t)rows[row] * ldy + k);
        return make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
    }
};

}  // namespace

// =====================================================================================================================================
// forward
// =====================================================================================================================================
template <int LEVEL>
__global__ __launch_bounds__(HC_THREADS) void heads_fwd_level_kernel(const mmbert_heads_step p) {
#if defined(__gfx950__)
    extern __shared__ __attribute__((aligned(16))) float hc_sm[];
    HcCtx c; c.nwg = gridDim.x; c.red = hc_sm; c.wg = blockIdx.x;
    float* lds2 = hc_sm + HC_WAVES * 256;                              // behind the reduction buffer: the loss phase's partial sums
    const int B = p.B, H = p.H, R = 3 * B, tH = H >> 4, tB = (B + 15) >> 4, tR = (R + 15) >> 4, t2B = (2 * B + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const HcWs w = hc_ws(B, H, p.ws, nullptr);
    const HcMem mem = {};
    const HcFirst first = {p.first, (const bf16_t*)p.y, (const long long*)p.first_rows, p.ldy, H, R};

    if constexpr (LEVEL == 1) {
    // ---- F1: P = tanh(first Wp^T + bp)  (HF:457-463);  rel = first[B:] Wal^T + bal  (REF :297-298);  fp32 copy of gathered rows
    {
        const int nP = tR * tH;
        for (int t = c.wg; t < nP + t2B; t += c.nwg) {
            if (t < nP) {
                hc_wg_tile(c, t / tH, t % tH, H, first, [&](int n, int k) { return hc_ld4(p.Wp + (size_t)n * H + k); },
                           [&](int row, int n, float s) { if (row < R) hc_stc(w.P + (size_t)row * H + n, tanhf(s + p.bp[n])); });
            } else {
                hc_wg_tile(c, t - nP, 0, H, [&](int row, int k) { return first(B + row, k); },
                           [&](int n, int k) { return hc_ld4(p.Wal + (size_t)min(n, 1) * H + k); },
                           [&](int row, int n, float s) { if (row < 2 * B && n < 2) hc_stc(p.rel + row * 2 + n, s + p.bal[n]); });
            }
        }
        if (!p.first) {                                                  // (read by the backward launch only: plain stores)
            for (int row = c.wg * HC_WAVES + wave; row < R; row += c.nwg * HC_WAVES)
                for (int k = 4 * lane; k < H; k += 256) *(float4*)(w.first + (size_t)row * H + k) = first(row, k);
        }
    }
    }
    if constexpr (LEVEL == 2) {
    // ---- F2: Apre = attn(cat(P, P)) = P (W[:, :H] + W[:, H:])^T + b  (REF :407-409);  t_rel = P[:B] Wsr^T + bsr  (REF :301: never in a loss)
    {
        const int nA = tR * tH;
        for (int t = c.wg; t < nA + tB; t += c.nwg) {
            if (t < nA) {
                hc_wg_tile(c, t / tH, t % tH, H, [&](int row, int k) { return mem.ld4(w.P + (size_t)min(row, R - 1) * H + k); },
                           [&](int n, int k) { const float* q = p.Wat + (size_t)n * 2 * H + k; return hc_add4(hc_ld4(q), hc_ld4(q + H)); },
                           [&](int row, int n, float s) { if (row < R) hc_stc(w.Apre + (size_t)row * H + n, s + p.bat[n]); });
            } else {
                hc_wg_tile(c, t - nA, 0, H, [&](int row, int k) { return mem.ld4(w.P + (size_t)min(row, B - 1) * H + k); },
                           [&](int n, int k) { return hc_ld4(p.Wsr + (size_t)min(n, 1) * H + k); },
                           [&](int row, int n, float s) { if (row < B && n < 2) p.t_rel[row * 2 + n] = s + p.bsr[n]; });
            }
        }
    }
    }
    if constexpr (LEVEL == 3) {
    // ---- F3 (rows): g[m,b] = relu(Apre[m,b,:]) . v_m + vb_m  (REF :407-409);  nx = |P[m,b,:]|
    for (int row = c.wg * HC_WAVES + wave; row < R; row += c.nwg * HC_WAVES) {
        const int m = row / B;
        const float* vw = hc_pick(p.vw, m);
        float sg = 0.f, sn = 0.f;
        for (int k = 4 * lane; k < H; k += 256) {
            const float4 a = mem.ld4(w.Apre + (size_t)row * H + k), v = hc_ld4(vw + k), x = mem.ld4(w.P + (size_t)row * H + k);
            sg += fmaxf(a.x, 0.f) * v.x + fmaxf(a.y, 0.f) * v.y + fmaxf(a.z, 0.f) * v.z + fmaxf(a.w, 0.f) * v.w;
            sn += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        sg = wave_sum(sg); sn = wave_sum(sn);
        if (lane == 0) { hc_stc(w.g + row, sg + hc_pick(p.vb, m)[0]); hc_stc(w.nx + row, sqrtf(sn)); }
    }
    }
    if constexpr (LEVEL == 4) {
    // ---- F4: T = classifier1_1(cat_m(P_m * g_m))  (REF :411-414), the gated concatenation formed by the operand loader
    for (int t = c.wg; t < tB * tH; t += c.nwg) {
        hc_wg_tile<10>(c, t / tH, t % tH, 3 * H,
                   [&](int b, int k) { const int m = k / H, row = m * B + min(b, B - 1); return hc_scale4(mem.ld4(w.P + (size_t)row * H + (k - m * H)), hc_ldc(w.g + row)); },
                   [&](int n, int k) { return hc_ld4(p.Wc1 + (size_t)n * 3 * H + k); },
                   [&](int b, int n, float s) { if (b < B) hc_stc(w.T + (size_t)b * H + n, s + p.bc1[n]); });
    }
    }
    if constexpr (LEVEL == 5) {
    // ---- F5: XP_m = cpc_m.net(T)  (REF:MMBertEmbedding.py:22);  lo = classifier1_2(T)  (REF :415)
    {
        const int nX = tB * 3 * tH;
        for (int t = c.wg; t < nX + tB; t += c.nwg) {
            if (t < nX) {
                const int tb = t / (3 * tH), tn = t % (3 * tH), m = tn / tH;
                const float* Wq = hc_pick(p.Wq, m); const float* bq = hc_pick(p.bq, m);
                hc_wg_tile(c, tb, tn, H, [&](int b, int k) { return mem.ld4(w.T + (size_t)min(b, B - 1) * H + k); },
                           [&](int n, int k) { return hc_ld4(Wq + (size_t)(n - m * H) * H + k); },
                           [&](int b, int n, float s) { if (b < B) hc_stc(w.XP + ((size_t)m * B + b) * H + (n - m * H), s + bq[n - m * H]); });
            } else {
                hc_wg_tile(c, t - nX, 0, H, [&](int b, int k) { return mem.ld4(w.T + (size_t)min(b, B - 1) * H + k); },
                           [&](int n, int k) { return hc_ld4(p.Wc2 + k); },
                           [&](int b, int n, float s) {
                               if (b < B && n == 0) { const float v = s + p.bc2[0]; hc_stc(w.lo + b, v); p.logits[b] = p.tanh_lo ? tanhf(v) : v; }
                           });
            }
        }
    }
    }
    if constexpr (LEVEL == 6) {
    // ---- F6: raw similarities S_m = P_m XP_m^T  [B, B] per modality;  ny = |XP[m,b,:]|
    {
        const int nS = 3 * tB * tB;
        for (int t = c.wg; t < nS; t += c.nwg) {
            const int m = t / (tB * tB), r = t - m * tB * tB;
            hc_wg_tile(c, r / tB, r % tB, H, [&](int b, int k) { return mem.ld4(w.P + ((size_t)m * B + min(b, B - 1)) * H + k); },
                       [&](int b2, int k) { return mem.ld4(w.XP + ((size_t)m * B + min(b2, B - 1)) * H + k); },
                       [&](int b, int b2, float s) { if (b < B && b2 < B) hc_stc(w.S + ((size_t)m * B + b) * B + b2, s); });
        }
        for (int row = c.wg * HC_WAVES + wave; row < R; row += c.nwg * HC_WAVES) {
            float sn = 0.f;
            for (int k = 4 * lane; k < H; k += 256) { const float4 x = mem.ld4(w.XP + (size_t)row * H + k); sn += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
            sn = wave_sum(sn);
            if (lane == 0) hc_stc(w.ny + row, sqrtf(sn));
        }
    }
    }
    if constexpr (LEVEL == 7) {
    // ---- F7: per modality (workgroups 0-2): cosine similarities, row logsumexp, InfoNCE part, dS and the sums of S o dS (the seeds of the
    // backward; REF:MMBertEmbedding.py:24-31); workgroup 3: alignment CE (REF :428), label loss (REF :430-441) and their seeds; the last of the
    // four to finish assembles the losses (REF :427, :443)
    if (c.wg < 3) {
        // (the [B, B] matrix stays in global memory -- normalised in place --: in LDS it would be 66 KB at B = 128, and a workgroup of this
        // kernel must stay small enough for TWO per CU: two processes' heads kernels may share the GPU, and a grid that spins on a barrier
        // must always fit beside another one)
        const int m = c.wg;
        float* S = w.S + (size_t)m * B * B; float* colp = lds2; float* nw = colp + HC_WAVES * B;
        for (int i = tid; i < B * B; i += HC_THREADS) {
            const int b = i / B, j = i - b * B;
            hc_stc(S + i, hc_ldc(S + i) / (hc_ldc(w.nx + m * B + b) * hc_ldc(w.ny + m * B + j)));
        }
        hc_sync_stores();
        const float wgt = -p.beta / (float)B;
        float cs[2] = {0.f, 0.f}, nacc = 0.f;
        for (int b = wave; b < B; b += HC_WAVES) {
            float sv[2] = {0.f, 0.f}, mx = -INFINITY;
            for (int u = 0; u < 2; ++u) { const int j = lane + 64 * u; if (j < B) { sv[u] = hc_ldc(S + b * B + j); mx = fmaxf(mx, sv[u]); } }
            mx = wave_max(mx);
            float se = 0.f;
            for (int u = 0; u < 2; ++u) { const int j = lane + 64 * u; if (j < B) se += expf(sv[u] - mx); }
            se = wave_sum(se);
            float rs = 0.f, diag = 0.f;
            for (int u = 0; u < 2; ++u) {
                const int j = lane + 64 * u;
                if (j < B) {
                    const float ds = wgt * (expf(sv[u] - mx) / se - (j == b ? 1.f : 0.f));
                    w.dS[((size_t)m * B + b) * B + j] = ds;
                    rs += ds * sv[u]; cs[u] += ds * sv[u];
                    if (j == b) diag = sv[u];
                }
            }
            rs = wave_sum(rs); diag = wave_sum(diag);
            if (lane == 0) { w.rsum[m * B + b] = rs; nacc += (mx + logf(se) - diag) / (float)B; }
        }
        for (int u = 0; u < 2; ++u) { const int j = lane + 64 * u; if (j < B) colp[wave * B + j] = cs[u]; }
        if (lane == 0) nw[wave] = nacc;
        __syncthreads();
        if (tid < B) {
            float s = 0.f;
            for (int q = 0; q < HC_WAVES; ++q) s += colp[q * B + tid];
            w.csum[m * B + tid] = s;
        }
        if (tid == 0) {
            float s = 0.f;
            for (int q = 0; q < HC_WAVES; ++q) s += nw[q];
            hc_sta(w.part + m, s);
        }
    } else if (c.wg == 3) {
        float* red = lds2;
        float ce = 0.f, se = 0.f;
        for (int i = tid; i < 2 * B; i += HC_THREADS) {
            const float a = hc_ldc(p.rel + i * 2), cc = hc_ldc(p.rel + i * 2 + 1);
            const float mx = fmaxf(a, cc), lse = mx + logf(expf(a - mx) + expf(cc - mx));
            const int y = (int)((p.ap2 && i >= B) ? p.ap2[i - B] : p.ap[i]);
            const float sc = 0.5f / (float)B;
            ce += (lse - (y ? cc : a)) * sc;
            w.drel[i * 2] = sc * (expf(a - lse) - (y == 0 ? 1.f : 0.f));
            w.drel[i * 2 + 1] = sc * (expf(cc - lse) - (y == 1 ? 1.f : 0.f));
        }
        for (int b = tid; b < B; b += HC_THREADS) {
            const float lo = hc_ldc(w.lo + b);
            const float v = p.tanh_lo ? tanhf(lo) : lo;
            const float d = v - p.sent[b];
            se += d * d / (float)B;
            w.dlo[b] = 2.f * d / (float)B * (p.tanh_lo ? 1.f - v * v : 1.f);
        }
        ce = wave_sum(ce); se = wave_sum(se);
        if (lane == 0) { red[wave] = ce; red[HC_WAVES + wave] = se; }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f, b2 = 0.f;
            for (int q = 0; q < HC_WAVES; ++q) { a += red[q]; b2 += red[HC_WAVES + q]; }
            hc_sta(w.part + 3, a); hc_sta(w.part + 4, b2);
        }
    }
    if (c.wg < 4) {
        hc_sync_stores();                                                // (the part stores above have completed)
        if (tid == 0 && atomicAdd(p.sync + 1, 1u) == 3u) {               // the last of the four: every part is written
            const float nce = hc_lda(w.part) + hc_lda(w.part + 1) + hc_lda(w.part + 2), ce = hc_lda(w.part + 3), se = hc_lda(w.part + 4);
            const float heads = ce + se - p.beta * nce;
            float ms = 0.f;
            for (int i = 0; i < p.nmlm; ++i) ms += p.mlm[i];
            const float joint = p.nmlm > 0 ? p.alpha * (ms / (float)p.nmlm) + heads : heads;
            p.out5[0] = ce; p.out5[1] = se; p.out5[2] = nce; p.out5[3] = heads; p.out5[4] = joint;
            *p.loss = joint;
            p.aux[0] = ce; p.aux[1] = se; p.aux[2] = nce;
            atomicExch(p.sync + 1, 0u);
        }
    }
    }
#endif
}

// =====================================================================================================================================
// backward.  d = the upstream gradient of the joint loss (device scalar); everything below is linear in it, so it is applied once, where a
// result leaves: dfirst, every parameter gradient (+=), dmlm.
//   dS = (-beta / B)(softmax_b'(S[b]) - delta)                       (saved by forward, with rsum = rows of S o dS, csum = its columns)
//   dPc  = (dS XPn - Xn o rsum) / nx      dXP = (dS^T Xn - XPn o csum) / ny           (through y = x / |x|: dx = (dy - y <y, dy>) / |x|)
//   dT = sum_m dXP_m Wq_m + dlo Wc2       dC = dT Wc1       dg = <dC_m, P_m>       dP0 = dC_m g + dPc
//   dA = dg v_m (Apre > 0)                E = dg relu(Apre)  dP = dP0 + dA (W1 + W2)           dpre = dP (1 - P^2)
//   dfirst = dpre Wp + [rows >= B] drel Wal
// =====================================================================================================================================
template <int LEVEL>
__global__ __launch_bounds__(HC_THREADS) void heads_bwd_level_kernel(const mmbert_heads_step p) {
#if defined(__gfx950__)
    extern __shared__ __attribute__((aligned(16))) float hc_sm[];
    HcCtx c; c.nwg = gridDim.x; c.red = hc_sm; c.wg = blockIdx.x;
    const int B = p.B, H = p.H, R = 3 * B, tH = H >> 4, tB = (B + 15) >> 4, tR = (R + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int gwave = c.wg * HC_WAVES + wave, nwave = c.nwg * HC_WAVES;
    const int gthread = c.wg * HC_THREADS + tid, nthread = c.nwg * HC_THREADS;
    // wave jobs and column jobs are handed out from the LAST workgroup down: the first ones hold the level's workgroup tiles
    const int rwave = nwave - 1 - gwave, rthread = nthread - 1 - gthread;
    const HcWs w = hc_ws(B, H, p.ws, nullptr);
    const HcMem mem = {};
    const float* firstf = p.first ? p.first : w.first;
    const float d = *p.dloss;
    // (what the FORWARD launch left in the workspace -- P, Apre, XP, T, g, nx, ny, dS, rsum, csum, drel, dlo, first -- is read with plain loads;
    //  what this launch writes in one phase and reads in a later one -- dPc, dXP, dT, dC, dP0, dA, E, dgv, dpre -- with agent-scope accesses)

    if constexpr (LEVEL == 1) {
    // ---- B1 (wave tiles, inner = batch): the CPC seeds dPc and dXP
    {
        const int per = tB * tH;
        for (int t = gwave; t < 6 * per; t += nwave) {
            const int which = t / (3 * per), r0 = t - which * 3 * per, m = r0 / per, r = r0 - m * per, tb = r / tH, tn = r - tb * tH;
            const float* dS = w.dS + (size_t)m * B * B;
            if (which == 0) {
                hc_wave_tile(tb, tn, B,
                             [&](int b, int j) { const float* q = dS + (size_t)min(b, B - 1) * B;
                                                 return make_float4(j < B ? q[min(j, B - 1)] : 0.f, j + 1 < B ? q[min(j + 1, B - 1)] : 0.f,
                                                                    j + 2 < B ? q[min(j + 2, B - 1)] : 0.f, j + 3 < B ? q[min(j + 3, B - 1)] : 0.f); },
                             [&](int n, int j) { float4 v = hc_col4_lim(w.XP + (size_t)m * B * H, H, j, n, B);
                                                 const float* ny = w.ny + m * B;
                                                 v.x /= ny[min(j, B - 1)]; v.y /= ny[min(j + 1, B - 1)]; v.z /= ny[min(j + 2, B - 1)]; v.w /= ny[min(j + 3, B - 1)];
                                                 return v; },
                             [&](int b, int n, float s) {
                                 if (b < B) { const int row = m * B + b; const float nx = w.nx[row];
                                              hc_stc(w.dPc + (size_t)row * H + n, (s - w.P[(size_t)row * H + n] / nx * w.rsum[row]) / nx); } });
            } else {
                hc_wave_tile(tb, tn, B,
                             [&](int b2, int j) { return hc_col4_lim(dS, B, j, min(b2, B - 1), B); },          // dS^T: dS[j][b2]
                             [&](int n, int j) { float4 v = hc_col4_lim(w.P + (size_t)m * B * H, H, j, n, B);
                                                 const float* nx = w.nx + m * B;
                                                 v.x /= nx[min(j, B - 1)]; v.y /= nx[min(j + 1, B - 1)]; v.z /= nx[min(j + 2, B - 1)]; v.w /= nx[min(j + 3, B - 1)];
                                                 return v; },
                             [&](int b2, int n, float s) {
                                 if (b2 < B) { const int row = m * B + b2; const float ny = w.ny[row];
                                               hc_stc(w.dXP + (size_t)row * H + n, (s - w.XP[(size_t)row * H + n] / ny * w.csum[row]) / ny); } });
            }
        }
    }
    }
    if constexpr (LEVEL == 2) {
    // ---- B2: dT = sum_m dXP_m Wq_m + dlo Wc2;  weight gradients of the CPC projections (their operands are final)
    {
        for (int t = c.wg; t < tB * tH; t += c.nwg) {
            hc_wg_tile<12>(c, t / tH, t % tH, 3 * H,
                       [&](int b, int k) { const int m = k / H; return mem.ld4(w.dXP + ((size_t)m * B + min(b, B - 1)) * H + (k - m * H)); },
                       [&](int n, int k) { const int m = k / H; return hc_col4(hc_pick(p.Wq, m), H, k - m * H, n); },
                       [&](int b, int n, float s) { if (b < B) hc_stc(w.dT + (size_t)b * H + n, s + w.dlo[b] * p.Wc2[n]); });
        }
        for (int t = rwave; t < 3 * tH * tH; t += 2 * nwave) {             // gWq_m[n][k] += d sum_b dXP_m[b][n] T[b][k]   (rows n' = m H + n of one 3H x H job)
            hc_wave_tile_pair(t, t + nwave, t + nwave < 3 * tH * tH, tH, B,
                              [&](int n, int b) { const int m = n / H; return hc_col4_lim<true>(w.dXP + (size_t)m * B * H, H, b, n - m * H, B); },
                              [&](int k, int b) { return hc_col4_lim(w.T, H, b, k, B); },
                              [&](int n, int k, float s) { const int m = n / H; hc_pick(p.gWq, m)[(size_t)(n - m * H) * H + k] += d * s; });
        }
        for (int i = rthread; i < 3 * H; i += nthread) {                   // gbq_m[n] += d sum_b dXP_m[b][n]
            const int m = i / H, n = i - m * H;
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += hc_ldc(w.dXP + ((size_t)m * B + b) * H + n);
            hc_pick(p.gbq, m)[n] += d * s;
        }
    }
    }
    if constexpr (LEVEL == 3) {
    // ---- B3: dC = dT Wc1;  weight gradients of classifier1_1 / classifier1_2
    {
        for (int t = c.wg; t < tB * 3 * tH; t += c.nwg) {
            hc_wg_tile(c, t / (3 * tH), t % (3 * tH), H, [&](int b, int k) { return mem.ld4(w.dT + (size_t)min(b, B - 1) * H + k); },
                       [&](int n, int k) { return hc_col4(p.Wc1, 3 * H, k, n); },
                       [&](int b, int n, float s) { if (b < B) hc_stc(w.dC + (size_t)b * 3 * H + n, s); });
        }
        for (int t = rwave; t < tH * 3 * tH; t += 2 * nwave) {             // gWc1[n][k'] += d sum_b dT[b][n] C[b][k'],  C = gated concatenation
            hc_wave_tile_pair(t, t + nwave, t + nwave < tH * 3 * tH, 3 * tH, B, [&](int n, int b) { return hc_col4_lim<true>(w.dT, H, b, n, B); },
                         [&](int k, int b) { const int m = k / H; float4 v = hc_col4_lim(w.P + (size_t)m * B * H, H, b, k - m * H, B);
                                             const float* g = w.g + m * B;
                                             v.x *= g[min(b, B - 1)]; v.y *= g[min(b + 1, B - 1)]; v.z *= g[min(b + 2, B - 1)]; v.w *= g[min(b + 3, B - 1)];
                                             return v; },
                         [&](int n, int k, float s) { p.gWc1[(size_t)n * 3 * H + k] += d * s; });
        }
        for (int i = rthread; i < 2 * H + 1; i += nthread) {
            float s = 0.f;
            if (i < H) { for (int b = 0; b < B; ++b) s += hc_ldc(w.dT + (size_t)b * H + i); p.gbc1[i] += d * s; }
            else if (i < 2 * H) { const int k = i - H; for (int b = 0; b < B; ++b) s += w.dlo[b] * w.T[(size_t)b * H + k]; p.gWc2[k] += d * s; }
            else { for (int b = 0; b < B; ++b) s += w.dlo[b]; p.gbc2[0] += d * s; }
        }
    }
    }
    if constexpr (LEVEL == 4) {
    // ---- B4 (rows): through the gates
    for (int row = gwave; row < R; row += nwave) {
        const int lane = tid & 63, m = row / B, b = row - m * B;
        const float* dc = w.dC + (size_t)b * 3 * H + (size_t)m * H;
        const float* vw = hc_pick(p.vw, m);
        float s = 0.f;
        for (int k = 4 * lane; k < H; k += 256) {
            const float4 a = mem.ld4(dc + k), x = hc_ld4(w.P + (size_t)row * H + k);
            s += a.x * x.x + a.y * x.y + a.z * x.z + a.w * x.w;
        }
        const float dg = wave_sum(s), gv = w.g[row];
        if (lane == 0) hc_stc(w.dgv + row, dg);
        for (int k = 4 * lane; k < H; k += 256) {
            const float4 a = mem.ld4(dc + k), pc = mem.ld4(w.dPc + (size_t)row * H + k), ap = hc_ld4(w.Apre + (size_t)row * H + k), v = hc_ld4(vw + k);
            mem.st4(w.dP0 + (size_t)row * H + k, make_float4(a.x * gv + pc.x, a.y * gv + pc.y, a.z * gv + pc.z, a.w * gv + pc.w));
            mem.st4(w.dA + (size_t)row * H + k, make_float4(ap.x > 0.f ? dg * v.x : 0.f, ap.y > 0.f ? dg * v.y : 0.f, ap.z > 0.f ? dg * v.z : 0.f, ap.w > 0.f ? dg * v.w : 0.f));
            mem.st4(w.E + (size_t)row * H + k, make_float4(dg * fmaxf(ap.x, 0.f), dg * fmaxf(ap.y, 0.f), dg * fmaxf(ap.z, 0.f), dg * fmaxf(ap.w, 0.f)));
        }
    }
    }
    if constexpr (LEVEL == 5) {
    // ---- B5: dP = dP0 + dA (W1 + W2), dpre = dP (1 - P^2);  gradients of attn and of the three gate vectors
    {
        for (int t = c.wg; t < tR * tH; t += c.nwg) {
            hc_wg_tile(c, t / tH, t % tH, H, [&](int row, int k) { return mem.ld4(w.dA + (size_t)min(row, R - 1) * H + k); },
                       [&](int n, int k) { return hc_add4(hc_col4(p.Wat, 2 * H, k, n), hc_col4(p.Wat, 2 * H, k, H + n)); },
                       [&](int row, int n, float s) {
                           if (row < R) { const float x = w.P[(size_t)row * H + n]; hc_stc(w.dpre + (size_t)row * H + n, (s + hc_ldc(w.dP0 + (size_t)row * H + n)) * (1.f - x * x)); } });
        }
        for (int t = rwave; t < tH * tH; t += 2 * nwave) {                  // gWat[n][k] and [n][H + k] += d sum_rows dA[row][n] P[row][k]
            hc_wave_tile_pair(t, t + nwave, t + nwave < tH * tH, tH, R, [&](int n, int r) { return hc_col4_lim<true>(w.dA, H, r, n, R); },
                         [&](int k, int r) { return hc_col4_lim(w.P, H, r, k, R); },
                         [&](int n, int k, float s) { float* q = p.gWat + (size_t)n * 2 * H + k; q[0] += d * s; q[H] += d * s; });
        }
        for (int i = rthread; i < 4 * H + 3; i += nthread) {
            float s = 0.f;
            if (i < H) { for (int r = 0; r < R; ++r) s += hc_ldc(w.dA + (size_t)r * H + i); p.gbat[i] += d * s; }
            else if (i < 4 * H) { const int m = (i - H) / H, k = i - H - m * H; for (int b = 0; b < B; ++b) s += hc_ldc(w.E + ((size_t)m * B + b) * H + k); hc_pick(p.gvw, m)[k] += d * s; }
            else { const int m = i - 4 * H; for (int b = 0; b < B; ++b) s += hc_ldc(w.dgv + m * B + b); hc_pick(p.gvb, m)[0] += d * s; }
        }
    }
    }
    if constexpr (LEVEL == 6) {
    // ---- B6: dfirst = d (dpre Wp + [rows >= B] drel Wal);  gradients of the pooler and of align;  the gradient of the per-pass MLM losses
    {
        for (int t = c.wg; t < tR * tH; t += c.nwg) {
            hc_wg_tile(c, t / tH, t % tH, H, [&](int row, int k) { return mem.ld4(w.dpre + (size_t)min(row, R - 1) * H + k); },
                       [&](int n, int k) { return hc_col4(p.Wp, H, k, n); },
                       [&](int row, int n, float s) {
                           if (row < R) {
                               if (row >= B) s += w.drel[(row - B) * 2] * p.Wal[n] + w.drel[(row - B) * 2 + 1] * p.Wal[H + n];
                               p.dfirst[(size_t)row * H + n] = d * s;
                           } });
        }
        for (int t = rwave; t < tH * tH; t += 2 * nwave) {                  // gWp[n][k] += d sum_rows dpre[row][n] first[row][k]
            hc_wave_tile_pair(t, t + nwave, t + nwave < tH * tH, tH, R, [&](int n, int r) { return hc_col4_lim<true>(w.dpre, H, r, n, R); },
                         [&](int k, int r) { return hc_col4_lim(firstf, H, r, k, R); },
                         [&](int n, int k, float s) { p.gWp[(size_t)n * H + k] += d * s; });
        }
        for (int i = rthread; i < 3 * H + 2 + p.nmlm; i += nthread) {
            float s = 0.f;
            if (i < H) { for (int r = 0; r < R; ++r) s += hc_ldc(w.dpre + (size_t)r * H + i); p.gbp[i] += d * s; }
            else if (i < 3 * H) { const int n = (i - H) / H, k = i - H - n * H; for (int q = 0; q < 2 * B; ++q) s += w.drel[q * 2 + n] * firstf[(size_t)(B + q) * H + k]; p.gWal[(size_t)n * H + k] += d * s; }
            else if (i < 3 * H + 2) { const int n = i - 3 * H; for (int q = 0; q < 2 * B; ++q) s += w.drel[q * 2 + n]; p.gbal[n] += d * s; }
            else p.dmlm[i - 3 * H - 2] = d * (p.alpha / (float)p.nmlm);
        }
    }
    }
#endif
}

extern "C" {

int mmbert_heads_step_struct_size(void) { return (int)sizeof(mmbert_heads_step); }

size_t mmbert_heads_step_workspace(int B, int H) {
    if (B <= 0 || H <= 0) return 0;
    size_t total = 0;
    hc_ws(B, H, nullptr, &total);
    return total * sizeof(float);
}

static int hc_check(const mmbert_heads_step* p, int* cus) {
    if (!p || p->B <= 0) return -1;
    if (p->B > 128 || p->H < 16 || (p->H & 15) || p->nmlm < 0 || p->nmlm > 256 || !p->ws || !p->sync || (!p->first && (!p->y || !p->first_rows || (p->ldy & 3)))) return -1;
    *cus = mmb_device_cus();
    return 0;
}
__global__ void heads_dmlm_kernel(const float* __restrict__ dloss, float* __restrict__ dmlm, int nmlm, float alpha) {
    const float d = *dloss;
    for (int i = threadIdx.x; i < nmlm; i += blockDim.x) dmlm[i] = d * (alpha / (float)nmlm);      // (the expression of backward level 6)
}
// a level's grid: its 16 x 16 workgroup tiles PLUS enough 16-wave workgroups for its wave jobs (handed out from the last workgroup down, so that
// a workgroup holds a tile or wave jobs, not both in a row), at most one round of the chip
static inline int hc_grid(int wg_tiles, int wave_jobs, int cus, int floor_) {
    int g = wg_tiles + (wave_jobs + HC_WAVES - 1) / HC_WAVES;
    if (g < floor_) g = floor_;
    return g > cus ? cus : (g < 1 ? 1 : g);
}
#define HC_LAUNCH(K, L, GRID, LDS) do { hipLaunchKernelGGL((K<L>), dim3(GRID), dim3(HC_THREADS), (LDS), stream, *p); MMB_CHECK_LAUNCH(); } while (0)

// levels lo .. hi of the forward (1 .. 7) / backward (1 .. 6): the model runs the heads beside the MLM head's launches on a side stream and
// only the loss level (which reads the MLM losses) behind both
int mmbert_heads_step_fwd_levels(hipStream_t stream, const mmbert_heads_step* p, int lo, int hi) {
    int cus;
    if (hc_check(p, &cus)) return -1;
    if (lo < 1 || hi > 7 || lo > hi) return -1;
    if (!p->loss || !p->aux || !p->out5 || !p->logits || !p->t_rel || !p->rel || !p->ap || !p->sent || (hi == 7 && p->nmlm > 0 && !p->mlm)) return -1;
    const int B = p->B, H = p->H, R = 3 * B, tH = H >> 4, tB = (B + 15) >> 4, tR = (R + 15) >> 4, t2B = (2 * B + 15) >> 4;
    const int red = HC_WAVES * 256 * (int)sizeof(float);
    if (lo <= 1 && 1 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 1, hc_grid(tR * tH + t2B, p->first ? 0 : R, cus, 1), red);
    if (lo <= 2 && 2 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 2, hc_grid(tR * tH + tB, 0, cus, 1), red);
    if (lo <= 3 && 3 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 3, hc_grid(0, R, cus, 1), red);
    if (lo <= 4 && 4 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 4, hc_grid(tB * tH, 0, cus, 1), red);
    if (lo <= 5 && 5 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 5, hc_grid(tB * 3 * tH + tB, 0, cus, 1), red);
    if (lo <= 6 && 6 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 6, hc_grid(3 * tB * tB, R, cus, 1), red);
    if (lo <= 7 && 7 <= hi) HC_LAUNCH(heads_fwd_level_kernel, 7, 4, red + (int)((HC_WAVES * (size_t)B + 2 * HC_WAVES) * sizeof(float)));
    return 0;
}

int mmbert_heads_step_fwd(hipStream_t stream, const mmbert_heads_step* p) { return mmbert_heads_step_fwd_levels(stream, p, 1, 7); }

int mmbert_heads_step_bwd_levels(hipStream_t stream, const mmbert_heads_step* p, int lo, int hi) {
    int cus;
    if (hc_check(p, &cus)) return -1;
    if (lo < 1 || hi > 6 || lo > hi) return -1;
    if (!p->dloss || !p->dfirst || (p->nmlm > 0 && !p->dmlm)) return -1;
    const int B = p->B, H = p->H, R = 3 * B, tH = H >> 4, tB = (B + 15) >> 4, tR = (R + 15) >> 4;
    const int red = HC_WAVES * 256 * (int)sizeof(float);
    const int cols = (4 * H + 3 + HC_THREADS - 1) / HC_THREADS;         // workgroups that cover the widest column-sum job with one thread per column
    if (lo <= 1 && 1 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 1, hc_grid(0, 6 * tB * tH, cus, 1), red);
    if (lo <= 2 && 2 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 2, hc_grid(tB * tH, 3 * tH * tH, cus, cols), red);
    if (lo <= 3 && 3 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 3, hc_grid(tB * 3 * tH, tH * 3 * tH, cus, cols), red);
    if (lo <= 4 && 4 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 4, hc_grid(0, R, cus, 1), red);
    if (lo <= 5 && 5 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 5, hc_grid(tR * tH, tH * tH, cus, cols), red);
    if (lo <= 6 && 6 <= hi) HC_LAUNCH(heads_bwd_level_kernel, 6, hc_grid(tR * tH, tH * tH, cus, cols), red);
    return 0;
}

int mmbert_heads_step_bwd(hipStream_t stream, const mmbert_heads_step* p) { return mmbert_heads_step_bwd_levels(stream, p, 1, 6); }

// the gradient of the per-pass MLM losses alone (what backward level 6 also writes): d(joint) / d(mlm[i]) = alpha / nmlm, times the upstream
// gradient -- one tiny launch, so that the MLM head's backward can start on the current stream while the six levels run on another
int mmbert_heads_step_dmlm(hipStream_t stream, const mmbert_heads_step* p) {
    if (!p || p->nmlm <= 0) return 0;
    if (!p->dloss || !p->dmlm) return -1;
    hipLaunchKernelGGL(heads_dmlm_kernel, dim3(1), dim3(64), 0, stream, p->dloss, p->dmlm, p->nmlm, p->alpha);
    MMB_CHECK_LAUNCH();
    return 0;
}
#undef HC_LAUNCH

}  // extern "C"
