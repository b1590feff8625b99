// Pretraining heads of MMBertForPretraining (REF:MMBertForPretraining.py:293-301, 399-443; CPC REF:MMBertEmbedding.py:21-32):
// the fp32, [B,H]-sized arithmetic BETWEEN the dense products -- gates, gated concatenation, CPC normalisation / in-batch
// InfoNCE, the 2-way and regression losses, and their hand-derived backward.  The dense products themselves stay with the
// host (hipBLASLt through torch.addmm): what this file replaces is ~200 element-wise / reduction launches of 2-4 us each
// (1.1 ms of device time per step) by 6 kernels.  Everything here is HBM/latency trivial: a few [48, 768] fp32 arrays.
//
// Notation (B samples, H hidden, m = modality 0 text / 1 visual / 2 speech, rows of every [3B, H] array ordered m-major):
//   P   = tanh(first Wp^T + bp)                      pooled rows                 (host: addmm + tanh)
//   Apre= P [W1|W2]^T + bat                          attn(cat(x, x))             (host: two addmm)
//   g[m,b]   = relu(Apre[m,b,:]) . v_m + vb_m        gate                        heads_gate_fwd
//   C[b, mH+k] = P[m,b,k] * g[m,b]                   gated concatenation         heads_gate_fwd
//   T = C Wc1^T + bc1,  lo = T Wc2^T + bc2,  XP[m] = T Wq_m^T + bq_m            (host)
//   nce = sum_m mean_b ( logsumexp_b' <Xn[m,b], XPn[m,b']> - <Xn[m,b], XPn[m,b]> ),  Xn = P/|P|, XPn = XP/|XP|
//   heads_loss = ap_loss + label_loss - beta * nce                               heads_loss_fwd (+ the backward seeds)
#include "common.h"

__device__ __forceinline__ float block_sum_256(float v, float* red) {      // 256 threads; red: 4 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one workgroup per (m, b) row: g = relu(Apre) . v_m + vb_m ; C[b, mH + k] = P[m,b,k] * g
struct GateW { const float* vw[3]; const float* vb[3]; float* gvw[3]; float* gvb[3]; };   // vt / vv / vs weights [H], biases [1], their gradients
__global__ __launch_bounds__(256) void heads_gate_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Apre, const GateW w,
                                                             int B, int H, float* __restrict__ g, float* __restrict__ C) {
    __shared__ float red[4];
    const int row = blockIdx.x, m = row / B, b = row - m * B;
    const float* vw = m == 0 ? w.vw[0] : (m == 1 ? w.vw[1] : w.vw[2]);
    const float* vb = m == 0 ? w.vb[0] : (m == 1 ? w.vb[1] : w.vb[2]);
    float s = 0.f;
    for (int k = threadIdx.x; k < H; k += 256) s += fmaxf(Apre[(size_t)row * H + k], 0.f) * vw[k];
    const float gv = block_sum_256(s, red) + vb[0];
    if (threadIdx.x == 0) g[row] = gv;
    for (int k = threadIdx.x; k < H; k += 256) C[(size_t)b * 3 * H + (size_t)m * H + k] = P[(size_t)row * H + k] * gv;
}

// CPC terms and their backward seeds (for an upstream gradient of 1; backward scales them).  grid = 3: one workgroup per
// modality, everything for its [B,H] pair of arrays in LDS (rows padded by one float: column reads of different rows hit
// different banks), B <= 16, H <= 1024, 256 threads.
//   nce_part[m] = mean_b (logsumexp_b' S[b][b'] - S[b][b]),  S = Xn XPn^T,  Xn = P_m/|.|, XPn = XP_m/|.|
//   dS[b][b'] = (-beta/B) (softmax_b'(S[b])[b'] - delta);  dXn = dS XPn, dXPn = dS^T Xn;  through y = x/|x|: dx = (dy - y<y,dy>)/|x|
//   dPc [3,B,H] = d loss / d P (CPC part), dXP [3,B,H] = d loss / d XP
__global__ __launch_bounds__(256) void heads_loss_fwd_kernel(const float* __restrict__ P, const float* __restrict__ XP, int B, int H, float beta,
                                                             float* __restrict__ dXP, float* __restrict__ dPc, float* __restrict__ nce_part) {
    extern __shared__ float sm[];                       // Xn [16][HP] | XPn [16][HP] | dS [16][16] | n1 [16] | n2 [16] | red [32][4]
    const int HP = H + 1;
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Xn = sm; float* XPn = sm + 16 * HP; float* dS = XPn + 16 * HP; float* n1 = dS + 256; float* n2 = n1 + 16; float* red = n2 + 16;
    const float* Pm = P + (size_t)m * B * H; const float* XPm = XP + (size_t)m * B * H;
    const int r = tid >> 4, q = tid & 15;               // 16 threads per row
    // rows into LDS + norms (16-lane shuffles: a row's 16 threads sit in one wave)
    float sa = 0.f, sc = 0.f;
    for (int k = q; k < H; k += 16) {
        const float x = r < B ? Pm[(size_t)r * H + k] : 0.f, y = r < B ? XPm[(size_t)r * H + k] : 0.f;
        Xn[r * HP + k] = x; XPn[r * HP + k] = y;
        sa += x * x; sc += y * y;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { sa += __shfl_xor(sa, o, 64); sc += __shfl_xor(sc, o, 64); }
    const float nx = sqrtf(sa), ny = sqrtf(sc);
    if (q == 0) { n2[r] = nx; n1[r] = ny; }
    const float ix = r < B ? 1.f / nx : 0.f, iy = r < B ? 1.f / ny : 0.f;
    for (int k = q; k < H; k += 16) { Xn[r * HP + k] *= ix; XPn[r * HP + k] *= iy; }
    __syncthreads();
    // S[b = r][c = q]
    float sv = 0.f;
    for (int k = 0; k < H; ++k) sv += Xn[r * HP + k] * XPn[q * HP + k];
    const bool valid = r < B && q < B;
    float mx = valid ? sv : -INFINITY;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = valid ? expf(sv - mx) : 0.f, se = e;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
    const float neg = mx + logf(se);
    const float w = -beta / (float)B;
    dS[r * 16 + q] = valid ? w * (e / se - (q == r ? 1.f : 0.f)) : 0.f;
    float part = (valid && q == r) ? (neg - sv) / (float)B : 0.f;
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (tid == 0) nce_part[m] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    // gradients: thread owns columns k = tid + 256 i
    float av[4][16], cv[4][16], dx[16], dy[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) { dx[b] = 0.f; dy[b] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        if (k < H) {
            float x[16], y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { x[j] = Xn[j * HP + k]; y[j] = XPn[j * HP + k]; }
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                float a = 0.f, c = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) { a += dS[b * 16 + j] * y[j]; c += dS[j * 16 + b] * x[j]; }
                av[i][b] = a; cv[i][b] = c;
                dx[b] += a * x[b]; dy[b] += c * y[b];
            }
        }
    }
    // <y, dy> per row: 32 block-wide sums (wave shuffles, then 4 partials each through LDS)
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const float u = wave_sum(dx[b]), v = wave_sum(dy[b]);
        if (lane == 0) { red[b * 4 + wv] = u; red[(16 + b) * 4 + wv] = v; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        if (k < H) {
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                if (b < B) {
                    const float dotx = red[b * 4] + red[b * 4 + 1] + red[b * 4 + 2] + red[b * 4 + 3];
                    const float doty = red[(16 + b) * 4] + red[(16 + b) * 4 + 1] + red[(16 + b) * 4 + 2] + red[(16 + b) * 4 + 3];
                    dPc[((size_t)m * B + b) * H + k] = (av[i][b] - Xn[b * HP + k] * dotx) / n2[b];
                    dXP[((size_t)m * B + b) * H + k] = (cv[i][b] - XPn[b * HP + k] * doty) / n1[b];
                }
            }
        }
    }
}

// single workgroup: 2-way CE of the alignment scores (REF :297-298, :428), the label loss (:430-441, num_labels 7 / 1), sums.
//   rel [2B,2] (visual rows then speech rows), ap [2B] labels, lo [B] raw classifier output (tanh applied here when tanh_lo), sent [B]
//   out: [ap_loss, label_loss, nce, heads_loss];  seeds: drel [2B,2], dlo [B] (gradient w.r.t. the PRE-tanh output when tanh_lo)
__global__ __launch_bounds__(256) void heads_loss_finish_kernel(const float* __restrict__ rel, const int64_t* __restrict__ ap, const float* __restrict__ lo,
                                                                const float* __restrict__ sent, const float* __restrict__ nce_part, int B, float beta,
                                                                int tanh_lo, float* __restrict__ out, float* __restrict__ drel, float* __restrict__ dlo) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float ce = 0.f, se = 0.f;
    if (tid < 2 * B) {
        const float a = rel[tid * 2], c = rel[tid * 2 + 1];
        const float mx = fmaxf(a, c), lse = mx + logf(expf(a - mx) + expf(c - mx));
        const int y = (int)ap[tid];
        ce = (lse - (y ? c : a)) / (float)B * 0.5f;
        const float sc = 0.5f / (float)B;
        drel[tid * 2] = sc * (expf(a - lse) - (y == 0 ? 1.f : 0.f));
        drel[tid * 2 + 1] = sc * (expf(c - lse) - (y == 1 ? 1.f : 0.f));
    }
    if (tid < B) {
        const float v = tanh_lo ? tanhf(lo[tid]) : lo[tid];
        const float d = v - sent[tid];
        se = d * d / (float)B;
        dlo[tid] = 2.f * d / (float)B * (tanh_lo ? 1.f - v * v : 1.f);
    }
    ce = block_sum_256(ce, red);
    se = block_sum_256(se, red);
    if (tid == 0) {
        const float nce = nce_part[0] + nce_part[1] + nce_part[2];
        out[0] = ce; out[1] = se; out[2] = nce; out[3] = ce + se - beta * nce;
    }
}

// x[i] *= *s  (the seeds are linear in the upstream gradient of heads_loss)
__global__ void heads_scale_kernel(float* __restrict__ x, size_t n, const float* __restrict__ s) {
    const float f = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= f;
}

// gate backward; one workgroup per (m, b) row:
//   dC [B,3H], P, Apre [3B,H], g [3B], vw (3 pointers), dPc [3B,H] (CPC part)  ->
//   dg[m,b] = <dC[b, mH:], P[m,b]>;  dP [3B,H] = dC_m * g + dPc;  dApre [3B,H] = dg * v_m * (Apre > 0);
//   E [3B,H] = dg * relu(Apre)   (its column sums over b are the gradient of v_m: heads_colsum)
__global__ __launch_bounds__(256) void heads_gate_bwd_kernel(const float* __restrict__ dC, const float* __restrict__ P, const float* __restrict__ Apre,
                                                             const float* __restrict__ g, const GateW w, const float* __restrict__ dPc,
                                                             int B, int H, float* __restrict__ dP, float* __restrict__ dApre,
                                                             float* __restrict__ E, float* __restrict__ dg_out) {
    __shared__ float red[4];
    const int row = blockIdx.x, m = row / B, b = row - m * B, tid = threadIdx.x;
    const float* vw = m == 0 ? w.vw[0] : (m == 1 ? w.vw[1] : w.vw[2]);
    const float* dc = dC + (size_t)b * 3 * H + (size_t)m * H;
    float s = 0.f;
    for (int k = tid; k < H; k += 256) s += dc[k] * P[(size_t)row * H + k];
    const float dg = block_sum_256(s, red);
    const float gv = g[row];
    if (tid == 0) dg_out[row] = dg;
    for (int k = tid; k < H; k += 256) {
        dP[(size_t)row * H + k] = dc[k] * gv + dPc[(size_t)row * H + k];
        const float a = Apre[(size_t)row * H + k];
        dApre[(size_t)row * H + k] = a > 0.f ? dg * vw[k] : 0.f;
        E[(size_t)row * H + k] = dg * fmaxf(a, 0.f);
    }
}

// dpre = dP * (1 - P^2)
__global__ void heads_tanh_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ P, float* __restrict__ dpre, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = P[i];
        dpre[i] = dP[i] * (1.f - p * p);
    }
}

// bias (and gate-vector) gradients: dst[c] += sum_r src[r][c] for up to 16 (src, rows, cols, dst) segments in one launch; grid.y = segment
struct ColsumSeg { const float* src; float* dst; int rows, cols, ld; };
struct ColsumArgs { ColsumSeg s[16]; };
__global__ void heads_colsum_kernel(const ColsumArgs a) {
    const ColsumSeg s = a.s[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.cols) return;
    float t = 0.f;
    for (int r = 0; r < s.rows; ++r) t += s.src[(size_t)r * s.ld + c];
    s.dst[c] += t;
}

extern "C" {

int mmbert_heads_gate_fwd(hipStream_t stream, const float* P, const float* Apre, const float* const* vw3, const float* const* vb3, int B, int H, float* g, float* C) {
    if (B <= 0) return 0;
    GateW w = {};
    for (int i = 0; i < 3; ++i) { w.vw[i] = vw3[i]; w.vb[i] = vb3[i]; }
    hipLaunchKernelGGL(heads_gate_fwd_kernel, dim3(3 * B), dim3(256), 0, stream, P, Apre, w, B, H, g, C);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_loss_fwd(hipStream_t stream, const float* P, const float* XP, const float* rel, const int64_t* ap, const float* lo, const float* sent,
                          int B, int H, float beta, int tanh_lo, float* out4, float* dXP, float* dPc, float* drel, float* dlo, float* nce_part) {
    if (B <= 0) return 0;
    if (H > 1024 || B > 16) return -1;
    const size_t lds = ((size_t)2 * 16 * (H + 1) + 256 + 32 + 128) * sizeof(float);
    if (lds > 160 * 1024 - 256) return -1;                   // the kernel also has a few static LDS words
    constexpr int LDS_MAX = (2 * 16 * (1024 + 1) + 256 + 32 + 128) * (int)sizeof(float);     // the H = 1024 request: allowed once per device
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)heads_loss_fwd_kernel, LDS_MAX, attr_done)) return e;
    hipLaunchKernelGGL(heads_loss_fwd_kernel, dim3(3), dim3(256), lds, stream, P, XP, B, H, beta, dXP, dPc, nce_part);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(heads_loss_finish_kernel, dim3(1), dim3(256), 0, stream, rel, ap, lo, sent, (const float*)nce_part, B, beta, tanh_lo, out4, drel, dlo);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_scale(hipStream_t stream, float* x, size_t n, const float* s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(heads_scale_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, stream, x, n, s);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_gate_bwd(hipStream_t stream, const float* dC, const float* P, const float* Apre, const float* g, const float* const* vw3, const float* dPc,
                          int B, int H, float* dP, float* dApre, float* E, float* dg) {
    if (B <= 0) return 0;
    GateW w = {};
    for (int i = 0; i < 3; ++i) w.vw[i] = vw3[i];
    hipLaunchKernelGGL(heads_gate_bwd_kernel, dim3(3 * B), dim3(256), 0, stream, dC, P, Apre, g, w, dPc, B, H, dP, dApre, E, dg);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_tanh_bwd(hipStream_t stream, const float* dP, const float* P, float* dpre, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(heads_tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, stream, dP, P, dpre, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

// nseg <= 16 segments: dst_i[c] += column sums of src_i [rows_i, cols_i] (row pitch ld_i)
int mmbert_heads_colsum(hipStream_t stream, int nseg, const float* const* src, float* const* dst, const int* rows, const int* cols, const int* ld) {
    if (nseg <= 0) return 0;
    if (nseg > 16) return -1;
    ColsumArgs a;
    int maxc = 0;
    for (int i = 0; i < nseg; ++i) { a.s[i].src = src[i]; a.s[i].dst = dst[i]; a.s[i].rows = rows[i]; a.s[i].cols = cols[i]; a.s[i].ld = ld[i]; if (cols[i] > maxc) maxc = cols[i]; }
    hipLaunchKernelGGL(heads_colsum_kernel, dim3((maxc + 255) / 256, nseg), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
