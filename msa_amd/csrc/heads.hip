// Pretraining heads of MMBertForPretraining (REF:MMBertForPretraining.py:293-301, 399-443; CPC REF:MMBertEmbedding.py:21-32):
// the fp32, [B,H]-sized arithmetic BETWEEN the dense products -- gates, gated concatenation, CPC normalisation / in-batch
// InfoNCE, the 2-way and regression losses, and their hand-derived backward -- and, since round 2, the dense products themselves
// (skinny_mm / skinny_wgrad below: lists of <= 64-row fp32 products per launch; round 1 left them to torch.addmm -> hipBLASLt).
// Together: 16 launches per step where the eager autograd form needs ~250.  Everything here is HBM/latency trivial.
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

// CPC terms and their backward seeds (for an upstream gradient of 1; backward scales them).  grid = (3, ceil(H / 64)): a
// workgroup per modality and 64-column block.  Every workgroup builds the modality's 16 x 16 similarity matrix from ALL columns
// (both [B,H] arrays normalised into LDS, 98 KB from L2: the round-1 kernel did this in 3 workgroups and then walked the H
// gradient columns in them, 46 us of a 16.4-ms step), then writes the gradients of its own 64 columns.  B <= 16, H <= 1024, H % 4 == 0.
//   nce_part[m] = mean_b (logsumexp_b' S[b][b'] - S[b][b]),  S = Xn XPn^T,  Xn = P_m/|.|, XPn = XP_m/|.|
//   dS[b][b'] = (-beta/B) (softmax_b'(S[b])[b'] - delta);  dXn = dS XPn, dXPn = dS^T Xn;  through y = x/|x|: dx = (dy - y<y,dy>)/|x|
//   with <Xn[b], dXn[b]> = sum_j dS[b][j] S[b][j] and <XPn[b], dXPn[b]> = sum_j dS[j][b] S[j][b] (no pass over the columns)
//   dPc [3,B,H] = d loss / d P (CPC part), dXP [3,B,H] = d loss / d XP
__global__ __launch_bounds__(256) void heads_loss_fwd_kernel(const float* __restrict__ P, const float* __restrict__ XP, int B, int H, float beta,
                                                             float* __restrict__ dXP, float* __restrict__ dPc, float* __restrict__ nce_part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // Xn [16][HP] | XPn [16][HP] | dS [16][16] | SdS [16][16] | n1 [16] | n2 [16] | dots [32] | red [4]
    const int HP = H + 4;                               // rows stay 16-byte aligned; 4 floats of skew between rows
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Xn = sm; float* XPn = sm + 16 * HP; float* dS = XPn + 16 * HP; float* SdS = dS + 256;
    float* n1 = SdS + 256; float* n2 = n1 + 16; float* dots = n2 + 16; float* red = dots + 32;
    const float* Pm = P + (size_t)m * B * H; const float* XPm = XP + (size_t)m * B * H;
    const int r = tid >> 4, q = tid & 15;               // 16 threads per row
    // rows into LDS (float4, 256 contiguous bytes per row and step), norms by 16-lane shuffles (a row's 16 threads sit in one wave)
    float sa = 0.f, sc = 0.f;
    for (int k = 4 * q; k < H; k += 64) {
        const float4 z = {0.f, 0.f, 0.f, 0.f};
        const float4 x = r < B ? *(const float4*)(Pm + (size_t)r * H + k) : z, y = r < B ? *(const float4*)(XPm + (size_t)r * H + k) : z;
        *(float4*)(Xn + r * HP + k) = x; *(float4*)(XPn + r * HP + k) = y;
        sa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; sc += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { sa += __shfl_xor(sa, o, 64); sc += __shfl_xor(sc, o, 64); }
    const float nx = sqrtf(sa), ny = sqrtf(sc);
    if (q == 0) { n2[r] = nx; n1[r] = ny; }
    const float ix = r < B ? 1.f / nx : 0.f, iy = r < B ? 1.f / ny : 0.f;
    for (int k = 4 * q; k < H; k += 64) {               // each thread rescales what it wrote
        float4 x = *(float4*)(Xn + r * HP + k), y = *(float4*)(XPn + r * HP + k);
        x.x *= ix; x.y *= ix; x.z *= ix; x.w *= ix; y.x *= iy; y.y *= iy; y.z *= iy; y.w *= iy;
        *(float4*)(Xn + r * HP + k) = x; *(float4*)(XPn + r * HP + k) = y;
    }
    __syncthreads();
    // S[b = r][c = q]
    float sv = 0.f;
    for (int k = 0; k < H; k += 4) {
        const float4 x = *(const float4*)(Xn + r * HP + k), y = *(const float4*)(XPn + q * HP + k);
        sv += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    const bool valid = r < B && q < B;
    float mx = valid ? sv : -INFINITY;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = valid ? expf(sv - mx) : 0.f, se = e;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
    const float neg = mx + logf(se);
    const float w = -beta / (float)B;
    const float ds = valid ? w * (e / se - (q == r ? 1.f : 0.f)) : 0.f;
    dS[r * 16 + q] = ds;
    SdS[r * 16 + q] = valid ? ds * sv : 0.f;
    float part = (valid && q == r) ? (neg - sv) / (float)B : 0.f;
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (tid == 0 && blockIdx.y == 0) nce_part[m] = red[0] + red[1] + red[2] + red[3];
    if (tid < 32) {                                     // <Xn[b], dXn[b]> (row sums of S o dS) and <XPn[b], dXPn[b]> (column sums)
        const int b = tid & 15;
        float t = 0.f;
        for (int j2 = 0; j2 < 16; ++j2) t += tid < 16 ? SdS[b * 16 + j2] : SdS[j2 * 16 + b];
        dots[tid] = t;
    }
    __syncthreads();
    // gradients of this workgroup's 64 columns: thread = (column c, 4 rows b = 4 rg ..)
    const int c = lane, rg = wv, k = blockIdx.y * 64 + c;
    if (k < H) {
        float x[16], y[16];
#pragma unroll
        for (int j2 = 0; j2 < 16; ++j2) { x[j2] = Xn[j2 * HP + k]; y[j2] = XPn[j2 * HP + k]; }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int b = 4 * rg + bb;
            if (b < B) {
                float a = 0.f, cc = 0.f;
#pragma unroll
                for (int j2 = 0; j2 < 16; ++j2) { a += dS[b * 16 + j2] * y[j2]; cc += dS[j2 * 16 + b] * x[j2]; }
                dPc[((size_t)m * B + b) * H + k] = (a - x[b] * dots[b]) / n2[b];
                dXP[((size_t)m * B + b) * H + k] = (cc - y[b] * dots[16 + b]) / n1[b];
            }
        }
    }
}

// The same terms for 16 < B <= 32 (round 4: the reference's default batch is 32, REF:train.py:38).  Two [32, H] arrays no longer fit LDS
// at H = 1024, so the similarity matrix is built from RAW rows streamed in 128-column chunks (S = <x, xp> / (|x| |xp|): the
// normalisation is applied to the sums) and the gradient phase reads this workgroup's 64 columns of both arrays from global memory
// (L2-resident: 3 x 32 x H floats per array).  grid = (3, ceil(H / 64)) as above; thread (r = tid >> 3, q8 = tid & 7) owns S[r][4 q8 .. + 3].
__global__ __launch_bounds__(256) void heads_loss_fwd32_kernel(const float* __restrict__ P, const float* __restrict__ XP, int B, int H, float beta,
                                                               float* __restrict__ dXP, float* __restrict__ dPc, float* __restrict__ nce_part) {
    __shared__ __attribute__((aligned(16))) float Xc[32][132];     // raw chunk of P_m   (rows >= B: zeros)
    __shared__ __attribute__((aligned(16))) float Yc[32][132];     // raw chunk of XP_m
    __shared__ float dS[32][33], SdS[32][33], nx[32], ny[32], dots[64], red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* Pm = P + (size_t)m * B * H; const float* XPm = XP + (size_t)m * B * H;
    const int r = tid >> 3, q8 = tid & 7;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, sa = 0.f, sc = 0.f;
    for (int k0 = 0; k0 < H; k0 += 128) {
        __syncthreads();
        // stage: thread (row = tid >> 3, 16 columns at 16 (tid & 7)) of both arrays
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = k0 + 16 * q8 + 4 * v;
            const float4 z = {0.f, 0.f, 0.f, 0.f};
            const bool in = r < B && k < H;                      // H % 4 == 0
            const float4 x = in ? *(const float4*)(Pm + (size_t)r * H + k) : z, y = in ? *(const float4*)(XPm + (size_t)r * H + k) : z;
            *(float4*)&Xc[r][16 * q8 + 4 * v] = x; *(float4*)&Yc[r][16 * q8 + 4 * v] = y;
            sa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; sc += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 128; k += 4) {
            const float4 x = *(const float4*)&Xc[r][k];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 y = *(const float4*)&Yc[4 * q8 + c][k];
                s4[c] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            }
        }
    }
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) { sa += __shfl_xor(sa, o, 64); sc += __shfl_xor(sc, o, 64); }   // a row's 8 threads are consecutive lanes
    if (q8 == 0) { nx[r] = sqrtf(sa); ny[r] = sqrtf(sc); }
    __syncthreads();
    float sv[4], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cc = 4 * q8 + c;
        const bool valid = r < B && cc < B;
        sv[c] = valid ? s4[c] / (nx[r] * ny[cc]) : 0.f;
        if (valid) mx = fmaxf(mx, sv[c]);
    }
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e[4], se = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { e[c] = (r < B && 4 * q8 + c < B) ? expf(sv[c] - mx) : 0.f; se += e[c]; }
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
    const float neg = mx + logf(se), w = -beta / (float)B;
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cc = 4 * q8 + c;
        const bool valid = r < B && cc < B;
        const float ds = valid ? w * (e[c] / se - (cc == r ? 1.f : 0.f)) : 0.f;
        dS[r][cc] = ds;
        SdS[r][cc] = valid ? ds * sv[c] : 0.f;
        if (valid && cc == r) part = (neg - sv[c]) / (float)B;
    }
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (tid == 0 && blockIdx.y == 0) nce_part[m] = red[0] + red[1] + red[2] + red[3];
    if (tid < 64) {                                     // <Xn[b], dXn[b]> (row sums of S o dS) and <XPn[b], dXPn[b]> (column sums)
        const int b = tid & 31;
        float t = 0.f;
        for (int j2 = 0; j2 < 32; ++j2) t += tid < 32 ? SdS[b][j2] : SdS[j2][b];
        dots[tid] = t;
    }
    __syncthreads();
    // gradients of this workgroup's 64 columns: thread = (column c, 8 rows b = 8 wv ..)
    const int k = blockIdx.y * 64 + lane;
    if (k < H) {
        float x[32], y[32];
#pragma unroll
        for (int j2 = 0; j2 < 32; ++j2) {
            x[j2] = j2 < B ? Pm[(size_t)j2 * H + k] / nx[j2] : 0.f;
            y[j2] = j2 < B ? XPm[(size_t)j2 * H + k] / ny[j2] : 0.f;
        }
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const int b = 8 * wv + bb;
            if (b < B) {
                float a = 0.f, cc = 0.f;
#pragma unroll
                for (int j2 = 0; j2 < 32; ++j2) { a += dS[b][j2] * y[j2]; cc += dS[j2][b] * x[j2]; }
                dPc[((size_t)m * B + b) * H + k] = (a - x[b] * dots[b]) / nx[b];
                dXP[((size_t)m * B + b) * H + k] = (cc - y[b] * dots[32 + b]) / ny[b];
            }
        }
    }
}

// single workgroup: 2-way CE of the alignment scores (REF :297-298, :428), the label loss (:430-441, num_labels 7 / 1), sums.
//   rel [2B,2] (visual rows then speech rows), ap [2B] labels, lo [B] raw classifier output (tanh applied here when tanh_lo), sent [B]
//   out: [ap_loss, label_loss, nce, heads_loss];  seeds: drel [2B,2], dlo [B] (gradient w.r.t. the PRE-tanh output when tanh_lo)
__global__ __launch_bounds__(256) void heads_loss_finish_kernel(const float* __restrict__ rel, const int64_t* __restrict__ ap, const float* __restrict__ lo,
                                                                const float* __restrict__ sent, const float* __restrict__ nce_part, int B, float beta,
                                                                int tanh_lo, float* __restrict__ out, float* __restrict__ drel, float* __restrict__ dlo,
                                                                const float* __restrict__ mlm, int nmlm, float alpha,
                                                                float* __restrict__ loss_out, float* __restrict__ aux_out) {
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
        const float heads = ce + se - beta * nce;
        out[0] = ce; out[1] = se; out[2] = nce; out[3] = heads;
        float ms = 0.f;                                  // joint = alpha * mean(mlm) + heads   (REF :427, :443)
        for (int i = 0; i < nmlm; ++i) ms += mlm[i];
        const float joint = nmlm > 0 ? alpha * (ms / (float)nmlm) + heads : heads;
        out[4] = joint;
        // the same values once more in buffers of their own: what the caller returns (a slice of `out` would be a VIEW, and autograd
        // refuses in-place operations on view outputs of a custom function; round 3 cloned the slices: two copy launches)
        if (loss_out) *loss_out = joint;
        if (aux_out) { aux_out[0] = ce; aux_out[1] = se; aux_out[2] = nce; }
    }
}

// x[i] *= *s  (the seeds are linear in the upstream gradient of heads_loss)
__global__ void heads_scale_kernel(float* __restrict__ x, size_t n, const float* __restrict__ s) {
    const float f = *s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= f;
}

// the start of the heads' backward in ONE launch (round 4; before: a clone of the seeds, heads_scale in place, a zero fill and a multiply):
// dst[i] = src[i] * *s (the seeds are kept: backward may run twice), zero[j] = 0 (the buffers the backward products are summed into),
// dmlm[k] = *s * coef (the upstream gradient of the per-pass MLM losses: alpha / passes)
__global__ void heads_seed_kernel(const float* __restrict__ src, size_t n, const float* __restrict__ s, float* __restrict__ dst,
                                  float* __restrict__ zero, size_t nzero, float* __restrict__ dmlm, int nmlm, float coef) {
    const float f = *s;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) dst[i] = src[i] * f;
    for (size_t i = tid; i < nzero; i += stride) zero[i] = 0.f;
    if (tid < (size_t)nmlm) dmlm[tid] = f * coef;
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


// --------------------------------------------------------------------------------------------------------------------------------
// The heads' dense layers themselves: fp32 products with at most 128 rows ([3B, H] pooled vectors against [H..3H, H] weights, B <= 32).
// Round 1 sent each through torch.addmm -> hipBLASLt: ~45 launches of 6-27 us per step for 0.9 GFLOP, plus the host cost of as many
// library calls.  Here a launch is a LIST of independent products (one dependency level of the heads' graph), memory-bound on the
// weights it streams once:
//   skinny_mm    Y[M, N] += bias + sum_j X_j . op(W_j)   (Y zeroed by the caller, or holding what to add to)   forward layers and input gradients
//                (op = W^T for a Linear weight [N, inner] -- "y = x W^T" --, or W itself [inner, N] -- "dx = dy W"; a source may
//                 cover a row range only: the alignment head reads / feeds the visual and speech rows)
//   skinny_wgrad dW[N, K] += dY[M, N]^T . X[M, K],  db[N] += column sums of dY                    weight and bias gradients
// Workgroup = 256 threads = one 16-column output tile x all rows (mm) or one 16 x 64 tile of dW (wgrad); operands staged through LDS
// in 64-deep chunks, float4 LDS reads (row pitch 68 floats: 16-byte aligned and bank-spread).
// --------------------------------------------------------------------------------------------------------------------------------
#define SK_MAXOPS 12
#define SK_MAXSRC 4
struct SkSrc { const float* X; const float* W; int ldx, ldw, inner, row0, rows, w_inner_major; };
struct SkOp { float* Y; const float* bias; int ldy, M, N, nsrc, act, accumulate, tile0, pad_; SkSrc src[SK_MAXSRC]; };
struct SkArgs { SkOp op[SK_MAXOPS]; int nops; long long ws_off[SK_MAXOPS + 1]; float* ws; };

// ORDERED (deterministic mode, mmbert_skinny_mm_ordered): a workgroup STORES its partial tile into the caller's slab
// ws[op][chunk][row][n] instead of adding it to Y with atomics; skinny_mm_fold_kernel then adds bias and the chunks in ascending order.
template <bool ORDERED>
__global__ __launch_bounds__(256) void skinny_mm_kernel(const SkArgs a) {
    // One workgroup = one 16-column output tile x ONE 64-deep chunk of one source: a few hundred to a few thousand independent
    // workgroups per launch, each a single round trip to memory (a first version walked all chunks of a tile in one workgroup:
    // 12-36 dependent round trips on 48-144 workgroups, 130 us per launch).  The partial products are added to Y with fp32 atomics,
    // so Y must hold zeros (or the value to accumulate onto) before the launch; the chunk 0 workgroup of a tile also adds the bias.
    __shared__ __attribute__((aligned(16))) float Xs[64][68];
    __shared__ __attribute__((aligned(16))) float Ws[16][68];
    int oi = 0;
    for (int q = 1; q < a.nops; ++q) if ((int)blockIdx.x >= a.op[q].tile0) oi = q;
    const SkOp& op = a.op[oi];
    int t = (int)blockIdx.x - op.tile0;
    const int chunks = op.pad_;                                    // 64-deep chunks over all sources of this op
    // (round 4: M up to 128 = the reference's default batch 32 x 3 modalities: the tile index carries a 64-row block as its slowest part)
    const int per_rb = ((op.N + 15) / 16) * chunks;
    const int rb0 = (t / per_rb) * 64;
    t -= (t / per_rb) * per_rb;
    const int n0 = (t / chunks) * 16;
    int ch = t - (t / chunks) * chunks, j = 0;
    const int ch_all = ch;                                          // chunk index over all sources of this op
    while (j + 1 < op.nsrc && ch >= (op.src[j].inner + 63) / 64) { ch -= (op.src[j].inner + 63) / 64; ++j; }
    const SkSrc& sc = op.src[j];
    const int c0 = ch * 64;
    const int tid = threadIdx.x, tn = tid & 15, tg = tid >> 4;
    // all 20 global loads of the workgroup's two operand chunks are issued before the first LDS store: one round trip
    float xv[16], wv[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                 // X chunk: 64 rows x 64 inner (rows outside the source's range: 0)
        const int idx = tid + 256 * i, row = rb0 + (idx >> 6), kk = idx & 63;
        const int r = row - sc.row0;
        xv[i] = (r >= 0 && r < sc.rows && c0 + kk < sc.inner) ? sc.X[(size_t)r * sc.ldx + c0 + kk] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                  // W chunk: 16 output columns x 64 inner
        const int idx = tid + 256 * i;
        int c, kk;
        if (sc.w_inner_major) { kk = idx >> 4; c = idx & 15; } else { c = idx >> 6; kk = idx & 63; }
        wv[i] = 0.f;
        if (n0 + c < op.N && c0 + kk < sc.inner)
            wv[i] = sc.w_inner_major ? sc.W[(size_t)(c0 + kk) * sc.ldw + n0 + c] : sc.W[(size_t)(n0 + c) * sc.ldw + c0 + kk];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int idx = tid + 256 * i; Xs[idx >> 6][idx & 63] = xv[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        if (sc.w_inner_major) Ws[idx & 15][idx >> 4] = wv[i]; else Ws[idx >> 6][idx & 63] = wv[i];
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
        const float4 w4 = *(const float4*)&Ws[tn][kk];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 x4 = *(const float4*)&Xs[tg + 16 * r][kk];
            acc[r] += x4.x * w4.x + x4.y * w4.y + x4.z * w4.z + x4.w * w4.w;
        }
    }
    const int n = n0 + tn;
    if (n < op.N) {
        const float b = (op.bias && j == 0 && ch == 0) ? op.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rb0 + tg + 16 * r;
            if (row >= op.M) continue;
            if constexpr (ORDERED) a.ws[a.ws_off[oi] + ((size_t)ch_all * op.M + row) * op.N + n] = acc[r];
            else atomicAdd(op.Y + (size_t)row * op.ldy + n, acc[r] + b);
        }
    }
}

// Y[row][n] += bias[n] + sum over the chunks (ascending) of the slab: one thread per output element, one add per element
__global__ __launch_bounds__(256) void skinny_mm_fold_kernel(const SkArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    int oi = 0;
    long long base = 0;
    for (int q = 0; q < a.nops; ++q) {
        const long long cnt = (long long)a.op[q].M * a.op[q].N;
        if (e >= base + cnt) { base += cnt; oi = q + 1; } else break;
    }
    if (oi >= a.nops) return;
    const SkOp& op = a.op[oi];
    const long long r = e - base;
    const int row = (int)(r / op.N), n = (int)(r - (long long)row * op.N);
    float t = op.bias ? op.bias[n] : 0.f;
    const float* w = a.ws + a.ws_off[oi] + (size_t)row * op.N + n;
    const size_t pitch = (size_t)op.M * op.N;
    int c = 0;
    for (; c + 4 <= op.pad_; c += 4) {                               // four chunks' loads together, added in ascending order
        const float w0 = w[(size_t)c * pitch], w1 = w[(size_t)(c + 1) * pitch], w2 = w[(size_t)(c + 2) * pitch], w3 = w[(size_t)(c + 3) * pitch];
        t += w0; t += w1; t += w2; t += w3;
    }
    for (; c < op.pad_; ++c) t += w[(size_t)c * pitch];
    op.Y[(size_t)row * op.ldy + n] += t;
}

struct SkWOp { const float* dY; const float* X; float* dW; float* db; int ldy, ldx, ldw, M, N, K, tile0, tiles_k; };
struct SkWArgs { SkWOp op[SK_MAXOPS]; int nops; };

__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const SkWArgs a) {
    __shared__ float dYs[64][17];
    __shared__ __attribute__((aligned(16))) float Xs[64][68];
    int oi = 0;
    for (int q = 1; q < a.nops; ++q) if ((int)blockIdx.x >= a.op[q].tile0) oi = q;
    const SkWOp& op = a.op[oi];
    const int t = (int)blockIdx.x - op.tile0;
    const int n0 = (t / op.tiles_k) * 16, k0 = (t % op.tiles_k) * 64;
    const int tid = threadIdx.x;
    const int tk4 = tid & 15, tn = tid >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    for (int m0 = 0; m0 < op.M; m0 += 64) {                        // (round 4: M up to 128 rows, 64 at a time through LDS)
        if (m0) __syncthreads();
        // all 20 loads of the block are issued before the first LDS store, from addresses clamped into range and zeroed by a
        // PRODUCT: as "in range ? load : 0" hipcc put every load into a branch of its own that ended in s_waitcnt vmcnt(0) --
        // 20 memory round trips in a row per block (the kernel took 32 us)
        float dyv[4], xv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, m = m0 + (idx >> 4), c = idx & 15;
            dyv[i] = op.dY[(size_t)min(m, op.M - 1) * op.ldy + min(n0 + c, op.N - 1)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = tid + 256 * i, m = m0 + (idx >> 6), kk = idx & 63;
            xv[i] = op.X[(size_t)min(m, op.M - 1) * op.ldx + min(k0 + kk, op.K - 1)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                // (the products here, behind the last load: hipcc keeps a product next to its load)
            const int idx = tid + 256 * i, m = m0 + (idx >> 4), c = idx & 15;
            dYs[idx >> 4][c] = dyv[i] * ((m < op.M && n0 + c < op.N) ? 1.f : 0.f);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = tid + 256 * i, m = m0 + (idx >> 6), kk = idx & 63;
            Xs[idx >> 6][kk] = xv[i] * ((m < op.M && k0 + kk < op.K) ? 1.f : 0.f);
        }
        __syncthreads();
        const int mend = min(64, op.M - m0);
        for (int m = 0; m < mend; ++m) {
            const float dy = dYs[m][tn];
            const float4 x4 = *(const float4*)&Xs[m][4 * tk4];
            acc.x += dy * x4.x; acc.y += dy * x4.y; acc.z += dy * x4.z; acc.w += dy * x4.w;
            bsum += dy;
        }
    }
    const int n = n0 + tn;
    if (n < op.N) {
        float* w = op.dW + (size_t)n * op.ldw + k0 + 4 * tk4;
        const float v[4] = {acc.x, acc.y, acc.z, acc.w};
        if (k0 + 4 * tk4 + 3 < op.K && !((size_t)w & 15)) {            // one 16-byte read-modify-write (four scalar ones waited for each other)
            float4 o = *(const float4*)w;
            o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
            *(float4*)w = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k0 + 4 * tk4 + e < op.K) w[e] += v[e];
        }
        if (op.db && k0 == 0 && tk4 == 0) op.db[n] += bsum;
    }
}

__global__ void heads_tanh_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = tanhf(x[i]);
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
                          int B, int H, float beta, int tanh_lo, float* out5, float* dXP, float* dPc, float* drel, float* dlo, float* nce_part,
                          const float* mlm, int nmlm, float alpha, float* loss_out, float* aux_out) {
    if (B <= 0) return 0;
    if (H > 1024 || (H & 3) || B > 32 || nmlm < 0 || (nmlm > 0 && !mlm)) return -1;
    if (B > 16) {
        hipLaunchKernelGGL(heads_loss_fwd32_kernel, dim3(3, (H + 63) / 64), dim3(256), 0, stream, P, XP, B, H, beta, dXP, dPc, nce_part);
        MMB_CHECK_LAUNCH();
        hipLaunchKernelGGL(heads_loss_finish_kernel, dim3(1), dim3(256), 0, stream, rel, ap, lo, sent, (const float*)nce_part, B, beta, tanh_lo, out5, drel, dlo,
                           mlm, nmlm, alpha, loss_out, aux_out);
        MMB_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = ((size_t)2 * 16 * (H + 4) + 512 + 32 + 32 + 4) * sizeof(float);
    constexpr int LDS_MAX = (2 * 16 * (1024 + 4) + 512 + 32 + 32 + 4) * (int)sizeof(float);     // the H = 1024 request: allowed once per device
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = mmb_allow_lds((const void*)heads_loss_fwd_kernel, LDS_MAX, attr_done)) return e;
    hipLaunchKernelGGL(heads_loss_fwd_kernel, dim3(3, (H + 63) / 64), dim3(256), lds, stream, P, XP, B, H, beta, dXP, dPc, nce_part);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(heads_loss_finish_kernel, dim3(1), dim3(256), 0, stream, rel, ap, lo, sent, (const float*)nce_part, B, beta, tanh_lo, out5, drel, dlo,
                       mlm, nmlm, alpha, loss_out, aux_out);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_scale(hipStream_t stream, float* x, size_t n, const float* s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(heads_scale_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, stream, x, n, s);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_seed(hipStream_t stream, const float* src, size_t n, const float* s, float* dst, float* zero, size_t nzero, float* dmlm, int nmlm, float coef) {
    const size_t m = n > nzero ? n : nzero;
    if (m == 0 && nmlm <= 0) return 0;
    if (nmlm > 256) return -1;
    const size_t blocks = (m + 255) / 256;
    hipLaunchKernelGGL(heads_seed_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < 512 ? blocks : 512))), dim3(256), 0, stream, src, n, s, dst, zero, nzero, dmlm, nmlm, coef);
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

int mmbert_heads_tanh(hipStream_t stream, float* x, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(heads_tanh_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, stream, x, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_heads_tanh_bwd(hipStream_t stream, const float* dP, const float* P, float* dpre, size_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(heads_tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, stream, dP, P, dpre, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

// see include/mmbert_hip.h for the two op structs (mirrored here field by field)
struct mmbert_skinny_src { const float* X; const float* W; int ldx, ldw, inner, row0, rows, w_inner_major; };
struct mmbert_skinny_op { float* Y; const float* bias; int ldy, M, N, nsrc, act, accumulate; mmbert_skinny_src src[4]; };
struct mmbert_skinny_wgrad_op { const float* dY; const float* X; float* dW; float* db; int ldy, ldx, ldw, M, N, K; };

static int skinny_mm_args(int nops, const mmbert_skinny_op* ops, SkArgs& a, int& tiles, long long& elems) {
    if (nops > SK_MAXOPS) return -1;
    a = SkArgs{};
    tiles = 0; elems = 0;
    long long ws = 0;
    for (int i = 0; i < nops; ++i) {
        const mmbert_skinny_op& o = ops[i];
        if (o.M < 0 || o.M > 128 || o.N <= 0 || o.nsrc < 1 || o.nsrc > SK_MAXSRC || !o.Y) return -1;
        SkOp& d = a.op[i];
        if (o.act != 0) return -1;                                 // (an activation cannot follow a sum that is still being added to)
        d.Y = o.Y; d.bias = o.bias; d.ldy = o.ldy; d.M = o.M; d.N = o.N; d.nsrc = o.nsrc; d.act = 0; d.accumulate = o.accumulate; d.tile0 = tiles;
        int chunks = 0;
        for (int j = 0; j < o.nsrc; ++j) {
            const mmbert_skinny_src& sc = o.src[j];
            if (sc.rows < 0 || sc.row0 < 0 || sc.row0 + sc.rows > 128 || sc.inner <= 0 || !sc.X || !sc.W) return -1;
            d.src[j].X = sc.X; d.src[j].W = sc.W; d.src[j].ldx = sc.ldx; d.src[j].ldw = sc.ldw; d.src[j].inner = sc.inner;
            d.src[j].row0 = sc.row0; d.src[j].rows = sc.rows; d.src[j].w_inner_major = sc.w_inner_major;
            chunks += (sc.inner + 63) / 64;
        }
        d.pad_ = chunks;
        tiles += ((o.N + 15) / 16) * chunks * ((o.M + 63) / 64 > 0 ? (o.M + 63) / 64 : 1);
        a.ws_off[i] = ws;
        ws += (long long)chunks * o.M * o.N;
        elems += (long long)o.M * o.N;
    }
    a.ws_off[nops] = ws;
    a.nops = nops;
    return 0;
}

int mmbert_skinny_mm(hipStream_t stream, int nops, const mmbert_skinny_op* ops) {
    if (nops <= 0) return 0;
    if (mmb_deterministic()) return -4;                            // deterministic mode: mmbert_skinny_mm_ordered (needs the caller's slab)
    SkArgs a;
    int tiles;
    long long elems;
    if (skinny_mm_args(nops, ops, a, tiles, elems)) return -1;
    hipLaunchKernelGGL(skinny_mm_kernel<false>, dim3(tiles), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

size_t mmbert_skinny_mm_workspace(int nops, const mmbert_skinny_op* ops) {
    SkArgs a;
    int tiles;
    long long elems;
    if (nops <= 0 || skinny_mm_args(nops, ops, a, tiles, elems)) return 0;
    return (size_t)a.ws_off[nops] * sizeof(float);
}

int mmbert_skinny_mm_ordered(hipStream_t stream, int nops, const mmbert_skinny_op* ops, void* workspace) {
    if (nops <= 0) return 0;
    SkArgs a;
    int tiles;
    long long elems;
    if (skinny_mm_args(nops, ops, a, tiles, elems) || !workspace) return -1;
    a.ws = (float*)workspace;
    hipLaunchKernelGGL(skinny_mm_kernel<true>, dim3(tiles), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(skinny_mm_fold_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_skinny_wgrad(hipStream_t stream, int nops, const mmbert_skinny_wgrad_op* ops) {
    if (nops <= 0) return 0;
    if (nops > SK_MAXOPS) return -1;
    SkWArgs a = {};
    int tiles = 0;
    for (int i = 0; i < nops; ++i) {
        const mmbert_skinny_wgrad_op& o = ops[i];
        if (o.M < 0 || o.M > 128 || o.N <= 0 || o.K <= 0 || !o.dY || !o.X || !o.dW) return -1;
        SkWOp& d = a.op[i];
        d.dY = o.dY; d.X = o.X; d.dW = o.dW; d.db = o.db; d.ldy = o.ldy; d.ldx = o.ldx; d.ldw = o.ldw; d.M = o.M; d.N = o.N; d.K = o.K;
        d.tile0 = tiles; d.tiles_k = (o.K + 63) / 64;
        tiles += ((o.N + 15) / 16) * d.tiles_k;
    }
    a.nops = nops;
    hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(tiles), dim3(256), 0, stream, a);
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
