// HBM-bound row-wise kernels of the MMBert train step (gfx950): LayerNorm forward/backward with
// the reference's dropout placements fused, embedding gather / scatter-add, the JointEmbeddings
// pair projection (K = 35/74/47/371/81: far too thin for MFMA, plain FMA), vocabulary
// cross-entropy over bf16 logits, the flat AdamW update (+ bf16 weight refresh), and the batched
// fp32 -> bf16 transpose that maintains the dgrad weight copies.
//
// Conventions: one wave (64 lanes) per row, 8-byte bf16x4 accesses, fp32 statistics, row-index
// maps (int32) where a kernel gathers or scatters rows of the packed token matrix.
#include "common.h"

std::atomic<int> g_mmb_deterministic{0};     // mmbert_set_deterministic (common.h: every translation unit of the library reads it)

// Row maps of the valid-first packing (ops.SplitLayout): packed row of every original row (inv) and back (perm), from the
// per-sequence unmasked length `valid`, the region starts and the row -> (sequence, position) tables.  mode 0: masked-out rows
// go to region B in order; 1: all masked-out rows of a sequence share its ONE region-B row (inference); 2: they are left out
// (inv = rows_a, one past the packed matrix).
// ``rank`` (optional, mmbert_prologue's row-set mode): the position of every row in its sequence's OWN valid-first order (active
// rows first, order kept) -- it replaces row_pos, so that the leading ``valid[s]`` rows of sequence s are its active rows wherever
// they sit in the original sequence (the fused text | visual | speech sequence has its visual padding in the middle).
__global__ void split_rows_kernel(const int64_t* __restrict__ row_seq, const int64_t* __restrict__ row_pos, const int* __restrict__ start_a,
                                  const int* __restrict__ start_b, const int* __restrict__ valid, int mode, int M, int rows_a,
                                  int64_t* __restrict__ perm, int64_t* __restrict__ inv, const int* __restrict__ rank,
                                  int* __restrict__ perm32, int* __restrict__ inv32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int s = (int)row_seq[i], p = rank ? rank[i] : (int)row_pos[i], v = valid[s];
    int n = rows_a;
    bool own = false;                                          // does packed row n stand for original row i?
    if (p < v) { n = start_a[s] + p; own = true; }
    else if (mode == 0) { n = start_b[s] + p - v; own = true; }
    else if (mode == 1) { n = start_b[s]; own = p == v; }
    inv[i] = n;
    if (inv32) inv32[i] = n;
    if (own) { perm[n] = i; if (perm32) perm32[n] = i; }
}

// The valid-first packing's per-sequence starts and attention tile lists, built on the DEVICE from the prologue's ``valid`` counts
// (round 3; ops.SplitLayout's numpy form needs them on the host: one blocking round trip per forward pass).  One workgroup:
//   v[s] = min(valid[s], len[s]);  start_a = exclusive prefix sum of v;  start_b = rows_a + exclusive prefix sum of (len - v);
//   region-A tiles (s, k), k < ceil(v / rows), then region-B tiles (s, k), k < ceil((len - v) / rows) -- each list ordered
//   longest-work-first with the sequences interleaved in groups of xs (ops.SplitLayout's rule: rank = stable order of -v; key =
//   (rank / xs, k, rank % xs));
//   unused list entries (the lists are sized for the worst case) get sequence -1: the attention kernels leave at once.
// out (int32): ftile_seq | ftile_r0 | ftile_qshift | ftile_qend  (nf_max each)  |  tile_seq | tile_r0 | qtile_qshift | qtile_qend
// (nq_max each)  |  start_a | v | start_b (nseq each)  |  counts: nf, nq, rows_a, 0.
#define SL_MAXSEQ 1024
__global__ __launch_bounds__(1024) void split_layout_kernel(const int* __restrict__ lens, const int* __restrict__ valid, int nseq, int rows, int xs,
                                                            int nf_max, int nq_max, int* __restrict__ out) {
    __shared__ int v[SL_MAXSEQ], ln[SL_MAXSEQ], rk[SL_MAXSEQ], sa[SL_MAXSEQ], sb[SL_MAXSEQ], seq_of_rank[SL_MAXSEQ], ntr[SL_MAXSEQ];
    __shared__ int tot[3];
    const int tid = threadIdx.x;
    if (tid < 3) tot[tid] = 0;
    for (int s = tid; s < nseq; s += 1024) { ln[s] = lens[s]; v[s] = min(valid[s], lens[s]); }
    __syncthreads();
    // Every sequence's thread walks the others itself -- stable rank of -v and both exclusive prefix sums in one pass of independent
    // LDS reads.  (Thread 0 alone used to run the two prefix sums and the two tile totals as dependent LDS chains: 4 x nseq steps
    // of a read-add-write each, most of the kernel's 23 us in front of every forward pass.)
    for (int s = tid; s < nseq; s += 1024) {
        int r = 0, a = 0, b = 0;
        const int vs = v[s];
        for (int t = 0; t < nseq; ++t) {
            const int vt = v[t];
            r += (vt > vs || (vt == vs && t < s)) ? 1 : 0;
            if (t < s) { a += vt; b += ln[t] - vt; }
        }
        rk[s] = r;
        seq_of_rank[r] = s;
        sa[s] = a; sb[s] = b;
        if (s == nseq - 1) tot[2] = a + vs;
    }
    __syncthreads();
    for (int s = tid; s < nseq; s += 1024) sb[s] += tot[2];
    __syncthreads();
    int* f_seq = out; int* f_r0 = out + nf_max; int* f_sh = out + 2 * nf_max; int* f_end = out + 3 * nf_max;
    int* q_seq = out + 4 * nf_max; int* q_r0 = q_seq + nq_max; int* q_sh = q_seq + 2 * nq_max; int* q_end = q_seq + 3 * nq_max;
    int* o_sa = q_seq + 4 * nq_max; int* o_v = o_sa + nseq; int* o_sb = o_v + nseq; int* o_cnt = o_sb + nseq;
    for (int s = tid; s < nseq; s += 1024) { o_sa[s] = sa[s]; o_v[s] = v[s]; o_sb[s] = sb[s]; }
    // position of tile (s, k) of a list with nt[.] tiles per sequence: tiles of earlier groups, then inside the group by (k, rank % xs)
    auto place = [&](bool regionB, int base, int* cnt_out) {
        auto nt = [&](int s) { const int c = regionB ? ln[s] - v[s] : v[s]; return (c + rows - 1) / rows; };
        for (int s = tid; s < nseq; s += 1024) {                 // tiles per sequence in RANK order (one division per sequence), and their total
            const int n = nt(s);
            ntr[rk[s]] = n;
            if (n) atomicAdd(cnt_out, n);
        }
        __syncthreads();
        for (int s = tid; s < nseq; s += 1024) {
            const int r = rk[s], n = ntr[r];
            if (n == 0) continue;
            const int g = r / xs, m = r - g * xs;
            int before_groups = 0;
            for (int r2 = 0; r2 < g * xs; ++r2) before_groups += ntr[r2];
            for (int k = 0; k < n; ++k) {
                int pos = before_groups;
                for (int m2 = 0; m2 < xs && g * xs + m2 < nseq; ++m2) {
                    const int n2 = ntr[g * xs + m2];
                    pos += min(n2, k) + ((n2 > k && m2 < m) ? 1 : 0);
                }
                const int first = regionB ? v[s] : 0, shift = regionB ? sb[s] - v[s] : sa[s], end = regionB ? ln[s] : v[s];
                f_seq[base + pos] = s; f_r0[base + pos] = first + k * rows; f_sh[base + pos] = shift; f_end[base + pos] = end;
                if (!regionB) { q_seq[pos] = s; q_r0[pos] = k * rows; q_sh[pos] = shift; q_end[pos] = end; }
            }
        }
        __syncthreads();                                          // (ntr is rewritten by the next region; the total is complete)
    };
    place(false, 0, &tot[0]);
    __syncthreads();
    const int nA = tot[0];
    place(true, nA, &tot[1]);
    __syncthreads();
    const int nf = nA + tot[1];
    for (int i = nf + tid; i < nf_max; i += 1024) { f_seq[i] = -1; f_r0[i] = 0; f_sh[i] = 0; f_end[i] = 0; }
    for (int i = nA + tid; i < nq_max; i += 1024) { q_seq[i] = -1; q_r0[i] = 0; q_sh[i] = 0; q_end[i] = 0; }
    if (tid == 0) { o_cnt[0] = nf; o_cnt[1] = nA; o_cnt[2] = tot[2]; o_cnt[3] = 0; }
}

extern "C" {
int mmbert_gemm_tn(hipStream_t stream, const void* A, int lda, const void* B, int ldb, float* W, int ldw,
                   int M, int N, int K, int accumulate, float alpha, const float* alpha_dev, void* slab, float* bias_out);
size_t mmbert_gemm_tn_workspace(int M, int N, int K, int* splits_out);
}

#define LN_MAXV 4   // 4 chunks x 64 lanes x 4 elements = 1024 columns max

// --------------------------------------------------------------------------------------------
// LayerNorm forward: y[out_row(i)] = dropout(LN(x[in_row(i)]))   (dropout index = i*H + col)
// --------------------------------------------------------------------------------------------
// R rows per wave and trip: their loads are issued together (a wave with one 1.5-KB row in flight at a time leaves the memory
// system idle between its dependent reductions; see mmbert_ln_fwd for the measured choice of R).
template <int NV, int R>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, int ldx, const int* __restrict__ in_rows,
                                                     bf16_t* __restrict__ y, int ldy, const int* __restrict__ out_rows,
                                                     int M, int H, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     uint32_t dstream, uint32_t dthr, float dscale, int drop_row0) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    // Software pipeline: the rows of trip t + 1 are requested BEFORE trip t is reduced, normalised and stored, so a wave always has
    // reads in flight under its arithmetic and its stores (one trip per wave -- the round-1/2 launch shape -- made the whole launch
    // "every read, then every store": 2.8 TB/s on cold operands where a plain copy of the same bytes reaches 5.2,
    // tools/ubench/stream_rate.py).
    const int step = gridDim.x * wpb * R;
    // gamma / beta live in registers: loaded inside the loop they would sit BEHIND the next trip's row requests in the in-order
    // vmcnt queue, and waiting for them would wait for those rows too (no pipeline left)
    float4 gam[NV], bet[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = c * 256 + lane * 4;
        gam[c] = col < H ? *(const float4*)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        bet[c] = col < H ? *(const float4*)(beta + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bf16x4 nx[R][NV];
    auto request = [&](int i0) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = min(i0 + rr, M - 1);                 // (a row past the end is computed again and not stored)
            const bf16_t* xr = x + (size_t)(in_rows ? in_rows[i] : i) * ldx;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < H) nx[rr][c] = *(const bf16x4*)(xr + col);
            }
        }
    };
    int i0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * R;
    if (i0 < M) request(i0);
    for (; i0 < M; i0 += step) {
        float v[R][NV][4];
        float s[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            s[rr] = 0.f;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[rr][c][r] = col < H ? bf2f(nx[rr][c][r]) : 0.f; s[rr] += v[rr][c][r]; }
            }
        }
        if (i0 + step < M) request(i0 + step);
        float mean[R], q[R], rstd[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) mean[rr] = wave_sum(s[rr]) / H;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            q[rr] = 0.f;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < H) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = v[rr][c][r] - mean[rr]; q[rr] += d * d; }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) rstd[rr] = rsqrtf(wave_sum(q[rr]) / H + eps);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = c * 256 + lane * 4;
            if (col < H) {
                const float4 g = gam[c], b = bet[c];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int i = i0 + rr;
                    if (i < M) {
                        float o[4] = {(v[rr][c][0] - mean[rr]) * rstd[rr] * g.x + b.x, (v[rr][c][1] - mean[rr]) * rstd[rr] * g.y + b.y,
                                      (v[rr][c][2] - mean[rr]) * rstd[rr] * g.z + b.z, (v[rr][c][3] - mean[rr]) * rstd[rr] * g.w + b.w};
                        if (dthr) {
                            bool k[4];
                            mmb_keep4(dstream, (uint64_t)(i + drop_row0) * H + col, dthr, k);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = k[r] ? o[r] * dscale : 0.f;
                        }
                        bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                        *(bf16x4*)(y + (size_t)(out_rows ? out_rows[i] : i) * ldy + col) = ob;
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = i0 + rr;
            if (lane == 0 && i < M) { if (mean_out) mean_out[i] = mean[rr]; if (rstd_out) rstd_out[i] = rstd[rr]; }
        }
    }
}

// --------------------------------------------------------------------------------------------
// LayerNorm backward.
//   g  = dy[dy_row(i)] (* post-LN dropout mask, index i*H+col, when post_thr)        -- d(LN out)
//   dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
//   dx  -> dx[dx_row(i)]                                                             -- d(LN in)
//   dx2 -> dx * pre-LN branch dropout mask (index i*H+col) when dx2 != null          -- d(GEMM out)
//   dgamma += sum_i g*xhat, dbeta += sum_i g   (fp32 atomics, one per column per workgroup)
// --------------------------------------------------------------------------------------------
// NV = 256-column chunks per row (ceil(H / 256)): accumulators and row registers are sized to it; R = rows per wave and trip
// (their loads are issued together, as in ln_fwd_kernel).
template <int NV, int R>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const int* __restrict__ dy_rows,
                                                     const bf16_t* __restrict__ x, int ldx, const int* __restrict__ x_rows,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, int M, int H,
                                                     bf16_t* __restrict__ dx, int lddx, const int* __restrict__ dx_rows,
                                                     bf16_t* __restrict__ dx2, int lddx2,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias2,
                                                     float* __restrict__ partial,
                                                     uint32_t post_stream, uint32_t post_thr, float post_scale,
                                                     uint32_t pre_stream, uint32_t pre_thr, float pre_scale,
                                                     const int* __restrict__ drop_rows, int dy_row_limit) {
    __shared__ float red[2][4][NV * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[NV][4], ab[NV][4], ad[NV][4];
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ag[c][r] = 0.f; ab[c][r] = 0.f; ad[c][r] = 0.f; }
    const int stride = gridDim.x * 4;
    float4 gam[NV];                                            // in registers, see ln_fwd_kernel
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = c * 256 + lane * 4;
        gam[c] = col < H ? *(const float4*)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // software pipeline as in ln_fwd_kernel: the operands of trip t + 1 are requested before trip t is evaluated and stored
    bf16x4 ndl[R][NV], ntl[R][NV];
    float nmean[R], nrstd[R];
    uint64_t ndi[R];
    bool nlive[R];
    auto request = [&](int i0) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int i = i0 + rr * stride;
            nlive[rr] = i < M;
            const int ic = nlive[rr] ? i : i0;
            // dy_row_limit > 0: a mapped dy row at or past the limit does not exist -- its gradient is zero (the rows the valid-first
            // packing leaves out of backward; wave-uniform)
            const int dr = dy_rows ? dy_rows[ic] : ic;
            const bool has_dy = !(dy_row_limit > 0 && dr >= dy_row_limit);
            const bf16_t* dyr = dy + (size_t)(has_dy ? dr : 0) * lddy;
            const bf16_t* xr = x + (size_t)(x_rows ? x_rows[ic] : ic) * ldx;
            nmean[rr] = mean_in[ic]; nrstd[rr] = rstd_in[ic];
            ndi[rr] = drop_rows ? (uint64_t)drop_rows[ic] : (uint64_t)ic;           // the row the dropout masks were drawn for
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < H) {
                    ndl[rr][c] = has_dy ? *(const bf16x4*)(dyr + col) : (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                    ntl[rr][c] = *(const bf16x4*)(xr + col);
                }
            }
        }
    };
    int i0 = blockIdx.x * 4 + w;
    if (i0 < M) request(i0);
    for (; i0 < M; i0 += stride * R) {
        bf16x4 dl[R][NV], tl[R][NV];
        float mean[R], rstd[R];
        uint64_t di[R];
        bool live[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            mean[rr] = nmean[rr]; rstd[rr] = nrstd[rr]; di[rr] = ndi[rr]; live[rr] = nlive[rr];
#pragma unroll
            for (int c = 0; c < NV; ++c) { dl[rr][c] = ndl[rr][c]; tl[rr][c] = ntl[rr][c]; }
        }
        if (i0 + stride * R < M) request(i0 + stride * R);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (!live[rr]) continue;                           // wave-uniform
            const int i = i0 + rr * stride;
            float g[NV][4], xh[NV][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < H) {
                    const float4 gm = gam[c];
                    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
                    bool k[4] = {true, true, true, true};
                    if (post_thr) mmb_keep4(post_stream, di[rr] * H + col, post_thr, k);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float dv = bf2f(dl[rr][c][r]);
                        if (post_thr) dv = k[r] ? dv * post_scale : 0.f;
                        xh[c][r] = (bf2f(tl[rr][c][r]) - mean[rr]) * rstd[rr];
                        ag[c][r] += dv * xh[c][r];
                        ab[c][r] += dv;
                        g[c][r] = dv * gmv[r];
                        s1 += g[c][r]; s2 += g[c][r] * xh[c][r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { g[c][r] = 0.f; xh[c][r] = 0.f; }
                }
            }
            s1 = wave_sum(s1) / H; s2 = wave_sum(s2) / H;
            bf16_t* dxr = dx + (size_t)(dx_rows ? dx_rows[i] : i) * lddx;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int col = c * 256 + lane * 4;
                if (col < H) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = rstd[rr] * (g[c][r] - s1 - xh[c][r] * s2);
                    bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    *(bf16x4*)(dxr + col) = ob;
                    if (dx2) {
                        if (pre_thr) {
                            bool k[4];
                            mmb_keep4(pre_stream, di[rr] * H + col, pre_thr, k);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = k[r] ? o[r] * pre_scale : 0.f;
                        }
                        bf16x4 o2 = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                        *(bf16x4*)(dx2 + (size_t)i * lddx2 + col) = o2;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) ad[c][r] += o[r];       // column sums of the dense layer's output gradient = its bias gradient
                }
            }
        }
    }
    // workgroup reduction of the gamma/beta partials, then one atomic per column
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[0][w][c * 256 + lane * 4 + r] = ag[c][r];
            red[1][w][c * 256 + lane * 4 + r] = ab[c][r];
        }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
        const float sg = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
        const float sb = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
        // every workgroup adding into the same 3*H addresses is contention-bound (~0.09 TB/s): with a
        // workspace the partials are stored plainly and summed by ln_bwd_reduce_kernel
        if (partial) { partial[((size_t)blockIdx.x * 3 + 0) * H + col] = sg; partial[((size_t)blockIdx.x * 3 + 1) * H + col] = sb; }
        else { if (dgamma) atomicAdd(dgamma + col, sg); if (dbeta) atomicAdd(dbeta + col, sb); }
    }
    if (dbias2) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[0][w][c * 256 + lane * 4 + r] = ad[c][r];
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += 256) {
            const float sd = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
            if (partial) partial[((size_t)blockIdx.x * 3 + 2) * H + col] = sd;
            else atomicAdd(dbias2 + col, sd);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Round 3: the LayerNorm pair for H = NV * 256 (every shipped width: 256 ... 1024), written for the instruction stream.
// tools/ubench/stream_rate.py put the round-1/2 kernels above at 2.4-2.8 TB/s on cold operands where a plain streaming kernel of the
// same bytes reaches 5.2-5.5: (1) one trip per wave made a launch "all reads, then all stores"; (2) gamma / beta / mean / rstd / the
// row maps were vector loads INSIDE the loop -- queued behind the next rows' requests in the in-order vmcnt queue, so every wait for
// them was a wait for those rows; (3) ~590 instructions per backward row (per-column predicates, 64-bit dropout indices, fp32
// selects, ds_bpermute reductions).  Here: rows are requested one trip ahead, everything wave-uniform (row maps, mean, rstd, the
// dropout row) travels through the SCALAR path two trips ahead, gamma / beta sit in registers, dropout is decided on the packed
// bf16 words with a per-lane seed (same words as mmb_keep4: pair index (row * H + col) / 2 in 32-bit arithmetic), arithmetic on
// float pairs, reductions by DPP.  The kernels above stay as the path for every other H.
// --------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
// sum over the 64 lanes, wave-uniform result: quad swaps, half-row and row mirrors, then the two row broadcasts; lane 63 holds the total
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_get<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xF>(v);       // row_half_mirror
    v += dpp_get<0x140, 0xF>(v);       // row_mirror: every lane of a row holds the row's sum
    v += dpp_get<0x142, 0xA>(v);       // row_bcast15 into rows 1, 3
    v += dpp_get<0x143, 0xC>(v);       // row_bcast31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ f32x2 bf2_unpack(uint32_t w) {
    return (f32x2){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
}
__device__ __forceinline__ uint32_t bf2_pack(f32x2 v) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int NV, bool DROP>
__global__ __launch_bounds__(256) void ln_fwd_lean_kernel(const bf16_t* __restrict__ x, int ldx, const int* __restrict__ in_rows,
                                                          bf16_t* __restrict__ y, int ldy, const int* __restrict__ out_rows,
                                                          int M, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          uint32_t dstream, uint32_t dthr, float dscale, int drop_row0) {
    constexpr int H = NV * 256;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int step = gridDim.x * 4;
    int i = blockIdx.x * 4 + w;                                  // wave-uniform: the row maps below are scalar loads
    if (i >= M) return;
    f32x2 gam[NV][2], bet[NV][2];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float4 g = *(const float4*)(gamma + c * 256 + lane * 4), b = *(const float4*)(beta + c * 256 + lane * 4);
        gam[c][0] = (f32x2){g.x, g.y}; gam[c][1] = (f32x2){g.z, g.w};
        bet[c][0] = (f32x2){b.x, b.y}; bet[c][1] = (f32x2){b.z, b.w};
    }
    const uint32_t thr_pk = mmb_thr_packed(dthr);
    const uint32_t lane_seed = (uint32_t)(lane * 2) * MMB_WEYL + dstream;
    u32x2_t nx[NV];
    auto request = [&](int src) {
        const bf16_t* xr = x + (size_t)src * ldx + lane * 4;
#pragma unroll
        for (int c = 0; c < NV; ++c) nx[c] = *(const u32x2_t*)(xr + c * 256);
    };
    auto map_in = [&](int r) { r = min(r, M - 1); return in_rows ? in_rows[r] : r; };
    auto map_out = [&](int r) { r = min(r, M - 1); return out_rows ? out_rows[r] : r; };
    request(map_in(i));
    int src1 = map_in(i + step), dst0 = map_out(i), dst1 = map_out(i + step);     // one and two trips ahead (scalar path)
    for (; i < M; i += step) {
        f32x2 v[NV][2], s2 = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) { v[c][k] = bf2_unpack(nx[c][k]); s2 += v[c][k]; }
        if (i + step < M) request(src1);
        const int dst = dst0;
        dst0 = dst1;
        src1 = map_in(i + 2 * step); dst1 = map_out(i + 2 * step);
        const float mean = wave_sum_dpp(s2.x + s2.y) * (1.0f / H);
        f32x2 q2 = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) { v[c][k] -= mean; q2 = fma2(v[c][k], v[c][k], q2); }
        const float rstd = rsqrtf(wave_sum_dpp(q2.x + q2.y) * (1.0f / H) + eps);
        bf16_t* yr = y + (size_t)dst * ldy + lane * 4;
        const uint32_t row_seed = (uint32_t)(i + drop_row0) * (uint32_t)((H / 2) * MMB_WEYL) + lane_seed;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            u32x2_t ow;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x2 o = fma2(v[c][k] * rstd, gam[c][k], bet[c][k]);
                if (DROP) o *= dscale;
                uint32_t pw = bf2_pack(o);
                if (DROP) pw &= ~mmb_drop_mask2(mmb_pair_mix(row_seed + (uint32_t)(c * 128 + k) * MMB_WEYL), thr_pk);
                ow[k] = pw;
            }
            *(u32x2_t*)(yr + c * 256) = ow;
        }
        if (lane == 0) { if (mean_out) mean_out[i] = mean; if (rstd_out) rstd_out[i] = rstd; }
    }
}

template <int NV, bool POST, bool DX2, bool PRE, int WPB>
__global__ __launch_bounds__(WPB * 64) void ln_bwd_lean_kernel(const bf16_t* __restrict__ dy, int lddy, const int* __restrict__ dy_rows,
                                                          const bf16_t* __restrict__ x, int ldx, const int* __restrict__ x_rows,
                                                          const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                          const float* __restrict__ gamma, int M,
                                                          bf16_t* __restrict__ dx, int lddx, const int* __restrict__ dx_rows,
                                                          bf16_t* __restrict__ dx2, int lddx2,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias2,
                                                          float* __restrict__ partial,
                                                          uint32_t post_stream, uint32_t post_thr, float post_scale,
                                                          uint32_t pre_stream, uint32_t pre_thr, float pre_scale,
                                                          const int* __restrict__ drop_rows, int dy_row_limit) {
    constexpr int H = NV * 256;
    __shared__ float red[WPB][H];                              // WPB waves per workgroup: one column-sum epilogue per WPB waves
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = gridDim.x * WPB;
    const bool want_ad = DX2 || dbias2 != nullptr;
    f32x2 gam[NV][2], ag[NV][2], ab[NV][2], ad[NV][2];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float4 g = *(const float4*)(gamma + c * 256 + lane * 4);
        gam[c][0] = (f32x2){g.x, g.y}; gam[c][1] = (f32x2){g.z, g.w};
#pragma unroll
        for (int k = 0; k < 2; ++k) { ag[c][k] = (f32x2){0.f, 0.f}; ab[c][k] = (f32x2){0.f, 0.f}; ad[c][k] = (f32x2){0.f, 0.f}; }
    }
    const uint32_t post_pk = mmb_thr_packed(post_thr), pre_pk = mmb_thr_packed(pre_thr);
    const uint32_t lane_post = (uint32_t)(lane * 2) * MMB_WEYL + post_stream, lane_pre = (uint32_t)(lane * 2) * MMB_WEYL + pre_stream;
    // everything wave-uniform about a row (scalar path); `dyr` < 0: the mapped dy row does not exist (dy_row_limit) -- zero gradient
    struct Row { int dyr, xr, dxr; uint32_t dr; float mean, rstd; };
    auto row_of = [&](int r) {
        r = min(r, M - 1);
        Row o;
        const int d = dy_rows ? dy_rows[r] : r;
        o.dyr = (dy_row_limit > 0 && d >= dy_row_limit) ? -1 : d;
        o.xr = x_rows ? x_rows[r] : r;
        o.dxr = dx_rows ? dx_rows[r] : r;
        o.dr = drop_rows ? (uint32_t)drop_rows[r] : (uint32_t)r;
        o.mean = mean_in[r]; o.rstd = rstd_in[r];
        return o;
    };
    u32x2_t ndl[NV], ntl[NV];
    auto request = [&](const Row& r) {
        const bf16_t* xr = x + (size_t)r.xr * ldx + lane * 4;
#pragma unroll
        for (int c = 0; c < NV; ++c) ntl[c] = *(const u32x2_t*)(xr + c * 256);
        if (r.dyr >= 0) {                                        // wave-uniform
            const bf16_t* dyr = dy + (size_t)r.dyr * lddy + lane * 4;
#pragma unroll
            for (int c = 0; c < NV; ++c) ndl[c] = *(const u32x2_t*)(dyr + c * 256);
        } else {
#pragma unroll
            for (int c = 0; c < NV; ++c) ndl[c] = (u32x2_t){0u, 0u};
        }
    };
    int i = blockIdx.x * WPB + w;
    if (i < M) {
        Row r0 = row_of(i);
        request(r0);
        Row r1 = row_of(i + stride);
        for (; i < M; i += stride) {
            u32x2_t dl[NV], tl[NV];
#pragma unroll
            for (int c = 0; c < NV; ++c) { dl[c] = ndl[c]; tl[c] = ntl[c]; }
            const Row r = r0;
            r0 = r1;
            if (i + stride < M) request(r0);
            r1 = row_of(i + 2 * stride);
            const uint32_t row_w = r.dr * (uint32_t)((H / 2) * MMB_WEYL);
            const float nmr = -r.mean * r.rstd;
            f32x2 g[NV][2], xh[NV][2], s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    uint32_t dw = dl[c][k];
                    if (POST) dw &= ~mmb_drop_mask2(mmb_pair_mix(row_w + lane_post + (uint32_t)(c * 128 + k) * MMB_WEYL), post_pk);
                    f32x2 dv = bf2_unpack(dw);
                    if (POST) dv *= post_scale;
                    xh[c][k] = fma2(bf2_unpack(tl[c][k]), (f32x2){r.rstd, r.rstd}, (f32x2){nmr, nmr});
                    ag[c][k] = fma2(dv, xh[c][k], ag[c][k]);
                    ab[c][k] += dv;
                    g[c][k] = dv * gam[c][k];
                    s1 += g[c][k];
                    s2 = fma2(g[c][k], xh[c][k], s2);
                }
            const float m1 = wave_sum_dpp(s1.x + s1.y) * (1.0f / H), m2 = wave_sum_dpp(s2.x + s2.y) * (1.0f / H);
            const float c1 = -m1 * r.rstd, c2 = -m2 * r.rstd;
            bf16_t* dxr = dx + (size_t)r.dxr * lddx + lane * 4;
            bf16_t* dx2r = DX2 ? dx2 + (size_t)i * lddx2 + lane * 4 : nullptr;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                u32x2_t ow, ow2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x2 o = fma2(xh[c][k], (f32x2){c2, c2}, fma2(g[c][k], (f32x2){r.rstd, r.rstd}, (f32x2){c1, c1}));
                    ow[k] = bf2_pack(o);
                    if (DX2 && PRE) {
                        const f32x2 o2 = o * pre_scale;
                        const uint32_t dm = mmb_drop_mask2(mmb_pair_mix(row_w + lane_pre + (uint32_t)(c * 128 + k) * MMB_WEYL), pre_pk);
                        ow2[k] = bf2_pack(o2) & ~dm;
                        // the dense layer's bias gradient in fp32: selects on the dropped halves' flag bits.  (hipcc trap: written as
                        // bit_cast(o2.y) & ~mask the second select read o2.x -- both lanes of the pair got the first element)
                        const float klo = (dm & 0x8000u) ? 0.f : o2[0], khi = (dm & 0x80000000u) ? 0.f : o2[1];
                        ad[c][k] += (f32x2){klo, khi};
                    } else {
                        if (DX2) ow2[k] = ow[k];
                        if (want_ad) ad[c][k] += o;
                    }
                }
                *(u32x2_t*)(dxr + c * 256) = ow;
                if (DX2) *(u32x2_t*)(dx2r + c * 256) = ow2;
            }
        }
    }
    // workgroup reduction of the column sums (one quantity at a time through the same LDS block), then one store (or atomic) per column
    auto fold = [&](f32x2 (&acc)[NV][2], int slot, float* out) {
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) *(f32x2*)&red[w][c * 256 + lane * 4 + 2 * k] = acc[c][k];
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += WPB * 64) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < WPB; ++q) t += red[q][col];
            if (partial) partial[((size_t)blockIdx.x * 3 + slot) * H + col] = t;
            else if (out) atomicAdd(out + col, t);
        }
        __syncthreads();
    };
    fold(ag, 0, dgamma);
    fold(ab, 1, dbeta);
    if (dbias2) fold(ad, 2, dbias2);
}

// Round 2: 16-byte-per-lane forms of the two LayerNorm kernels (half a wave per row, bf16x8 accesses, two rows in flight per wave)
// were built and timed against the 8-byte one-wave-per-row kernels above in one process (tools/bench_ln.py at the time): forward
// 13.6 vs 13.7 us, backward in the encoder's form (dx, dx2) 25.6 vs 22.8 us at 18 400 rows, 21.6 vs 20.2 us at 13 745 -- the wider
// accesses buy nothing here (the kernels run 12-25 us: launch ramp and drain, not the access width, separate them from the copy
// ceiling) and the 200+ VGPRs of the wide backward cost occupancy.  Not kept.  What did pay: the encoder's bias gradients ride on
// the weight-gradient GEMM and the gamma / beta partial sums of all its LayerNorms are folded in by ONE launch (below).

// out_k[col] += sum_b partial[b][k][col]  for k = 0 (dgamma), 1 (dbeta), 2 (dbias2, optional), batched: up to 32 LayerNorm backward
// passes of one launch sequence (same H, same block count) are folded into their gradients by ONE launch.
// grid (ceil(H/64), items * 3, 8): a workgroup = 64 columns x 4 interleaved row groups over 1/8 of the partials, LDS-reduced, then
// ONE atomic per column per workgroup (8 adders per address: no contention to speak of).
struct LnReduceBatch { float* out[32][3]; const float* partial[32]; int nblocks[32]; int H, nq, items, ordered; };   // (nblocks per item: the calls of one launch may differ in rows)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_batch_kernel(const LnReduceBatch b) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6, nsub = (int)blockDim.x >> 6;
    const int item = blockIdx.y / b.nq, k = blockIdx.y - item * b.nq;
    float* out = b.out[item][k];
    if (out == nullptr) return;
    if (b.ordered) {
        // Deterministic mode (one z slice, 1024 threads).  Items that share a gradient (the joint embedding's LayerNorm runs once per pair
        // modality, the MLM head's once per row chunk) are folded by the FIRST of them, in item order, and reach the gradient as ONE add:
        // with one adder per item the arrival order decided the rounding as soon as the gradient held a value to accumulate onto
        // (the second micro-batch of an accumulation step) or three items met.
        for (int j = 0; j < item; ++j)
            if (b.out[j][k] == out) return;                             // (workgroup-uniform)
        float total = 0.f;
        for (int it = item; it < b.items; ++it) {
            if (b.out[it][k] != out) continue;
            const float* partial = b.partial[it];
            const int nblocks = b.nblocks[it];
            float s = 0.f;
            if (col < b.H)
                for (int q = sub; q < nblocks; q += nsub) s += partial[((size_t)q * b.nq + k) * b.H + col];
            red[sub][threadIdx.x & 63] = s;
            __syncthreads();
            if (sub == 0) {
                float t = red[0][threadIdx.x];
                for (int q = 1; q < nsub; ++q) t += red[q][threadIdx.x];
                total += t;
            }
            __syncthreads();
        }
        if (sub == 0 && col < b.H) out[col] += total;                  // the only adder of this address
        return;
    }
    const float* partial = b.partial[item];
    const int nblocks = b.nblocks[item];
    const int per = (nblocks + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (col < b.H)
        for (int q = b0 + sub; q < b1; q += nsub) s += partial[((size_t)q * b.nq + k) * b.H + col];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && col < b.H) {
        float t = red[0][threadIdx.x];
        for (int q = 1; q < nsub; ++q) t += red[q][threadIdx.x];
        atomicAdd(out + col, t);
    }
}

// --------------------------------------------------------------------------------------------
// Embedding gather: out[i] = word[ids[i]] + type[tt[i]] + pos[i % T]      (fp32 tables -> bf16)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tts,
                                                           const float* __restrict__ word, const float* __restrict__ type,
                                                           const float* __restrict__ pos, int n, int T, int H, int V,
                                                           bf16_t* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        long id = ids[i]; if (id < 0 || id >= V) id = 0;
        const long tt = tts ? (tts[i] != 0) : 0;
        const float* wr = word + (size_t)id * H;
        const float* tr = type + (size_t)tt * H;
        const float* pr = pos + (size_t)(i % T) * H;
        for (int col = lane * 4; col < H; col += 256) {
            const float4 a = *(const float4*)(wr + col), b = *(const float4*)(tr + col), c = *(const float4*)(pr + col);
            bf16x4 o = {f2bf(a.x + b.x + c.x), f2bf(a.y + b.y + c.y), f2bf(a.z + b.z + c.z), f2bf(a.w + b.w + c.w)};
            *(bf16x4*)(out + (size_t)i * ldo + col) = o;
        }
    }
}

// scatter-add of d(embedding sum): word rows (not row 0: padding_idx), position rows, and the two token-type rows.
// One workgroup per POSITION p (rows p, p + T, p + 2T, ...: the same position of every sequence), a thread owns 4 columns: the
// position row and the token-type rows are summed in registers over the sequences and leave as one add per column (the first
// form gave every row its own atomics: 48 sequences contending for each of the T position rows, 75 us at 2400 rows); only the
// word rows, which rarely collide, take an atomic per row.
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tts,
                                                            const bf16_t* __restrict__ d, int ldd, int n, int T, int H, int V,
                                                            float* __restrict__ gword, float* __restrict__ gtype, float* __restrict__ gpos,
                                                            float* __restrict__ type_slab) {
    // type_slab (deterministic mode, one slice): the position's two token-type sums are STORED to type_slab[p][0 / 1][col] and folded in
    // position order by embed_type_fold_kernel, instead of T workgroups adding to the same two rows in arrival order.
    // (round 4) The loop over the sequences was one dependent memory round trip per row (id -> branch -> atomics: 47 us for 48 rows per
    // workgroup, on the critical path in front of the optimizer): the rows now come in BATCHES of 8 whose loads are issued together, and
    // blockIdx.y takes every gridDim.y-th batch -- one slice by default: more slices LOSE (every slice adds its token-type sums to the
    // same two rows: T x slices same-address atomics per element; 8 slices with a row per trip: 85 us), the rest is the word rows' atomics.
    const int p = blockIdx.x;
    for (int col = threadIdx.x * 4; col < H; col += 1024) {
        float ps[4] = {0.f, 0.f, 0.f, 0.f}, t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = p + (int)blockIdx.y * 8 * T; i0 < n; i0 += 8 * T * (int)gridDim.y) {
            long id[8]; bool tt[8]; bf16x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * T, n - 1);              // (clamped: a row past the end is loaded and not used)
                id[u] = ids[i];
                tt[u] = tts ? (tts[i] != 0) : false;
                v[u] = *(const bf16x4*)(d + (size_t)i * ldd + col);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u * T >= n) break;
                const bool word = gword != nullptr && id[u] > 0 && id[u] < V;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = bf2f(v[u][r]);
                    ps[r] += f;
                    if (tt[u]) t1[r] += f; else t0[r] += f;
                    if (word) atomicAdd(gword + (size_t)id[u] * H + col + r, f);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            atomicAdd(gpos + (size_t)p * H + col + r, ps[r]);       // (one slice: the only adder of this address)
            if (type_slab) { type_slab[((size_t)p * 2) * H + col + r] = t0[r]; type_slab[((size_t)p * 2 + 1) * H + col + r] = t1[r]; }
            else {
                if (t0[r] != 0.f) atomicAdd(gtype + col + r, t0[r]);
                if (t1[r] != 0.f) atomicAdd(gtype + H + col + r, t1[r]);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// JointEmbeddings pair projection: out[out_row0 + b*(T+P) + T + p] = relu(W . feat[b,p] + bias)
// feat fp32 [B*P, D]; W fp32 [H, D].
// --------------------------------------------------------------------------------------------
// On the fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products summed as an fmaf chain over ascending k, as the scalar kernels
// before it summed): one workgroup = 128 hidden columns x 64 feature rows, 4 waves of 64 x 32 (4 x 2 MFMA tiles),
// M = hidden column, N = feature row, so that a lane ends with 4 consecutive columns of one row.  Both operand tiles are copied to LDS
// as they lie in memory (row-major, k padded to a multiple of 4 with zeros; row pitch = 2 mod 16 floats: the 16 rows x 2 k of a
// half-wave's ds_read_b32 fall in 32 different banks).  D beyond 128 runs in chunks of 128.  The sums are fmaf chains over ascending k.
// (Round 1: 16 rows per workgroup, every thread streaming its own W row from global memory, 27 / 54 us for D = 35 / 74 at 8000 rows;
// round 2: 64 x 64 tile with scalar LDS reads, 28 / 43 us; round 4: 128 x 64 tile, K-major LDS, three ds_read_b128 per 32 FMAs,
// 23 / 35 us stand-alone, 30 / 43 us in the step's trace, against a scalar-FMA floor of 5.5 / 11.6 us; this form: 15 / 25 us by
// rocprofv3 (tools/r4_s3_pairprof.sh, profiles/r4_pair_proj.log).)
// FT = float, or double: the reference's collate hands the pair features over as float64 (REF:model_utils.py:94-99) and casts them in
// JointEmbeddings.forward (REF:MMBertEmbedding.py:62,64: `.float()`): read as they are and rounded on load, the cast launches are gone.
template <int KSHIFT, typename FT>
__global__ __launch_bounds__(256) void pair_proj_fwd_kernel(const FT* __restrict__ feat, int n_rows, int P, int D,
                                                            const float* __restrict__ W, const float* __restrict__ bias, int H,
                                                            bf16_t* __restrict__ out, int ldo, int T, int KC, int pitch) {
#if defined(__gfx950__)
    extern __shared__ __attribute__((aligned(16))) float pair_sm[];
    float* Ws = pair_sm;                         // [128][pitch]
    float* Fs = pair_sm + 128 * pitch;           // [64][pitch]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 64, h0 = blockIdx.y * 128;
    const int wh = (wave & 1) * 64, wr = (wave >> 1) * 32;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the bias of the lane's 16 columns, fetched first (clamped in range; unused past H): its latency passes under the staging
    float bcol[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) bcol[i][q] = bias[min(h0 + wh + 16 * i + 4 * g + q, H - 1)];
    // staging: a wave copies whole rows, lane = k (64 or 128 lanes of k per row), up to 32 rows in flight per wave (batches of 8 were
    // a memory round trip each).  Per-lane 32-bit offsets on the two base pointers: with wave-uniform row pointers hipcc runs out of
    // SGPRs, parks them in VGPR lanes and ends up waiting for every load before it issues the next (48 round trips, 20 us)
    constexpr int kw = 1 << KSHIFT, rstep = 256 >> KSHIFT;
    const int kl = tid & (kw - 1), rsub = tid >> KSHIFT;
    for (int k0 = 0; k0 < D; k0 += KC) {
        const int kc = min(KC, D - k0), kc4 = (kc + 3) & ~3;
        if (k0) __syncthreads();
        const float kmask = kl < kc ? 1.f : 0.f;
        const int kg = min(k0 + kl, D - 1);                             // always in range: the load is not predicated (a product, not a
        const bool kst = kl < kc4;                                       // select, zeroes the padding: see pair_wgrad_kernel)
        auto stage = [&](auto base, int g0, int lim, int lrow0, auto nrows_c) {
            constexpr int N = decltype(nrows_c)::value / rstep, NB = N < 32 ? N : 32;
#pragma unroll 1
            for (int ub = 0; ub < N; ub += NB) {
                std::remove_cv_t<std::remove_reference_t<decltype(base[0])>> v[NB];       // (float, or double: converted behind the last load)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int gr = g0 + rsub + (ub + u) * rstep;
                    v[u] = base[(uint32_t)(min(gr, lim - 1) * D + kg)];
                }
                if (kst) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const int r = rsub + (ub + u) * rstep;
                        pair_sm[(lrow0 + r) * pitch + kl] = (float)v[u] * (g0 + r < lim ? kmask : 0.f);
                    }
                }
            }
        };
        stage(W, h0, H, 0, std::integral_constant<int, 128>{});
        stage(feat, row0, n_rows, 128, std::integral_constant<int, 64>{});
        __syncthreads();
        const float* wa = Ws + (wh + c) * pitch + g;
        const float* fb = Fs + (wr + c) * pitch + g;
        float av[4], bv[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = wa[16 * i * pitch];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = fb[16 * j * pitch];
        for (int kk = 0; kk < kc4; kk += 4) {
            float an[4], bn[2];
            const int kn = kk + 4 < kc4 ? kk + 4 : kk;                  // the next step's fragments are read under this step's MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = wa[16 * i * pitch + kn];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = fb[16 * j * pitch + kn];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = an[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bn[j];
        }
    }
    // acc[i][j][q] = column h0 + wh + 16 i + 4 g + q of row row0 + wr + 16 j + c.  Stored from there a wave-instruction is 16 rows x 32
    // bytes, and the kernel is bound by issuing those (20 us at 8000 rows whatever D); so the tile goes through LDS (bf16, [64][128 + 8])
    // and leaves as whole 256-byte rows, 16 bytes per lane
    if (h0 + 128 <= H && !(ldo & 7)) {
        __syncthreads();
        bf16_t* Os = (bf16_t*)pair_sm;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 ov = {f2bf(fmaxf(acc[i][j][0] + bcol[i][0], 0.f)), f2bf(fmaxf(acc[i][j][1] + bcol[i][1], 0.f)),
                             f2bf(fmaxf(acc[i][j][2] + bcol[i][2], 0.f)), f2bf(fmaxf(acc[i][j][3] + bcol[i][3], 0.f))};
                *(bf16x4*)(Os + (wr + 16 * j + c) * 136 + wh + 16 * i + 4 * g) = ov;
            }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int rl = (tid >> 4) + 16 * ps, row = row0 + rl;
            if (row >= n_rows) continue;
            const int b_ = row / P, p_ = row - b_ * P;
            *(bf16x8*)(out + (size_t)(b_ * (T + P) + T + p_) * ldo + h0 + 8 * (tid & 15)) = *(const bf16x8*)(Os + rl * 136 + 8 * (tid & 15));
        }
        return;
    }
    // acc[i][j][q] = column h0 + wh + 16 i + 4 g + q of row row0 + wr + 16 j + c
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + wr + 16 * j + c;
        if (row >= n_rows) continue;
        const int b_ = row / P, p_ = row - b_ * P;
        bf16_t* orow = out + (size_t)(b_ * (T + P) + T + p_) * ldo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = h0 + wh + 16 * i + 4 * g;
            if (h + 3 < H && !(ldo & 3)) {
                bf16x4 ov = {f2bf(fmaxf(acc[i][j][0] + bcol[i][0], 0.f)), f2bf(fmaxf(acc[i][j][1] + bcol[i][1], 0.f)),
                             f2bf(fmaxf(acc[i][j][2] + bcol[i][2], 0.f)), f2bf(fmaxf(acc[i][j][3] + bcol[i][3], 0.f))};
                *(bf16x4*)(orow + h) = ov;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (h + q < H) orow[h + q] = f2bf(fmaxf(acc[i][j][q] + bcol[i][q], 0.f));
            }
        }
    }
#endif
}

// backward: dpre = dJ * (J > 0); dW[h][k] += sum_rows dpre[row][h]*feat[row][k]; db[h] += sum_rows dpre[row][h].
// A tokens-contracted product with a tiny output ([H, D], D = 35 / 74) and fp32 features: it runs on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 products, an fmaf chain over the rows), operands straight from global memory in fragment order:
//   * contraction = rows, 4 per MFMA (lane group g = lane >> 4 holds row r + g); M = 16 hidden columns, N = 16 feature columns;
//   * the A fragment of M-tile i takes column hb + 4 m + i from lane m, so ONE 8-byte load per lane of J and of dJ (columns
//     hb + 4 m .. + 3 of its row: 128 contiguous bytes per row and instruction) feeds the four M-tiles of a wave's 64 columns;
//   * feature column D is the constant 1: the bias gradient is column D of the product, no separate column sum;
//   * a workgroup = 64 hidden columns x one range of rows, its 4 waves a quarter of the range each, summed through LDS in wave order;
//     the per-range partial tiles go to a slab [S][H][D + 1] and pair_wgrad_reduce_kernel adds them to dW / db in range order
//     (deterministic: no atomics).
// (Round 1: a scalar-FMA kernel with one atomicAdd per weight and row chunk, 300 us for D = 74 at 8000 rows; rounds 2-4: a bf16 hi / lo
// split of the features through the grouped TN kernel, four launches and 49 us; this form: two launches, 26 + 4.5 us.)
struct PairBwdArgs { const void* feat; const bf16_t* J; const bf16_t* dJ; float* slab; int n, P, D, T, H, ld, rows_per_wg, ntiles; };
constexpr int PAIR_NT = 5;                       // feature-column tiles (16 columns) per workgroup: D + 1 <= 80 in one pass, more on grid.z

template <typename FT>
__global__ __launch_bounds__(256) void pair_wgrad_kernel(const PairBwdArgs a) {
#if defined(__gfx950__)
    __shared__ float red[3][PAIR_NT * 16][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int hb = blockIdx.x * 64, s = blockIdx.y, jt0 = blockIdx.z * PAIR_NT;
    const int njt = min(PAIR_NT, a.ntiles - jt0);
    const int rpw = a.rows_per_wg >> 2;                                  // rows per wave: a multiple of 16
    const int r0 = s * a.rows_per_wg + wave * rpw, r1 = min(r0 + rpw, a.n);
    const int hcol = hb + 4 * c;
    const bool hok = hcol + 3 < a.H;
    struct Stage { bf16x4 j, d; float f[PAIR_NT]; };
    auto load = [&](Stage& st, int r) {
        const bool ok = r < r1;
        const int rc = ok ? r : 0;
        const int b_ = rc / a.P, p_ = rc - b_ * a.P;
        const size_t off = (size_t)(b_ * (a.T + a.P) + a.T + p_) * a.ld + (hok ? hcol : 0);
        st.j = *(const bf16x4*)(a.J + off);
        st.d = *(const bf16x4*)(a.dJ + off);
        const FT* fr = (const FT*)a.feat + (size_t)rc * a.D;
#pragma unroll
        for (int j = 0; j < PAIR_NT; ++j) {
            const int k = (jt0 + j) * 16 + c;
            // (arithmetic, not a select, on the loaded value: hipcc sinks a load whose only use is one arm of a select into a branch
            // of its own, and every such branch ends in s_waitcnt vmcnt(0) -- the ring would hold one load at a time)
            const float live = (ok && j < njt) ? 1.0f : 0.0f;
            st.f[j] = fmaf((float)fr[min(k, a.D - 1)], k < a.D ? live : 0.0f, k == a.D ? live : 0.0f);
        }
        if (!(ok && hok)) { st.d = (bf16x4){(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f}; }
    };
    f32x4 acc[PAIR_NT][4];
#pragma unroll
    for (int j = 0; j < PAIR_NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // four steps of 4 rows in flight per wave (one wave per SIMD: nothing else hides the memory latency); the ring is unrolled so
    // that a stage is refilled in place -- rotating it through register copies would wait for the newest load in every step.
    // (hipcc still moves a stage's loads from the end of one turn to the top of the next; inline-asm loads behind counted waits ran
    // 21 instead of 26 us, and with eight stages the register allocator copied an asm output before its wait: the load landed in a
    // register that by then held an address -- a memory fault.  Asynchronous asm outputs are not worth 5 us: plain loads stay.)
    Stage st[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(st[u], r0 + 4 * u + g);
#pragma unroll 1
    for (int r = r0; r < r1; r += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = bf2f(st[u].j[i]) > 0.f ? bf2f(st[u].d[i]) : 0.f;
            float fv[PAIR_NT];
#pragma unroll
            for (int j = 0; j < PAIR_NT; ++j) fv[j] = st[u].f[j];
            load(st[u], r + 16 + 4 * u + g);
#pragma unroll
            for (int j = 0; j < PAIR_NT; ++j) {
                if (j >= njt) continue;                               // (wave-uniform: D = 35 has three column tiles, not five)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], fv[j], acc[j][i], 0, 0, 0);
            }
        }
    }
    // waves 1-3 hand their tiles to wave 0, which adds them in wave order and writes the workgroup's partial product
    if (wave) {
#pragma unroll
        for (int j = 0; j < PAIR_NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave - 1][(j * 4 + i) * 4 + q][lane] = acc[j][i][q];
    }
    __syncthreads();
    if (wave == 0) {
        const int W1 = a.D + 1;
        float* out = a.slab + (size_t)s * a.H * W1;
#pragma unroll
        for (int j = 0; j < PAIR_NT; ++j) {
            const int k = (jt0 + j) * 16 + c;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = (j * 4 + i) * 4 + q;
                    const float v = ((acc[j][i][q] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane];
                    const int h = hb + 4 * (4 * g + q) + i;              // D tile: row m = 4 g + q of M-tile i is column hb + 4 m + i
                    if (j < njt && k < W1 && h < a.H) out[(size_t)h * W1 + k] = v;
                }
        }
    }
#endif
}

// dW[h][k] += sum_s slab[s][h][k] (k < D), db[h] += sum_s slab[s][h][D], ranges in ascending order
__global__ void pair_wgrad_reduce_kernel(const float* __restrict__ slab, int S, int H, int D, float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W1 = D + 1;
    if (i >= H * W1) return;
    float t = 0.f;
    int s_ = 0;
    for (; s_ + 8 <= S; s_ += 8) {                                     // eight loads in flight, added in range order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(s_ + u) * H * W1 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; s_ < S; ++s_) t += slab[(size_t)s_ * H * W1 + i];
    const int h = i / W1, k = i - h * W1;
    if (k < D) dW[(size_t)h * D + k] += t; else db[h] += t;
}

// --------------------------------------------------------------------------------------------
// Vocabulary cross-entropy over bf16 logits [M, ldv] (first V columns valid), ignore_index -100.
// Rows are grouped into up to 4 segments (the text / text+visual / text+speech passes): each
// segment's loss is a mean over ITS non-ignored rows (REF get_outputs, one CrossEntropyLoss per pass).
//   ce_count : inv_count[s] = 1/max(1,#valid rows in segment s); loss_sum[s] = 0
//   ce_row<0>: forward  (loss per segment, per-row logsumexp kept for backward)
//   ce_row<1>: backward (dlogits, the upstream gradient of each segment's loss folded in, so the two
//              vocabulary GEMMs of backward run once over all rows instead of once per pass)
// --------------------------------------------------------------------------------------------
// Compact list of the rows that carry a label (0 <= label < V), in row order, plus their count: one workgroup, block scan.
// The MLM head's backward runs on these rows only: every other row of dlogits is exactly zero (ignore_index), so the two
// vocabulary-sized products, LayerNorm', GELU' and the transform products of backward need ~2 % of the rows.
__global__ __launch_bounds__(1024) void active_rows_kernel(const int64_t* __restrict__ labels, int M, int V, int* __restrict__ idx, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 1024) {
        const int i = i0 + tid;
        const bool act = i < M && labels[i] >= 0 && labels[i] < V;
        const unsigned long long bal = __ballot(act);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wsum[q];
        if (act) idx[off + before] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += wsum[q]; base += t; }
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

// One predicate for "row i carries a label" in every kernel of the MLM head (ce_count / ce_row / active_rows): 0 <= label < V.
// -100 is the reference's ignore_index; any other value outside the vocabulary makes torch's CrossEntropyLoss raise in the
// reference -- here the host side counts such labels (model.mlm_active_rows) and raises when it reads the count.
__global__ void ce_count_kernel(const int64_t* __restrict__ labels, int M, int V, const int* __restrict__ seg_bounds, int nseg,
                                float* __restrict__ inv_count, float* __restrict__ loss_sum) {
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    // eight labels per thread in flight, counted in registers (one label at a time, with the segment search reading seg_bounds behind
    // each, the single workgroup made 18 dependent round trips at the headline size: 12 us in front of the loss)
    const int b1 = nseg > 1 ? seg_bounds[1] : 0x7fffffff, b2 = nseg > 2 ? seg_bounds[2] : 0x7fffffff, b3 = nseg > 3 ? seg_bounds[3] : 0x7fffffff;
    int c[4] = {0, 0, 0, 0};
    for (int i0 = threadIdx.x; i0 < M; i0 += (int)blockDim.x * 8) {
        int64_t lab[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lab[u] = labels[min(i0 + u * (int)blockDim.x, M - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            const int ok = (i < M && lab[u] >= 0 && lab[u] < V) ? 1 : 0;
            const int sg = (i >= b1 ? 1 : 0) + (i >= b2 ? 1 : 0) + (i >= b3 ? 1 : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] += (sg == q) ? ok : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) if (c[q]) atomicAdd(&cnt[q], c[q]);
    __syncthreads();
    if (threadIdx.x < nseg) {
        inv_count[threadIdx.x] = 1.0f / (float)max(cnt[threadIdx.x], 1);
        // torch CrossEntropyLoss(mean) over zero valid rows gives nan; the reference never hits it
        loss_sum[threadIdx.x] = 0.f;
    }
}

// Deterministic mode: loss_sum[seg] = the sum of the rows' loss terms (ce_row_kernel left them in row_loss, 0 for unlabelled rows) in a FIXED
// order -- thread t takes rows t, t + 1024, ... ascending, then a fixed-shape tree over the 1024 partial sums -- instead of one atomic per
// row in arrival order.
__global__ __launch_bounds__(1024) void ce_loss_sum_ordered_kernel(const float* __restrict__ row_loss, int M, const int* __restrict__ seg_bounds, int nseg,
                                                                   float* __restrict__ loss_sum) {
    __shared__ float red[4][1024];
    const int b1 = nseg > 1 ? seg_bounds[1] : 0x7fffffff, b2 = nseg > 2 ? seg_bounds[2] : 0x7fffffff, b3 = nseg > 3 ? seg_bounds[3] : 0x7fffffff;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < M; i0 += 1024 * 8) {            // eight rows per thread and trip: their loads go out together
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row_loss[min(i0 + u * 1024, M - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            const int sg = (i >= b1 ? 1 : 0) + (i >= b2 ? 1 : 0) + (i >= b3 ? 1 : 0);
            const float x = i < M ? v[u] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] += (sg == q) ? x : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = c[q];
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < nseg) loss_sum[threadIdx.x] = red[threadIdx.x][0];
}

// Deterministic scatter-add of rows keyed by an id: dst[row_of(id)][:] += sum of src[i][:] over all rows i that carry that id, added in a
// FIXED association -- row group g (of G = blockDim / (H / 4)) sums the id's rows number g, g + G, ... (ascending i), then the G partial
// sums are added in the order g = 0 .. G - 1 -- with exactly one add per destination element, whatever the scheduling.  One workgroup
// per row i: it scans all n ids (a few KB, L2-resident), leaves at once unless i is the FIRST row of its id, and otherwise collects the
// id's rows in ascending order in LDS (wave 0, ballot compaction) and sums them.  No sort, no compaction on the host, nothing read back.
// row_of(id) = id (uni == null: the word-embedding table) or the id's index in the ascending list `uni` (the data-parallel row block);
// ids outside (0, V) or not in the list are skipped.  n <= 8192 (the LDS list).  Replaces the atomics of mmbert_embed_scatter's word rows
// and of mmbert_rows_to_block in deterministic mode.
#define RUNS_MAXN 8192
__global__ __launch_bounds__(1024) void id_runs_sum_rows_kernel(const void* __restrict__ src, int src_bf16, int lds, const int64_t* __restrict__ ids,
                                                                int n, int H, int V, const int64_t* __restrict__ uni, int U,
                                                                float* __restrict__ dst, int ldd) {
    __shared__ int list[RUNS_MAXN];
    __shared__ float red[1024 * 4];
    __shared__ int s_count, s_earlier;
    const int i0 = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const long long key = ids[i0];
    if (key <= 0 || key >= V) return;
    if (tid == 0) { s_count = 0; s_earlier = 0; }
    __syncthreads();
    int earlier = 0;
    for (int j = tid; j < i0; j += nthr) earlier |= (ids[j] == key) ? 1 : 0;
    if (earlier) s_earlier = 1;                                    // (benign: every writer stores 1)
    __syncthreads();
    if (s_earlier) return;
    long long dr = key;
    if (uni) {
        int lo = 0, hi = U - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (uni[mid] < key) lo = mid + 1; else hi = mid; }
        if (U <= 0 || uni[lo] != key) return;
        dr = lo;
    }
    if (tid < 64) {                                                // wave 0: the id's rows from i0 on, ascending, into the list
        int count = 0;
        for (int base = i0; base < n; base += 64) {
            const int j = base + tid;
            const bool hit = j < n && ids[j] == key;
            const unsigned long long m = __ballot(hit);
            if (hit) list[count + __popcll(m & ((1ull << tid) - 1ull))] = j;
            count += __popcll(m);
        }
        if (tid == 0) s_count = count;
    }
    __syncthreads();
    const int cnt = s_count;
    const int CT = H >> 2, G = nthr / CT;                          // column threads, row groups
    const int g = tid / CT, c = tid - g * CT;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G) {
#pragma unroll 4
        for (int k = g; k < cnt; k += G) {
            const size_t r = (size_t)list[k];
            if (src_bf16) {
                const bf16x4 x = *(const bf16x4*)((const bf16_t*)src + r * lds + 4 * c);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] += bf2f(x[q]);
            } else {
                const float4 x = *(const float4*)((const float*)src + r * lds + 4 * c);
                a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(g * CT + c) * 4 + q] = a[q];
    }
    __syncthreads();
    if (g == 0) {
        float4* d = (float4*)(dst + (size_t)dr * ldd + 4 * c);
        float4 o = *d;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += red[(gg * CT + c) * 4 + q];
        o.x += t[0]; o.y += t[1]; o.z += t[2]; o.w += t[3];
        *d = o;
    }
}

// token-type rows of the embedding scatter, deterministic mode: gtype[t][col] += sum over the positions p (ascending) of the per-position
// partial sums the scatter kernel left in slab[p][t][col]
__global__ void embed_type_fold_kernel(const float* __restrict__ slab, int T, int H, float* __restrict__ gtype) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * H) return;
    float t = 0.f;
    for (int p = 0; p < T; ++p) t += slab[(size_t)p * 2 * H + e];
    gtype[e] += t;
}

#define CE_MAXC 16
// One workgroup per row, the row held in registers (<= 16 x 16 B per lane).
//   MODE 0 (forward): row_lse[i] = logsumexp(row); loss_sum[seg] += (lse - logit[label]) * inv_count[seg]
//   MODE 1 (backward): dlogits = (exp(logit - row_lse) - onehot) * inv_count[seg] * gscale[seg]
//                      (0 for ignored rows and for the pad columns V..ldv); gscale = upstream d(loss)/d(loss_seg)
// Ignored rows (label -100) are never read.
// F32: the logits are fp32 (the vocabulary GEMM's EPI_OUT_F32 output, handed to the caller as the reference's fp32 prediction scores);
// they are rounded to bf16 as they are loaded, so losses and gradients are bit-identical to the bf16-logits path.
template <int MODE, bool F32>
__global__ __launch_bounds__(256) void ce_row_kernel(const void* __restrict__ logits_, int ldv, int V, const int64_t* __restrict__ labels,
                                                     const int* __restrict__ seg_bounds, int nseg, const float* __restrict__ inv_count,
                                                     float* __restrict__ loss_sum, float* __restrict__ row_lse, const float* __restrict__ gscale,
                                                     bf16_t* __restrict__ dlogits, int ldd, const int* __restrict__ rows, float* __restrict__ row_loss) {
    __shared__ float red[4];
    __shared__ float lab_logit;
    const int i = rows ? rows[blockIdx.x] : blockIdx.x;        // backward over a compact row list: output row = blockIdx.x
    const int64_t lab = labels[i];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nchunk = ldv >> 3;
    if (lab < 0 || lab >= V) {
        if (MODE == 0) { if (tid == 0) { row_lse[i] = 0.f; if (row_loss) row_loss[i] = 0.f; } }
        else {
            bf16_t* drow = dlogits + (size_t)(rows ? blockIdx.x : i) * ldd;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = tid; c < nchunk; c += 256) *(bf16x8*)(drow + c * 8) = z;
        }
        return;
    }
    int s = 0;
    while (s + 1 < nseg && i >= seg_bounds[s + 1]) ++s;
    bf16x8 v[CE_MAXC];
    if constexpr (F32) {
        const float* row = (const float*)logits_ + (size_t)i * ldv;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            const int ch = c * 256 + tid;
            if (ch < nchunk) {
                const float4 lo = *(const float4*)(row + ch * 8), hi = *(const float4*)(row + ch * 8 + 4);
                v[c] = (bf16x8){f2bf(lo.x), f2bf(lo.y), f2bf(lo.z), f2bf(lo.w), f2bf(hi.x), f2bf(hi.y), f2bf(hi.z), f2bf(hi.w)};
            }
        }
    } else {
        const bf16_t* row = (const bf16_t*)logits_ + (size_t)i * ldv;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            const int ch = c * 256 + tid;
            if (ch < nchunk) v[c] = *(const bf16x8*)(row + ch * 8);
        }
    }
    if (MODE == 0) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            const int ch = c * 256 + tid;
            if (ch < nchunk) {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (ch * 8 + r < V) mx = fmaxf(mx, bf2f(v[c][r]));
            }
        }
        mx = wave_max(mx);
        if (lane == 0) red[w] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            const int ch = c * 256 + tid;
            if (ch < nchunk) {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (ch * 8 + r < V) se += __expf(bf2f(v[c][r]) - mx);
                if ((int)(lab >> 3) == ch) lab_logit = bf2f(v[c][lab & 7]);
            }
        }
        se = wave_sum(se);
        if (lane == 0) red[w] = se;
        __syncthreads();
        if (tid == 0) {
            const float lse = mx + __logf(red[0] + red[1] + red[2] + red[3]);
            row_lse[i] = lse;
            if (row_loss) row_loss[i] = (lse - lab_logit) * inv_count[s];                 // deterministic mode: summed in order by ce_loss_sum_ordered_kernel
            else atomicAdd(loss_sum + s, (lse - lab_logit) * inv_count[s]);
        }
    } else {
        const float lse = row_lse[i];
        const float scale = inv_count[s] * gscale[s];
        bf16_t* drow = dlogits + (size_t)(rows ? blockIdx.x : i) * ldd;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) {
            const int ch = c * 256 + tid;
            if (ch < nchunk) {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int col = ch * 8 + r;
                    float pr = col < V ? __expf(bf2f(v[c][r]) - lse) : 0.f;
                    if (col == (int)lab) pr -= 1.0f;
                    o[r] = f2bf(pr * scale);
                }
                *(bf16x8*)(drow + ch * 8) = o;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Flat AdamW over the model's contiguous fp32 parameter / gradient / moment buffers.
// flags[block of 256 elements]: 0 = no weight decay, 1 = weight decay, 2 = frozen (the
// reference's never-differentiated parameters keep grad None and are skipped by the optimizer);
// + 4 = "the next backward OVERWRITES this block's gradient" (flat.FlatParams lazy zero: the fused zero_grad
// leaves the block alone -- torch's zero_grad(set_to_none=True) does not touch the old gradient either).
// mode 0 = transformers-2.8 AdamW (decay after the update), 1 = torch.optim.AdamW.
// Also writes the bf16 working copy and (optionally) zeroes the gradient: zero_grad fused.
// --------------------------------------------------------------------------------------------
// The scalar coefficients arrive ready-made (mmbert_adamw forms them in double): omb1 / omb2 = 1 - beta1 / 1 - beta2; step_size =
// lr sqrt(bc2) / bc1 (mode 0) or lr / bc1 (mode 1); rsbc2 = 1 / sqrt(bc2) (mode 1); lrwd = lr * wd.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, const uint8_t* __restrict__ flags, size_t n,
                                                    float beta1, float beta2, float omb1, float omb2, float eps, float step_size, float rsbc2, float lrwd,
                                                    float gscale, int mode, int zero_grad) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const uint8_t fb = flags[i >> 8];
    const uint8_t f = fb & 3;
    float4 P = *(float4*)(p + i);
    if (f != 2) {
        float4 G = *(float4*)(g + i), Mm = *(float4*)(m + i), Vv = *(float4*)(v + i);
        float pa[4] = {P.x, P.y, P.z, P.w}, ga[4] = {G.x, G.y, G.z, G.w}, ma[4] = {Mm.x, Mm.y, Mm.z, Mm.w}, va[4] = {Vv.x, Vv.y, Vv.z, Vv.w};
        const float decay = (f == 1) ? lrwd : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gr = ga[r] * gscale;
            if (mode == 1) pa[r] *= (1.0f - decay);
            ma[r] = beta1 * ma[r] + omb1 * gr;
            va[r] = beta2 * va[r] + omb2 * gr * gr;
            if (mode == 0) {
                pa[r] -= step_size * ma[r] / (sqrtf(va[r]) + eps);
                pa[r] -= decay * pa[r];
            } else {
                pa[r] -= step_size * ma[r] / (sqrtf(va[r]) * rsbc2 + eps);
            }
        }
        P = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *(float4*)(p + i) = P;
        *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
    }
    if (zero_grad && !(fb & 4)) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pb) { bf16x4 o = {f2bf(P.x), f2bf(P.y), f2bf(P.z), f2bf(P.w)}; *(bf16x4*)(pb + i) = o; }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const float4 a = *(const float4*)(x + i);
        bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
        *(bf16x4*)(y + i) = o;
    }
}

__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const bf16x4 a = *(const bf16x4*)(x + i);
        *(float4*)(y + i) = make_float4(bf2f(a[0]), bf2f(a[1]), bf2f(a[2]), bf2f(a[3]));
    }
}

// du = dy * gelu'(u)   (MLM head transform, HF:476-480; the encoder's FFN fuses this into its dgrad GEMM)
__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ du, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const bf16x4 a = *(const bf16x4*)(dy + i), b = *(const bf16x4*)(u + i);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(bf2f(a[r]) * gelu_erf_grad(bf2f(b[r])));
        *(bf16x4*)(du + i) = o;
    }
}

// batched transpose-cast: for each descriptor d: dst[c][r] (bf16, ld = dst_ld) = src[r][c] (fp32 or bf16 [rows, cols])
// (the bf16 source is the working copy AdamW has just written: half the bytes to read, the same values)
struct TransDesc { long long src_off, dst_off; int rows, cols, dst_ld, tile0; };
template <typename SRC>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const SRC* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const TransDesc* __restrict__ descs, int ndesc) {
    __shared__ float t[64][65];
    // the matrix this tile belongs to: binary search over the descriptors' first-tile numbers (a linear scan of the ~50 matrices of a
    // 12-layer model was ~25 dependent loads in front of every 16-KB tile: 150 -> 118 us per step for the search alone)
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= descs[mid].tile0) lo = mid; else hi = mid - 1;
    }
    const TransDesc ds = descs[lo];
    const int tl = blockIdx.x - ds.tile0;
    const int tc = (ds.cols + 63) >> 6;
    const int r0 = (tl / tc) << 6, c0 = (tl % tc) << 6;
    const SRC* s = src + ds.src_off;
    bf16_t* o = dst + ds.dst_off;
    const bool full = r0 + 64 <= ds.rows && c0 + 64 <= ds.cols && r0 + 64 <= ds.dst_ld && !(ds.cols & 3) && !(ds.dst_ld & 7) && !(ds.src_off & 3) && !(ds.dst_off & 7);
    if (full) {                                  // interior tile: 16-byte reads along a source row, 16-byte writes along a destination row
        // the thread's four reads are issued together (as a loop hipcc left it rolled: read, wait, LDS store, four times -- four
        // memory round trips per 8-KB tile; the launch took 100 us for 340 MB)
        typedef typename std::conditional<sizeof(SRC) == 4, float4, bf16x4>::type V4;
        V4 v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = threadIdx.x + 256 * q, r = e >> 4, c = (e & 15) << 2;
            v4[q] = *(const V4*)(s + (size_t)(r0 + r) * ds.cols + c0 + c);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = threadIdx.x + 256 * q, r = e >> 4, c = (e & 15) << 2;
            if constexpr (sizeof(SRC) == 4) {
                t[r][c] = v4[q].x; t[r][c + 1] = v4[q].y; t[r][c + 2] = v4[q].z; t[r][c + 3] = v4[q].w;
            } else {
                t[r][c] = bf2f(v4[q][0]); t[r][c + 1] = bf2f(v4[q][1]); t[r][c + 2] = bf2f(v4[q][2]); t[r][c + 3] = bf2f(v4[q][3]);
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 512; e += 256) {
            const int c = e >> 3, r = (e & 7) << 3;
            bf16x8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = f2bf(t[r + j][c]);
            *(bf16x8*)(o + (size_t)(c0 + c) * ds.dst_ld + r0 + r) = ov;
        }
        return;
    }
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        t[r][c] = (r0 + r < ds.rows && c0 + c < ds.cols) ? (float)s[(size_t)(r0 + r) * ds.cols + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (c0 + c < ds.cols && r0 + r < ds.dst_ld) o[(size_t)(c0 + c) * ds.dst_ld + r0 + r] = f2bf(r0 + r < ds.rows ? t[r][c] : 0.f);
    }
}

// Batched row gather: dst_k[i][:] = src_k[idx[i]][:] for up to 12 matrices that share ONE row list (the sparse backward paths gather
// the same few hundred rows of every saved activation: 9 + 6 index_select launches of ~5 us per step before).  Rows are byte
// strings (row_bytes % 4 == 0, 16-byte pieces where the pointers and pitches allow); grid (rows, matrices).
struct GatherSeg { const char* src; char* dst; long long src_pitch, dst_pitch; int row_bytes, vec16; };
struct GatherArgs { GatherSeg s[12]; };
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherArgs a, const int* __restrict__ idx) {
    const GatherSeg s = a.s[blockIdx.y];
    const int i = blockIdx.x;
    const char* src = s.src + (long long)idx[i] * s.src_pitch;
    char* dst = s.dst + (long long)i * s.dst_pitch;
    if (s.vec16) {
        for (int b = threadIdx.x * 16; b < s.row_bytes; b += 256 * 16) *(uint4*)(dst + b) = *(const uint4*)(src + b);
    } else {
        for (int b = threadIdx.x * 4; b < s.row_bytes; b += 256 * 4) *(uint32_t*)(dst + b) = *(const uint32_t*)(src + b);
    }
}

// Concatenation of up to 12 int64 segments into one buffer (token ids, token types and labels of the step's passes: three
// torch.cat launches, two zero fills and their dtype conversions before); a segment without a source is filled with a constant.
struct PackSeg { const int64_t* src; long long dst0, n, fill; };
struct PackArgs { PackSeg s[12]; };
__global__ __launch_bounds__(256) void pack_i64_kernel(const PackArgs a, int64_t* __restrict__ out) {
    const PackSeg s = a.s[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.n; i += (long long)gridDim.x * 256)
        out[s.dst0 + i] = s.src ? s.src[i] : (int64_t)s.fill;
}

// test/debug: keep mask of a dropout site as bytes (so the CPU oracle can replay the same mask)
__global__ void dropout_mask_kernel(uint8_t* __restrict__ out, size_t n, uint32_t stream, uint32_t thr) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = mmb_keep(stream, i, thr) ? 1 : 0;
}

// MLM masking on the device (REF:model_utils.py:6-39): element i draws ONE 32-bit word from the counter RNG; its low half decides
// "selected" (probability sel_thr16 / 65536, never for the special ids), its high half "replaced by [MASK]" (rep_thr16 / 65536 of
// the selected).  labels[i] = the original id where selected, -100 elsewhere; ids are rewritten in place like the reference does.
__global__ void mlm_mask_kernel(int64_t* __restrict__ ids, int64_t* __restrict__ labels, size_t n, uint32_t stream, uint32_t sel_thr16,
                                uint32_t rep_thr16, int64_t special0, int64_t special1, int64_t special2, int64_t mask_id) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        const uint32_t h = mmb_pair_bits(stream, (uint32_t)i);
        const bool special = (id == special0) | (id == special1) | (id == special2);
        const bool sel = !special && (h & 0xFFFFu) < sel_thr16;
        labels[i] = sel ? id : (int64_t)-100;
        if (sel && (h >> 16) < rep_thr16) ids[i] = mask_id;
    }
}


// --------------------------------------------------------------------------------------------
// Step prologue: from the caller's attention masks and MLM labels to everything the encoder's launches need, in TWO launches
// (round 1 spent ~45 element-wise torch launches and two device->host copies on this: 0.8 ms of a 17.5 ms step).
//   A (one workgroup per sequence): the padded additive key bias (REF:MMBertForPretraining.py:57-154: (1 - mask) * -10000 per key,
//     -1e30 in the padding slots), kv_len (keys behind it are all masked out: attn_kv_len_kernel's rule), valid = max(kv_len, last
//     labelled position + 1) (the rows backward must visit: model._split_layout), and per-sequence label counts;
//   B (one workgroup per sequence): the ascending list of labelled rows (mmbert_active_rows' output) and the host words
//     [valid[0..nseq), #labelled rows, #labelled first rows, #labels outside {-100} u [0, V)] -- ONE device->host copy.
// A mask segment = positions [offset, offset + len) of every sequence of one pass, element (b, p) at ptr + b*sb + p*sp bytes, any of
// the dtypes the reference's collate produces (float64 / int64) or torch hands over.
// --------------------------------------------------------------------------------------------
struct MaskSeg { const char* ptr; long long sb, sp; int dtype, pass, offset, len; };
#define PRO_MAXSEG 12
#define PRO_MAXPASS 4
struct PrologueArgs {
    MaskSeg seg[PRO_MAXSEG];
    int nseg, npass, B, vocab;
    int pass_len[PRO_MAXPASS], pass_row0[PRO_MAXPASS], pass_bias0[PRO_MAXPASS];
    const int64_t* labels;
    float* key_bias; int* kv_len; int* valid; int* seq_cnt;      // seq_cnt: [3][nseq] labelled rows / labelled first row / bad labels
    int* idx; int* words;
    // row-set mode (both or neither): rank[row] = position of the row in its sequence's valid-first order, key_bias_perm = the padded
    // key bias in that order; valid[s] then COUNTS the active rows (unmasked key, or labelled) instead of bounding a prefix
    int* rank; float* key_bias_perm;
};

__device__ __forceinline__ float mask_value(const MaskSeg& g, int b, int p) {
    const char* a = g.ptr + (long long)b * g.sb + (long long)p * g.sp;
    switch (g.dtype) {
        case 0: return *(const float*)a;
        case 1: return (float)*(const double*)a;
        case 2: return (float)*(const int64_t*)a;
        case 3: return (float)*(const int*)a;
        case 4: return bf2f(*(const bf16_t*)a);
        case 5: return (float)*(const uint8_t*)a;
        default: return (float)*(const _Float16*)a;
    }
}

__global__ __launch_bounds__(256) void prologue_seq_kernel(const PrologueArgs a) {
    __shared__ int red[4][5];
    const int s = blockIdx.x, p = s / a.B, b = s - p * a.B;
    const int S = a.pass_len[p], row0 = a.pass_row0[p] + b * S;
    const int slots = (S + 127) & ~127;
    float* kb = a.key_bias + a.pass_bias0[p] + b * slots;
    int last_key = -1, last_lab = -1, cnt = 0, first = 0, bad = 0;
    for (int pos = threadIdx.x; pos < slots; pos += 256) {
        float bias = -1.0e30f;
        if (pos < S) {
            float m = 1.0f;
            for (int q = 0; q < a.nseg; ++q) {
                const MaskSeg& g = a.seg[q];
                if (g.pass == p && pos >= g.offset && pos < g.offset + g.len) { m = mask_value(g, b, pos - g.offset); break; }
            }
            bias = (1.0f - m) * -10000.0f;
            if (bias > -10000.0f) last_key = pos;
            if (a.labels) {
                const int64_t lab = a.labels[row0 + pos];
                if (lab != -100) {
                    last_lab = pos;
                    if (lab >= 0 && lab < a.vocab) { ++cnt; if (pos == 0) first = 1; } else ++bad;
                }
            }
        }
        kb[pos] = bias;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        last_key = max(last_key, __shfl_xor(last_key, o, 64)); last_lab = max(last_lab, __shfl_xor(last_lab, o, 64));
        cnt += __shfl_xor(cnt, o, 64); first += __shfl_xor(first, o, 64); bad += __shfl_xor(bad, o, 64);
    }
    if (lane == 0) { red[w][0] = last_key; red[w][1] = last_lab; red[w][2] = cnt; red[w][3] = first; red[w][4] = bad; }
    __syncthreads();
    __shared__ int all_active;
    if (threadIdx.x == 0) {
        const int nseq = a.npass * a.B;
        int lk = -1, ll = -1, c = 0, f = 0, bd = 0;
        for (int q = 0; q < 4; ++q) { lk = max(lk, red[q][0]); ll = max(ll, red[q][1]); c += red[q][2]; f += red[q][3]; bd += red[q][4]; }
        const int kv = lk < 0 ? S : lk + 1;                     // no unmasked key at all: the sequence keeps its full length
        a.kv_len[s] = kv;
        if (!a.rank) a.valid[s] = max(kv, ll + 1);
        a.seq_cnt[s] = c; a.seq_cnt[nseq + s] = f; a.seq_cnt[2 * nseq + s] = bd;
        all_active = lk < 0;
    }
    if (!a.rank) return;
    // ---- row-set mode: a row is ACTIVE iff it is an unmasked key or carries a label (a query with a gradient); the active rows of the
    // sequence come first, both groups keep their order.  Two sweeps over the positions: count, then place (ballot prefix sums).
    __shared__ int wcount[4];
    __shared__ int nact_s;
    __syncthreads();                                               // kb[] of this sequence (pass 1) and all_active are visible
    const bool every = all_active != 0;
    auto active = [&](int pos) {                                   // position 0 always: the heads read the [CLS] row of every sequence
        if (every || pos == 0 || kb[pos] > -10000.0f) return true;
        return a.labels && a.labels[row0 + pos] != -100;
    };
    int mine = 0;
    for (int pos = threadIdx.x; pos < S; pos += 256) mine += active(pos) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) wcount[w] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { nact_s = wcount[0] + wcount[1] + wcount[2] + wcount[3]; a.valid[s] = nact_s; }
    __syncthreads();
    const int nact = nact_s;
    float* kp = a.key_bias_perm + a.pass_bias0[p] + b * slots;
    int base = 0;                                                  // active rows before this chunk
    for (int p0 = 0; p0 < slots; p0 += 256) {
        const int pos = p0 + threadIdx.x;
        const bool in = pos < S, act = in && active(pos);
        const unsigned long long bal = __ballot(act);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wcount[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wcount[q];
        if (in) {
            const int ab = off + before;                           // active rows in front of this position
            const int np = act ? ab : nact + (pos - ab);
            a.rank[row0 + pos] = np;
            kp[np] = kb[pos];
        } else if (pos < slots) {
            kp[pos] = -1.0e30f;
        }
        base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    }
}

__global__ __launch_bounds__(256) void prologue_rows_kernel(const PrologueArgs a) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int s = blockIdx.x, p = s / a.B, b = s - p * a.B, nseq = a.npass * a.B;
    const int S = a.pass_len[p], row0 = a.pass_row0[p] + b * S;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        int off = 0;
        for (int q = 0; q < s; ++q) off += a.seq_cnt[q];
        base_s = off;
        a.words[s] = a.valid[s];
        if (s == 0) {
            int t0 = 0, t1 = 0, t2 = 0;
            for (int q = 0; q < nseq; ++q) { t0 += a.seq_cnt[q]; t1 += a.seq_cnt[nseq + q]; t2 += a.seq_cnt[2 * nseq + q]; }
            a.words[nseq] = t0; a.words[nseq + 1] = t1; a.words[nseq + 2] = t2;
        }
    }
    __syncthreads();
    if (!a.labels || a.seq_cnt[s] == 0) return;
    int base = base_s;
    for (int p0 = 0; p0 < S; p0 += 256) {
        const int pos = p0 + tid;
        bool act = false;
        if (pos < S) { const int64_t lab = a.labels[row0 + pos]; act = lab >= 0 && lab < a.vocab; }
        const unsigned long long bal = __ballot(act);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wsum[q];
        if (act) a.idx[off + before] = row0 + pos;
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

static inline int grid_for(size_t work_items, int per_block, int cap = 2048) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > (size_t)cap ? cap : b);
}

// Compact exchange of embedding-row gradients (data parallel, parallel.exchange_rows): block[pos(ids[i])] += rows[i], pos = the index of
// ids[i] in the ascending list `uni` (the union of touched table rows over all ranks); ids outside (0, V) or not in the list carry
// nothing.  One workgroup per row, the binary search is wave-uniform; rows bf16 or fp32.  (Round 3: eleven ATen launches before.)
__global__ __launch_bounds__(256) void rows_to_block_kernel(const int64_t* __restrict__ ids, const void* __restrict__ rows, int rows_bf16, int ldr,
                                                            int n, int H, const int64_t* __restrict__ uni, int U, int V, float* __restrict__ block) {
    const int i = blockIdx.x;
    if (i >= n || U <= 0) return;
    const long long id = ids[i];
    if (id <= 0 || id >= V) return;
    int lo = 0, hi = U - 1;
    while (lo < hi) {                                          // first position with uni[pos] >= id
        const int mid = (lo + hi) >> 1;
        if (uni[mid] < id) lo = mid + 1; else hi = mid;
    }
    if (uni[lo] != id) return;
    float* dst = block + (size_t)lo * H;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float v = rows_bf16 ? bf2f(((const bf16_t*)rows)[(size_t)i * ldr + c]) : ((const float*)rows)[(size_t)i * ldr + c];
        atomicAdd(dst + c, v);
    }
}

template <typename FT>
static int pair_proj_fwd_launch(hipStream_t stream, const FT* feat, int n, int P, int D, const float* W, const float* bias, int H, void* out, int ldo, int T,
                                int KC, int pitch, int lds) {
    static std::atomic<unsigned long long> attr_done{0};
    if (KC <= 64) {                                                // 64 lanes of k per row (four rows per wave-instruction pair) or 128
        hipLaunchKernelGGL((pair_proj_fwd_kernel<6, FT>), dim3((n + 63) / 64, (H + 127) / 128), dim3(256), lds, stream, feat, n, P, D, W, bias, H, (bf16_t*)out, ldo, T,
                           KC, pitch);
    } else {
        if (int e = mmb_allow_lds((const void*)pair_proj_fwd_kernel<7, FT>, 192 * 130 * (int)sizeof(float), attr_done)) return e;
        hipLaunchKernelGGL((pair_proj_fwd_kernel<7, FT>), dim3((n + 63) / 64, (H + 127) / 128), dim3(256), lds, stream, feat, n, P, D, W, bias, H, (bf16_t*)out, ldo, T,
                           KC, pitch);
    }
    MMB_CHECK_LAUNCH();
    return 0;
}

extern "C" {

uint32_t mmbert_rng_stream(uint64_t seed, uint32_t site) {
    uint32_t a = mmb_hash32((uint32_t)seed ^ 0xA511E9B3u);
    uint32_t b = mmb_hash32((uint32_t)(seed >> 32) + 0x7F4A7C15u);
    return mmb_hash32(a ^ (b * 0x9E3779B1u) ^ mmb_hash32(site * 0x85EBCA6Bu + 0x27D4EB2Fu));
}

uint32_t mmbert_dropout_thr16(float p) {
    if (p <= 0.f) return 0;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

// the lean LayerNorm pair applies (H a multiple of 256)
static inline bool ln_lean(int H) { return (H & 255) == 0 && H <= LN_MAXV * 256; }

int mmbert_ln_fwd(hipStream_t stream, const void* x, int ldx, const int* in_rows, void* y, int ldy, const int* out_rows,
                  int M, int H, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                  uint32_t dstream, uint32_t dthr, float dscale, int drop_row0) {
    if (M <= 0) return 0;
    if (H > LN_MAXV * 256 || (H & 3) || (ldx & 3) || (ldy & 3)) return -1;
    constexpr int cap = 0;
    if (ln_lean(H)) {
        // H = NV * 256: the lean kernel, one row per wave and trip on a grid of <= 1024 workgroups (4 per CU: every wave resident,
        // 4-5 trips per wave at the step's 14-18 k rows; swept 256 ... 4096 with tools/ubench/stream_rate.py)
        const int NVL = H >> 8;
        const dim3 grid(grid_for(M, 4, cap > 0 ? cap : 1024));
#define LN_FWD_LEAN(NVV, DD) hipLaunchKernelGGL((ln_fwd_lean_kernel<NVV, DD>), grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, in_rows, \
                                                 (bf16_t*)y, ldy, out_rows, M, gamma, beta, eps, mean, rstd, dstream, dthr, dscale, drop_row0)
        if (dthr) { if (NVL == 1) LN_FWD_LEAN(1, true); else if (NVL == 2) LN_FWD_LEAN(2, true); else if (NVL == 3) LN_FWD_LEAN(3, true); else LN_FWD_LEAN(4, true); }
        else { if (NVL == 1) LN_FWD_LEAN(1, false); else if (NVL == 2) LN_FWD_LEAN(2, false); else if (NVL == 3) LN_FWD_LEAN(3, false); else LN_FWD_LEAN(4, false); }
#undef LN_FWD_LEAN
        MMB_CHECK_LAUNCH();
        return 0;
    }
    // rows per wave, measured at 18 400 x 768 (stand-alone, same box): 1 / 2 / 4 -> 18.3 / 14.9 / 20.9 us
    const int R = M >= 8192 ? 2 : 1;
    const int NV = (H + 255) / 256;
#define LN_FWD_LAUNCH(NVV, RR) hipLaunchKernelGGL((ln_fwd_kernel<NVV, RR>), dim3(grid_for(M, 4 * RR, cap > 0 ? cap : 2048)), dim3(256), 0, stream, (const bf16_t*)x, ldx, in_rows, \
                                                  (bf16_t*)y, ldy, out_rows, M, H, gamma, beta, eps, mean, rstd, dstream, dthr, dscale, drop_row0)
    if (R >= 2) { if (NV == 1) LN_FWD_LAUNCH(1, 2); else if (NV == 2) LN_FWD_LAUNCH(2, 2); else if (NV == 3) LN_FWD_LAUNCH(3, 2); else LN_FWD_LAUNCH(4, 2); }
    else { if (NV == 1) LN_FWD_LAUNCH(1, 1); else if (NV == 2) LN_FWD_LAUNCH(2, 1); else if (NV == 3) LN_FWD_LAUNCH(3, 1); else LN_FWD_LAUNCH(4, 1); }
#undef LN_FWD_LAUNCH
    MMB_CHECK_LAUNCH();
    return 0;
}

// (block cap swept in round 2 at 13 745 rows, encoder form: 256 / 512 / 1024 / 2048 / 4096 blocks -> 37.6 / 24.3 / 18.5 / 20.3 / 23.3 us)
constexpr int LN_BWD_LEAN_WPB = 8;
static inline int ln_bwd_lean_wpb() { return LN_BWD_LEAN_WPB; }
// workgroups of a backward launch (also the number of partial-sum rows the launch leaves in its workspace)
static inline int ln_bwd_blocks(int M, int H) {
    constexpr int cap_env = 0;
    if (ln_lean(H)) {
        const int wpb = ln_bwd_lean_wpb();
        return grid_for(M, wpb, cap_env > 0 ? cap_env : (wpb == 8 ? 256 : 512));
    }
    return grid_for(M, 4, cap_env > 0 ? cap_env : 1024);
}

int mmbert_ln_bwd(hipStream_t stream, const void* dy, int lddy, const int* dy_rows, const void* x, int ldx, const int* x_rows,
                  const float* mean, const float* rstd, const float* gamma, int M, int H,
                  void* dx, int lddx, const int* dx_rows, void* dx2, int lddx2, float* dgamma, float* dbeta, float* dbias2,
                  uint32_t post_stream, uint32_t post_thr, float post_scale,
                  uint32_t pre_stream, uint32_t pre_thr, float pre_scale, float* partial_ws, const int* drop_rows, int defer_reduce,
                  int dy_row_limit) {
    if (M <= 0) return 0;
    if (H > LN_MAXV * 256 || (H & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) || (lddx2 & 3)) return -1;
    if (dy_row_limit > 0 && !dy_rows) return -1;
    if (defer_reduce && !partial_ws) return -1;
    const int nblocks = ln_bwd_blocks(M, H);
    if (ln_lean(H)) {
        const int NVL = H >> 8;
        const bool post = post_thr != 0, has2 = dx2 != nullptr, pre = has2 && pre_thr != 0;
        // waves per workgroup (one column-sum epilogue per workgroup) and grid: swept with tools/ubench/stream_rate.py on cold operands at
        // 14 000 / 18 400 rows -- 8 waves x 256 workgroups 21.6 / 25.8 us, 4 x 512 22.6 / 26.7, 4 x 1024 24.7 / 28.5, 16 x 256 22.6 / 26.7
        const int wpb = ln_bwd_lean_wpb();
#define LN_BWD_LEAN(NVV, PO, D2, PR, WW) hipLaunchKernelGGL((ln_bwd_lean_kernel<NVV, PO, D2, PR, WW>), dim3(nblocks), dim3(WW * 64), 0, stream, (const bf16_t*)dy, lddy, dy_rows, \
        (const bf16_t*)x, ldx, x_rows, mean, rstd, gamma, M, (bf16_t*)dx, lddx, dx_rows, (bf16_t*)dx2, lddx2, dgamma, dbeta, dbias2, partial_ws, \
        post_stream, post_thr, post_scale, pre_stream, pre_thr, pre_scale, drop_rows, dy_row_limit)
#define LN_BWD_LEAN_W(NVV, WW) { if (post) { if (!has2) LN_BWD_LEAN(NVV, true, false, false, WW); else if (pre) LN_BWD_LEAN(NVV, true, true, true, WW); else LN_BWD_LEAN(NVV, true, true, false, WW); } \
                                 else { if (!has2) LN_BWD_LEAN(NVV, false, false, false, WW); else if (pre) LN_BWD_LEAN(NVV, false, true, true, WW); else LN_BWD_LEAN(NVV, false, true, false, WW); } }
#define LN_BWD_LEAN_NV(NVV) { if (wpb == 8) LN_BWD_LEAN_W(NVV, 8) else LN_BWD_LEAN_W(NVV, 4) }
        if (NVL == 1) LN_BWD_LEAN_NV(1) else if (NVL == 2) LN_BWD_LEAN_NV(2) else if (NVL == 3) LN_BWD_LEAN_NV(3) else LN_BWD_LEAN_NV(4)
#undef LN_BWD_LEAN_NV
#undef LN_BWD_LEAN_W
#undef LN_BWD_LEAN
    } else {
    // measured at 18 400 x 768 (stand-alone, same box): registers sized for 1024 columns (round 1: 142 VGPRs, 3 waves per SIMD) 35.0 us;
    // sized to the row (NV = 3: 116 VGPRs, 4 waves per SIMD = all 4096 waves of the launch resident) 28.5; two rows per trip 34.0
    const int R = 1;
    const int NV = (H + 255) / 256;
#define LN_BWD_LAUNCH(NVV, RR) hipLaunchKernelGGL((ln_bwd_kernel<NVV, RR>), dim3(nblocks), dim3(256), 0, stream, (const bf16_t*)dy, lddy, dy_rows, \
        (const bf16_t*)x, ldx, x_rows, mean, rstd, gamma, M, H, (bf16_t*)dx, lddx, dx_rows, (bf16_t*)dx2, lddx2, dgamma, dbeta, dbias2, partial_ws, \
        post_stream, post_thr, post_scale, pre_stream, pre_thr, pre_scale, drop_rows, dy_row_limit)
    if (R >= 2) { if (NV == 1) LN_BWD_LAUNCH(1, 2); else if (NV == 2) LN_BWD_LAUNCH(2, 2); else if (NV == 3) LN_BWD_LAUNCH(3, 2); else LN_BWD_LAUNCH(4, 2); }
    else { if (NV == 1) LN_BWD_LAUNCH(1, 1); else if (NV == 2) LN_BWD_LAUNCH(2, 1); else if (NV == 3) LN_BWD_LAUNCH(3, 1); else LN_BWD_LAUNCH(4, 1); }
#undef LN_BWD_LAUNCH
    }
    MMB_CHECK_LAUNCH();
    if (partial_ws && !defer_reduce) {
        LnReduceBatch b = {};
        b.out[0][0] = dgamma; b.out[0][1] = dbeta; b.out[0][2] = dbias2; b.partial[0] = partial_ws;
        b.nblocks[0] = nblocks; b.H = H; b.nq = 3; b.items = 1; b.ordered = mmb_deterministic() ? 1 : 0;
        hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((H + 63) / 64, 3, b.ordered ? 1 : 8), dim3(b.ordered ? 1024 : 256), 0, stream, b);
        MMB_CHECK_LAUNCH();
    }
    return 0;
}

// floats the caller must provide as partial_ws for mmbert_ln_bwd (0 = use the atomic path)
size_t mmbert_ln_bwd_workspace(int M, int H) { return (size_t)ln_bwd_blocks(M, H) * 3 * H; }

// Deferred reduction (mmbert_ln_bwd(..., defer_reduce = 1)): folds the partial sums of `items` (<= 32) earlier mmbert_ln_bwd calls
// -- same H, each with its own partial_ws -- into their gradients in one launch.  mmbert_ln_bwd_reduce: all calls had M rows;
// mmbert_ln_bwd_reduce_rows: call i had M[i] rows (the sparse start of backward, the dense layers and the embedding stage in one list).
static int ln_bwd_reduce_launch(hipStream_t stream, int items, const float* const* partial_ws, float* const* dgamma, float* const* dbeta,
                                float* const* dbias2, const int* Ms, int M_all, int H) {
    if (items <= 0) return 0;
    if (items > 32) return -1;
    LnReduceBatch b = {};
    for (int i = 0; i < items; ++i) {
        const int M = Ms ? Ms[i] : M_all;
        b.out[i][0] = M > 0 ? dgamma[i] : nullptr; b.out[i][1] = M > 0 ? dbeta[i] : nullptr; b.out[i][2] = (M > 0 && dbias2) ? dbias2[i] : nullptr;
        b.partial[i] = partial_ws[i];
        b.nblocks[i] = M > 0 ? ln_bwd_blocks(M, H) : 0;
    }
    b.H = H; b.nq = 3; b.items = items; b.ordered = mmb_deterministic() ? 1 : 0;
    hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((H + 63) / 64, items * 3, b.ordered ? 1 : 8), dim3(b.ordered ? 1024 : 256), 0, stream, b);
    MMB_CHECK_LAUNCH();
    return 0;
}
int mmbert_ln_bwd_reduce(hipStream_t stream, int items, const float* const* partial_ws, float* const* dgamma, float* const* dbeta,
                         float* const* dbias2, int M, int H) {
    if (M <= 0) return 0;
    return ln_bwd_reduce_launch(stream, items, partial_ws, dgamma, dbeta, dbias2, nullptr, M, H);
}
int mmbert_ln_bwd_reduce_rows(hipStream_t stream, int items, const float* const* partial_ws, float* const* dgamma, float* const* dbeta,
                              float* const* dbias2, const int* M, int H) {
    if (!M) return -1;
    return ln_bwd_reduce_launch(stream, items, partial_ws, dgamma, dbeta, dbias2, M, 0, H);
}

int mmbert_embed_gather(hipStream_t stream, const int64_t* ids, const int64_t* tts, const float* word, const float* type, const float* pos,
                        int n, int T, int H, int V, void* out, int ldo) {
    if (n <= 0) return 0;
    if ((H & 3) || (ldo & 3)) return -1;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for(n, 4)), dim3(256), 0, stream, ids, tts, word, type, pos, n, T, H, V, (bf16_t*)out, ldo);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_embed_scatter(hipStream_t stream, const int64_t* ids, const int64_t* tts, const void* d, int ldd, int n, int T, int H, int V,
                         float* gword, float* gtype, float* gpos, float* type_slab) {
    if (n <= 0) return 0;
    if (H > LN_MAXV * 256 || (H & 3) || (ldd & 3) || T <= 0) return -1;
    // deterministic mode: the word rows cannot come from this kernel (a word at several positions: atomics in arrival order -- the caller
    // uses mmbert_id_runs_sum_rows) and the token-type sums need the slab (2 * T * H floats)
    if (mmb_deterministic() && (gword != nullptr || type_slab == nullptr)) return -4;
    constexpr int slices = 1;   // grid.y slices of the sequences; measured at the headline shape: 1 / 2 / 4 / 8 = 37.8 / 39.1 / 49.0 / 59.2 us (same-address atomics on the two token-type rows)
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(T < n ? T : n, slices), dim3(256), 0, stream, ids, tts, (const bf16_t*)d, ldd, n, T, H, V, gword, gtype, gpos, type_slab);
    MMB_CHECK_LAUNCH();
    if (type_slab) {
        hipLaunchKernelGGL(embed_type_fold_kernel, dim3((2 * H + 255) / 256), dim3(256), 0, stream, type_slab, T < n ? T : n, H, gtype);
        MMB_CHECK_LAUNCH();
    }
    return 0;
}

int mmbert_pair_proj_fwd(hipStream_t stream, const void* feat, int feat_f64, int B, int P, int D, const float* W, const float* bias, int H,
                         void* out, int ldo, int T) {
    const int n = B * P;
    if (n <= 0) return 0;
    if (D < 1 || H < 1) return -1;
    const int KC = D < 128 ? (D + 3) & ~3 : 128;                 // k chunk in LDS (a multiple of 4); row pitch = 2 mod 16 floats
    const int pitch = KC + ((18 - (KC & 15)) & 15);
    const int lds = 192 * pitch * (int)sizeof(float);             // <= 99 840 bytes
    return feat_f64 ? pair_proj_fwd_launch(stream, (const double*)feat, n, P, D, W, bias, H, out, ldo, T, KC, pitch, lds)
                    : pair_proj_fwd_launch(stream, (const float*)feat, n, P, D, W, bias, H, out, ldo, T, KC, pitch, lds);
}

// workspace: the slab [S][H][D + 1] fp32 of per-row-range partial products.  S row ranges (a multiple of 16 rows each) so that the
// launch has about one workgroup per CU, i.e. one wave per SIMD: twenty independent accumulators keep the matrix pipe busy from one wave
static void pair_bwd_split(int n, int D, int H, int* rows_per_wg, int* S) {
    const int htiles = (H + 63) / 64, zt = ((D + 1 + 15) / 16 + PAIR_NT - 1) / PAIR_NT;
    int want = (mmb_device_cus() + htiles * zt - 1) / (htiles * zt);
    if (want < 1) want = 1;
    int rp = ((n + want - 1) / want + 63) / 64 * 64;             // a multiple of 64: 16 per wave and ring turn
    if (rp < 64) rp = 64;
    *rows_per_wg = rp;
    *S = (n + rp - 1) / rp;
}
size_t mmbert_pair_proj_bwd_workspace(int B, int P, int D, int H) {
    if (B <= 0 || P <= 0 || D < 1 || H < 1) return 0;
    int rp, S;
    pair_bwd_split(B * P, D, H, &rp, &S);
    return (size_t)S * H * (D + 1) * sizeof(float);
}

int mmbert_pair_proj_bwd(hipStream_t stream, const void* feat, int feat_f64, int B, int P, int D, const void* J, const void* dJ, int ld, int T,
                         float* dW, float* db, int H, void* workspace) {
    const int n = B * P;
    if (n <= 0) return 0;
    if (!workspace || (H & 7) || (ld & 7) || D < 1) return -1;
    PairBwdArgs a = {};
    a.feat = feat; a.J = (const bf16_t*)J; a.dJ = (const bf16_t*)dJ; a.slab = (float*)workspace;
    a.n = n; a.P = P; a.D = D; a.T = T; a.H = H; a.ld = ld;
    int S;
    pair_bwd_split(n, D, H, &a.rows_per_wg, &S);
    a.ntiles = (D + 1 + 15) / 16;
    const dim3 grid((H + 63) / 64, S, (a.ntiles + PAIR_NT - 1) / PAIR_NT);
    if (feat_f64) hipLaunchKernelGGL(pair_wgrad_kernel<double>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pair_wgrad_kernel<float>, grid, dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(pair_wgrad_reduce_kernel, dim3((H * (D + 1) + 255) / 256), dim3(256), 0, stream, (const float*)workspace, S, H, D, dW, db);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_ce_fwd(hipStream_t stream, const void* logits, int ldv, int V, const int64_t* labels, int M,
                  const int* seg_bounds, int nseg, float* inv_count, float* loss_sum, float* row_lse, int logits_f32, float* row_loss) {
    if (M <= 0) return 0;
    if (nseg < 1 || nseg > 4 || (ldv & 7) || ldv > CE_MAXC * 256 * 8 || V > ldv) return -1;
    const bool det = mmb_deterministic();
    if (det && !row_loss) return -4;                              // deterministic mode needs the per-row loss buffer (M floats)
    if (!det) row_loss = nullptr;
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(1024), 0, stream, labels, M, V, seg_bounds, nseg, inv_count, loss_sum);
    MMB_CHECK_LAUNCH();
    if (logits_f32)
        hipLaunchKernelGGL((ce_row_kernel<0, true>), dim3(M), dim3(256), 0, stream, logits, ldv, V, labels, seg_bounds, nseg, inv_count, loss_sum,
                           row_lse, (const float*)nullptr, (bf16_t*)nullptr, 0, (const int*)nullptr, row_loss);
    else
        hipLaunchKernelGGL((ce_row_kernel<0, false>), dim3(M), dim3(256), 0, stream, logits, ldv, V, labels, seg_bounds, nseg, inv_count, loss_sum,
                           row_lse, (const float*)nullptr, (bf16_t*)nullptr, 0, (const int*)nullptr, row_loss);
    MMB_CHECK_LAUNCH();
    if (det) {
        hipLaunchKernelGGL(ce_loss_sum_ordered_kernel, dim3(1), dim3(1024), 0, stream, (const float*)row_loss, M, seg_bounds, nseg, loss_sum);
        MMB_CHECK_LAUNCH();
    }
    return 0;
}

void mmbert_set_deterministic(int on) { g_mmb_deterministic.store(on != 0); }
int mmbert_get_deterministic(void) { return g_mmb_deterministic.load(); }

int mmbert_id_runs_sum_rows(hipStream_t stream, const void* src, int src_bf16, int lds, const int64_t* ids, int n,
                            int H, int V, const int64_t* uni, int U, float* dst, int ldd) {
    if (n <= 0) return 0;
    if ((H & 3) || H > 4096 || n > RUNS_MAXN || (lds & 3) || (ldd & 3) || !src || !ids || !dst) return -1;
    const int CT = H >> 2;
    int G = 1024 / CT; if (G > 8) G = 8; if (G < 1) G = 1;
    int threads = G * CT; if (threads < 64) threads = 64;
    hipLaunchKernelGGL(id_runs_sum_rows_kernel, dim3(n), dim3(threads), 0, stream, src, src_bf16, lds, ids, n, H, V, uni, U, dst, ldd);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_ce_bwd(hipStream_t stream, const void* logits, int ldv, int V, const int64_t* labels, int M,
                  const int* seg_bounds, int nseg, const float* inv_count, const float* gscale, const float* row_lse, void* dlogits, int ldd,
                  const int* rows, int nrows, int logits_f32) {
    if (M <= 0 || (rows && nrows <= 0)) return 0;
    if (nseg < 1 || nseg > 4 || (ldv & 7) || (ldd & 7) || ldv > CE_MAXC * 256 * 8 || V > ldv) return -1;
    if (logits_f32)
        hipLaunchKernelGGL((ce_row_kernel<1, true>), dim3(rows ? nrows : M), dim3(256), 0, stream, logits, ldv, V, labels, seg_bounds, nseg, inv_count,
                           (float*)nullptr, (float*)row_lse, gscale, (bf16_t*)dlogits, ldd, rows, (float*)nullptr);
    else
        hipLaunchKernelGGL((ce_row_kernel<1, false>), dim3(rows ? nrows : M), dim3(256), 0, stream, logits, ldv, V, labels, seg_bounds, nseg, inv_count,
                           (float*)nullptr, (float*)row_lse, gscale, (bf16_t*)dlogits, ldd, rows, (float*)nullptr);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_split_rows(hipStream_t stream, const int64_t* row_seq, const int64_t* row_pos, const int* start_a, const int* start_b, const int* valid,
                      int mode, int M, int rows_a, int64_t* perm, int64_t* inv, const int* rank, int* perm32, int* inv32) {
    if (M <= 0) return 0;
    if (mode < 0 || mode > 2) return -1;
    hipLaunchKernelGGL(split_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, row_seq, row_pos, start_a, start_b, valid, mode, M, rows_a, perm, inv, rank,
                       perm32, inv32);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_split_layout(hipStream_t stream, const int* seq_len, const int* valid, int nseq, int tile_rows, int xs, int nf_max, int nq_max, int* out) {
    if (nseq <= 0) return 0;
    if (nseq > SL_MAXSEQ || tile_rows <= 0 || xs <= 0 || !out) return -1;
    hipLaunchKernelGGL(split_layout_kernel, dim3(1), dim3(1024), 0, stream, seq_len, valid, nseq, tile_rows, xs, nf_max, nq_max, out);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_prologue(hipStream_t stream, int nseg, const void* const* seg_ptr, const long long* seg_stride_b, const long long* seg_stride_p,
                    const int* seg_dtype, const int* seg_pass, const int* seg_offset, const int* seg_len,
                    int npass, const int* pass_len, int B, const int64_t* labels, int vocab,
                    float* key_bias, int* kv_len, int* valid, int* seq_cnt, int* idx, int* words, int* rank, float* key_bias_perm) {
    if (npass <= 0 || B <= 0) return 0;
    if (nseg < 0 || nseg > PRO_MAXSEG || npass > PRO_MAXPASS || ((rank == nullptr) != (key_bias_perm == nullptr))) return -1;
    PrologueArgs a = {};
    for (int q = 0; q < nseg; ++q) {
        if (seg_dtype[q] < 0 || seg_dtype[q] > 6 || seg_pass[q] < 0 || seg_pass[q] >= npass) return -1;
        a.seg[q].ptr = (const char*)seg_ptr[q]; a.seg[q].sb = seg_stride_b[q]; a.seg[q].sp = seg_stride_p[q];
        a.seg[q].dtype = seg_dtype[q]; a.seg[q].pass = seg_pass[q]; a.seg[q].offset = seg_offset[q]; a.seg[q].len = seg_len[q];
    }
    int row = 0, slot = 0;
    for (int p = 0; p < npass; ++p) {
        a.pass_len[p] = pass_len[p]; a.pass_row0[p] = row; a.pass_bias0[p] = slot;
        row += B * pass_len[p]; slot += B * ((pass_len[p] + 127) & ~127);
    }
    a.nseg = nseg; a.npass = npass; a.B = B; a.vocab = vocab; a.labels = labels;
    a.key_bias = key_bias; a.kv_len = kv_len; a.valid = valid; a.seq_cnt = seq_cnt; a.idx = idx; a.words = words;
    a.rank = rank; a.key_bias_perm = key_bias_perm;
    hipLaunchKernelGGL(prologue_seq_kernel, dim3(npass * B), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    hipLaunchKernelGGL(prologue_rows_kernel, dim3(npass * B), dim3(256), 0, stream, a);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_active_rows(hipStream_t stream, const int64_t* labels, int M, int V, int* idx, int* count) {
    if (M < 0) return -1;
    hipLaunchKernelGGL(active_rows_kernel, dim3(1), dim3(1024), 0, stream, labels, M, V, idx, count);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_adamw(hipStream_t stream, float* p, float* g, float* m, float* v, void* p_bf16, const uint8_t* flags, size_t n,
                 double lr, double beta1, double beta2, double eps, double wd, int step, double gscale, int mode, int zero_grad) {
    if (n == 0) return 0;
    if (n & 255) return -1;
    // every derived coefficient in double, rounded to fp32 once (the reference's optimizer computes them as Python floats)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = mode == 0 ? lr * sqrt(bc2) / bc1 : lr / bc1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, flags, n,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)step_size, (float)(1.0 / sqrt(bc2)),
                       (float)(lr * wd), (float)gscale, mode, zero_grad);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_gelu_bwd(hipStream_t stream, const void* dy, const void* u, void* du, size_t n) {
    if (n == 0) return 0;
    if (n & 3) return -1;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)du, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_cast_f32_bf16(hipStream_t stream, const float* x, void* y, size_t n) {
    if (n == 0) return 0;
    if (n & 3) return -1;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, x, (bf16_t*)y, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_cast_bf16_f32(hipStream_t stream, const void* x, float* y, size_t n) {
    if (n == 0) return 0;
    if (n & 3) return -1;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
    MMB_CHECK_LAUNCH();
    return 0;
}

// descs: device array of {src_off, dst_off, rows, cols, dst_ld, tile0}; total_tiles = sum of 64x64 tiles
int mmbert_transpose_cast(hipStream_t stream, const float* src, void* dst, const void* descs, int ndesc, int total_tiles) {
    if (ndesc <= 0 || total_tiles <= 0) return 0;
    hipLaunchKernelGGL(transpose_cast_kernel<float>, dim3(total_tiles), dim3(256), 0, stream, src, (bf16_t*)dst, (const TransDesc*)descs, ndesc);
    MMB_CHECK_LAUNCH();
    return 0;
}

// the same from a bf16 source laid out like the fp32 one (same element offsets)
int mmbert_transpose_bf16(hipStream_t stream, const void* src, void* dst, const void* descs, int ndesc, int total_tiles) {
    if (ndesc <= 0 || total_tiles <= 0) return 0;
    hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, (const TransDesc*)descs, ndesc);
    MMB_CHECK_LAUNCH();
    return 0;
}

// out[i] = map[ i < n ? rows[i] : extra[i - n] ]  as int64 AND int32 (map optional): the row list of the top encoder layer's sparse
// backward -- labelled rows, then the [CLS] rows, in the encoder's packed order -- in one launch (round 3: an int32->int64 cast, a cat,
// an index_select through the packing's inverse map and an int64->int32 cast)
// (inv_stamp, optional: the INVERSE of the list for mmbert_scatter_rows_zero -- inv_stamp[out[i]] = stamp << 32 | i in a persistent
// int64 array over all rows that is never cleared: an entry counts only while its upper half equals the current call's stamp)
__global__ void compact_rows_kernel(const int* __restrict__ a32, const int64_t* __restrict__ a64, int n, const int64_t* __restrict__ extra, int nextra,
                                    const int64_t* __restrict__ map, int64_t* __restrict__ out64, int* __restrict__ out32,
                                    long long* __restrict__ inv_stamp, long long stamp_hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + nextra) return;
    int64_t r = i < n ? (a32 ? (int64_t)a32[i] : a64[i]) : extra[i - n];
    if (map) r = map[r];
    out64[i] = r;
    out32[i] = (int)r;
    if (inv_stamp) inv_stamp[r] = stamp_hi | (long long)i;
}
static int compact_rows_launch(hipStream_t stream, const int* rows32, const int64_t* rows64, int n, const int64_t* extra, int nextra, const int64_t* map,
                               int64_t* out64, int* out32, long long* inv_stamp, unsigned stamp) {
    if (n < 0 || nextra < 0 || (n > 0 && !rows32 && !rows64) || (nextra > 0 && !extra)) return -1;
    if (n + nextra == 0) return 0;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((n + nextra + 255) / 256), dim3(256), 0, stream, rows32, rows64, n, extra, nextra, map, out64, out32,
                       inv_stamp, (long long)((unsigned long long)stamp << 32));
    MMB_CHECK_LAUNCH();
    return 0;
}
int mmbert_compact_rows(hipStream_t stream, const int* rows32, const int64_t* rows64, int n, const int64_t* extra, int nextra, const int64_t* map,
                        int64_t* out64, int* out32) {
    return compact_rows_launch(stream, rows32, rows64, n, extra, nextra, map, out64, out32, nullptr, 0u);
}
int mmbert_compact_rows_inv(hipStream_t stream, const int* rows32, const int64_t* rows64, int n, const int64_t* extra, int nextra, const int64_t* map,
                            int64_t* out64, int* out32, int64_t* inv_stamp, unsigned stamp) {
    if (!inv_stamp || stamp == 0u) return -1;
    return compact_rows_launch(stream, rows32, rows64, n, extra, nextra, map, out64, out32, (long long*)inv_stamp, stamp);
}

// dst_k[r] = src_k[i] where the stamped inverse says row r is entry i of the current list, ZERO otherwise, for r < nrows and up to 4
// matrices k: a zero fill and an index_copy_ per matrix in one launch (the sparse top layer's gradients going back to full height).
struct ScatterSeg { const char* src; char* dst; long long src_pitch, dst_pitch; int row_bytes; };
struct ScatterArgs { ScatterSeg s[4]; };
__global__ __launch_bounds__(256) void scatter_rows_zero_kernel(const ScatterArgs a, const long long* __restrict__ inv_stamp, long long stamp_hi, int nlist) {
    const ScatterSeg s = a.s[blockIdx.y];
    const int r = blockIdx.x;
    const long long v = inv_stamp[r];
    const int i = (int)(v & 0xffffffffll);
    const bool hit = (v & ~0xffffffffll) == stamp_hi && i < nlist;
    char* dst = s.dst + (long long)r * s.dst_pitch;
    const char* src = s.src + (long long)(hit ? i : 0) * s.src_pitch;
    for (int b = threadIdx.x * 16; b < s.row_bytes; b += 256 * 16) {
        uint4 x = {0u, 0u, 0u, 0u};
        if (hit) x = *(const uint4*)(src + b);
        *(uint4*)(dst + b) = x;
    }
}
int mmbert_scatter_rows_zero(hipStream_t stream, int nseg, const void* const* src, void* const* dst, const long long* src_pitch, const long long* dst_pitch,
                             const int* row_bytes, const int64_t* inv_stamp, unsigned stamp, int nlist, int nrows) {
    if (nseg <= 0 || nrows <= 0) return 0;
    if (nseg > 4 || !inv_stamp || stamp == 0u || nlist < 0) return -1;
    ScatterArgs a = {};
    for (int k = 0; k < nseg; ++k) {
        if (!dst[k] || (nlist > 0 && !src[k]) || row_bytes[k] <= 0 || ((row_bytes[k] | src_pitch[k] | dst_pitch[k]) & 15) ||
            (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15)) return -1;
        a.s[k].src = (const char*)src[k]; a.s[k].dst = (char*)dst[k]; a.s[k].src_pitch = src_pitch[k]; a.s[k].dst_pitch = dst_pitch[k];
        a.s[k].row_bytes = row_bytes[k];
    }
    hipLaunchKernelGGL(scatter_rows_zero_kernel, dim3(nrows, nseg), dim3(256), 0, stream, a, (const long long*)inv_stamp,
                       (long long)((unsigned long long)stamp << 32), nlist);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_gather_rows(hipStream_t stream, int nseg, const void* const* src, void* const* dst, const long long* src_pitch, const long long* dst_pitch,
                       const int* row_bytes, const int* idx, int nrows) {
    if (nseg <= 0 || nrows <= 0) return 0;
    if (nseg > 12) return -1;
    GatherArgs a = {};
    for (int k = 0; k < nseg; ++k) {
        if (!src[k] || !dst[k] || row_bytes[k] <= 0 || (row_bytes[k] & 3) || (src_pitch[k] & 3) || (dst_pitch[k] & 3) ||
            ((uintptr_t)src[k] & 3) || ((uintptr_t)dst[k] & 3)) return -1;
        a.s[k].src = (const char*)src[k]; a.s[k].dst = (char*)dst[k]; a.s[k].src_pitch = src_pitch[k]; a.s[k].dst_pitch = dst_pitch[k];
        a.s[k].row_bytes = row_bytes[k];
        a.s[k].vec16 = !((row_bytes[k] | src_pitch[k] | dst_pitch[k]) & 15) && !(((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15);
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows, nseg), dim3(256), 0, stream, a, idx);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_rows_to_block(hipStream_t stream, const int64_t* ids, const void* rows, int rows_bf16, int ldr, int n, int H,
                         const int64_t* uni, int U, int V, float* block) {
    if (n <= 0 || U <= 0) return 0;
    if (!ids || !rows || !uni || !block || H <= 0 || ldr < H) return -1;
    hipLaunchKernelGGL(rows_to_block_kernel, dim3(n), dim3(256), 0, stream, ids, rows, rows_bf16, ldr, n, H, uni, U, V, block);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_pack_i64(hipStream_t stream, int nseg, const int64_t* const* src, const long long* dst_offset, const long long* count,
                    const long long* fill, int64_t* out) {
    if (nseg <= 0) return 0;
    if (nseg > 12 || !out) return -1;
    PackArgs a = {};
    long long most = 0;
    for (int k = 0; k < nseg; ++k) {
        if (count[k] < 0 || dst_offset[k] < 0) return -1;
        a.s[k].src = src[k]; a.s[k].dst0 = dst_offset[k]; a.s[k].n = count[k]; a.s[k].fill = fill[k];
        most = count[k] > most ? count[k] : most;
    }
    if (most == 0) return 0;
    hipLaunchKernelGGL(pack_i64_kernel, dim3(grid_for((size_t)most, 256, 256), nseg), dim3(256), 0, stream, a, out);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_mlm_mask(hipStream_t stream, int64_t* ids, int64_t* labels, size_t n, uint32_t rng_stream, uint32_t select_thr16,
                    uint32_t replace_thr16, int64_t special0, int64_t special1, int64_t special2, int64_t mask_id) {
    if (n == 0) return 0;
    if (n >= ((size_t)1 << 32) || select_thr16 > 65536u || replace_thr16 > 65536u) return -1;
    hipLaunchKernelGGL(mlm_mask_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, stream, ids, labels, n, rng_stream, select_thr16,
                       replace_thr16, special0, special1, special2, mask_id);
    MMB_CHECK_LAUNCH();
    return 0;
}

int mmbert_dropout_mask(hipStream_t stream, uint8_t* out, size_t n, uint32_t rng_stream, uint32_t thr16) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, out, n, rng_stream, thr16);
    MMB_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
