/* mmbert_hip.h -- C ABI of libmmbert_hip.so: the MI355X (gfx950) kernels behind the MMBert train step.
 *
 * The reference (kimkyeonghun/MSA) is pure Python and has no FFI: its hot path is the Python module
 * API of MMBertForPretraining / MMBertModel / JointEmbeddings whose arithmetic
 * runs inside HuggingFace BERT modules -> ATen/cuBLAS.  This header is the boundary a maintainer
 * binds instead of those ATen calls (INTEGRATION.md shows the ctypes stubs).  Every entry point:
 *   - is extern "C", takes plain device pointers / sizes / strides (elements) and a hipStream_t,
 *   - returns 0, a negative value for rejected arguments, or a positive hipError_t,
 *   - allocates no device memory, synchronises nothing and keeps no per-call state (graph-capturable): every workspace,
 *     the tile queue of the persistent GEMM included, is passed in by the caller.  Process-global and documented as such:
 *     the one-time per-device opt-in of kernels to > 64 KiB of LDS (hipFuncSetAttribute on first use -- warm each kernel up once
 *     before capturing a graph), a cached CU count per device, and the two test / A-B knobs mmbert_gemm_nt_force and
 *     mmbert_gemm_tn_force_splits (atomics, default 0 = choose by shape; every choice computes the same product).
 *   - reads NO environment variable (round 5: the ~25 MMBERT_* measurement switches of rounds 1-4 are gone with the kernels and
 *     schedules they selected between; the rules they settled are in csrc/gemm.hip nt_choose / tn_plan).  The switches that remain
 *     belong to the Python host side and are read once at import / model construction (msa_amd/_lib.py, ops.py, model.py, build.py):
 *     MMBERT_LIB_PATH (another build of this library), MMBERT_NT_DYNAMIC=1 (tile queue on every stream), MMBERT_DEFER_WGRADS=0/1,
 *     MMBERT_FUSED_HEADS=0, MMBERT_DETERMINISTIC=1 (model.deterministic), MMBERT_HIPCC_FLAGS (build only).
 * bf16 tensors are row-major `uint16` storage; "ld*" are leading dimensions in elements.
 * REF: = /root/reference/<file>:<line>;  HF: = transformers models/bert/modeling_bert.py (5.15.0).
 */
#ifndef MMBERT_HIP_H
#define MMBERT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mmbert_stream_t;   /* == hipStream_t */

/* ---- GEMM epilogue flags (mmbert_gemm_nt.epi) ---- */
#define MMBERT_EPI_BIAS 1      /* + bias[N] (fp32)                                             */
#define MMBERT_EPI_GELU 2      /* out = gelu_erf(v); aux (optional) = v      HF:334-337         */
#define MMBERT_EPI_RESID 4     /* out = dropout(v) + R                       HF:289-293,347-351 */
#define MMBERT_EPI_GELU_BWD 8  /* out = v * gelu'(U)   (dgrad of the FFN down projection)       */
#define MMBERT_EPI_OUT_F32 16  /* C is fp32 instead of bf16                                     */

/* C[M,N] = epi(alpha * alpha_dev[0] * A[M,K] . B[N,K]^T).  Replaces nn.Linear forward (HF:175-177,
 * 289, 334, 347, 476, 493) and, with the transposed bf16 weight copy as B, its input gradient.
 * K % 64 == 0, N % 4 == 0.  Accepted epi: 0, 1, 1|2, 1|4, 4, 8, 16, 1|16. */
/* tile_queue (may be NULL): 16 ints (64 bytes) in device memory, zero before the first launch that uses them -- eight fetch
 * counters, one per XCD (a workgroup draws tiles of its own XCD's share, in order, so the L2 locality of the static walk holds),
 * an exit counter, padding.  The persistent kernel then draws its tiles from this queue instead of the static b, b+G, ...
 * schedule and leaves all of it zero again when it exits, so one 64-byte buffer serves all launches of ONE stream (launches
 * that share a queue must not overlap).  For processes whose
 * GEMMs share the CUs with concurrently running kernels of another stream (data-parallel training: RCCL's channel kernels hold
 * CUs while the gradient all-reduce overlaps backward; a late workgroup would otherwise run its whole static share alone). */
int mmbert_gemm_nt(mmbert_stream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                   int M, int N, int K, int epi, const float* bias, const void* R, int ldr, void* aux, int ldaux,
                   const void* U, int ldu, float alpha, const float* alpha_dev,
                   uint32_t drop_stream, uint32_t drop_thr16, float drop_scale, int* tile_queue);

/* Split-K form of the plain product (C bf16 = A . B^T, + R[M,N] bf16 when R is not null) for long K with few output tiles: fp32
 * partial slabs in the caller's workspace (mmbert_gemm_nt_splitk_workspace() bytes), reduced deterministically. */
int mmbert_gemm_nt_splitk(mmbert_stream_t stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                          int M, int N, int K, void* workspace, const void* R, int ldr);
size_t mmbert_gemm_nt_splitk_workspace(int M, int N, int K);

/* Kernel selection for mmbert_gemm_nt: 0 = by shape (default), 1 = the 128x128-tile kernel, 8 = the 8-phase kernel wherever it is
 * eligible (K % 128 == 0, K >= 256, N % 8 == 0; tile height by the shape rules), 128 / 192 / 224 / 256 = the 8-phase kernel on that
 * tile height (one tile per workgroup, or -- more tiles than CUs -- its multi-tile form; 128-row tiles always one tile per workgroup).
 * For tests and A/B benchmarking; results are identical up to fp32 summation order, and bit-identical between the tile heights. */
void mmbert_gemm_nt_force(int mode);
/* Which kernel mmbert_gemm_nt launches for a shape on the current device (contiguous operands), without launching anything:
 * out[0] kernel (0: the 128x128-tile kernel, 3: the 8-phase kernel), out[1] tile rows (128 / 192 / 224 / 256), out[2] tile columns,
 * out[3] output tiles, out[4] workgroups launched (fewer than tiles: the multi-tile form), out[5] tile rounds x 100 over the
 * device's CUs, out[6] group_m of the tile walk, out[7] CUs.  with_queue: as if a tile_queue were passed.  Host-only; bench.py
 * reports it per shape of the reference's default model (REF:train.py:28,32,38), tests pin the headline shapes. */
int mmbert_gemm_nt_describe(int M, int N, int K, int epi, int with_queue, int* out);
/* Split count of the token axis in mmbert_gemm_tn / _grouped: 0 = by shape (default), > 0 forced.  Tests and A/B benchmarking. */
void mmbert_gemm_tn_force_splits(int splits);
/* 1: a grouped call of more long tiles than CUs goes out as ONE launch instead of one launch per round of CUs-many tiles.  A/B benchmarking. */
void mmbert_gemm_tn_force_one_launch(int on);

/* Weight gradients autograd computes for nn.Linear (REF:trainer.py:83):
 *   W[N,K] (fp32, contiguous: ldw == K) = (accumulate ? W : 0) + alpha * alpha_dev[0] * A[M,N]^T . B[M,K]
 *   bias_out[N] (optional)             += alpha * alpha_dev[0] * column sums of A      (the bias gradient)
 * The token axis may be split into fp32 slabs (deterministic reduce): `slab` must hold *_workspace() bytes.
 * The grouped form runs up to 52 problems that share M in ONE call (the four dense layers of one, two or -- round 4, when nothing needs a
 * layer's weight gradients before the optimizer -- up to twelve encoder layers); host arrays of length nprob.  A call of more than 8
 * problems and more tiles than CUs never splits the token axis and goes out as whole rounds of CUs-many tiles, one launch per round. */
size_t mmbert_gemm_tn_workspace(int M, int N, int K, int* splits_out);
int mmbert_gemm_tn(mmbert_stream_t stream, const void* A, int lda, const void* B, int ldb, float* W, int ldw,
                   int M, int N, int K, int accumulate, float alpha, const float* alpha_dev, void* slab, float* bias_out);
size_t mmbert_gemm_tn_grouped_workspace(int nprob, const int* N, const int* K, int M, int* splits_out);
int mmbert_gemm_tn_grouped(mmbert_stream_t stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                           float* const* W, float* const* bias, const int* N, const int* K, int M,
                           int accumulate, float alpha, const float* alpha_dev, void* slab);
/* ... with a row count per problem (round 6): the problems of M[0] rows first, problems of FEWER rows behind them -- the weight gradients
 * of the few hundred rows that carry a loss (tied decoder, MLM transform: HF:466-496; the top encoder layer's row-sparse sublayers), which
 * autograd computes like all others (REF:trainer.py:83).  Their tiles ride behind the long tiles of the call's LAST launch (its idle CUs
 * work through them); the token axis is not split in such a call.  accumulate_each (may be null: `accumulate` for all): per problem,
 * whether it adds to W or overwrites it. */
int mmbert_gemm_tn_grouped_rows(mmbert_stream_t stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* W, float* const* bias, const int* N, const int* K, const int* M,
                                int accumulate, const int* accumulate_each, float alpha, const float* alpha_dev, void* slab);

/* out[n] += alpha * alpha_dev[0] * sum_m X[m][n]   (bias gradients) */
int mmbert_colsum(mmbert_stream_t stream, const void* X, int ldx, int M, int N, float* out, float alpha, const float* alpha_dev);

/* ---- dropout RNG (counter based; forward and backward regenerate the same mask) ---- */
uint32_t mmbert_rng_stream(uint64_t seed, uint32_t site);
uint32_t mmbert_dropout_thr16(float p);          /* keep iff 16 random bits >= thr16; 0 = no dropout */
int mmbert_dropout_mask(mmbert_stream_t stream, uint8_t* out, size_t n, uint32_t rng_stream, uint32_t thr16);

/* MLM masking of a batch of token ids on the device (REF:model_utils.py:6-39): position i is selected with probability
 * select_thr16 / 65536 unless its id is one of the three special ids (pass an id twice, or -1, for fewer); labels[i] = the id where
 * selected and -100 elsewhere; a selected position becomes mask_id with probability replace_thr16 / 65536 (ids rewritten in
 * place, like the reference).  Counter RNG: the same (rng_stream, n) gives the same masks.  n < 2^32. */
int mmbert_mlm_mask(mmbert_stream_t stream, int64_t* ids, int64_t* labels, size_t n, uint32_t rng_stream, uint32_t select_thr16,
                    uint32_t replace_thr16, int64_t special0, int64_t special1, int64_t special2, int64_t mask_id);

/* ---- LayerNorm (+ the reference's dropout placements) ----
 * fwd: y[out_rows[i]] = dropout(LN(x[in_rows[i]]))              BertEmbeddings HF:104-107,
 *      JointEmbeddings REF:MMBertEmbedding.py:69-70, BertSelfOutput/BertOutput LN HF:292,350.  The dropout mask of row i is the
 *      one of row i + drop_row0 of the dropout site (a launch over a slice of the rows the site covers).
 * bwd: see rowwise.hip; dx2 = dx * (pre-LN branch dropout mask) feeds the dense layer's gradients.  The dropout masks are
 *      functions of (row, column): drop_rows[i] (may be null: i) names the row of the forward pass that compact row i stands for. */
int mmbert_ln_fwd(mmbert_stream_t stream, const void* x, int ldx, const int* in_rows, void* y, int ldy, const int* out_rows,
                  int M, int H, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                  uint32_t dstream, uint32_t dthr, float dscale, int drop_row0);
int mmbert_ln_bwd(mmbert_stream_t stream, const void* dy, int lddy, const int* dy_rows, const void* x, int ldx, const int* x_rows,
                  const float* mean, const float* rstd, const float* gamma, int M, int H,
                  void* dx, int lddx, const int* dx_rows, void* dx2, int lddx2, float* dgamma, float* dbeta, float* dbias2,
                  uint32_t post_stream, uint32_t post_thr, float post_scale,
                  uint32_t pre_stream, uint32_t pre_thr, float pre_scale,
                  float* partial_ws /* mmbert_ln_bwd_workspace() floats, or NULL: contended atomics */, const int* drop_rows,
                  int defer_reduce /* != 0: leave the per-block partial sums in partial_ws for mmbert_ln_bwd_reduce */,
                  int dy_row_limit /* > 0 (needs dy_rows): a mapped dy row >= the limit does not exist, its gradient is zero -- the rows
                                      the valid-first packing leaves out of backward (mmbert_split_rows' inv32 as dy_rows, limit rows_a) */);
size_t mmbert_ln_bwd_workspace(int M, int H);
/* One launch that folds the partial sums of `items` (<= 32) deferred mmbert_ln_bwd calls into their dgamma / dbeta (/ dbias2)
 * gradients (+=).  The calls must share M and H.  Host arrays of `items` pointers; dbias2 (or single entries of it) may be NULL. */
int mmbert_ln_bwd_reduce(mmbert_stream_t stream, int items, const float* const* partial_ws, float* const* dgamma, float* const* dbeta,
                         float* const* dbias2, int M, int H);
/* The same for calls that share H only: call i had M[i] rows (host array).  One list can then hold the LayerNorm' calls of the sparse
 * start of backward (a few hundred rows), of the dense layers and of the embedding stage. */
int mmbert_ln_bwd_reduce_rows(mmbert_stream_t stream, int items, const float* const* partial_ws, float* const* dgamma, float* const* dbeta,
                              float* const* dbias2, const int* M, int H);

/* ---- embeddings ----
 * gather: out[i] = word[ids[i]] + type[tts[i]] + pos[i % T]     HF:96-102 via REF:MMBertForPretraining.py:264
 * scatter: the matching scatter-add of the gradient (row 0 of word excluded: padding_idx, HF:58); gword may be NULL (position and
 *          token-type gradients only).  type_slab (may be NULL; required in deterministic mode): 2 * T * H floats -- the token-type sums
 *          are then stored per position and folded in position order instead of added with atomics. */
int mmbert_embed_gather(mmbert_stream_t stream, const int64_t* ids, const int64_t* tts, const float* word, const float* type,
                        const float* pos, int n, int T, int H, int V, void* out, int ldo);
int mmbert_embed_scatter(mmbert_stream_t stream, const int64_t* ids, const int64_t* tts, const void* d, int ldd, int n, int T, int H, int V,
                         float* gword, float* gtype, float* gpos, float* type_slab);

/* JointEmbeddings pair projection relu(W.feat + b) written after the text rows of each sample:
 * out[(b*(T+P) + T + p)] (REF:MMBertEmbedding.py:61-68); bwd accumulates dW, db.
 * feat: [B*P, D] contiguous, fp32 -- or float64 with feat_f64 != 0: the reference's collate produces float64 features
 * (REF:model_utils.py:94-99) and JointEmbeddings casts them with .float() (REF:MMBertEmbedding.py:62,64); the kernels round on load. */
int mmbert_pair_proj_fwd(mmbert_stream_t stream, const void* feat, int feat_f64, int B, int P, int D, const float* W, const float* bias, int H,
                         void* out, int ldo, int T);
int mmbert_pair_proj_bwd(mmbert_stream_t stream, const void* feat, int feat_f64, int B, int P, int D, const void* J, const void* dJ, int ld, int T,
                         float* dW, float* db, int H, void* workspace /* mmbert_pair_proj_bwd_workspace() bytes */);
size_t mmbert_pair_proj_bwd_workspace(int B, int P, int D, int H);

/* ---- attention (head dim 64) over packed variable-length sequences ----
 * softmax(q.k^T/8 + key_bias) -> dropout -> .v   (HF:111-136); REF mask plumbing: key_bias is
 * (1-mask)*-10000 per key token (REF:MMBertForPretraining.py:57-154,246-250).
 * key_bias is PADDED per sequence: sequence s owns entries bias_start[s] .. bias_start[s] + ceil128(len) - 1, and the
 * entries past its length must be <= -1e30 (they stand for "no such key"; the kernels carry no range logic).
 * tile_seq/tile_r0 list the row tiles (sequence id, first row) of mmbert_attn_tile_rows() rows each, one
 * list for forward and one for backward; elem_base[s] (multiples of 4) are the dropout index bases.
 * kv_len (may be null): per sequence, the count of leading keys behind which EVERY key is masked out (bias <= -10000, the
 * reference's value for a masked key); mmbert_attn_kv_len derives it from the padded key bias.  Those trailing keys have
 * probability exactly 0 in fp32 -- in the reference's softmax too --, so the kernels skip their tiles: same result.
 * Split ("valid-first") layout, optional: the rows of a sequence need not be contiguous for the QUERY side -- tile t reads
 * its queries (and writes their outputs) at packed row tile_qshift[t] + (index in the sequence), for indices below
 * tile_qend[t]; keys/values stay at seq_start[s] + index.  bwd with split != 0 (needs kv_len): a sequence owns only its first
 * kv_len[s] rows at seq_start[s], and only those queries carry a gradient (model.py packs the masked-out rows of all
 * sequences behind the others and runs backward on the leading rows only). */
int mmbert_attn_tile_rows(int which);   /* rows per entry of the tile lists: which = 0 forward, 1 backward */
int mmbert_attn_kv_len(mmbert_stream_t stream, const float* key_bias, const int* bias_start, const int* seq_len, int nseq, int* kv_len);
int mmbert_attn_fwd(mmbert_stream_t stream, const void* qkv, void* ctx, float* lse, const float* key_bias, const int* bias_start, int H, int heads,
                    const int* seq_start, const int* seq_len, const unsigned* elem_base, const int* tile_seq, const int* tile_r0, int ntiles,
                    uint32_t dstream, uint32_t dthr, float dscale, const int* kv_len, const int* tile_qshift, const int* tile_qend);
int mmbert_attn_bwd(mmbert_stream_t stream, const void* qkv, const void* ctx, const void* dctx, void* dqkv, const float* lse, float* delta,
                    const float* key_bias, const int* bias_start, int H, int heads, const int* seq_start, const int* seq_len, const unsigned* elem_base,
                    const int* qtile_seq, const int* qtile_r0, int nqtiles,      /* query tiles: mmbert_attn_tile_rows(0) rows */
                    const int* tile_seq, const int* tile_r0, int ntiles,         /* key tiles:   mmbert_attn_tile_rows(1) rows */
                    uint32_t dstream, uint32_t dthr, float dscale, const int* kv_len,
                    const int* qtile_qshift, const int* qtile_qend, int split,
                    const int* q_limit);     /* optional, per sequence: query rows at index >= q_limit[s] have dO == 0 exactly (see below) */
/* q_limit[s] = 1 + the largest query index (packed row - seq_start[s]) of the int32 row list `rows` inside sequence s (0: none): the
 * rows that CAN have a non-zero output gradient.  For the top encoder layer those are the MLM-labelled rows and the [CLS] rows
 * (REF:MMBertForPretraining.py:381-384, 406-415 read nothing else), all within a sequence's first rows, so its attention backward
 * stops every query loop there: dQ = 0 beyond, and those rows add exact zeros to dK / dV.  seq_start ascending. */
int mmbert_attn_q_limit(mmbert_stream_t stream, const int* rows, int n, const int* seq_start, int nseq, int* q_limit);
int mmbert_attn_dropout_mask(mmbert_stream_t stream, uint8_t* out, int S, unsigned elem_base, int head, uint32_t rng_stream, uint32_t thr16);

/* ---- vocabulary cross-entropy (ignore_index -100), per-pass means ----
 * REF:MMBertForPretraining.py:381-384 (one CrossEntropyLoss(mean) per pass).  Segment s = rows
 * seg_bounds[s]..seg_bounds[s+1].  fwd: loss_sum[s] = mean loss of segment s, row_lse[i] = logsumexp(row i),
 * inv_count[s] = 1/#valid rows.  bwd: dlogits = d(sum_s gscale[s]*loss_sum[s])/d(logits) (may alias logits). */
int mmbert_ce_fwd(mmbert_stream_t stream, const void* logits, int ldv, int V, const int64_t* labels, int M,
                  const int* seg_bounds, int nseg, float* inv_count, float* loss_sum, float* row_lse,
                  int logits_f32 /* logits are fp32 (rounded to bf16 as loaded: same losses as the bf16 form) */,
                  float* row_loss /* M floats, may be NULL; required in deterministic mode: the rows' loss terms, summed in a fixed order */);
int mmbert_ce_bwd(mmbert_stream_t stream, const void* logits, int ldv, int V, const int64_t* labels, int M,
                  const int* seg_bounds, int nseg, const float* inv_count, const float* gscale, const float* row_lse, void* dlogits, int ldd,
                  const int* rows /* NULL: all M rows; else a row list: dlogits row j = gradient of row rows[j] */, int nrows,
                  int logits_f32 /* logits are fp32; dlogits stays bf16 */);
/* Row maps of the valid-first packing (msa_amd/ops.py SplitLayout): inv[original row] = packed row, perm = the
 * inverse, from the per-sequence unmasked lengths.  mode 0: masked-out rows behind all others, in order; 1: ONE shared row per
 * sequence (inference); 2: left out (inv = rows_a).  rank (optional, from mmbert_prologue's row-set mode): the position of every
 * row in its sequence's own valid-first order, used instead of row_pos -- the leading valid[s] rows of a sequence are then its
 * ACTIVE rows wherever they sit (a sequence whose padding lies in the middle: the fused text | visual | speech extension). */
int mmbert_split_rows(mmbert_stream_t stream, const int64_t* row_seq, const int64_t* row_pos, const int* start_a, const int* start_b,
                      const int* valid, int mode, int M, int rows_a, int64_t* perm, int64_t* inv, const int* rank,
                      int* perm32 /* optional int32 copies of the two maps (row lists for mmbert_ln_fwd / _bwd) */, int* inv32);

/* The same packing's per-sequence starts and attention tile lists built on the device from `valid` (mmbert_prologue's output) -- no
 * host round trip in the forward pass.  tile_rows = mmbert_attn_tile_rows(); xs = 8 / gcd(heads, 8) (sequences interleaved in groups of
 * xs so that one (sequence, head) stays on one XCD); lists are sized for the worst case: nq_max = sum ceil(len / tile_rows),
 * nf_max = nq_max + nseq; unused entries carry sequence -1 (mmbert_attn_fwd / _bwd skip them).  out (int32):
 * ftile_seq | ftile_r0 | ftile_qshift | ftile_qend (nf_max each) | tile_seq | tile_r0 | qtile_qshift | qtile_qend (nq_max each) |
 * start_a | valid (clamped) | start_b (nseq each) | nf, nq, rows_a, 0.  nseq <= 1024. */
int mmbert_split_layout(mmbert_stream_t stream, const int* seq_len, const int* valid, int nseq, int tile_rows, int xs, int nf_max, int nq_max,
                        int* out);

/* Step prologue (two launches): from the caller's attention masks and MLM labels to what the encoder's launches need.
 * Sequences: npass passes x B samples, pass p has pass_len[p] positions per sequence; packed rows pass-major, then sample, then
 * position (the order of the token matrix).  Mask segment q (host arrays of nseg <= 12 entries) covers positions
 * [seg_offset, seg_offset + seg_len) of every sequence of pass seg_pass; its element (b, pos) is at
 * seg_ptr + b * seg_stride_b + pos * seg_stride_p BYTES, dtype code 0 f32 / 1 f64 / 2 i64 / 3 i32 / 4 bf16 / 5 u8 / 6 f16
 * (REF:model_utils.py:118-136 hands over float64 and int64 masks; joint passes use feature 0 of the [B,P,D] pair mask,
 * REF:MMBertForPretraining.py:76: pass the tensor's own strides).  labels: int64 per packed row, -100 = none (may be NULL).
 * Outputs (device): key_bias in the padded layout mmbert_attn_fwd reads ((1 - mask) * -10000 per key; ceil128(len) slots per
 * sequence, the padding <= -1e30); kv_len[s] (mmbert_attn_kv_len's rule); valid[s] = max(kv_len[s], last labelled position + 1);
 * idx = the rows with a label in [0, vocab), ascending (mmbert_active_rows' list; room for every row); seq_cnt: 3 * nseq ints of
 * scratch; words = [valid[0..nseq), #labelled rows, #labelled position-0 rows, #labels that are neither -100 nor in [0, vocab)]
 * -- nseq + 3 ints, contiguous for one device->host copy.
 * Row-set mode (rank and key_bias_perm both non-NULL; both NULL otherwise): a row is ACTIVE iff its key is unmasked, it carries a
 * label or it is position 0 of its sequence (the row the heads read); valid[s] = the COUNT of active rows of sequence s, rank[row] = the row's position in the order "active rows first, both
 * groups in their original order" (one int per packed row), key_bias_perm = the padded key bias in that order (same size as
 * key_bias, which still receives the original order).  With mmbert_split_rows(..., rank) the packed sequences then look like
 * prefix-valid ones to every other entry point (attention reads key_bias_perm; its dropout indices follow the new order). */
int mmbert_prologue(mmbert_stream_t stream, int nseg, const void* const* seg_ptr, const long long* seg_stride_b, const long long* seg_stride_p,
                    const int* seg_dtype, const int* seg_pass, const int* seg_offset, const int* seg_len,
                    int npass, const int* pass_len, int B, const int64_t* labels, int vocab,
                    float* key_bias, int* kv_len, int* valid, int* seq_cnt, int* idx, int* words, int* rank, float* key_bias_perm);

/* Data parallel, compact exchange of the embedding lookup's row gradients (msa_amd/parallel.py::exchange_rows; there is no reference
 * counterpart: REF:train.py:22,75 is single-GPU): block[pos(ids[i])][0..H) += rows[i], pos = index of ids[i] in the ascending list
 * uni[0..U) (the union of touched table rows over all ranks); ids outside (0, V) or not in the list are skipped.  rows: bf16
 * (rows_bf16 = 1) or fp32, row pitch ldr elements; block fp32 [U, H], zeroed by the caller. */
int mmbert_rows_to_block(mmbert_stream_t stream, const int64_t* ids, const void* rows, int rows_bf16, int ldr, int n, int H,
                         const int64_t* uni, int U, int V, float* block);

/* out[dst_offset[k] + i] = src[k][i] (or fill[k] where src[k] is NULL), i < count[k], for up to 12 int64 segments in ONE launch: the
 * token ids / token types / MLM labels of a step's passes packed into the order of the token matrix (REF:trainer.py:49-64 hands
 * them over as separate tensors per pass). */
int mmbert_pack_i64(mmbert_stream_t stream, int nseg, const int64_t* const* src, const long long* dst_offset, const long long* count,
                    const long long* fill, int64_t* out);

/* idx[0..count) = the rows with a label in [0, V), ascending; every other row of the CE gradient is exactly zero (ignore_index),
 * so the head's backward may run on this list alone.  idx has room for M entries; count is one int on the device. */
int mmbert_active_rows(mmbert_stream_t stream, const int64_t* labels, int M, int V, int* idx, int* count);

/* ---- pretraining heads (REF:MMBertForPretraining.py:293-301, 399-443; CPC REF:MMBertEmbedding.py:21-32) ----
 * The fp32 [B,H]-sized arithmetic between the heads' dense products (which stay with the caller's BLAS).  Row order of every
 * [3B, H] array: modality-major (text, visual, speech).  See csrc/heads.hip for the notation.
 *   gate_fwd : g[m,b] = relu(Apre[m,b,:]).vw3[m] + vb3[m][0];  C[b, mH+k] = P[m,b,k] * g[m,b]   (vw3/vb3: host arrays of 3 device
 *              pointers -- the vt / vv / vs layers)
 *   loss_fwd : out5 = {ap_loss, label_loss, nce, heads = ap_loss + label_loss - beta*nce, joint = alpha*mean(mlm[0..nmlm)) + heads}
 *              (mlm: device array of the per-pass MLM losses, nmlm = 0: joint = heads; REF :427, :443); seeds of the backward for
 *              upstream 1: dXP, dPc [3B,H] (CPC), drel [2B,2] (alignment CE), dlo [B] (label loss; pre-tanh when tanh_lo);
 *              nce_part: 3 floats
 *   scale    : x *= *s (device scalar: the upstream gradient)
 *   gate_bwd : dg = <dC[b,mH:], P[m,b]> [3B]; dP = dC_m*g + dPc; dApre = dg*vw3[m]*(Apre>0); E = dg*relu(Apre) [3B,H] (the column
 *              sums of E and dg over b are the gradients of the vt / vv / vs layers: mmbert_heads_colsum)
 *   tanh_bwd : dpre = dP*(1-P^2);   colsum: dst_i[c] += sum_r src_i[r][c] for up to 16 segments (bias and gate-vector gradients) */
int mmbert_heads_gate_fwd(mmbert_stream_t stream, const float* P, const float* Apre, const float* const* vw3, const float* const* vb3, int B, int H, float* g, float* C);
int mmbert_heads_loss_fwd(mmbert_stream_t stream, const float* P, const float* XP, const float* rel, const int64_t* ap, const float* lo, const float* sent,
                          int B, int H, float beta, int tanh_lo, float* out5, float* dXP, float* dPc, float* drel, float* dlo, float* nce_part,
                          const float* mlm, int nmlm, float alpha, float* loss_out /* optional: out5[4] once more */, float* aux_out /* optional: out5[0..2] */);
int mmbert_heads_scale(mmbert_stream_t stream, float* x, size_t n, const float* s);
/* The start of the heads' backward in one launch: dst[i] = src[i] * s[0] (i < n: the seeds scaled by the upstream gradient, out of place),
 * zero[j] = 0 (j < nzero: the buffers the backward products are summed into), dmlm[k] = s[0] * coef (k < nmlm <= 256: the gradient handed to
 * the per-pass MLM losses, coef = alpha / passes; REF:MMBertForPretraining.py:427,443). */
int mmbert_heads_seed(mmbert_stream_t stream, const float* src, size_t n, const float* s, float* dst, float* zero, size_t nzero, float* dmlm, int nmlm, float coef);
int mmbert_heads_gate_bwd(mmbert_stream_t stream, const float* dC, const float* P, const float* Apre, const float* g, const float* const* vw3, const float* dPc,
                          int B, int H, float* dP, float* dApre, float* E, float* dg);
int mmbert_heads_tanh(mmbert_stream_t stream, float* x, size_t n);            /* x = tanh(x) in place (the pooler's activation, HF:457-463) */
int mmbert_heads_tanh_bwd(mmbert_stream_t stream, const float* dP, const float* P, float* dpre, size_t n);
int mmbert_heads_colsum(mmbert_stream_t stream, int nseg, const float* const* src, float* const* dst, const int* rows, const int* cols, const int* ld);

/* ---- the pretraining heads, one launch per dependency level (round 6; csrc/heads_coop.hip) ----
 * Everything downstream of the [CLS] rows (pooler HF:457-463; REF:MMBertForPretraining.py:293-301 align / seq_relationship, :399-443 gates,
 * gated concatenation, classifier1_1 / 1_2, losses; CPC REF:MMBertEmbedding.py:21-32) from ONE call that issues seven launches, and its whole
 * backward -- the gradient of the [CLS] rows and of every head parameter -- from one more call (six launches).
 * fp32 throughout (products on the fp32 MFMA), no data atomics: results do not depend on scheduling.
 * Row order of every [3B, H] array: modality-major (text, visual, speech).  B <= 128, H % 16 == 0, the regression head (one output).
 *   first      fp32 [3B, H] (ld H) -- or NULL: row i is row first_rows[i] of the bf16 matrix y (ldy elements per row; ldy % 4 == 0)
 *   ap         int64 [2B] alignment labels (visual rows, then speech rows) -- or ap [B] (visual) and ap2 [B] (speech) when ap2 != NULL; sent fp32 [B]; mlm fp32 [nmlm] per-pass MLM losses (nmlm may be 0)
 *   parameters fp32 in PyTorch's Linear layout [out, in]: Wp (pooler), Wal (align [2,H]), Wsr (seq_relationship [2,H]), Wat (attn [H,2H]),
 *              vw[m] / vb[m] (vt, vv, vs: [H] / [1]), Wc1 (classifier1_1 [H,3H]), Wc2 (classifier1_2 [1,H]), Wq[m] (cpc_z{t,v,a}.net [H,H])
 *   forward    loss[1] = alpha * mean(mlm) + ap_loss + label_loss - beta * nce (REF :427,:443), aux[3] = {ap_loss, label_loss, nce},
 *              out5[5] = {ap, label, nce, heads, joint}, logits[B] (tanh applied when tanh_lo), t_rel [B,2], rel [2B,2]
 *   ws         mmbert_heads_step_workspace(B, H) bytes; written by forward, read and extended by backward (same B, H)
 *   backward   dloss: device scalar (upstream gradient of `loss`); dfirst fp32 [3B, H]; dmlm [nmlm] = dloss * alpha / nmlm;
 *              g*: the parameters' gradients, ACCUMULATED (+=); gWat has row pitch 2H like Wat
 *   sync       4 zero-initialised uint32 in device memory (the loss level's "last workgroup assembles the losses" counter); left zeroed;
 *              one buffer per stream in flight */
typedef struct {
    int B, H, tanh_lo, nmlm;
    float alpha, beta;
    const float* first; const void* y; const int64_t* first_rows; int ldy, pad0_;
    const int64_t* ap; const int64_t* ap2; const float* sent; const float* mlm;
    const float *Wp, *bp, *Wal, *bal, *Wsr, *bsr, *Wat, *bat, *vw[3], *vb[3], *Wc1, *bc1, *Wc2, *bc2, *Wq[3], *bq[3];
    float *loss, *aux, *out5, *logits, *t_rel, *rel;
    float* ws;
    const float* dloss; float* dfirst; float* dmlm;
    float *gWp, *gbp, *gWal, *gbal, *gWat, *gbat, *gvw[3], *gvb[3], *gWc1, *gbc1, *gWc2, *gbc2, *gWq[3], *gbq[3];
    unsigned* sync;
} mmbert_heads_step;
int mmbert_heads_step_struct_size(void);          /* sizeof(mmbert_heads_step): bindings check their mirror of the struct against it */
size_t mmbert_heads_step_workspace(int B, int H);
int mmbert_heads_step_fwd(mmbert_stream_t stream, const mmbert_heads_step* p);
int mmbert_heads_step_bwd(mmbert_stream_t stream, const mmbert_heads_step* p);
/* levels lo .. hi only (forward 1 .. 7, backward 1 .. 6; the calls above = all of them): the heads of REF:MMBertForPretraining.py:293-443 run
 * beside the MLM head's launches on a side stream; forward level 7 (the losses; the only one that reads `mlm`) behind both. */
int mmbert_heads_step_fwd_levels(mmbert_stream_t stream, const mmbert_heads_step* p, int lo, int hi);
int mmbert_heads_step_bwd_levels(mmbert_stream_t stream, const mmbert_heads_step* p, int lo, int hi);
/* dmlm[i] = dloss * alpha / nmlm alone (backward level 6 writes the same): the MLM head's backward needs nothing else of the heads' to start */
int mmbert_heads_step_dmlm(mmbert_stream_t stream, const mmbert_heads_step* p);

/* ---- the heads' dense layers: lists of fp32 products with at most 64 rows, one launch per dependency level ----
 * mmbert_skinny_mm: for every op, Y[M, N] += bias + sum_j X_j . op(W_j) -- the sum is split over workgroups and added with fp32
 * atomics, so Y must hold zeros (or the value to add to) before the call; `act` must be 0 and `accumulate` is informational
 * (an activation goes in its own pass: mmbert_heads_tanh).  Source j
 * contributes to the output rows [row0, row0 + rows) from X_j [rows, inner] (ldx); W_j is a Linear weight [N, inner] (ldw) applied
 * as y = x W^T (w_inner_major = 0: forward layers, HF:457-463, REF:MMBertForPretraining.py:293-301,406-415, REF:MMBertEmbedding.py:22),
 * or [inner, N] applied as y = x W (w_inner_major = 1: the input gradients dX = dY W of the same layers).
 * mmbert_skinny_wgrad: for every op, dW[N, K] (ldw) += dY[M, N]^T . X[M, K] and db[N] += column sums of dY (db may be NULL).
 * nops <= 12, M <= 128 (round 4; 64 before: the reference's default batch of 32 gives [96, H] pooled rows), at most 4 sources per op; ops of one call must not write the same memory. */
typedef struct { const float* X; const float* W; int ldx, ldw, inner, row0, rows, w_inner_major; } mmbert_skinny_src;
typedef struct { float* Y; const float* bias; int ldy, M, N, nsrc, act, accumulate; mmbert_skinny_src src[4]; } mmbert_skinny_op;
typedef struct { const float* dY; const float* X; float* dW; float* db; int ldy, ldx, ldw, M, N, K; } mmbert_skinny_wgrad_op;
int mmbert_skinny_mm(mmbert_stream_t stream, int nops, const mmbert_skinny_op* ops);
/* The same products in DETERMINISTIC form (mmbert_set_deterministic(1) makes mmbert_skinny_mm refuse with -4 and callers use this):
 * every workgroup stores its partial tile into the caller's slab (mmbert_skinny_mm_workspace() bytes) and a second launch adds
 * bias + the chunks in ascending order -- one add per output element. */
size_t mmbert_skinny_mm_workspace(int nops, const mmbert_skinny_op* ops);
int mmbert_skinny_mm_ordered(mmbert_stream_t stream, int nops, const mmbert_skinny_op* ops, void* workspace);
int mmbert_skinny_wgrad(mmbert_stream_t stream, int nops, const mmbert_skinny_wgrad_op* ops);

/* ---- composite encoder-layer calls (round 5): the launches of a layer's forward / of the dense part of its backward from ONE C call ----
 * The same kernels through the same entry points, in the same order and with the same arguments as calling them one by one (bit-identical
 * results); what they save is host time (HF:374-416 BertLayer).  All matrices contiguous bf16 unless a leading dimension is given;
 * weights W* are [out, in], the transposed copies W*T [in, out]. */
typedef struct { uint32_t stream, thr16; float scale; } mmbert_drop;          /* a dropout site: mmbert_rng_stream / mmbert_dropout_thr16 / 1 / (1 - p) */
typedef struct {                                                             /* the attention kernels' view of the packed token matrix */
    const float* key_bias; const int* bias_start; const int* seq_start; const int* seq_len; const unsigned* elem_base;
    const int* ftile_seq; const int* ftile_r0; const int* ftile_qshift; const int* ftile_qend;      /* forward query tiles */
    const int* qtile_seq; const int* qtile_r0; const int* qtile_qshift; const int* qtile_qend;      /* backward query tiles */
    const int* tile_seq; const int* tile_r0;                                                        /* backward key tiles */
    const int* kv_len;
    int nftiles, nqtiles, ntiles, split, heads, pad_;
} mmbert_attn_layout;
typedef struct {
    const void* x;                                                           /* layer input [rows, H], row pitch ldx */
    const void* Wqkv; const void* Wo; const void* W1; const void* W2;
    const float* bqkv; const float* bo; const float* b1; const float* b2; const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
    void* qkv; void* actx; float* lse; void* z1; void* y1; float* m1; float* r1; void* u /* may be NULL */; void* g; void* z2; void* y2; float* m2; float* r2;
    const int* y2_rows;                                                      /* LayerNorm 2 stores row i at y2[y2_rows[i]] (NULL: row i), pitch ldy2 */
    int* tile_queue;
    mmbert_drop att, h1, h2;                                                 /* attention probabilities, the two hidden dropouts */
    int rows, H, I, ldx, ldy2; float ln_eps;
} mmbert_layer_fwd_args;
typedef struct {
    const void* dy; const int* dy_rows;                                      /* gradient of the layer output (pitch lddy; read through dy_rows when given) */
    const void* z2; const float* m2; const float* r2; const float* ln2_g; float* g_ln2_g; float* g_ln2_b; float* ln2_ws;   /* ln*_ws: mmbert_ln_bwd_workspace() floats, */
    const void* z1; const float* m1; const float* r1; const float* ln1_g; float* g_ln1_g; float* g_ln1_b; float* ln1_ws;   /* folded later by mmbert_ln_bwd_reduce_rows  */
    const void* u; const void* qkv; const void* actx; const float* lse;
    const void* W2T; const void* W1T; const void* WoT; const void* WqkvT;
    void* dz2; void* dz2d /* NULL without hidden dropout */; void* du; void* dy1; void* dz1; void* dz1d /* NULL without */; void* dctx; void* dqkv; float* delta; void* dx;
    int* tile_queue;
    mmbert_drop att, h1, h2;
    int rows, H, I, lddy;
} mmbert_layer_bwd_args;
int mmbert_layer_fwd(mmbert_stream_t stream, const mmbert_attn_layout* layout, const mmbert_layer_fwd_args* a);
/* dx = the gradient of the layer input; the caller then launches the layer's weight gradients from (du, y1), (dz2d, g), (dqkv, x), (dz1d, actx). */
int mmbert_layer_bwd(mmbert_stream_t stream, const mmbert_attn_layout* layout, const mmbert_layer_bwd_args* a);
int mmbert_layer_struct_sizes(int* out /* [3]: sizeof the three structures above, for a binding's self-check */);

/* ---- deterministic mode (round 5) ----
 * mmbert_set_deterministic(1): every fp32 sum of the library is formed in an order that does not depend on how workgroups are scheduled --
 * the CE loss sums (an ordered one-workgroup sum instead of an atomic per row), the weight-gradient kernel (no token split: one adder per
 * bias column), the LayerNorm partial-sum fold and mmbert_colsum (one adder per address), mmbert_skinny_mm (-> _ordered).  The two
 * scatter-adds with data-dependent collisions -- mmbert_embed_scatter's word / token-type rows and mmbert_rows_to_block -- have no ordered
 * form of their own: a deterministic caller uses mmbert_id_runs_sum_rows;
 * mmbert_embed_scatter then takes a slab for its token-type sums and refuses a word table (-4).  Process-global like the force
 * knobs; the same seeded step then gives bit-identical losses and gradients run to run (tests/test_train_gpu.py); default 0. */
void mmbert_set_deterministic(int on);
int mmbert_get_deterministic(void);
/* dst[row_of(id)][0..H) += the sum of src[i][0..H) over all rows i with ids[i] == id, for every id, each sum formed in a fixed association
 * (ascending i) and landing with ONE add per destination element.  row_of(id) = id (uni == NULL) or the id's index in the ascending list
 * uni[0..U); ids outside (0, V) or not in the list are skipped.  src bf16 (src_bf16 = 1) or fp32 with row pitch lds; dst fp32 with row
 * pitch ldd; H % 4 == 0, H <= 4096, n <= 8192.  No sort, no host read. */
int mmbert_id_runs_sum_rows(mmbert_stream_t stream, const void* src, int src_bf16, int lds, const int64_t* ids, int n,
                            int H, int V, const int64_t* uni, int U, float* dst, int ldd);

/* ---- optimizer: flat AdamW (REF:train.py:76-97; mode 0 = transformers-2.8 AdamW, 1 = torch.optim.AdamW) ----
 * flags[i/256]: 0 no decay, 1 decay, 2 frozen; + 4 = "the next backward overwrites this block's gradient": zero_grad leaves it alone.
 * n % 256 == 0.  Also refreshes the bf16 copy, and zeroes g.  The hyper-parameters are DOUBLES, as the Python floats of the reference's
 * optimizer are: 1 - beta, the bias corrections 1 - beta^step and the step size are formed in double and rounded to fp32 once, the way
 * `exp_avg_sq.mul_(beta2).addcmul_(1 - beta2, grad, grad)` hands a double scalar to an fp32 tensor (round 5: formed in fp32, 1 - 0.999f
 * was 1.3e-5 off -- found by the hand-computed vector tests/golden/hf_adamw_hand.py). */
int mmbert_adamw(mmbert_stream_t stream, float* p, float* g, float* m, float* v, void* p_bf16, const uint8_t* flags, size_t n,
                 double lr, double beta1, double beta2, double eps, double wd, int step, double gscale, int mode, int zero_grad);

/* du = dy * gelu_erf'(u), contiguous bf16 (BertPredictionHeadTransform backward, HF:476-480) */
int mmbert_gelu_bwd(mmbert_stream_t stream, const void* dy, const void* u, void* du, size_t n);

int mmbert_cast_f32_bf16(mmbert_stream_t stream, const float* x, void* y, size_t n);
int mmbert_cast_bf16_f32(mmbert_stream_t stream, const void* x, float* y, size_t n);
/* descs: device array of {int64 src_off, int64 dst_off, int rows, cols, dst_ld, tile0} (64x64 tiles) */
int mmbert_transpose_cast(mmbert_stream_t stream, const float* src, void* dst, const void* descs, int ndesc, int total_tiles);
/* Row list of a sparse backward in one launch: out[i] = map[i < n ? rows[i] : extra[i - n]] written as int64 (out64) and int32 (out32);
 * rows come as int32 (rows32) or int64 (rows64, used when rows32 is NULL); map (int64, e.g. the valid-first packing's inverse) may be NULL.
 * Replaces the cast / cat / index_select / cast chain in front of the top encoder layer's sparse backward (REF: autograd of
 * MMBertForPretraining.py:406-445 -- only the MLM-labelled rows and the [CLS] rows of the last layer's output carry a gradient). */
int mmbert_compact_rows(mmbert_stream_t stream, const int* rows32, const int64_t* rows64, int n, const int64_t* extra, int nextra, const int64_t* map,
                        int64_t* out64, int* out32);
/* The same, and the list's INVERSE for mmbert_scatter_rows_zero: inv_stamp[out[i]] = (int64) stamp << 32 | i.  inv_stamp is a persistent
 * int64 array over all rows of the matrices the list indexes, zero-filled ONCE by its owner and never cleared: an entry counts only while
 * its upper half equals the stamp of the current call (stamp != 0, a new one per call). */
int mmbert_compact_rows_inv(mmbert_stream_t stream, const int* rows32, const int64_t* rows64, int n, const int64_t* extra, int nextra, const int64_t* map,
                            int64_t* out64, int* out32, int64_t* inv_stamp, unsigned stamp);
/* Back to full height: dst_k[r] = src_k[i] where inv_stamp[r] = stamp << 32 | i with i < nlist, zero otherwise, for r < nrows and up to 4
 * matrices k (rows of row_bytes[k] bytes; pointers, pitches and row_bytes multiples of 16).  One launch for the zero fill + index_copy_
 * pairs that hand the sparse top layer's gradients (attention output, pre-LayerNorm sum) to the dense kernels below it. */
int mmbert_scatter_rows_zero(mmbert_stream_t stream, int nseg, const void* const* src, void* const* dst, const long long* src_pitch,
                             const long long* dst_pitch, const int* row_bytes, const int64_t* inv_stamp, unsigned stamp, int nlist, int nrows);
/* Batched row gather: dst_k[i] = src_k[idx[i]] for i < nrows and up to 12 matrices k that share the row list idx (int32, device); rows
 * are row_bytes[k] bytes (a multiple of 4) at byte pitches src_pitch[k] / dst_pitch[k].  The sparse backward paths use it to pull
 * the labelled rows out of every saved activation in one launch. */
int mmbert_gather_rows(mmbert_stream_t stream, int nseg, const void* const* src, void* const* dst, const long long* src_pitch, const long long* dst_pitch,
                       const int* row_bytes, const int* idx, int nrows);
/* the same from a bf16 source with the fp32 source's element offsets (the working copy mmbert_adamw has just written) */
int mmbert_transpose_bf16(mmbert_stream_t stream, const void* src, void* dst, const void* descs, int ndesc, int total_tiles);

#ifdef __cplusplus
}
#endif
#endif /* MMBERT_HIP_H */
