// Composite entry points (round 5): ONE C call issues the launches of an encoder layer's forward (7) or of the dense part of its backward
// (7) -- the same kernels through the same entry points, in the same order and with the same arguments as msa_amd/model.py's per-launch
// path (results are bit-identical: tests/test_model_gpu.py::test_composite_layer_calls_are_bit_identical).  What they save is host time:
// a ctypes call marshals ~25 arguments in ~10 us, and a layer makes 14 of them; the composite path fills two structures once per layer.
// HF:374-416 (BertLayer), HF:175-177,289-293,334-351 (the dense layers with their dropout / residual / LayerNorm placements).
#include <hip/hip_runtime.h>
#include "../../include/mmbert_hip.h"

extern "C" {

int mmbert_layer_struct_sizes(int* out) {
    if (!out) return -1;
    out[0] = (int)sizeof(mmbert_attn_layout); out[1] = (int)sizeof(mmbert_layer_fwd_args); out[2] = (int)sizeof(mmbert_layer_bwd_args);
    return 0;
}

#define MMB_TRY(call) do { const int e_ = (call); if (e_) return e_; } while (0)

int mmbert_layer_fwd(mmbert_stream_t s, const mmbert_attn_layout* L, const mmbert_layer_fwd_args* a) {
    if (!L || !a || a->rows <= 0) return a && a->rows == 0 ? 0 : -1;
    const int M = a->rows, H = a->H, I = a->I;
    // q | k | v = x . Wqkv^T + bqkv                                                            (HF:175-177)
    MMB_TRY(mmbert_gemm_nt(s, a->x, a->ldx, a->Wqkv, H, a->qkv, 3 * H, M, 3 * H, H, MMBERT_EPI_BIAS, a->bqkv, nullptr, 0, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    MMB_TRY(mmbert_attn_fwd(s, a->qkv, a->actx, a->lse, L->key_bias, L->bias_start, H, L->heads, L->seq_start, L->seq_len, L->elem_base,
                            L->ftile_seq, L->ftile_r0, L->nftiles, a->att.stream, a->att.thr16, a->att.scale, L->kv_len, L->ftile_qshift, L->ftile_qend));
    // z1 = dropout(ctx . Wo^T + bo) + x ; y1 = LayerNorm(z1)                                   (HF:289-293)
    MMB_TRY(mmbert_gemm_nt(s, a->actx, H, a->Wo, H, a->z1, H, M, H, H, MMBERT_EPI_BIAS | MMBERT_EPI_RESID, a->bo, a->x, a->ldx, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, a->h1.stream, a->h1.thr16, a->h1.scale, a->tile_queue));
    MMB_TRY(mmbert_ln_fwd(s, a->z1, H, nullptr, a->y1, H, nullptr, M, H, a->ln1_g, a->ln1_b, a->ln_eps, a->m1, a->r1, 0, 0, 1.0f, 0));
    // g = gelu(y1 . W1^T + b1), u = the pre-activation                                         (HF:334-337)
    MMB_TRY(mmbert_gemm_nt(s, a->y1, H, a->W1, H, a->g, I, M, I, H, MMBERT_EPI_BIAS | MMBERT_EPI_GELU, a->b1, nullptr, 0, a->u, a->u ? I : 0, nullptr, 0,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    // z2 = dropout(g . W2^T + b2) + y1 ; y2 = LayerNorm(z2) (stored through y2_rows when given)  (HF:347-351)
    MMB_TRY(mmbert_gemm_nt(s, a->g, I, a->W2, I, a->z2, H, M, H, I, MMBERT_EPI_BIAS | MMBERT_EPI_RESID, a->b2, a->y1, H, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, a->h2.stream, a->h2.thr16, a->h2.scale, a->tile_queue));
    MMB_TRY(mmbert_ln_fwd(s, a->z2, H, nullptr, a->y2, a->ldy2, a->y2_rows, M, H, a->ln2_g, a->ln2_b, a->ln_eps, a->m2, a->r2, 0, 0, 1.0f, 0));
    return 0;
}

int mmbert_layer_bwd(mmbert_stream_t s, const mmbert_attn_layout* L, const mmbert_layer_bwd_args* a) {
    if (!L || !a || a->rows <= 0) return a && a->rows == 0 ? 0 : -1;
    const int M = a->rows, H = a->H, I = a->I;
    // output sublayer: dz2 = LayerNorm'(dy) (+ dz2d = its dropped copy, the FFN-down input gradient's operand)
    MMB_TRY(mmbert_ln_bwd(s, a->dy, a->lddy, a->dy_rows, a->z2, H, nullptr, a->m2, a->r2, a->ln2_g, M, H, a->dz2, H, nullptr, a->dz2d, a->dz2d ? H : 0,
                          a->g_ln2_g, a->g_ln2_b, nullptr, 0, 0, 1.0f, a->h2.stream, a->dz2d ? a->h2.thr16 : 0, a->h2.scale, a->ln2_ws, nullptr, 1, 0));
    const void* dz2d = a->dz2d ? a->dz2d : a->dz2;
    MMB_TRY(mmbert_gemm_nt(s, dz2d, H, a->W2T, H, a->du, I, M, I, H, MMBERT_EPI_GELU_BWD, nullptr, nullptr, 0, nullptr, 0, a->u, I,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    MMB_TRY(mmbert_gemm_nt(s, a->du, I, a->W1T, I, a->dy1, H, M, H, I, MMBERT_EPI_RESID, nullptr, a->dz2, H, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    // attention sublayer
    MMB_TRY(mmbert_ln_bwd(s, a->dy1, H, nullptr, a->z1, H, nullptr, a->m1, a->r1, a->ln1_g, M, H, a->dz1, H, nullptr, a->dz1d, a->dz1d ? H : 0,
                          a->g_ln1_g, a->g_ln1_b, nullptr, 0, 0, 1.0f, a->h1.stream, a->dz1d ? a->h1.thr16 : 0, a->h1.scale, a->ln1_ws, nullptr, 1, 0));
    const void* dz1d = a->dz1d ? a->dz1d : a->dz1;
    MMB_TRY(mmbert_gemm_nt(s, dz1d, H, a->WoT, H, a->dctx, H, M, H, H, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    MMB_TRY(mmbert_attn_bwd(s, a->qkv, a->actx, a->dctx, a->dqkv, a->lse, a->delta, L->key_bias, L->bias_start, H, L->heads, L->seq_start, L->seq_len,
                            L->elem_base, L->qtile_seq, L->qtile_r0, L->nqtiles, L->tile_seq, L->tile_r0, L->ntiles,
                            a->att.stream, a->att.thr16, a->att.scale, L->kv_len, L->qtile_qshift, L->qtile_qend, L->split, nullptr));
    MMB_TRY(mmbert_gemm_nt(s, a->dqkv, 3 * H, a->WqkvT, 3 * H, a->dx, H, M, H, 3 * H, MMBERT_EPI_RESID, nullptr, a->dz1, H, nullptr, 0, nullptr, 0,
                           1.0f, nullptr, 0, 0, 1.0f, a->tile_queue));
    return 0;
}

}  // extern "C"
