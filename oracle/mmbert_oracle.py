"""CPU oracle for the MMBert train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32, autograd) restatement of the arithmetic that
kimkyeonghun/MSA's ``MMBertForPretraining.forward`` + ``trainer.train_epoch`` trigger, written
functionally over a ``dict`` of tensors keyed by the reference's state-dict names.  It imports
neither ``transformers`` nor anything from the reference tree, so it travels to the GPU box.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
-- as the checker / timed CPU baseline only.  Nothing under ``msa_amd/`` imports it; the product
path fails loudly when the HIP library is missing.

Parity pinning: the reference ships no tests and no golden vectors.  The oracle is
pinned against outputs of the reference itself, imported in the build container with the shim of
SURVEY.md Appendix A; the generating script is ``tests/golden/make_golden.py`` and the vectors are
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks every one of them).  The HF-AdamW
update (``transformers==2.8.0 optimization.AdamW``, the class ``train.py:10,92`` imports) is absent
from the installed transformers 5.15.0, so that single function is restated from its published
algorithm and is *parity unpinned* against live reference code.

Citations: ``REF:`` = /root/reference/<file>:<line>; ``HF:`` = transformers 5.15.0
``models/bert/modeling_bert.py`` (third-party, not vendored by the reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

LN_EPS_BERT = 1e-12      # HF BertConfig.layer_norm_eps default (HF:64)
LN_EPS_JOINT = 1e-5      # nn.LayerNorm default, REF:MMBertEmbedding.py:54
MASK_NEG = -10000.0      # REF:MMBertForPretraining.py:153

# modality feature dims, REF:config.py:13-17
MODALITY_DIMS = {"mosi": (47, 74), "mosei": (35, 74), "ur_funny": (371, 81)}


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
_EMULATE_BF16 = False      # tolerance calibrator only: see bf16_storage_emulation()


def _r(x):
    """bf16 storage emulation with a straight-through gradient (identity unless the calibrator is on)."""
    if not _EMULATE_BF16:
        return x
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


class bf16_storage_emulation:
    """Context manager: round activations and GEMM weights to bf16 at the points where the HIP path
    stores bf16 (fp32 accumulation everywhere).  NOT part of the parity pin -- the tests use it to
    measure how far bf16 storage alone moves each loss / gradient away from the fp32 oracle, and
    bound the HIP path's deviation by a small multiple of that (ill-conditioned head gradients)."""

    def __enter__(self):
        global _EMULATE_BF16
        self.prev, _EMULATE_BF16 = _EMULATE_BF16, True

    def __exit__(self, *a):
        global _EMULATE_BF16
        _EMULATE_BF16 = self.prev


def _linear(x, p: Params, name: str, big: bool = False):
    """``big`` marks the projections the HIP path runs as bf16 GEMMs (weights rounded in emulation)."""
    w = _r(p[name + ".weight"]) if big else p[name + ".weight"]
    return F.linear(x, w, p[name + ".bias"])


def _layer_norm(x, p: Params, name: str, eps: float):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def _dropout(x, prob: float, train: bool, masks: Optional[dict], key: str):
    """Inverted dropout.  ``masks[key]`` (a 0/1 keep tensor broadcastable to x) makes the oracle
    replay a mask produced elsewhere (the HIP kernels' counter RNG) -- scale is 1/(1-prob)."""
    if masks is not None and key in masks:
        return x * masks[key].to(x.dtype) / (1.0 - prob)
    if not train or prob == 0.0:
        return x
    return F.dropout(x, prob, True)


# ----------------------------------------------------------------------------------------------
# a4: additive key mask
# ----------------------------------------------------------------------------------------------
def extended_attention_mask(attention_mask: torch.Tensor, joint: bool, dtype=torch.float32):
    """REF:MMBertForPretraining.py:57-154 (is_decoder=False branches only).

    joint & 3-D: feature 0 of the [B,P,D] mask (narrow(...,2,0,1), :76); non-joint 3-D: mean over
    features (:111); 2-D: as is.  Result [B,1,1,S] = (1-m)*-10000 in the model dtype (:152-153)."""
    if attention_mask.dim() == 3:
        if joint:
            m = torch.narrow(attention_mask, 2, 0, 1).squeeze(-1)
        else:
            m = attention_mask.mean(2)
    elif attention_mask.dim() == 2:
        m = attention_mask
    else:
        raise ValueError("attention_mask must be 2-D or 3-D")
    ext = m[:, None, None, :].to(dtype)
    return (1.0 - ext) * MASK_NEG


# ----------------------------------------------------------------------------------------------
# a7: BertEmbeddings, a6: JointEmbeddings
# ----------------------------------------------------------------------------------------------
def bert_embeddings(p: Params, input_ids, token_type_ids, *, hidden_dropout=0.1, train=False,
                    masks=None, tag=""):
    """HF:53-108 via REF:MMBertForPretraining.py:264-265: word+type+pos -> LN(1e-12) -> dropout."""
    T = input_ids.shape[1]
    # padding_idx=pad_token_id=0 (HF:58): the lookup sends NO gradient to row 0; the tied decoder still does
    e = F.embedding(input_ids.long(), p["bert.embeddings.word_embeddings.weight"], padding_idx=0)
    e = e + p["bert.embeddings.token_type_embeddings.weight"][token_type_ids.long()]
    e = e + p["bert.embeddings.position_embeddings.weight"][:T][None]
    e = _layer_norm(_r(e), p, "bert.embeddings.LayerNorm", LN_EPS_BERT)
    return _r(_dropout(e, hidden_dropout, train, masks, tag + "emb"))


def joint_embeddings(p: Params, input_embs, pair_ids, dims: Tuple[int, int], *, joint_dropout=0.5,
                     train=False, masks=None, tag=""):
    """REF:MMBertEmbedding.py:57-72.  Dispatch on the feature dim (:61-66), relu(Linear(.float())),
    concat after the text rows (:68), LayerNorm eps 1e-5 (:54,69), Dropout(0.5) (:70)."""
    vdim, sdim = dims
    d = pair_ids.shape[-1]
    if d == vdim:
        pe = F.relu(_linear(pair_ids.float(), p, "bert.jointEmbeddings.Wv"))
    elif d == sdim:
        pe = F.relu(_linear(pair_ids.float(), p, "bert.jointEmbeddings.Ws"))
    else:
        raise Exception("Wrong Dimension")                      # REF:MMBertEmbedding.py:66
    x = _r(torch.cat((input_embs, pe), dim=1))
    x = _layer_norm(x, p, "bert.jointEmbeddings.LayerNorm", LN_EPS_JOINT)
    return _r(_dropout(x, joint_dropout, train, masks, tag + "joint"))


# ----------------------------------------------------------------------------------------------
# a8: encoder stack
# ----------------------------------------------------------------------------------------------
def encoder_layer(p: Params, i: int, x, ext_mask, n_heads: int, *, hidden_dropout=0.1,
                  attn_dropout=0.1, train=False, masks=None, tag=""):
    """One HF BertLayer (HF:374-416): self-attention (HF:164-203, eager HF:111-136),
    BertSelfOutput (HF:289-293), BertIntermediate gelu-erf (HF:334-337), BertOutput (HF:347-351)."""
    pre = f"bert.encoder.layer.{i}."
    B, S, H = x.shape
    dh = H // n_heads
    q = _r(_linear(x, p, pre + "attention.self.query", True)).view(B, S, n_heads, dh).transpose(1, 2)
    k = _r(_linear(x, p, pre + "attention.self.key", True)).view(B, S, n_heads, dh).transpose(1, 2)
    v = _r(_linear(x, p, pre + "attention.self.value", True)).view(B, S, n_heads, dh).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3)) * (dh ** -0.5)
    if ext_mask is not None:
        w = w + ext_mask
    w = F.softmax(w, dim=-1)
    w = _dropout(w, attn_dropout, train, masks, f"{tag}l{i}.attn")
    ctx = _r(torch.matmul(_r(w), v).transpose(1, 2).reshape(B, S, H))
    a = _linear(ctx, p, pre + "attention.output.dense", True)
    a = _dropout(a, hidden_dropout, train, masks, f"{tag}l{i}.h1")
    y = _r(_layer_norm(_r(a + x), p, pre + "attention.output.LayerNorm", LN_EPS_BERT))
    u = _r(F.gelu(_r(_linear(y, p, pre + "intermediate.dense", True))))
    o = _linear(u, p, pre + "output.dense", True)
    o = _dropout(o, hidden_dropout, train, masks, f"{tag}l{i}.h2")
    return _r(_layer_norm(_r(o + y), p, pre + "output.LayerNorm", LN_EPS_BERT))


def encoder(p: Params, x, ext_mask, n_layers: int, n_heads: int, *, collect=None, **kw):
    for i in range(n_layers):
        x = encoder_layer(p, i, x, ext_mask, n_heads, **kw)
        if collect is not None:
            collect.append(x)
    return x


# ----------------------------------------------------------------------------------------------
# a3/a9: MMBertModel.forward
# ----------------------------------------------------------------------------------------------
def mmbert_model(p: Params, cfg: dict, input_ids, attention_mask, token_type_ids, joint: bool, *,
                 train=False, masks=None, tag="", collect=None):
    """REF:MMBertForPretraining.py:216-285.  joint: input_ids=(text_ids, pair_feats),
    attention_mask=(text_mask, pair_mask); token types forced to zero (:223); masks concatenated
    on the key axis (:248-250)."""
    hd, ad = cfg.get("hidden_dropout", 0.1), cfg.get("attn_dropout", 0.1)
    if joint:
        text_ids, pair = input_ids
        tmask, pmask = attention_mask
        tt = torch.zeros_like(text_ids, dtype=torch.long)
        ext = torch.cat((extended_attention_mask(tmask, True), extended_attention_mask(pmask, True)), dim=-1)
    else:
        text_ids, tmask, tt = input_ids, attention_mask, token_type_ids
        if tt is None:
            tt = torch.zeros_like(text_ids, dtype=torch.long)
        ext = extended_attention_mask(tmask, False)
    x = bert_embeddings(p, text_ids, tt, hidden_dropout=hd, train=train, masks=masks, tag=tag)
    inter = collect if isinstance(collect, dict) else None      # dict form: {tag + "emb" / "jemb" / "hidden": ...} of every pass
    if inter is not None:
        inter[tag + "emb"] = x
    if joint:
        x = joint_embeddings(p, x, pair, MODALITY_DIMS[cfg["dataset"]],
                             joint_dropout=cfg.get("joint_dropout", 0.5), train=train, masks=masks, tag=tag)
        if inter is not None:
            inter[tag + "jemb"] = x
    hidden = inter.setdefault(tag + "hidden", []) if inter is not None else collect
    seq = encoder(p, x, ext, cfg["layers"], cfg["heads"], hidden_dropout=hd, attn_dropout=ad,
                  train=train, masks=masks, tag=tag, collect=hidden)
    pooled = torch.tanh(_linear(seq[:, 0], p, "bert.pooler.dense"))          # HF:457-463
    return seq, pooled


# ----------------------------------------------------------------------------------------------
# a10: heads, a12: CPC
# ----------------------------------------------------------------------------------------------
def mlm_scores(p: Params, seq):
    """HF:466-496: decoder(LN(gelu(dense(seq)))); decoder.weight tied to the word embeddings
    (HF:728-731), decoder.bias = cls.predictions.bias."""
    t = _r(F.gelu(_r(_linear(seq, p, "cls.predictions.transform.dense", True))))
    t = _r(_layer_norm(t, p, "cls.predictions.transform.LayerNorm", LN_EPS_BERT))
    return _r(F.linear(t, _r(p["bert.embeddings.word_embeddings.weight"]), p["cls.predictions.bias"]))


def pretraining_heads(p: Params, seq, pooled, joint: bool):
    """REF:MMBertForPretraining.py:292-302."""
    scores = mlm_scores(p, seq)
    if joint:
        return scores, _linear(seq[:, 0], p, "cls.align")
    return scores, _linear(pooled, p, "cls.seq_relationship")


def cpc(p: Params, name: str, x, y):
    """REF:MMBertEmbedding.py:21-32 (InfoNCE over the in-batch negatives)."""
    x_pred = _linear(y, p, name + ".net")
    x_pred = x_pred / x_pred.norm(dim=1, keepdim=True)
    x = x / x.norm(dim=1, keepdim=True)
    pos = torch.sum(x * x_pred, dim=-1)
    neg = torch.logsumexp(torch.matmul(x, x_pred.t()), dim=-1)
    return -(pos - neg).mean()


# ----------------------------------------------------------------------------------------------
# a1/a2/a11/a13: MMBertForPretraining.forward
# ----------------------------------------------------------------------------------------------
def _one_pass(p, cfg, input_ids, attention_mask, token_type_ids, labels, ap_label, joint, **kw):
    """REF:MMBertForPretraining.py:353-390 (get_bert_output + get_outputs)."""
    seq, pooled = mmbert_model(p, cfg, input_ids, attention_mask, token_type_ids, joint, **kw)
    scores, rel = pretraining_heads(p, seq, pooled, joint)
    V = scores.shape[-1]
    mlm = F.cross_entropy(scores.view(-1, V), labels.view(-1).long()) if labels is not None else 0
    ap = F.cross_entropy(rel.view(-1, 2), ap_label.view(-1).long()) if ap_label is not None else 0
    return mlm, ap, pooled, scores, rel


def pretraining_forward(p: Params, cfg: dict, input_ids, token_type_ids, attention_mask, masked_labels,
                        ap_label, sentiment, *, train=False, masks=None, collect=None):
    """REF:MMBertForPretraining.py:392-449.  Returns (outputs_tuple_of_13, logits) exactly as the
    reference does (text/visual/speech_loss slots are None, :394,445)."""
    text_ids, visual, speech, twv, tws = input_ids
    tt_t, tt_v, tt_s = token_type_ids
    am_t, am_v, am_s = attention_mask
    lab_t, lab_v, lab_s = masked_labels
    ap_v, ap_s = ap_label
    kw = dict(train=train, masks=masks)
    t_mlm, _, pt, t_sc, t_rel = _one_pass(p, cfg, text_ids, am_t, tt_t, lab_t, None, False, tag="t.", collect=collect, **kw)
    cj = collect if isinstance(collect, dict) else None       # (a list collects the text pass only; a dict every pass)
    v_mlm, v_ap, pv, v_sc, v_rel = _one_pass(p, cfg, (twv, visual), am_v, tt_v, lab_v, ap_v, True, tag="v.", collect=cj, **kw)
    s_mlm, s_ap, ps, s_sc, s_rel = _one_pass(p, cfg, (tws, speech), am_s, tt_s, lab_s, ap_s, True, tag="s.", collect=cj, **kw)

    ap, label, nce, logits = fusion_objective(p, cfg, pt, pv, ps, v_ap, s_ap, sentiment)
    mlm = (t_mlm + v_mlm + s_mlm) / 3.0                      # :427
    joint_loss = cfg.get("alpha", 1.0) * mlm + ap + label - cfg.get("beta", 1.0) * nce   # :443
    outputs = (joint_loss, None, None, None, ap, label, nce, t_sc, t_rel, v_sc, v_rel, s_sc, s_rel)
    return outputs, logits


def fusion_objective(p: Params, cfg: dict, pt, pv, ps, v_ap, s_ap, sentiment):
    """REF:MMBertForPretraining.py:406-436 -- everything downstream of the three pooled vectors and the two alignment losses:
    gates, gated concatenation, classifier, the three CPC terms, ap / label losses.  Returns (ap, label, nce, logits)."""
    def gate(x, vname):                                      # :407-409
        a = F.relu(_linear(torch.cat((x, x), dim=1), p, "attn"))
        return _linear(a, p, vname)
    pooled = torch.cat((pt * gate(pt, "vt"), pv * gate(pv, "vv"), ps * gate(ps, "vs")), dim=1)
    temp = _linear(pooled, p, "classifier1_1")               # :414
    logits = _linear(temp, p, "classifier1_2")               # :415
    nce = cpc(p, "cpc_zt", pt, temp) + cpc(p, "cpc_zv", pv, temp) + cpc(p, "cpc_za", ps, temp)
    ap = (v_ap + s_ap) / 2.0                                 # :428
    num_labels = cfg.get("num_labels", 7)
    if num_labels == 1:
        logits = torch.tanh(logits)                          # :434-435
    label = F.mse_loss(logits.view(-1), sentiment.view(-1).float())      # :433-436
    return ap, label, nce, logits


def heads_from_cls(p: Params, cfg: dict, first, ap_v, ap_s, sentiment):
    """The objective's part that hangs off the [CLS] rows, given those rows: ``first`` [3B, H] = the encoder outputs at position 0
    of the text / visual / speech passes (in that order).  pooler (HF:457-463), ``align`` on the joint passes' rows
    (REF:MMBertForPretraining.py:297-298), their 2-way CE (:385-388), then fusion_objective -- the same functions
    pretraining_forward runs (pinned by the golden full-forward fixtures).  Returns (ap + label - beta * nce, ap, label, nce,
    logits): what the HIP path's fused heads (csrc/heads.hip) are checked against in fp32, free of the encoder's bf16 noise."""
    B = first.shape[0] // 3
    pooled = torch.tanh(_linear(first, p, "bert.pooler.dense"))
    v_rel, s_rel = _linear(first[B:2 * B], p, "cls.align"), _linear(first[2 * B:], p, "cls.align")
    v_ap = F.cross_entropy(v_rel.view(-1, 2), ap_v.view(-1).long())
    s_ap = F.cross_entropy(s_rel.view(-1, 2), ap_s.view(-1).long())
    ap, label, nce, logits = fusion_objective(p, cfg, pooled[:B], pooled[B:2 * B], pooled[2 * B:], v_ap, s_ap, sentiment)
    return ap + label - cfg.get("beta", 1.0) * nce, ap, label, nce, logits


def fused_forward(p: Params, cfg: dict, input_ids, token_type_ids, attention_mask, masked_labels, ap_label, sentiment, *,
                  train=False, masks=None):
    """NOT IN THE REFERENCE -- the declared fused-sequence extension (SURVEY S8(d) ``fused1050``), restated here only so that
    the HIP path's ``forward_fused`` has a CPU checker: text | visual | speech in one sequence (JointEmbeddings' ``cat`` with
    both projected modalities, REF:MMBertEmbedding.py:61-70), one encoder pass, and the objective of
    REF:MMBertForPretraining.py:392-449 with that pass in all three modality slots.  Parity for this function is pinned by
    nothing in the reference (there is nothing to pin it to); it reuses the pinned building blocks above."""
    text_ids, visual, speech = input_ids
    am_t, am_v, am_s = attention_mask
    ap_v, ap_s = ap_label
    hd, ad = cfg.get("hidden_dropout", 0.1), cfg.get("attn_dropout", 0.1)
    tt = token_type_ids if token_type_ids is not None else torch.zeros_like(text_ids, dtype=torch.long)
    x = bert_embeddings(p, text_ids, tt, hidden_dropout=hd, train=train, masks=masks, tag="f.")
    pv = F.relu(_linear(visual.float(), p, "bert.jointEmbeddings.Wv"))
    ps = F.relu(_linear(speech.float(), p, "bert.jointEmbeddings.Ws"))
    x = _r(torch.cat((x, pv, ps), dim=1))
    x = _layer_norm(x, p, "bert.jointEmbeddings.LayerNorm", LN_EPS_JOINT)
    x = _r(_dropout(x, cfg.get("joint_dropout", 0.5), train, masks, "f.joint"))
    ext = torch.cat((extended_attention_mask(am_t, True), extended_attention_mask(am_v, True), extended_attention_mask(am_s, True)), dim=-1)
    seq = encoder(p, x, ext, cfg["layers"], cfg["heads"], hidden_dropout=hd, attn_dropout=ad, train=train, masks=masks, tag="f.")
    pooled1 = torch.tanh(_linear(seq[:, 0], p, "bert.pooler.dense"))
    scores = mlm_scores(p, seq)
    rel = _linear(seq[:, 0], p, "cls.align")
    V = scores.shape[-1]
    mlm = F.cross_entropy(scores.view(-1, V), masked_labels.view(-1).long())
    ap = (F.cross_entropy(rel.view(-1, 2), ap_v.view(-1).long()) + F.cross_entropy(rel.view(-1, 2), ap_s.view(-1).long())) / 2.0

    def gate(xx, vname):
        a = F.relu(_linear(torch.cat((xx, xx), dim=1), p, "attn"))
        return _linear(a, p, vname)
    pooled = torch.cat((pooled1 * gate(pooled1, "vt"), pooled1 * gate(pooled1, "vv"), pooled1 * gate(pooled1, "vs")), dim=1)
    temp = _linear(pooled, p, "classifier1_1")
    logits = _linear(temp, p, "classifier1_2")
    nce = cpc(p, "cpc_zt", pooled1, temp) + cpc(p, "cpc_zv", pooled1, temp) + cpc(p, "cpc_za", pooled1, temp)
    if cfg.get("num_labels", 7) == 1:
        logits = torch.tanh(logits)
    label = F.mse_loss(logits.view(-1), sentiment.view(-1).float())
    joint_loss = cfg.get("alpha", 1.0) * mlm + ap + label - cfg.get("beta", 1.0) * nce
    return (joint_loss, None, None, None, ap, label, nce, scores, rel), logits


# ----------------------------------------------------------------------------------------------
# parameter construction (shapes + init of the reference flow, used by tests and cpu_baseline)
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """Every state-dict tensor of the reference model, CPC x_size = H
    (declared generalisation of REF:MMBertForPretraining.py:327-344)."""
    H, L, V, I = cfg["hidden"], cfg["layers"], cfg["vocab"], cfg["intermediate"]
    vd, sd = MODALITY_DIMS[cfg["dataset"]]
    s: Dict[str, Tuple[int, ...]] = {}
    s["bert.embeddings.word_embeddings.weight"] = (V, H)
    s["bert.embeddings.position_embeddings.weight"] = (cfg.get("max_pos", 512), H)
    s["bert.embeddings.token_type_embeddings.weight"] = (2, H)
    s["bert.embeddings.LayerNorm.weight"] = (H,)
    s["bert.embeddings.LayerNorm.bias"] = (H,)
    for i in range(L):
        pre = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[pre + f"attention.self.{n}.weight"] = (H, H)
            s[pre + f"attention.self.{n}.bias"] = (H,)
        s[pre + "attention.output.dense.weight"] = (H, H)
        s[pre + "attention.output.dense.bias"] = (H,)
        s[pre + "attention.output.LayerNorm.weight"] = (H,)
        s[pre + "attention.output.LayerNorm.bias"] = (H,)
        s[pre + "intermediate.dense.weight"] = (I, H)
        s[pre + "intermediate.dense.bias"] = (I,)
        s[pre + "output.dense.weight"] = (H, I)
        s[pre + "output.dense.bias"] = (H,)
        s[pre + "output.LayerNorm.weight"] = (H,)
        s[pre + "output.LayerNorm.bias"] = (H,)
    s["bert.pooler.dense.weight"] = (H, H)
    s["bert.pooler.dense.bias"] = (H,)
    s["bert.jointEmbeddings.W_cv.weight"] = (H, vd + H)
    s["bert.jointEmbeddings.W_cv.bias"] = (H,)
    s["bert.jointEmbeddings.W_cs.weight"] = (H, sd + H)
    s["bert.jointEmbeddings.W_cs.bias"] = (H,)
    s["bert.jointEmbeddings.Wv.weight"] = (H, vd)
    s["bert.jointEmbeddings.Wv.bias"] = (H,)
    s["bert.jointEmbeddings.Ws.weight"] = (H, sd)
    s["bert.jointEmbeddings.Ws.bias"] = (H,)
    s["bert.jointEmbeddings.LayerNorm.weight"] = (H,)
    s["bert.jointEmbeddings.LayerNorm.bias"] = (H,)
    s["cls.predictions.bias"] = (V,)
    s["cls.predictions.transform.dense.weight"] = (H, H)
    s["cls.predictions.transform.dense.bias"] = (H,)
    s["cls.predictions.transform.LayerNorm.weight"] = (H,)
    s["cls.predictions.transform.LayerNorm.bias"] = (H,)
    s["cls.seq_relationship.weight"] = (2, H)
    s["cls.seq_relationship.bias"] = (2,)
    s["cls.align.weight"] = (2, H)
    s["cls.align.bias"] = (2,)
    s["classifier1_1.weight"] = (H, 3 * H)
    s["classifier1_1.bias"] = (H,)
    s["classifier1_2.weight"] = (1, H)
    s["classifier1_2.bias"] = (1,)
    s["attn.weight"] = (H, 2 * H)
    s["attn.bias"] = (H,)
    for n in ("vt", "vs", "vv"):
        s[n + ".weight"] = (1, H)
        s[n + ".bias"] = (1,)
    for n in ("cpc_zt", "cpc_zv", "cpc_za"):
        s[n + ".net.weight"] = (H, H)
        s[n + ".net.bias"] = (H,)
    return s


def seeded_params(cfg: dict, seed: int = 0) -> Params:
    """Deterministic weights that do NOT depend on torch's RNG stream layout: numpy PCG64, one
    stream per tensor name.  N(0,0.02) matrices, zero biases, LN=(1,0) like HF's _init_weights;
    the post-init modules (jointEmbeddings, CPC nets) get U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
    The same function feeds the golden generator (which load_state_dict()s it into the real
    reference model), the oracle tests and the HIP parity tests, so no weights are committed."""
    import numpy as np
    import zlib
    out: Params = {}
    for name, shape in param_shapes(cfg).items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        if "LayerNorm.weight" in name:
            a = np.ones(shape, np.float32)
            # perturb so LN gamma/beta parity is actually exercised
            a += 0.1 * rng.standard_normal(shape).astype(np.float32)
        elif "LayerNorm.bias" in name:
            a = 0.1 * rng.standard_normal(shape).astype(np.float32)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shape).astype(np.float32)
        elif "jointEmbeddings" in name or name.startswith("cpc_"):
            bound = 1.0 / math.sqrt(shape[-1])
            a = rng.uniform(-bound, bound, shape).astype(np.float32)
        else:
            a = (0.02 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = torch.from_numpy(a)
    out["bert.embeddings.word_embeddings.weight"][0].zero_()          # padding_idx=0 (HF:58)
    return out


def count_params(cfg: dict) -> int:
    n = 0
    for shape in param_shapes(cfg).values():
        k = 1
        for d in shape:
            k *= d
        n += k
    return n


# ----------------------------------------------------------------------------------------------
# a14/a15: optimizer + schedule + the train_epoch stepping rule
# ----------------------------------------------------------------------------------------------
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")                 # REF:train.py:78


def decays(name: str) -> bool:
    """REF:train.py:79-91: weight decay 0.01 unless the NAME contains one of NO_DECAY."""
    return not any(nd in name for nd in NO_DECAY)


def linear_schedule_lambda(step: int, warmup: int, total: float) -> float:
    """transformers get_linear_schedule_with_warmup (published algorithm); the reference passes
    warmup=N, total=warmup_proportion*N (REF:train.py:93-97)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_step(p, g, m, v, step: int, lr: float, wd: float, *, beta1=0.9, beta2=0.999, eps=1e-6,
               mode="hf"):
    """One in-place AdamW update on CPU tensors.

    mode "hf": transformers==2.8.0 ``optimization.AdamW.step`` (published algorithm; parity
    unpinned, see header): m,v EMA; denom = sqrt(v)+eps; step_size = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= step_size*m/denom; then p -= lr*wd*p (decay AFTER the update, on the updated p).
    mode "torch": ``torch.optim.AdamW`` (decay first; denom = sqrt(v)/sqrt(1-b2^t)+eps) -- the
    optimizer the golden fixture G8 is generated with."""
    if mode == "torch":
        p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if mode == "hf":
        denom = v.sqrt().add_(eps)
        p.addcdiv_(m, denom, value=-(lr * math.sqrt(bc2) / bc1))
        if wd > 0.0:
            p.add_(p, alpha=-lr * wd)
    else:
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def should_step(step_idx: int, gas: int, quirk: bool = True) -> bool:
    """REF:trainer.py:96 ``(step + 1)& args.gradient_accumulation_step == 0`` -- a bitwise AND, so at
    gas=1 the optimizer steps on every second micro-batch.  quirk=False gives the intended ``%``."""
    if quirk:
        return ((step_idx + 1) & gas) == 0
    return ((step_idx + 1) % gas) == 0


# ----------------------------------------------------------------------------------------------
# a16: MLM masking rule (input contract)
# ----------------------------------------------------------------------------------------------
def mask_tokens_rule(inputs: torch.Tensor, select: torch.Tensor, replace: torch.Tensor,
                     special_ids: Sequence[int] = (101, 102), mask_id: int = 103):
    """REF:model_utils.py:6-39 with the two Bernoulli draws passed in (``select`` ~ B(p=0.15),
    ``replace`` ~ B(0.8)): special tokens are never selected (:17-23), labels = -100 where not
    selected (:28), 80% of selected -> [MASK] (:30-32); the 10% random branch is commented out.
    Special = [CLS], [SEP]: what the pinned transformers-2.8 ``get_special_tokens_mask`` flags; the PAD branch (:24-26) is a
    non-in-place ``masked_fill`` whose result is dropped, so [PAD] stays selectable (pinned by tests/golden/mask_tokens.npz)."""
    special = torch.zeros_like(inputs, dtype=torch.bool)
    for s in special_ids:
        special |= inputs == s
    masked = select.bool() & ~special
    labels = inputs.clone()
    labels[~masked] = -100
    out = inputs.clone()
    out[replace.bool() & masked] = mask_id
    return out, labels
