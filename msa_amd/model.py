"""Drop-in MI355X implementation of the reference's model API:

    MMBertForPretraining(config) / .from_pretrained(path) / .bert.set_joint_embeddings(dataset) /
    .set_alpha_beta(alpha, beta) / .forward(input_ids, token_type_ids, attention_mask,
    masked_labels, ap_label, sentiment) -> (outputs_tuple_of_13, logits)      REF:MMBertForPretraining.py:304-449
    MMBertModel.forward(..., joint) -> (sequence_output, pooled_output)       REF:MMBertForPretraining.py:13-285
    MMBertPreTrainingHeads.forward(sequence_output, pooled_output, joint)     REF:MMBertForPretraining.py:287-302
    JointEmbeddings(hidden_size, dropout_prob, dataset).forward(embs, pair)   REF:MMBertEmbedding.py:34-72
    CPC(x_size, y_size, n_layers, activation).forward(x, y)                   REF:MMBertEmbedding.py:7-32

Parameter names / shapes / state-dict keys are the reference's.  The arithmetic is NOT torch's:
the three encoder passes (text, text+visual, text+speech) are packed into one variable-length
token matrix and run through hand-written gfx950 kernels (msa_amd/csrc) via the C ABI in
include/mmbert_hip.h -- bf16 MFMA GEMMs with fused epilogues, flash-style attention, fused
LayerNorm/dropout/residual, vocabulary cross-entropy -- with fp32 master weights, fp32 gradient
accumulation and fp32 statistics.  The <0.01 %-of-FLOPs heads on the [B,H] pooled vectors (pooler,
gates, CPC, 2-way CE, MSE) are torch-ROCm glue in fp32.  There is no CPU path: a missing
libmmbert_hip.so or a CPU tensor raises.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .flat import FlatParams

MODALITY_DIMS = {"mosi": (47, 74), "mosei": (35, 74), "ur_funny": (371, 81)}      # REF:config.py:13-17
LN_EPS_JOINT = 1e-5                                                                # nn.LayerNorm default, REF:MMBertEmbedding.py:54
MASK_NEG = -10000.0                                                                # REF:MMBertForPretraining.py:153


class MMBertConfig:
    """The BertConfig fields the hot path reads (any object with these attributes works, including a
    HuggingFace BertConfig)."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12,
                 initializer_range=0.02, pad_token_id=0, hidden_act="gelu", is_decoder=False, **unused):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.layer_norm_eps, self.initializer_range = layer_norm_eps, initializer_range
        self.pad_token_id, self.hidden_act, self.is_decoder = pad_token_id, hidden_act, is_decoder
        self.output_attentions = False
        self.output_hidden_states = False


# ================================================================================================
# parameter-holder modules (names = the reference's state-dict keys)
# ================================================================================================
class _Box(nn.Module):
    pass


def _make_bert_holder(cfg) -> nn.Module:
    H, I = cfg.hidden_size, cfg.intermediate_size
    bert = _Box()
    emb = _Box()
    emb.word_embeddings = nn.Embedding(cfg.vocab_size, H, padding_idx=cfg.pad_token_id)
    emb.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H)
    emb.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
    emb.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
    bert.embeddings = emb
    enc = _Box()
    layers = []
    for _ in range(cfg.num_hidden_layers):
        lay = _Box()
        att = _Box()
        att.self = _Box()
        att.self.query, att.self.key, att.self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        att.output = _Box()
        att.output.dense = nn.Linear(H, H)
        att.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        lay.attention = att
        lay.intermediate = _Box()
        lay.intermediate.dense = nn.Linear(H, I)
        lay.output = _Box()
        lay.output.dense = nn.Linear(I, H)
        lay.output.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        layers.append(lay)
    enc.layer = nn.ModuleList(layers)
    bert.encoder = enc
    bert.pooler = _Box()
    bert.pooler.dense = nn.Linear(H, H)
    return bert


def _hf_init(module: nn.Module, std: float, skip: Optional[nn.Module] = None):
    """HF ``_init_weights``: N(0, std) Linear/Embedding weights, zero biases, LayerNorm (1, 0),
    zero padding row -- what ``init_weights()`` does at REF:MMBertForPretraining.py:22,347.  ``skip``: a submodule whose
    (already initialised or loaded) weights are left alone."""
    skipped = set(id(x) for x in skip.modules()) if skip is not None else ()
    for m in module.modules():
        if id(m) in skipped:
            continue
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=std)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class CPC(nn.Module):
    """Contrastive predictive coding head (REF:MMBertEmbedding.py:7-32): holds the one projection ``net`` (state-dict keys
    ``cpc_z*.net.{weight,bias}``, y_size -> x_size) and evaluates InfoNCE with in-batch negatives.  In the model the three CPC terms
    run batched inside the heads kernels (``_HeadsFn`` / ``_heads``); this ``forward`` serves callers that use the module alone:
    with S the [B,B] matrix of cosine similarities between x and net(y), the loss is mean_i(logsumexp_j S_ij - S_ii)."""

    def __init__(self, x_size, y_size, n_layers=1, activation="Tanh"):
        super().__init__()
        if n_layers != 1:
            raise NotImplementedError("the reference builds a projection for n_layers == 1 only (REF:MMBertEmbedding.py:15-19)")
        self.x_size, self.y_size, self.layers = x_size, y_size, n_layers
        self.activation = getattr(nn, activation)              # attribute kept for checkpoints / introspection; never applied
        self.net = nn.Linear(y_size, x_size)

    def forward(self, x, y):
        unit = lambda t: t / torch.linalg.vector_norm(t, dim=1, keepdim=True)
        sim = unit(x) @ unit(self.net(y)).t()
        return (torch.logsumexp(sim, dim=1) - torch.diagonal(sim)).mean()


class JointEmbeddings(nn.Module):
    """REF:MMBertEmbedding.py:34-72.  ``TEXTDIM`` is the hidden size (the reference hard-codes 1024,
    config.py:12).  W_cv / W_cs exist (state-dict compatibility) and are unused."""

    def __init__(self, hidden_size, dropout_prob, dataset):
        super().__init__()
        if dataset not in MODALITY_DIMS:
            raise KeyError(dataset)
        self.VISUALDIM, self.SPEECHDIM = MODALITY_DIMS[dataset]
        H = hidden_size
        self.W_cv = nn.Linear(self.VISUALDIM + H, H)
        self.W_cs = nn.Linear(self.SPEECHDIM + H, H)
        self.Wv = nn.Linear(self.VISUALDIM, H)
        self.Ws = nn.Linear(self.SPEECHDIM, H)
        self.LayerNorm = nn.LayerNorm(H)
        self.dropout_prob = dropout_prob
        self.dropout = nn.Dropout(dropout_prob)
        self._owner = None

    def which(self, pair_ids) -> str:
        d = pair_ids.size()[-1]
        if d == self.VISUALDIM:
            return "Wv"
        if d == self.SPEECHDIM:
            return "Ws"
        raise Exception("Wrong Dimension")                                 # REF:MMBertEmbedding.py:66

    def forward(self, input_embs, pair_ids):
        assert input_embs is not None, "You miss input_embs"
        assert pair_ids is not None, "You miss pair_ids"
        if self._owner is None:
            raise RuntimeError("JointEmbeddings must be attached with MMBertModel.set_joint_embeddings()")
        top = self._owner()
        top._ensure_ready(input_embs.device)
        B, T, H = input_embs.shape
        e1 = input_embs.reshape(B * T, H).to(torch.bfloat16)
        drop = ops.make_drop(self.dropout_prob if self.training else 0.0, top._next_seed(), 1001)
        out = _JointFn.apply(e1, self.LayerNorm.weight, top, (_pair_features(pair_ids, e1.device),), (self.which(pair_ids),), B, T, drop)
        return out.view(B, -1, H).float()


# ================================================================================================
# autograd functions over the C-ABI kernels.  Parameter gradients are accumulated straight into the
# flat fp32 gradient buffer (p.grad views); the ``anchor`` parameter only makes autograd call backward.
# ================================================================================================
class _JointFn(torch.autograd.Function):
    """cat(text_emb, relu(W.pair+b)) -> LayerNorm(1e-5) -> dropout(0.5)      (REF:MMBertEmbedding.py:57-72)
    ``feats`` / ``whichs`` are tuples: one modality in the reference's joint passes; both (text | visual | speech in ONE
    sequence) in the fused-sequence extension (forward_fused)."""

    @staticmethod
    def forward(ctx, e1, anchor, top, feats, whichs, B, T, drop):
        w = top._w
        H = e1.shape[1]
        S = T + sum(f.shape[1] for f in feats)
        j0 = torch.empty((B * S, H), device=e1.device, dtype=torch.bfloat16)
        j0.view(B, S, H)[:, :T].copy_(e1.view(B, T, H))
        off = T
        for feat, which in zip(feats, whichs):
            ops.pair_proj_fwd(feat, w[which + "_w"], w[which + "_b"], j0, T, seq_len=S, offset=off)
            off += feat.shape[1]
        x, mean, rstd = ops.ln_fwd(j0, w["joint_ln_g"], w["joint_ln_b"], LN_EPS_JOINT, drop=drop)
        ctx.top, ctx.whichs, ctx.B, ctx.T, ctx.drop, ctx.nf = top, whichs, B, T, drop, len(feats)
        ctx.save_for_backward(j0, mean, rstd, *feats)              # (feats: the caller's tensors; autograd's version check covers them)
        return x

    @staticmethod
    def backward(ctx, dx):
        j0, mean, rstd, *feats = ctx.saved_tensors
        w, B, T = ctx.top._w, ctx.B, ctx.T
        H = j0.shape[1]
        S = j0.shape[0] // B
        dj0 = ops.ln_bwd(dx.contiguous(), j0, mean, rstd, w["joint_ln_g"], w["g_joint_ln_g"], w["g_joint_ln_b"], post_drop=ctx.drop)
        off = T
        for feat, which in zip(feats, ctx.whichs):
            ops.pair_proj_bwd(feat, j0, dj0, T, w["g_" + which + "_w"], w["g_" + which + "_b"], seq_len=S, offset=off)
            off += feat.shape[1]
        de1 = dj0.view(B, -1, H)[:, :T].reshape(B * T, H)
        return de1, None, None, None, None, None, None, None


def _pair_features(f, dev):
    """The pair features as the projection kernels take them: contiguous fp32 or float64 on the device.  The reference's collate hands over
    float64 (REF:model_utils.py:94-99) and JointEmbeddings casts with ``.float()`` (REF:MMBertEmbedding.py:62,64); the kernels read float64
    directly and round on load (same values, no cast launch, no fp32 copy).  The tensor is the CALLER's: it is read again by backward
    (weight gradient of the projection), so an in-place change between forward and backward raises there, as a saved tensor would."""
    f = f.to(dev)
    if f.dtype not in (torch.float32, torch.float64):
        f = f.float()
    return f.contiguous()


def _check_unchanged(feats, versions):
    for f, v in zip(feats, versions):
        if f._version != v:
            raise RuntimeError("a pair-feature tensor needed for gradient computation has been modified by an inplace operation "
                               "between forward and backward (the projection's weight gradient reads the caller's tensor)")


_cu_count = {}
_defer_fits = {}
_DEFER_ENV = os.environ.get("MMBERT_DEFER_WGRADS")          # "0" / "1": whole-process A/B override, read ONCE at import (model.defer_wgrads overrides per model)


def _auto_defer_wgrads(H, I, L, x_dev, rows=None) -> bool:
    """Whether the encoder's weight gradients go out as ONE call of whole 256-tile rounds at the end of backward instead of one launch per
    layer pair (only without a gradient hook: one GPU): yes as soon as the one call has at least a full round of tiles.  Measured:
    bert-large, the reference's default -- a pair is 384 tiles = 1.5 rounds (0.75 full), 23 layers 17.25 rounds (0.96) -- -2.2 % of the step
    (19.55 -> 19.12 ms, profiles/r4_ab_refdef_wgrads.log); 12-layer d = 768 -- a pair is 216 tiles on 256 CUs (0.84), eleven layers 4.64
    rounds (0.93) -- -0.6 / -0.8 % with 40-step windows, -0.7 % with alternating 600-step processes (profiles/r4_ab_tn8_defer.log).
    The price is memory: the eight operands of every dense layer -- rows x (2 I + 8 H) bf16, 0.34 GB per layer at the headline shape, 3.7 GB
    for eleven layers -- stay alive until the end of backward.  ``rows`` (the backward's row count) given: the deferral is taken only
    while that fits into HALF of what the device still has (the driver's free memory + torch's cached-but-free blocks; one query per
    (shape, depth, 2048-row class), cached for 256 queries) -- a configuration that fits with the paired launches must not run out of
    memory because of a 0.7 % scheduling gain.  In deterministic mode the rule is the shape alone.  MMBERT_DEFER_WGRADS=0/1 (read at import) and model.defer_wgrads = True / False override the rule."""
    key = str(x_dev)
    cus = _cu_count.get(key)
    if cus is None:
        cus = _cu_count[key] = int(torch.cuda.get_device_properties(x_dev).multi_processor_count)
    c = lambda n: (n + 255) // 256
    t = 2 * c(I) * c(H) + c(3 * H) * c(H) + c(H) * c(H)            # 256 x 256 tiles of a layer's four weight gradients
    if not (L > 2 and (L - 1) * t >= cus):
        return False
    if rows is None or ops.deterministic():
        # deterministic mode: the two forms differ in fp32 summation order (the paired form splits layer 0's token axis), so the choice must
        # not depend on how much memory happens to be free at the first query -- by shape only (model.defer_wgrads overrides)
        return True
    fk = (key, H, I, L, -(-int(rows) // 2048))                       # (row counts vary from batch to batch: 2048-row classes)
    ent = _defer_fits.get(fk)
    if ent is None or ent[1] <= 0:                                    # (the answer is re-taken every 256 queries: memory fills up and empties over a run)
        need = L * (fk[4] * 2048) * (2 * I + 8 * H) * 2
        free, _total = torch.cuda.mem_get_info(x_dev)
        cached = torch.cuda.memory_reserved(x_dev) - torch.cuda.memory_allocated(x_dev)
        ent = _defer_fits[fk] = [need <= 0.5 * (free + cached), 256]
    ent[1] -= 1
    return ent[0]


def _wgrad(top, probs):
    """One weight-gradient launch for ``probs`` = [(dY, X, gW (view of the flat gradient buffer), gb)]: it OVERWRITES gradients the
    optimizer has dropped (lazy zero, flat.FlatParams.take_accumulate) and accumulates otherwise."""
    f = top._flat
    if not f.stale:
        ops.gemm_tn_grouped(probs, accumulate=True)
    else:
        # per problem: a gradient outside the lazy set (the MLM transform's: zeroed by the optimizer) in one call with dropped ones must not
        # make take_accumulate settle those with zero fills
        ops.gemm_tn_grouped(probs, accumulate=[f.take_accumulate([q[2]]) for q in probs])


def _late_wgrad(top, prob, defer: bool):
    """A weight gradient of the FEW rows that carry a loss (tied decoder, MLM transform, the top layer's row-sparse sublayers).  ``defer``:
    it waits in ``top._late_wgrads`` for the end of the trunk's backward, where _EncoderFn.run_backward hands it to the deferred
    multi-layer call -- its tiles then run on the CUs the call's last round leaves idle (mmbert_gemm_tn_grouped_rows) instead of as a launch
    of 9 - 360 short tiles on the serial tail of backward.  Otherwise (a gradient hook wants final gradients early, or no trunk backward
    follows): launched here."""
    if defer:
        pend = top.__dict__.setdefault("_late_wgrads", [])
        if any(q[2] is prob[2] for q in pend):                  # a second backward stage writing the same gradient: never two writers in one launch
            _flush_late_wgrads(top)
            pend = top.__dict__.setdefault("_late_wgrads", [])
        pend.append(prob)
    else:
        _wgrad(top, [prob])


def _flush_late_wgrads(top, long_probs=(), deferred=False):
    """The deferred dense problems (``long_probs``, all of the backward's row count) and whatever waits in ``top._late_wgrads``: as ONE
    call per 12 layers when ``deferred`` (the few-row problems behind the last one's), else the few-row problems one launch each."""
    late = top.__dict__.pop("_late_wgrads", None) or []
    long_probs = list(long_probs)
    if not deferred or not long_probs:
        for c in range(0, len(long_probs), 48):
            _wgrad(top, long_probs[c:c + 48])
        for q in late:
            _wgrad(top, [q])
        return
    chunks = [long_probs[c:c + 48] for c in range(0, len(long_probs), 48)]    # (12 layers per call: whole rounds, as measured in round 4)
    if late and len(chunks[-1]) + len(late) <= ops.TN_MAX_PROBLEMS:
        chunks[-1] = chunks[-1] + late
        late = []
    side = getattr(top, "wgrad_side_stream", True) and long_probs[0][0].is_cuda
    if side:
        # on the model's side stream, behind everything queued so far: nothing reads these gradients before the optimizer, and what the
        # caller (_TrunkFn.backward) queues next -- the embedding stage: LayerNorm', the pair projections' weight gradients, the reduce
        # launches; 110 us of launches that leave most of the chip empty -- runs beside the call's 1.8 ms instead of behind them.  The caller
        # joins (_join_wgrads) before anything else writes the tied table's gradient; the operands stay referenced until then (their
        # blocks belong to the current stream's pool)
        _join_wgrads(top)
        s = ops.side_stream("wgrads", long_probs[0][0].device)
        s.wait_stream(torch.cuda.current_stream())
        # (Measured: the small launches cannot share a CU with a weight-gradient workgroup -- 128 KB of LDS, half the registers -- and get CUs
        # where a round ends; cutting the call behind its last whole round and starting them beside the partly filled last round instead
        # was 0.4 % SLOWER than this, profiles/r6_ab_side_streams.log.)
        with torch.cuda.stream(s):
            for c in chunks:
                _wgrad(top, c)
            for q in late:
                _wgrad(top, [q])
            top.__dict__["_wgrad_join"] = (s.record_event(), chunks, late)
        return
    for c in chunks:
        _wgrad(top, c)
    for q in late:
        _wgrad(top, [q])


def _join_heads(top):
    """The current stream waits for the heads' backward levels queued on the side stream (_HeadsStepFn.backward; no-op otherwise)."""
    j = top.__dict__.pop("_heads_join", None)
    if j:
        torch.cuda.current_stream().wait_event(j[-1][0])


def _join_wgrads(top):
    """The current stream waits for the deferred weight-gradient call queued on the side stream (no-op otherwise); its operands are
    released behind the wait."""
    j = top.__dict__.pop("_wgrad_join", None)
    if j is not None:
        torch.cuda.current_stream().wait_event(j[0])


class _EncoderFn:
    """L x BertLayer (HF:374-416) over the packed token matrix: the forward / backward bodies that _TrunkFn runs between the
    embedding stage and the heads (plain functions: the whole trunk is ONE autograd node, so no [tokens, H] tensor crosses
    autograd between its stages)."""

    @staticmethod
    def run_forward(top, x, layout, key_bias, seed, kv_len, keep, y_out=None, y_rows=None):
        """Returns (y, saved).  ``y_out`` / ``y_rows``: the LAST layer's LayerNorm writes row i of its output to y_out[y_rows[i]] --
        the un-packing of the valid-first layout (ops.SplitLayout.perm32) rides on that store instead of a [tokens, H] gather."""
        cfg = top.config
        H, L = cfg.hidden_size, cfg.num_hidden_layers
        train = top.training
        ph = cfg.hidden_dropout_prob if train else 0.0
        pa = cfg.attention_probs_dropout_prob if train else 0.0
        saved = []
        # One C call per layer (mmbert_layer_fwd: the same seven launches with the same arguments, bit-identical) unless somebody wrapped
        # the per-launch functions (bench.py's event timing, tests that spy on launches) or switched it off (model.composite_layers)
        composite = getattr(top, "composite_layers", True) and ops.launches_unwrapped() and top.debug_hidden is None
        if composite:
            AL = ops.attn_layout_struct(key_bias, layout, backward=False)
            if kv_len is not None and not getattr(layout, "split", False):
                AL.kv_len = kv_len.data_ptr()
            I_, M_, dev = cfg.intermediate_size, x.shape[0], x.device
            heads = layout.heads
            bf, f32 = torch.bfloat16, torch.float32
            tq = ops._tile_queue(dev)
            for i in range(L):
                lw = top._lw[i]
                d_att, d_h1, d_h2 = (ops.make_drop(pa, seed, 8 * i), ops.make_drop(ph, seed, 8 * i + 1), ops.make_drop(ph, seed, 8 * i + 2))
                last = i == L - 1 and y_out is not None
                qkv, actx, z1, y1 = (torch.empty((M_, 3 * H), device=dev, dtype=bf), torch.empty((M_, H), device=dev, dtype=bf),
                                     torch.empty((M_, H), device=dev, dtype=bf), torch.empty((M_, H), device=dev, dtype=bf))
                g, z2 = torch.empty((M_, I_), device=dev, dtype=bf), torch.empty((M_, H), device=dev, dtype=bf)
                lse = torch.empty((M_, heads), device=dev, dtype=f32)
                u = torch.empty((M_, I_), device=dev, dtype=bf) if keep else None
                if keep:
                    m1, r1, m2, r2 = (torch.empty(M_, device=dev, dtype=f32) for _ in range(4))
                else:
                    m1 = r1 = m2 = r2 = None
                y2 = y_out if last else torch.empty((M_, H), device=dev, dtype=bf)
                a = lw.get("_fwd_args")
                if a is None:                                        # the layer's constants, once
                    a = lw["_fwd_args"] = ops._LayerFwd()
                    for n in ("Wqkv", "Wo", "W1", "W2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b"):
                        assert lw[n].is_contiguous()
                        setattr(a, n, lw[n].data_ptr())
                    a.H, a.I, a.ln_eps = H, I_, cfg.layer_norm_eps
                a.x, a.ldx, a.rows = x.data_ptr(), x.stride(0), M_
                a.qkv, a.actx, a.lse, a.z1, a.y1, a.g, a.z2 = qkv.data_ptr(), actx.data_ptr(), lse.data_ptr(), z1.data_ptr(), y1.data_ptr(), g.data_ptr(), z2.data_ptr()
                a.u, a.m1, a.r1, a.m2, a.r2 = ops._ptr(u), ops._ptr(m1), ops._ptr(r1), ops._ptr(m2), ops._ptr(r2)
                a.y2, a.ldy2, a.y2_rows = y2.data_ptr(), y2.stride(0), ops._ptr(y_rows) if last else None
                a.tile_queue = tq
                ops._set_drop(a.att, d_att); ops._set_drop(a.h1, d_h1); ops._set_drop(a.h2, d_h2)
                ops.layer_fwd(AL, a)
                if keep:
                    saved.append((x, qkv, actx, lse, z1, m1, r1, y1, u, g, z2, m2, r2, d_att, d_h1, d_h2))
                x = y2
            return x, saved
        for i in range(L):
            lw = top._lw[i]
            d_att, d_h1, d_h2 = (ops.make_drop(pa, seed, 8 * i), ops.make_drop(ph, seed, 8 * i + 1), ops.make_drop(ph, seed, 8 * i + 2))
            qkv = ops.gemm_nt(x, lw["Wqkv"], bias=lw["bqkv"])
            actx, lse = ops.attn_fwd(qkv, key_bias, layout, H, drop=d_att, kv_len=kv_len)
            z1 = ops.gemm_nt(actx, lw["Wo"], bias=lw["bo"], resid=x, drop=d_h1)
            y1, m1, r1 = ops.ln_fwd(z1, lw["ln1_g"], lw["ln1_b"], cfg.layer_norm_eps, stats=keep)
            u = torch.empty((x.shape[0], cfg.intermediate_size), device=x.device, dtype=torch.bfloat16) if keep else None
            g = ops.gemm_nt(y1, lw["W1"], bias=lw["b1"], gelu=True, aux=u)
            z2 = ops.gemm_nt(g, lw["W2"], bias=lw["b2"], resid=y1, drop=d_h2)
            last = i == L - 1 and y_out is not None
            y2, m2, r2 = ops.ln_fwd(z2, lw["ln2_g"], lw["ln2_b"], cfg.layer_norm_eps, stats=keep,
                                    out=y_out if last else None, out_rows=y_rows if last else None)
            if keep:
                saved.append((x, qkv, actx, lse, z1, m1, r1, y1, u, g, z2, m2, r2, d_att, d_h1, d_h2))
            x = y2
            if top.debug_hidden is not None:
                top.debug_hidden.setdefault("_layers_packed", []).append(None if last else y2.detach())
        return x, saved

    @staticmethod
    def run_backward(top, layout, key_bias, kv_len, saved, dy, dy_rows, top_rows, compact=None, lnd=None):
        """``dy``: gradient of the encoder output -- in the encoder's own row order, or (``dy_rows`` = the int32 row list
        ops.SplitLayout.perm32[:rows_a]) in the caller's order, read through the map by the top layer's LayerNorm'.  ``compact`` =
        (rows in the caller's order int64, their gradients [n, H] bf16): the only rows of the output that carry a gradient, handed
        over by the MLM head (no dense [tokens, H] gradient exists then; ``dy`` is ignored).  Returns dx for the leading rows_a rows."""
        H = top.config.hidden_size
        L = top.config.num_hidden_layers
        top._flat.grads_dirty = True
        top._flat.attach_lazy()
        top._flat.wait_transposes()                               # (the optimizer's transposed-copy launch runs on a side stream)
        _join_wgrads(top)                                         # (left by a backward that raised)
        _join_heads(top)
        # split (valid-first) layout: the rows behind rows_a have exactly-zero gradients in every layer (see _encode), so the
        # whole backward -- dgrads, weight gradients, LayerNorm', attention -- runs on the leading rows_a rows only
        M_all = saved[0][0].shape[0]
        ra = layout.rows_a if getattr(layout, "split", False) else M_all
        if dy is not None and dy_rows is None and ra < dy.shape[0]:
            dy = dy[:ra]
        # LayerNorm' leaves its gamma / beta partial sums in a workspace; ONE reduce launch folds all layers' sums into the gradients at
        # the end (with a data-parallel hook: per layer, before the layer's gradient slice is handed to the all-reduce); same-process
        # A/B against round 1's form (bias sums inside LayerNorm', one reduce launch per call): 16.58 vs 16.76 ms per step
        # (lnd handed in by the trunk's backward: ITS collector, which already holds the MLM head's call and goes on through the
        # embedding stage -- one reduce launch for all of them; it flushes at its end)
        own_lnd = lnd is None
        if own_lnd:
            lnd = ops.LnDeferred(2 * L)
        pair_wgrads, held = getattr(top, "pair_wgrads", True), None
        # (model.defer_wgrads: None = by shape (_auto_defer_wgrads), True / False forced; see the comment at its use)
        dw = getattr(top, "defer_wgrads", None)
        if dw is None and _DEFER_ENV:
            dw = _DEFER_ENV != "0"
        if dw is None:
            dw = _auto_defer_wgrads(H, top.config.intermediate_size, L, x_dev=saved[0][0].device, rows=ra)
        defer_wgrads, deferred = (top.grad_hook is None and bool(dw)), []
        composite_bwd = getattr(top, "composite_layers", True) and ops.launches_unwrapped()
        AL = tq = None
        for i in reversed(range(L)):
            lw = top._lw[i]
            saved_i = saved[i]
            saved[i] = None
            if ra < M_all:
                saved_i = tuple(t[:ra] if torch.is_tensor(t) else t for t in saved_i)
            x, qkv, actx, lse, z1, m1, r1, y1, u, g, z2, m2, r2, d_att, d_h1, d_h2 = saved_i
            if i == L - 1 and (top_rows is not None or compact is not None):
                R = None
                R32 = None
                if compact is not None:
                    rows_l, rows_f, dy_c = compact                # labelled rows (int32 / int64), [CLS] rows (int64), their gradients stacked
                    # (with the list's stamped inverse: the two gradients that go back to full height below do so in one launch)
                    rinv = top.__dict__.get("_row_inverse")
                    if rinv is None or rinv.table.numel() < M_all or rinv.table.device != rows_f.device:
                        rinv = top.__dict__["_row_inverse"] = ops.RowInverse(M_all, rows_f.device)
                    R, R32 = ops.compact_rows(rows_l, rows_f, layout.inv if getattr(layout, "split", False) else None, inverse=rinv)
                elif top_rows is not None:
                    Ro = _EncoderFn._sparse_rows(top_rows, None, ra)          # rows in the caller's order
                    if Ro is not None:
                        R = layout.inv.index_select(0, Ro) if getattr(layout, "split", False) else Ro
                        src_rows = Ro if dy_rows is not None else R            # dy is in the caller's order iff it comes with a map
                        (dy_c,) = ops.gather_rows([dy], src_rows.int())
                if R is not None:
                    dy = _EncoderFn._last_layer_sparse(top, lw, layout, key_bias, kv_len, saved_i, dy_c, R, H, ra, R32, lnd,
                                                       rinv if compact is not None else None, deferred if defer_wgrads else None)
                    if top.grad_hook is not None:
                        lnd.flush()
                    dy_rows = None
                    top._layer_grads_done(i)
                    continue
                if compact is not None:
                    raise RuntimeError("compact output gradient without the sparse top-layer path")
            if composite_bwd:
                # the seven launches of the dense backward body from ONE C call (mmbert_layer_bwd: same kernels, same arguments, bit-identical)
                dev, bf = z2.device, torch.bfloat16
                I_ = top.config.intermediate_size
                if AL is None:
                    AL = ops.attn_layout_struct(key_bias, layout, backward=True)
                    if kv_len is not None and not getattr(layout, "split", False):
                        AL.kv_len = kv_len.data_ptr()
                    tq = ops._tile_queue(dev)
                dz2, dz1, dy1, dctx, dx_ = (torch.empty((ra, H), device=dev, dtype=bf) for _ in range(5))
                dz2d = torch.empty((ra, H), device=dev, dtype=bf) if d_h2[1] else None
                dz1d = torch.empty((ra, H), device=dev, dtype=bf) if d_h1[1] else None
                du, dqkv = torch.empty((ra, I_), device=dev, dtype=bf), torch.empty((ra, 3 * H), device=dev, dtype=bf)
                delta = torch.empty((ra, layout.heads), device=dev, dtype=torch.float32)
                lnd.reserve(2, ra, H, dev)                           # (both slots are taken before their launches: no flush in between)
                ws2 = lnd.slot(ra, H, dev, lw["g_ln2_g"], lw["g_ln2_b"], None)
                ws1 = lnd.slot(ra, H, dev, lw["g_ln1_g"], lw["g_ln1_b"], None)
                b = lw.get("_bwd_args")
                if b is None:
                    b = lw["_bwd_args"] = ops._LayerBwd()
                    for n in ("W2T", "W1T", "WoT", "WqkvT", "ln2_g", "ln1_g", "g_ln2_g", "g_ln2_b", "g_ln1_g", "g_ln1_b"):
                        assert lw[n].is_contiguous()
                        setattr(b, n, lw[n].data_ptr())
                    b.H, b.I = H, I_
                b.dy, b.lddy, b.dy_rows, b.rows = dy.data_ptr(), dy.stride(0), ops._ptr(dy_rows), ra
                b.z2, b.m2, b.r2, b.ln2_ws, b.z1, b.m1, b.r1, b.ln1_ws = z2.data_ptr(), m2.data_ptr(), r2.data_ptr(), ws2, z1.data_ptr(), m1.data_ptr(), r1.data_ptr(), ws1
                b.u, b.qkv, b.actx, b.lse = u.data_ptr(), qkv.data_ptr(), actx.data_ptr(), lse.data_ptr()
                b.dz2, b.dz2d, b.du, b.dy1, b.dz1, b.dz1d = dz2.data_ptr(), ops._ptr(dz2d), du.data_ptr(), dy1.data_ptr(), dz1.data_ptr(), ops._ptr(dz1d)
                b.dctx, b.dqkv, b.delta, b.dx, b.tile_queue = dctx.data_ptr(), dqkv.data_ptr(), delta.data_ptr(), dx_.data_ptr(), tq
                ops._set_drop(b.att, d_att); ops._set_drop(b.h1, d_h1); ops._set_drop(b.h2, d_h2)
                ops.layer_bwd(AL, b)
                dy_rows = None
                dy = dx_
                if dz2d is None:
                    dz2d = dz2
                if dz1d is None:
                    dz1d = dz1
            else:
              # --- output sublayer: y2 = LN(dropout(g.W2^T + b2) + y1)      (b2 / bo gradients = column sums of dz2d / dz1d: they ride
              # on the weight-gradient launch below, like b1 and bqkv, on an all-ones MFMA operand)
              dz2d = torch.empty((ra, H), device=z2.device, dtype=torch.bfloat16) if d_h2[1] else None
              dz2 = ops.ln_bwd(dy, z2, m2, r2, lw["ln2_g"], lw["g_ln2_g"], lw["g_ln2_b"], dx2=dz2d, pre_drop=d_h2, deferred=lnd, dy_rows=dy_rows)
              dy_rows = None
              if dz2d is None:
                  dz2d = dz2
              du = ops.gemm_nt(dz2d, lw["W2T"], gelu_bwd_u=u)
              dy1 = ops.gemm_nt(du, lw["W1T"], resid=dz2)
              # --- attention sublayer: y1 = LN(dropout(ctx.Wo^T + bo) + x)
              dz1d = torch.empty((ra, H), device=z2.device, dtype=torch.bfloat16) if d_h1[1] else None
              dz1 = ops.ln_bwd(dy1, z1, m1, r1, lw["ln1_g"], lw["g_ln1_g"], lw["g_ln1_b"], dx2=dz1d, pre_drop=d_h1, deferred=lnd)
              if dz1d is None:
                  dz1d = dz1
              dctx = ops.gemm_nt(dz1d, lw["WoT"])
              dqkv = ops.attn_bwd(qkv, actx, dctx, lse, key_bias, layout, H, drop=d_att, kv_len=kv_len)
              dy = ops.gemm_nt(dqkv, lw["WqkvT"], resid=dz1)
            # --- all four weight gradients (+ their bias gradients) of the layer in ONE launch.  (They are off the critical path of backward;
            # a side stream for them was measured a loss in rounds 1, 3 and 4 -- the chip's power envelope is shared -- and is gone.)
            probs = [(du, y1, lw["g_W1"], lw["g_b1"]), (dz2d, g, lw["g_W2"], lw["g_b2"]),
                     (dqkv, x, lw["g_Wqkv"], lw["g_bqkv"]), (dz1d, actx, lw["g_Wo"], lw["g_bo"])]
            if defer_wgrads:
                # One GPU (no gradient hook): nothing needs this layer's weight gradients before the optimizer, so ALL dense layers'
                # weight gradients can go out as ONE call at the end of backward, in whole rounds of 256 tiles (11 layers = 1188
                # tiles = 4.64 rounds; the paired launches below fill 216 of 256 CUs each and layer 0, alone, splits the token axis
                # into slabs + a reduce).  The operands just stay alive until then (~190 MB per layer).  Round 4, same-process A/B of
                # the train step: 14.010 ms against 13.989 for the paired form (profiles/r4_ab_deferred_wgrads.log) -- filling the 40
                # idle CUs buys nothing: with 256 instead of 216 CUs multiplying, every tile takes proportionally longer (the chip
                # holds its clock down under this load: DESIGN 3.1), as the balanced-atomics and side-stream forms had hinted.  Where a
                # pair's launch fills its rounds badly (bert-large: 384 tiles = 1.5 rounds) the deferred call wins: _auto_defer_wgrads.
                deferred.extend(probs)
                top._layer_grads_done(i)
                continue
            if pair_wgrads:
                # Two layers per launch: a layer's 108 tiles leave the chip half empty, so a single layer splits the token axis in two
                # (fp32 slabs + a reduce launch, 12 us and 85 MB per layer); two layers' 216 tiles fill it in one round unsplit.  The
                # upper layer of a pair just keeps its operands alive for one more layer (fresh buffers from the caching allocator)
                if held is None and i > 0:
                    held = (i, probs)
                    continue
                if held is not None:
                    _wgrad(top, held[1] + probs)
                    if top.grad_hook is not None:
                        lnd.flush()
                    top._layer_grads_done(held[0])
                    held = None
                else:
                    _wgrad(top, probs)
            else:
                _wgrad(top, probs)
            if top.grad_hook is not None:
                lnd.flush()                                # ... and after its LayerNorm sums
            top._layer_grads_done(i)
        if own_lnd:
            lnd.flush()
        # (mmbert_gemm_tn_grouped_rows: 12 layers per call; round 6: the few-row weight gradients of the MLM head and of the sparse top layer
        # ride behind the last call's tiles)
        _flush_late_wgrads(top, deferred, deferred=defer_wgrads)
        return dy

    @staticmethod
    def _sparse_rows(top_rows, layout, ra):
        """The rows of the LAST layer's output that have a gradient: the MLM-labelled rows and the [CLS] rows (everything else
        feeds nothing but the returned scores).  (device int64 list in the encoder's packed order) or None if not worth it."""
        (idx, host, ev), first = top_rows
        n = _active_row_count(top_rows[0])                        # (long complete: the MLM head's backward ran before this)
        r = n + first.numel()
        if 4 * r > ra or int(host[1]) != 0:                       # a labelled [CLS] row would be gathered twice: dense backward
            return None
        R = torch.cat((idx[:n].long(), first))
        if layout is not None and getattr(layout, "split", False):
            R = layout.inv.index_select(0, R)
        return R

    @staticmethod
    def _last_layer_sparse(top, lw, layout, key_bias, kv_len, saved_i, dy_c, R, H, ra, R32=None, lnd=None, rinv=None, deferred=None):
        """Backward of the top encoder layer when only the rows R of its output carry a gradient: the output sublayer (LayerNorm',
        FFN-down and FFN-up input gradients, their weight gradients), LayerNorm' and the output projection of the attention
        sublayer run on those rows only (gathered operands, the dropout masks of the ORIGINAL rows); attention's backward is
        dense again (every unmasked key receives a gradient), as is everything below."""
        cfg = top.config
        x, qkv, actx, lse, z1, m1, r1, y1, u, g, z2, m2, r2, d_att, d_h1, d_h2 = saved_i
        if R32 is None:
            R32 = R.int()
        # the rows R of every saved activation this path reads, in ONE launch (9 index_select launches before)
        m2_c, r2_c, u_c, m1_c, r1_c, y1_c, g_c, actx_c = ops.gather_rows([m2, r2, u, m1, r1, y1, g, actx], R32)
        dy = dy_c
        dz2d_c = torch.empty_like(dy_c) if d_h2[1] else None
        dz2_c = ops.ln_bwd(dy_c, z2, m2_c, r2_c, lw["ln2_g"], lw["g_ln2_g"], lw["g_ln2_b"], x_rows=R32, dx2=dz2d_c, pre_drop=d_h2,
                           drop_rows=R32, deferred=lnd)
        if dz2d_c is None:
            dz2d_c = dz2_c
        du_c = ops.gemm_nt(dz2d_c, lw["W2T"], gelu_bwd_u=u_c)
        # K = 4H with a few hundred rows: 18 output tiles -> split-K (50 -> 15 us at the headline shape)
        dy1_c = ops.gemm_nt_splitk(du_c, lw["W1T"], resid=dz2_c) if du_c.shape[0] <= 1024 else ops.gemm_nt(du_c, lw["W1T"], resid=dz2_c)
        dz1d_c = torch.empty_like(dy_c) if d_h1[1] else None
        dz1_c = ops.ln_bwd(dy1_c, z1, m1_c, r1_c, lw["ln1_g"], lw["g_ln1_g"], lw["g_ln1_b"], x_rows=R32, dx2=dz1d_c, pre_drop=d_h1,
                           drop_rows=R32, deferred=lnd)
        if dz1d_c is None:
            dz1d_c = dz1_c
        dctx_c = ops.gemm_nt(dz1d_c, lw["WoT"])
        if rinv is not None:                                      # both full-height gradients in ONE launch (zero fill + index_copy_, twice, before)
            dctx, dz1 = ops.scatter_rows_zero([dctx_c, dz1_c], rinv, ra)
        else:
            dctx = torch.zeros((ra, H), device=dy.device, dtype=torch.bfloat16)
            dctx.index_copy_(0, R, dctx_c)
        # only the rows R of this layer's attention output have a gradient: the query loops of its backward stop at the last of them
        # in every sequence (the labelled text rows and [CLS] sit in a sequence's first rows: one 64-row query tile instead of ~7).
        # Exact: a query row with dO = 0 has delta = 0 and dS = 0 (round 4; model.top_layer_query_limit = False switches it off)
        qlim = ops.attn_q_limit(R32, layout) if getattr(top, "top_layer_query_limit", True) else None
        dqkv = ops.attn_bwd(qkv, actx, dctx, lse, key_bias, layout, H, drop=d_att, kv_len=kv_len, q_limit=qlim)
        if rinv is None:
            dz1 = torch.zeros((ra, H), device=dy.device, dtype=torch.bfloat16)
            dz1.index_copy_(0, R, dz1_c)
        out = ops.gemm_nt(dqkv, lw["WqkvT"], resid=dz1)
        # (bias gradients b1 / b2 / bo: column sums of the bf16 gradients on the ones-operand MFMA, as in the dense layers)
        few = [(du_c, y1_c, lw["g_W1"], lw["g_b1"]), (dz2d_c, g_c, lw["g_W2"], lw["g_b2"]), (dz1d_c, actx_c, lw["g_Wo"], lw["g_bo"])]
        if deferred is not None and getattr(top, "late_wgrads", True):
            # the deferred multi-layer call takes them: the QKV gradient (all rows of the backward) as one more of its problems -- it was a
            # launch of 27 tiles with the token axis split 8 ways + a reduce launch --, the three few-row ones behind its tiles
            deferred.append((dqkv, x, lw["g_Wqkv"], lw["g_bqkv"]))
            for q in few:
                _late_wgrad(top, q, True)
        else:
            _wgrad(top, few)
            _wgrad(top, [(dqkv, x, lw["g_Wqkv"], lw["g_bqkv"])])
        return out


def _trunk_static(plan, B, T, lens, joff, dev):
    """Shape-static row lists of the trunk (cached in the plan): ``jrows[k]`` = where the B*T text rows of joint pass k sit in the
    joint passes' pre-LayerNorm matrix J (row joff[k] + b * S_k + t), ``arange`` = 0 .. passes * B * T - 1 (the embedding dropout's row
    index of every text row), both int32 on the device."""
    key = ("trunk_static", T, tuple(sorted(joff.items())))
    st = plan.get(key)
    if st is None:
        jrows = {}
        for k, lo in joff.items():
            b = torch.arange(B, dtype=torch.int32)[:, None] * lens[k] + torch.arange(T, dtype=torch.int32)[None, :] + lo
            jrows[k] = b.reshape(-1).to(dev)
        groups, k = [], 0
        while k < len(lens):                                   # maximal runs of consecutive passes of one kind: one LayerNorm launch each
            k1 = k
            while k1 + 1 < len(lens) and ((k1 + 1) in joff) == (k in joff):
                k1 += 1
            groups.append((k in joff, k, k1 + 1, torch.cat([jrows[q] for q in range(k, k1 + 1)]) if k in joff else None))
            k = k1 + 1
        st = plan[key] = dict(jrows=jrows, groups=groups, arange=torch.arange(len(lens) * B * T, dtype=torch.int32, device=dev))
    return st


class _Trunk:
    """What one _TrunkFn call works on (plain attributes; built by _encode)."""
    __slots__ = ("ids", "tts", "B", "T", "lens", "pair_info", "feats", "feat_versions", "plan", "layout", "split", "key_bias", "kv_len", "seed", "top_rows",
                 "d_emb", "d_joint", "infer", "late_split", "compact", "J_pre")


class _TrunkFn(torch.autograd.Function):
    """Embeddings -> (JointEmbeddings) -> L encoder layers as ONE autograd node (the anchor parameter only makes autograd call
    backward; parameter gradients go straight into the flat gradient buffer).
      word+type+pos -> LayerNorm(1e-12) -> dropout                          (HF:53-108 via REF:MMBertForPretraining.py:264)
      cat(text_emb, relu(W.pair+b)) -> LayerNorm(1e-5) -> dropout(0.5)       (REF:MMBertEmbedding.py:57-72; joint passes)
      L x BertLayer                                                         (HF:374-416)
    Row maps instead of copies: the embedding LayerNorm stores the text-pass rows straight into the encoder's input X and the
    text rows of a joint pass into that pass's pre-LayerNorm matrix J (round 2: a copy per pass); the joint LayerNorm stores into X
    (round 2: torch.cat of the passes); with the valid-first packing both store through SplitLayout.inv32 (round 2: a [tokens, H]
    gather), the last encoder layer's LayerNorm un-packs through perm32 (another gather), and backward reads its output gradient
    through perm32 and hands the embedding stage its rows through inv32 with "past rows_a = zero" (two more gathers, a zero fill and
    the slice / accumulate kernels of autograd's cat / index backward).  Output: [tokens, H] bf16 in the caller's row order."""

    @staticmethod
    def forward(ctx, anchor, top, t):
        cfg, w = top.config, top._w
        H = cfg.hidden_size
        B, T, lens, plan, split = t.B, t.T, t.lens, t.plan, t.split
        bounds, tokens = plan["bounds"], plan["layout"].tokens
        dev = t.ids.device
        keep = ctx.needs_input_grad[0]
        late = t.layout                                           # callable: the packing is decided after the embedding launches
        npass = len(lens)
        joint = [k for k in range(npass) if t.pair_info[k] is not None]
        joff, jtot = {}, 0
        for k in joint:
            joff[k] = jtot
            jtot += B * lens[k]
        early = split is not None and not t.late_split and not t.infer          # the packing is known: store X in packed order right away
        dropped = split is not None and getattr(split, "dropped", False)
        Mx = split.rows_packed if early else tokens
        pad = 1 if (early and dropped) else 0                   # left-out rows (inv = rows_a) land in one scratch row behind X
        pre = t.J_pre                                             # (J, events): the pair rows of J are being written on side streams (_encode)
        t.J_pre = None
        if pre is not None and pre[0].shape[0] == jtot:
            buf = torch.empty((Mx + pad, H), device=dev, dtype=torch.bfloat16)
            X, J = buf[:Mx], pre[0]
        else:
            pre = None
            buf = torch.empty((Mx + pad + jtot, H), device=dev, dtype=torch.bfloat16)
            X, J = buf[:Mx], buf[Mx + pad:]
        st = _trunk_static(plan, B, T, lens, joff, dev)
        # ---- embeddings: e0 = word + type + pos for the text rows of all passes, LayerNorm + dropout stored where the rows are needed
        e0 = ops.embed_gather(t.ids, t.tts, w["word"], w["type"], w["pos"], T)
        mean0 = torch.empty(npass * B * T, device=dev, dtype=torch.float32)
        rstd0 = torch.empty_like(mean0)
        for is_joint, k0, k1, jr in st["groups"]:               # one launch per run of passes of one kind (2 at the headline shape)
            rows = slice(k0 * B * T, k1 * B * T)
            if is_joint:
                out, orows = J, jr
            elif early:
                out, orows = X, split.inv32[bounds[k0]:bounds[k1]]
            else:
                out, orows = X[bounds[k0]:bounds[k1]], None
            ops.ln_fwd(e0[rows], w["emb_ln_g"], w["emb_ln_b"], cfg.layer_norm_eps, out=out, out_rows=orows, drop=t.d_emb, drop_row0=k0 * B * T,
                       stats=(mean0[rows], rstd0[rows]))
        jstats = {}
        for n, k in enumerate(joint):
            S = lens[k]
            Jk = J[joff[k]:joff[k] + B * S]
            off = T
            if pre is not None:
                torch.cuda.current_stream().wait_event(pre[1][n])
            else:
                for feat, which in zip(t.feats[k], t.pair_info[k][1]):
                    ops.pair_proj_fwd(feat, w[which + "_w"], w[which + "_b"], Jk, T, seq_len=S, offset=off)
                    off += feat.shape[1]
            if early:
                out, orows = X, split.inv32[bounds[k]:bounds[k + 1]]
            else:
                out, orows = X[bounds[k]:bounds[k + 1]], None
            _, m_, r_ = ops.ln_fwd(Jk, w["joint_ln_g"], w["joint_ln_b"], LN_EPS_JOINT, out=out, out_rows=orows, drop=t.d_joint[k])
            jstats[k] = (m_, r_)
        if top.debug_hidden is not None:
            e1 = torch.cat([(J[joff[k]:joff[k] + B * lens[k]].view(B, lens[k], H)[:, :T].reshape(B * T, H) if k in joff else
                             (torch.cat((X, X.new_zeros((1, H)))).index_select(0, split.inv[bounds[k]:bounds[k + 1]]) if early else X[bounds[k]:bounds[k + 1]]))
                            for k in range(npass)])
            xo = torch.cat((X, X.new_zeros((1, H)))).index_select(0, split.inv) if early else X
            top.debug_hidden.update(emb=e1.detach().clone(), x=xo.detach().clone())
            top.debug_hidden.pop("_layers_packed", None)
        # ---- the packing, when it was not known before the embedding kernels were queued (synchronous prologue) or for inference
        if late is not None:
            split = t.split = late()
            t.layout = None
            dropped = split is not None and getattr(split, "dropped", False)
        x = X
        if split is not None and not early:
            x = X.index_select(0, split.perm)
        # ---- encoder; the last layer's LayerNorm un-packs (not for the inference packing: several rows share one there)
        layout = split if split is not None else plan["layout"]
        y_out = y_rows = None
        if split is not None and not t.infer:
            y_out = (torch.zeros if dropped else torch.empty)((tokens, H), device=dev, dtype=torch.bfloat16)
            y_rows = split.perm32
        y, saved = _EncoderFn.run_forward(top, x, layout, t.key_bias, t.seed, None if split is not None else t.kv_len, keep, y_out, y_rows)
        if split is not None and t.infer:
            y = y.index_select(0, split.inv)                      # every masked-out row reads its sequence's representative
        ctx.top, ctx.t, ctx.saved, ctx.joff, ctx.early = top, t, saved, joff, early
        if keep:
            ctx.save_for_backward(e0, mean0, rstd0, J, *[x_ for k in joint for x_ in jstats[k]])
        return y

    @staticmethod
    def backward(ctx, dy):
        top, t = ctx.top, ctx.t
        cfg, w = top.config, top._w
        H = cfg.hidden_size
        e0, mean0, rstd0, J, *js = ctx.saved_tensors
        B, T, lens, plan, split = t.B, t.T, t.lens, t.plan, t.split
        bounds = plan["bounds"]
        st = _trunk_static(plan, B, T, lens, ctx.joff, e0.device)
        npass = len(lens)
        layout = split if split is not None else plan["layout"]
        compact, t.compact = t.compact, None                      # (rows in the caller's order, their gradients): set by the MLM head
        # no MLM-head launch has overwritten a dropped table gradient (no labelled row): zero it before the embedding rows are added / the
        # slice is reduced -- unless that launch waits in the deferred call (then after the encoder's backward, below: a no-op)
        if not any(q[2] is w["g_word_pad"] for q in (top.__dict__.get("_late_wgrads") or ())):
            top._flat.settle([w["g_word_pad"]])
        # ONE collector for the LayerNorm' gamma / beta sums of the whole backward -- the MLM head's call (already in it), the sparse top
        # layer's two, the dense layers', the embedding stage's: one reduce launch at the end instead of seven (a data-parallel hook
        # flushes it wherever it needs final gradients)
        lnd = top._shared_lnd()
        if top.head_grad_hook is not None:
            lnd.flush()
            top.head_grad_hook()                                  # every gradient of the heads is final, the tied decoder's included
        dy_rows = None
        if compact is None:
            dy = dy.contiguous()
            if split is not None:
                dy_rows = split.perm32[:split.rows_a]             # dy is in the caller's order: the top LayerNorm' reads it through the map
        dx = _EncoderFn.run_backward(top, layout, t.key_bias, None if split is not None else t.kv_len, ctx.saved, dy, dy_rows, t.top_rows,
                                     compact, lnd)
        ctx.saved = None
        top._flat.settle([w["g_word_pad"]])
        if split is not None:
            top.last_backward_row_fraction = float(split.rows_a) / split.tokens
        # ---- embedding stage: dx holds the leading rows_a rows of the packed order (all rows without the packing)
        ra = dx.shape[0]
        inv32 = split.inv32 if split is not None else None
        limit = ra if split is not None else 0
        dJ = torch.empty_like(J)
        de0 = torch.empty_like(e0)
        joint = sorted(ctx.joff)
        for n, k in enumerate(joint):
            S = lens[k]
            lo = ctx.joff[k]
            m_, r_ = js[2 * n], js[2 * n + 1]
            if split is not None:
                ops.ln_bwd(dx, J[lo:lo + B * S], m_, r_, w["joint_ln_g"], w["g_joint_ln_g"], w["g_joint_ln_b"], dx=dJ[lo:lo + B * S],
                           dy_rows=inv32[bounds[k]:bounds[k + 1]], dy_row_limit=limit, post_drop=t.d_joint[k], deferred=lnd)
            else:
                ops.ln_bwd(dx[bounds[k]:bounds[k + 1]], J[lo:lo + B * S], m_, r_, w["joint_ln_g"], w["g_joint_ln_g"], w["g_joint_ln_b"],
                           dx=dJ[lo:lo + B * S], post_drop=t.d_joint[k], deferred=lnd)
        for n, k in enumerate(joint):
            S = lens[k]
            lo = ctx.joff[k]
            off = T
            _check_unchanged(t.feats[k], t.feat_versions[k])
            for feat, which in zip(t.feats[k], t.pair_info[k][1]):
                ops.pair_proj_bwd(feat, J[lo:lo + B * S], dJ[lo:lo + B * S], T, w["g_" + which + "_w"], w["g_" + which + "_b"], seq_len=S, offset=off)
                off += feat.shape[1]
        for is_joint, k0, k1, jr in st["groups"]:
            rows = slice(k0 * B * T, k1 * B * T)
            kw = dict(dx=de0[rows], post_drop=t.d_emb, drop_rows=st["arange"][rows] if t.d_emb[1] else None, deferred=lnd)
            if is_joint:
                ops.ln_bwd(dJ, e0[rows], mean0[rows], rstd0[rows], w["emb_ln_g"], w["g_emb_ln_g"], w["g_emb_ln_b"], dy_rows=jr, **kw)
            elif split is not None:
                ops.ln_bwd(dx, e0[rows], mean0[rows], rstd0[rows], w["emb_ln_g"], w["g_emb_ln_g"], w["g_emb_ln_b"],
                           dy_rows=inv32[bounds[k0]:bounds[k1]], dy_row_limit=limit, **kw)
            else:
                ops.ln_bwd(dx[bounds[k0]:bounds[k1]], e0[rows], mean0[rows], rstd0[rows], w["emb_ln_g"], w["g_emb_ln_g"], w["g_emb_ln_b"], **kw)
        lnd.flush()
        # the deferred call writes the tied table's gradient whole; the rows below add to it.  (Measured: that gradient on a side stream of
        # its own at the start of backward, so that embed_scatter could run beside the deferred call: +0.1 %, profiles/r6_ab_side_streams.log)
        _join_wgrads(top)
        if top.defer_embed_rows:
            ops.embed_scatter(t.ids, t.tts, de0, T, None, w["g_type"], w["g_pos"], vocab=cfg.vocab_size)
            # a LIST: every differentiated trunk backward between two finish_backward() calls hands over its rows (a second
            # backward must not overwrite the first one's -- parallel.DataParallel exchanges all of them)
            top.__dict__.setdefault("_deferred_embed_rows", []).append((t.ids, de0))
        else:
            ops.embed_scatter(t.ids, t.tts, de0, T, w["g_word"], w["g_type"], w["g_pos"])
        return None, None, None


def mlm_active_rows(labels, vocab, first=None):
    """(row list on the device, pinned host words, event): which packed rows carry an MLM label.  Host words (they travel by ONE
    async copy): [0] the COUNT of labelled rows, [1] how many of the ``first`` ([CLS]) rows are among them (the sparse top-layer
    backward gathers labelled rows and [CLS] rows as one list and must not see a row twice), [2] how many labels are neither
    -100 nor a vocabulary index (torch's CrossEntropyLoss raises on those; here the error surfaces when the words are read).
    Issue this as EARLY in the forward pass as the labels exist -- the sparse MLM backward synchronises on the event, and a copy
    issued at the end of forward would make the host wait at the start of every backward until the GPU has finished the whole
    forward pass (measured: no throughput difference on one GPU, where the host is far ahead anyway; kept early so that nothing
    depends on that)."""
    idx, cnt = ops.active_rows(labels, vocab)
    dup = (labels.index_select(0, first) != -100).sum().to(torch.int32).view(1) if first is not None else torch.zeros_like(cnt)
    bad = ((labels != -100) & ((labels < 0) | (labels >= vocab))).sum().to(torch.int32).view(1)
    host = torch.empty(3, dtype=torch.int32, pin_memory=True)
    host.copy_(torch.cat((cnt, dup, bad)), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return idx, host, ev


def _active_row_count(rows) -> int:
    """Waits for the words of mlm_active_rows() and returns the labelled-row count; labels outside the vocabulary raise like
    torch.nn.CrossEntropyLoss does in the reference (REF:MMBertForPretraining.py:381-384)."""
    _idx, host, ev = rows
    ev.synchronize()
    if int(host[2]) != 0:
        raise IndexError(f"masked_labels: {int(host[2])} label(s) are neither -100 nor in [0, vocab_size) -- Target out of bounds")
    return int(host[0])


_ones = {}


class _ScalarLoss(torch.Tensor):
    """The 0-dim joint loss as ``forward()`` returns it.  The reference's loop differentiates ``outputs[0].mean()`` (REF:trainer.py:83); on a
    0-dim tensor ``mean()`` is the identity, yet torch launches a reduction for it, a fill for the implicit unit gradient and a division in
    ``MeanBackward`` -- three ATen launches per step in front of the heads' backward.  Here ``mean()`` of the scalar returns the scalar and
    ``backward()`` without a gradient seeds a cached device-side 1.0: the same graph, the same values, no launch.  Everything else is
    ``torch.Tensor`` (``.item()``, arithmetic, ``.detach()``, a ``mean`` with arguments)."""

    def mean(self, *args, **kwargs):
        if self.dim() == 0 and not args and not kwargs:
            return self
        return super().mean(*args, **kwargs)

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if gradient is None and self.dim() == 0 and not create_graph:
            key = (self.dtype, self.device)
            gradient = _ones.get(key)
            if gradient is None:
                gradient = _ones[key] = torch.ones((), dtype=self.dtype, device=self.device)
        return torch.autograd.backward(self, gradient, retain_graph, create_graph, inputs=inputs)


def _scalar_loss(loss):
    return loss.as_subclass(_ScalarLoss) if (torch.is_tensor(loss) and loss.dim() == 0 and type(loss) is torch.Tensor) else loss


_dummies = {}


def _zero_dummy(M, H, dtype, device):
    """A [M, H] all-zero tensor of stride 0 (one element of storage): what autograd is handed for a tensor whose real gradient
    travels in compact form beside the graph."""
    key = (dtype, str(device))
    z = _dummies.get(key)
    if z is None:
        z = _dummies[key] = torch.zeros((1, 1), dtype=dtype, device=device)
    return z.expand(M, H)


class _MLMHeadFn(torch.autograd.Function):
    """decoder(LN(gelu(dense(seq)))) + per-pass CrossEntropy(ignore -100)
    (HF:466-496 via REF:MMBertForPretraining.py:293,381-384).  Returns (loss[nseg], logits or None, first): ``first`` = the rows
    ``first_rows`` of y in fp32 (the [CLS] rows the other heads read) -- handed out here so that y has ONE consumer in the autograd
    graph and its gradient is assembled once (labelled rows copied in, [CLS] rows added) instead of two [tokens, H] tensors and
    their sum."""

    @staticmethod
    def forward(ctx, y, anchor, top, labels, seg_bounds_host, seg_bounds, want_scores, rows=None, first_rows=None, trunk=None):
        """``trunk``: the _Trunk record of the _TrunkFn call that produced y (forward() / forward_fused() pass it): when the top encoder
        layer's sparse backward applies, backward hands it the gradient of y in COMPACT form -- the labelled rows and the [CLS] rows,
        the only ones with a gradient -- through ``trunk.compact`` and returns a stride-0 dummy for y (round 2: a zero-filled [tokens, H]
        tensor, an index_copy, an index_add, and a row gather to pick the rows out again)."""
        cfg, w = top.config, top._w
        ctx.trunk = trunk
        M, H = y.shape
        V, Vp = cfg.vocab_size, top._flat.vpad
        keep = ctx.needs_input_grad[0]
        y = y.contiguous()
        ctx.compact = False
        ctx.first_rows = first_rows
        first = None
        if first_rows is not None:
            if getattr(top, "_heads_read_rows", False):
                # the level-launch heads (_HeadsStepFn) read rows first_rows of y themselves: `first` only carries the autograd edge (an
                # uninitialised buffer: no gather, no cast); the real operand travels beside the graph, like the trunk's compact gradient
                first = torch.empty((first_rows.numel(), H), device=y.device, dtype=torch.float32)
                top.__dict__["_heads_src"] = (y, first_rows, first.data_ptr())
            else:
                first = y.index_select(0, first_rows).float()
        if (not want_scores) and keep and rows is not None and getattr(top, "sparse_mlm_backward", True):
            # the caller does not want the prediction scores: the whole head runs on the labelled rows only (the loss is a mean
            # over them; trainer.py never reads the scores) -- forward included
            idx_all, host, ev = rows
            n = _active_row_count(rows)                       # requested at the start of forward: long complete
            if 0 < n and 2 * n <= M:
                sel32 = idx_all[:n]
                sel = sel32.long()
                y_c = y.index_select(0, sel)
                pre_c = torch.empty_like(y_c)
                t0_c = ops.gemm_nt(y_c, w["Wt"], bias=w["bt"], gelu=True, aux=pre_c)
                t_c, mean_c, rstd_c = ops.ln_fwd(t0_c, w["mlm_ln_g"], w["mlm_ln_b"], cfg.layer_norm_eps, stats=True)
                logits_c = ops.gemm_nt(t_c, w["word_h"], bias=w["pred_bias"])          # [n, Vpad] bf16
                labels_c = labels.index_select(0, sel)
                nseg = len(seg_bounds_host) - 1
                bounds_c = torch.searchsorted(sel32, seg_bounds.to(sel32.dtype)).to(torch.int32)   # labelled rows are in row order
                loss, inv, lse = ops.ce_fwd(logits_c, V, labels_c, bounds_c, nseg)
                ctx.top, ctx.nseg, ctx.compact, ctx.M = top, nseg, True, M
                ctx.set_materialize_grads(False)
                ctx.save_for_backward(y_c, pre_c, t0_c, mean_c, rstd_c, t_c, logits_c, labels_c, bounds_c, inv, lse, sel)
                return loss, None, first
        pre = torch.empty_like(y) if keep else None
        t0 = ops.gemm_nt(y, w["Wt"], bias=w["bt"], gelu=True, aux=pre)
        t, mean, rstd = ops.ln_fwd(t0, w["mlm_ln_g"], w["mlm_ln_b"], cfg.layer_norm_eps, stats=keep)
        # [M, Vpad]: bf16, or fp32 straight from the accumulators when the caller wants the reference's fp32 prediction scores
        f32_scores = want_scores and getattr(top, "scores_dtype", torch.bfloat16) == torch.float32
        logits = ops.gemm_nt(t, w["word_h"], bias=w["pred_bias"], out_f32=f32_scores)
        nseg = len(seg_bounds_host) - 1
        loss, inv, lse = ops.ce_fwd(logits, V, labels, seg_bounds, nseg)
        ctx.top, ctx.nseg, ctx.keep_logits, ctx.M = top, nseg, want_scores, M
        ctx.set_materialize_grads(False)      # or autograd zero-fills a [tokens, vocab] gradient for the returned scores
        # Rows without a label have an exactly-zero CE gradient (ignore_index): backward needs the labelled rows only (~2 % of
        # the packed tokens).  ``rows`` = (device row list, pinned host count, event), normally made by mlm_active_rows() at
        # the START of the forward pass, so that the count is on the host long before backward asks for it.
        ctx.rows = None
        if keep and getattr(top, "sparse_mlm_backward", True):
            ctx.rows = rows if rows is not None else mlm_active_rows(labels, V)
        if keep:
            ctx.save_for_backward(y, pre, t0, mean, rstd, t, logits, labels, seg_bounds, inv, lse)
        out_logits = logits if want_scores else None
        if out_logits is not None:
            ctx.mark_non_differentiable(out_logits)
        return loss, out_logits, first

    @staticmethod
    def backward(ctx, dloss, _unused, dfirst=None):
        t = ctx.trunk
        # room for the [CLS] rows behind the labelled rows' gradients: the compact hand-over to the trunk is then ONE row list and ONE matrix
        # without a cat (the rows' gradients are written in place by the last product, the [CLS] rows by one converting copy)
        nf = t.top_rows[1].numel() if (t is not None and t.top_rows is not None and dfirst is not None) else 0
        res = _MLMHeadFn._backward(ctx, dloss, extra_rows=nf)
        _join_heads(ctx.top)                                  # (the heads' backward may still run on its side stream: dfirst is read from here on)
        nones = (None,) * 9
        if isinstance(res, tuple):                            # (labelled rows int32 or int64, their gradients [n (+ nf), H] bf16): the sparse paths
            sel, dy_all = res
            n = sel.numel()
            dyl = dy_all[:n]
            if nf and _MLMHeadFn._compact_ok(ctx, t, n, dfirst):
                first = t.top_rows[1]
                df = dfirst if dfirst.shape[0] == first.numel() else dfirst.view(-1, first.numel(), dfirst.shape[1]).sum(0)   # ([CLS] rows repeated: fused)
                dy_all[n:].copy_(df)
                t.compact = (sel, first, dy_all)
                return (_zero_dummy(ctx.M, dyl.shape[1], dyl.dtype, dyl.device),) + nones
            dy = torch.zeros((ctx.M, dyl.shape[1]), device=dyl.device, dtype=dyl.dtype)
            if n:
                dy.index_copy_(0, sel.long(), dyl)
        else:
            dy = res
        if dfirst is not None:                                # a [CLS] row may also carry a label: add, after the copy
            if dy is None:
                dy = torch.zeros((ctx.M, dfirst.shape[1]), device=dfirst.device, dtype=torch.bfloat16)
            dy.index_add_(0, ctx.first_rows, dfirst.to(dy.dtype))
        return (dy,) + nones

    @staticmethod
    def _compact_ok(ctx, t, n, dfirst):
        """The condition under which _EncoderFn.run_backward takes the sparse top-layer path (_sparse_rows): few rows, no labelled
        [CLS] row -- evaluated here with the same numbers, so that a compact hand-over is never refused there."""
        (idx, host, ev), first = t.top_rows
        ra = t.split.rows_a if t.split is not None else t.plan["layout"].tokens
        if ctx.first_rows is None or (ctx.first_rows.numel() % first.numel()) != 0:
            return False
        return 4 * (n + first.numel()) <= ra and int(host[1]) == 0 and n == _active_row_count(t.top_rows[0])

    @staticmethod
    def _backward(ctx, dloss, extra_rows=0):
        w = ctx.top._w
        V = ctx.top.config.vocab_size
        if dloss is None:
            return None
        ctx.top._flat.grads_dirty = True
        ctx.top._flat.attach_lazy()
        ctx.top._flat.wait_transposes()                           # (the optimizer's transposed-copy launch runs on a side stream)
        gs = dloss.contiguous().float()
        # the transform LayerNorm's gamma / beta sums join the trunk's collector when the trunk's backward follows (it flushes), else
        # they are folded right here
        lnd = ctx.top._shared_lnd() if (ctx.trunk is not None and ctx.needs_input_grad[0]) else None
        # the few-row weight gradients wait for the trunk's backward (which follows in this pass) unless a hook wants them final before it
        late = (ctx.trunk is not None and ctx.needs_input_grad[0] and ctx.top.grad_hook is None and ctx.top.head_grad_hook is None
                and getattr(ctx.top, "late_wgrads", True))
        if ctx.top.__dict__.get("_late_wgrads"):                 # (left by a backward whose trunk stage never ran)
            _flush_late_wgrads(ctx.top)
        if ctx.compact:
            y_c, pre_c, t0_c, mean_c, rstd_c, t_c, logits_c, labels_c, bounds_c, inv, lse, sel = ctx.saved_tensors
            dl = ops.ce_bwd(logits_c, V, labels_c, bounds_c, ctx.nseg, inv, gs, lse, logits_c)       # in place: the scores go nowhere
            _late_wgrad(ctx.top, (dl, t_c, w["g_word_pad"], w["g_pred_bias"]), late)
            dt = ops.gemm_nt_splitk(dl, w["wordT"])
            dt0 = ops.ln_bwd(dt, t0_c, mean_c, rstd_c, w["mlm_ln_g"], w["g_mlm_ln_g"], w["g_mlm_ln_b"], deferred=lnd)
            dpre = ops.gelu_bwd(dt0, pre_c)
            _late_wgrad(ctx.top, (dpre, y_c, w["g_Wt"], w["g_bt"]), late)
            dy_all = torch.empty((sel.numel() + extra_rows, y_c.shape[1]), device=y_c.device, dtype=torch.bfloat16)
            ops.gemm_nt(dpre, w["WtT"], out=dy_all[:sel.numel()])
            return sel, dy_all
        y, pre, t0, mean, rstd, t, logits, labels, seg_bounds, inv, lse = ctx.saved_tensors
        M = y.shape[0]
        if ctx.rows is not None:
            idx_all, host, ev = ctx.rows
            n = _active_row_count(ctx.rows)
            if 2 * n <= M:
                if n == 0:
                    return idx_all[:0], y.new_zeros((extra_rows, y.shape[1]))
                idx = idx_all[:n]
                dl = torch.empty((n, logits.shape[1]), device=y.device, dtype=torch.bfloat16)
                ops.ce_bwd(logits, V, labels, seg_bounds, ctx.nseg, inv, gs, lse, dl, rows=idx)
                t_c, t0_c, pre_c, y_c, mean_c, rstd_c = ops.gather_rows([t, t0, pre, y, mean, rstd], idx)      # one launch
                _late_wgrad(ctx.top, (dl, t_c, w["g_word_pad"], w["g_pred_bias"]), late)
                dt = ops.gemm_nt_splitk(dl, w["wordT"])                     # K = vocabulary, a few hundred rows: split-K
                dt0 = ops.ln_bwd(dt, t0_c, mean_c, rstd_c, w["mlm_ln_g"], w["g_mlm_ln_g"], w["g_mlm_ln_b"], deferred=lnd)
                dpre = ops.gelu_bwd(dt0, pre_c)
                _late_wgrad(ctx.top, (dpre, y_c, w["g_Wt"], w["g_bt"]), late)
                dy_all = torch.empty((n + extra_rows, y.shape[1]), device=y.device, dtype=torch.bfloat16)
                ops.gemm_nt(dpre, w["WtT"], out=dy_all[:n])
                return idx, dy_all
        # dense path: dlogits with the per-pass upstream gradients folded in; in place unless the scores were handed to the caller
        dl = torch.empty(logits.shape, device=logits.device, dtype=torch.bfloat16) if (ctx.keep_logits or logits.dtype != torch.bfloat16) else logits
        ops.ce_bwd(logits, V, labels, seg_bounds, ctx.nseg, inv, gs, lse, dl)
        ops.gemm_tn(dl, t, w["g_word_pad"], bias_out=w["g_pred_bias"], accumulate=ctx.top._flat.take_accumulate([w["g_word_pad"]]))   # tied decoder weight + prediction bias gradient
        dt = ops.gemm_nt(dl, w["wordT"])
        dt0 = ops.ln_bwd(dt, t0, mean, rstd, w["mlm_ln_g"], w["g_mlm_ln_g"], w["g_mlm_ln_b"], deferred=lnd)
        dpre = ops.gelu_bwd(dt0, pre)
        ops.gemm_tn(dpre, y, w["g_Wt"], bias_out=w["g_bt"])
        return ops.gemm_nt(dpre, w["WtT"])


# ================================================================================================
# the reference's module API
# ================================================================================================
class _GpuModelBase(nn.Module):
    """Shared machinery: flat storage, weight views, dropout seeds, pass packing."""

    def _init_runtime(self):
        self._flat: Optional[FlatParams] = None
        self._seed = 0x5EED
        self._calls = 0
        self._plans = {}
        self.grad_hook = None           # set by parallel.DataParallel: called as layers finish in backward
        self.head_grad_hook = None      # ... and once when the heads' backward is complete (the trunk's backward starts)
        # parallel.DataParallel: the embedding lookup's row gradients stay OUT of the word-embedding table's gradient; backward leaves
        # (ids, rows) in ``_deferred_embed_rows`` for the wrapper's compact exchange (the table itself is reduced early)
        self.defer_embed_rows = False
        self.embed_ids_hook = None      # ... called with the step's token ids at the start of a differentiated forward pass
        # tests only: a dict here collects, per _encode() call, "emb" (text embeddings of all passes), "x" (encoder input) and
        # "layers" (every encoder layer's output), all [tokens, H] bf16 in the ORIGINAL packed row order (pass, sample, position)
        self.debug_hidden = None

    def _shared_lnd(self):
        """The one ``ops.LnDeferred`` collector of a backward pass (MLM head -> trunk: see ``_TrunkFn.backward``)."""
        l = self.__dict__.get("_lnd")
        if l is None:
            l = self.__dict__["_lnd"] = ops.LnDeferred(32)
        return l

    @property
    def input_stream(self):
        """The stream the step prologue runs on with ``async_prologue`` (created on first use).  A data pipeline that builds its
        batches on the GPU does so on this stream (see ``async_prologue``)."""
        s = self.__dict__.get("_input_stream")
        if s is None:
            s = self.__dict__["_input_stream"] = torch.cuda.Stream()
        return s

    def _prologue_stream(self, first=None):
        """None, or the input stream when the step prologue may run there, AHEAD of the current stream -- which is the case
        * for a batch ``trainer.on_input_stream`` has just built on ``model.input_stream``: it tags the batch (``first``, the batch's
          first tensor, is the tagged one), so an eval_epoch or a plain train_epoch that follows in the same process -- batches built
          on the CURRENT stream -- takes the synchronous path again (round 2 set a sticky flag there: masks and labels of such
          batches were then read before they were written);
        * when ``model.async_prologue`` is set: the CALLER's statement that (a) the tensors it passes to forward() are not still
          being written by work queued on the current stream -- they are static or prefetched (and synchronised; bench.py) -- and
          (b) it runs the trainer's order, every forward followed by its backward before the next forward but one (the prologue's
          outputs live in two alternating buffer sets).
        The masks and labels are then read on the input stream without waiting for the current one."""
        tag = self.__dict__.get("_input_stream_batch")
        if getattr(self, "async_prologue", False) or (tag is not None and tag is first):
            return self.input_stream
        return None

    def manual_seed(self, seed: int):
        self._seed, self._calls = int(seed), 0

    def _next_seed(self) -> int:
        self._calls += 1
        return self._seed * 1000003 + self._calls

    def _layer_grads_done(self, i: int):
        if self.grad_hook is not None:
            self.grad_hook(i)

    def _bert(self):
        return self.bert if hasattr(self, "bert") else self

    def _ensure_ready(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("msa_amd runs on MI355X only: move the model and its inputs to 'cuda' (there is no CPU path)")
        if not hasattr(self._bert(), "jointEmbeddings"):
            raise RuntimeError("call .bert.set_joint_embeddings(dataset) first (REF:train.py:72)")
        p0 = next(self.parameters())
        if p0.device != device:
            raise RuntimeError(f"model is on {p0.device}, inputs on {device}")
        if self._flat is None or not self._flat.owns(self):
            self._flat = FlatParams(self, self.config, device)
            self._build_views()
        else:
            self._flat.maybe_refresh()

    def _build_views(self):
        f, cfg = self._flat, self.config
        H, I, V, L = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers
        pre = "bert." if hasattr(self, "bert") else ""
        if not pre:
            raise RuntimeError("flat storage is built from MMBertForPretraining")
        self._lw = []
        f.settle()
        f.lazy.clear()
        for i in range(L):
            p = f"bert.encoder.layer.{i}."
            d = {}
            for tag, buf in (("", f.half), ("g_", f.grads)):
                d[tag + "Wqkv"] = f.span(buf, p + "attention.self.query.weight", 3 * H * H, (3 * H, H))
                d[tag + "Wo"] = f.span(buf, p + "attention.output.dense.weight", H * H, (H, H))
                d[tag + "W1"] = f.span(buf, p + "intermediate.dense.weight", I * H, (I, H))
                d[tag + "W2"] = f.span(buf, p + "output.dense.weight", H * I, (H, I))
            for tag, buf in (("", f.params), ("g_", f.grads)):
                d[tag + "bqkv"] = f.span(buf, p + "attention.self.query.bias", 3 * H, (3 * H,))
                d[tag + "bo"] = f.span(buf, p + "attention.output.dense.bias", H, (H,))
                d[tag + "b1"] = f.span(buf, p + "intermediate.dense.bias", I, (I,))
                d[tag + "b2"] = f.span(buf, p + "output.dense.bias", H, (H,))
                d[tag + "ln1_g"] = f.span(buf, p + "attention.output.LayerNorm.weight", H, (H,))
                d[tag + "ln1_b"] = f.span(buf, p + "attention.output.LayerNorm.bias", H, (H,))
                d[tag + "ln2_g"] = f.span(buf, p + "output.LayerNorm.weight", H, (H,))
                d[tag + "ln2_b"] = f.span(buf, p + "output.LayerNorm.bias", H, (H,))
            d["WqkvT"], d["WoT"], d["W1T"], d["W2T"] = f.tview(p + "qkv"), f.tview(p + "o"), f.tview(p + "w1"), f.tview(p + "w2")
            # lazy zero (flat.FlatParams): every backward writes these whole with ONE weight-gradient launch each (_wgrad)
            f.register_lazy(d["g_Wqkv"], [p + "attention.self." + x + ".weight" for x in ("query", "key", "value")])
            f.register_lazy(d["g_Wo"], [p + "attention.output.dense.weight"])
            f.register_lazy(d["g_W1"], [p + "intermediate.dense.weight"])
            f.register_lazy(d["g_W2"], [p + "output.dense.weight"])
            self._lw.append(d)
        w = {}
        e = "bert.embeddings."
        j = "bert.jointEmbeddings."
        c = "cls.predictions."
        vd, sd = self.bert.jointEmbeddings.VISUALDIM, self.bert.jointEmbeddings.SPEECHDIM
        for tag, buf in (("", f.params), ("g_", f.grads)):
            w[tag + "word"] = f.span(buf, e + "word_embeddings.weight", V * H, (V, H))
            w[tag + "word_pad"] = f.span(buf, e + "word_embeddings.weight", f.vpad * H, (f.vpad, H))
            w[tag + "type"] = f.span(buf, e + "token_type_embeddings.weight", cfg.type_vocab_size * H, (cfg.type_vocab_size, H))
            w[tag + "pos"] = f.span(buf, e + "position_embeddings.weight", cfg.max_position_embeddings * H, (cfg.max_position_embeddings, H))
            w[tag + "emb_ln_g"] = f.span(buf, e + "LayerNorm.weight", H, (H,))
            w[tag + "emb_ln_b"] = f.span(buf, e + "LayerNorm.bias", H, (H,))
            w[tag + "joint_ln_g"] = f.span(buf, j + "LayerNorm.weight", H, (H,))
            w[tag + "joint_ln_b"] = f.span(buf, j + "LayerNorm.bias", H, (H,))
            w[tag + "Wv_w"] = f.span(buf, j + "Wv.weight", H * vd, (H, vd))
            w[tag + "Wv_b"] = f.span(buf, j + "Wv.bias", H, (H,))
            w[tag + "Ws_w"] = f.span(buf, j + "Ws.weight", H * sd, (H, sd))
            w[tag + "Ws_b"] = f.span(buf, j + "Ws.bias", H, (H,))
            w[tag + "pred_bias"] = f.span(buf, c + "bias", f.vpad, (f.vpad,))
            w[tag + "bt"] = f.span(buf, c + "transform.dense.bias", H, (H,))
            w[tag + "mlm_ln_g"] = f.span(buf, c + "transform.LayerNorm.weight", H, (H,))
            w[tag + "mlm_ln_b"] = f.span(buf, c + "transform.LayerNorm.bias", H, (H,))
        w["Wt"] = f.span(f.half, c + "transform.dense.weight", H * H, (H, H))
        w["g_Wt"] = f.span(f.grads, c + "transform.dense.weight", H * H, (H, H))
        w["word_h"] = f.span(f.half, e + "word_embeddings.weight", f.vpad * H, (f.vpad, H))
        w["WtT"], w["wordT"] = f.tview("transform"), f.tview("word")
        f.register_lazy(w["g_word_pad"], [e + "word_embeddings.weight"])     # the tied decoder's weight gradient: the MLM head's first launch
        self._w = w

    # ---- pass packing --------------------------------------------------------------------------
    def _plan(self, lens_per_pass, B, device):
        key = (tuple(lens_per_pass), B, str(device))
        pl = self._plans.get(key)
        if pl is None:
            lens, first, bounds = [], [], [0]
            row = 0
            for S in lens_per_pass:
                for b in range(B):
                    lens.append(S)
                    first.append(row + b * S)
                row += B * S
                bounds.append(row)
            lay = ops.SeqLayout(lens, self.config.num_attention_heads, device)
            pl = dict(layout=lay, first=torch.tensor(first, dtype=torch.int64, device=device),
                      bounds=bounds, bounds_dev=torch.tensor(bounds, dtype=torch.int32, device=device),
                      row_seq=torch.from_numpy(lay._row_seq).to(device), row_pos=torch.from_numpy(lay._row_pos).to(device))
            self._plans[key] = pl
        return pl

    @staticmethod
    def _mask2d(mask, joint_pair: bool, dev):
        """The [B, positions] mask whose (1 - m) * -10000 is the key bias (REF:MMBertForPretraining.py:57-154): 2-D masks as they
        are; 3-D joint masks -> feature 0 (a strided VIEW, :76); 3-D non-joint masks -> the mean over features (:111)."""
        mask = mask.to(dev)
        if mask.dim() == 3:
            return mask[:, :, 0] if joint_pair else mask.float().mean(2)
        if mask.dim() != 2:
            raise ValueError("You have so large dimension (), Check dimension or shape ")
        return mask

    def _pack_inputs(self, passes, label_parts=None):
        """(ids, token types, labels) of all passes in the order of the token matrix from ONE launch (mmbert_pack_i64; round 2:
        three torch.cat launches, two zero fills and the dtype conversions) -- on the input stream when the prologue runs there.
        None when an input is not an int64 GPU tensor (the callers then take the torch path)."""
        B, T = passes[0]["ids"].shape
        tens = [p["ids"] for p in passes] + [p["tt"] for p in passes if p.get("tt") is not None] + list(label_parts or ())
        if not all(torch.is_tensor(t_) and t_.is_cuda and t_.dtype == torch.int64 for t_ in tens):
            return None
        segs = [p["ids"] for p in passes] + [(p["tt"] if p.get("tt") is not None else (B * T, 0)) for p in passes] + list(label_parts or ())
        side = self._prologue_stream(passes[0]["ids"])
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            packed, _offs = ops.pack_i64(segs)
        if side is not None:
            packed.record_stream(torch.cuda.current_stream())
        n = len(passes) * B * T
        return packed[:n], packed[n:2 * n], (packed[2 * n:] if label_parts else None)

    def _encode(self, passes, labels=None, want_rows=False, rowset=False, packed=None):
        """passes: list of dict(ids[B,T], tt[B,T]|None, mask, pair[B,P,D]|None, pair_mask|None).
        Returns (Y [tokens,H] bf16, plan, lens_per_pass, rows) -- ``rows`` = (labelled-row list, host words, event) when asked for.
        ``rowset``: the valid-first packing over a row SET instead of a prefix per sequence (the prologue's row-set mode): for
        sequences whose masked-out rows do not sit at the end -- the fused text | visual | speech sequence has its visual padding in
        the middle.  Every sequence is then run in its own valid-first order (active rows = unmasked keys and labelled rows, first):
        attention is invariant under a permutation of the keys that carries the key bias along, every other operator is row-wise,
        and the un-packing gather restores the caller's order.  Only with a backward pass to save (labels, autograd on)."""
        bert = self._bert()
        dev = passes[0]["ids"].device
        self._ensure_ready(dev)
        lnd = self.__dict__.get("_lnd")
        if lnd is not None and lnd.items and torch.is_grad_enabled():
            lnd.drop()                 # a backward pass that raised half-way left its LayerNorm' sums behind: no pass is in flight at a forward
        B, T = passes[0]["ids"].shape
        cfg = self.config
        if T > cfg.max_position_embeddings:
            raise ValueError("text length exceeds max_position_embeddings")
        seed = self._next_seed()
        je = bert.jointEmbeddings
        rowset = bool(rowset and labels is not None and torch.is_grad_enabled() and getattr(self, "skip_padded_backward", True)
                      and getattr(self, "skip_masked_keys", True))
        # ---- masks and labels first: ONE prologue call (two launches) gives the padded key bias, the per-sequence unmasked
        # lengths, the rows backward must visit and the labelled-row list, and starts the one device -> host copy of the step
        # (the embedding kernels below keep the GPU busy while it travels and the host packs the layout)
        segs, lens, pair_info = [], [], []
        side = self._prologue_stream(passes[0]["ids"])
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):     # (a 3-D text mask is reduced by a kernel)
            for k, p in enumerate(passes):
                segs.append((self._mask2d(p["mask"], False, dev), k, 0))
                if p.get("pair") is not None:
                    pairs = p["pair"] if isinstance(p["pair"], (tuple, list)) else (p["pair"],)
                    pmasks = p["pair_mask"] if isinstance(p["pair_mask"], (tuple, list)) else (p["pair_mask"],)
                    off = T
                    for f, pm in zip(pairs, pmasks):
                        segs.append((self._mask2d(pm, True, dev), k, off))
                        off += f.shape[1]
                    lens.append(off)
                    pair_info.append((pairs, tuple(je.which(f) for f in pairs)))
                else:
                    lens.append(T)
                    pair_info.append(None)
        plan = self._plan(lens, B, dev)
        if side is None:
            pro = ops.prologue(segs, lens, B, labels, cfg.vocab_size, dev, rowset=rowset)
            nseq = pro.nseq
            host = torch.empty(nseq + 3, dtype=torch.int32, pin_memory=True)
            host.copy_(pro.words, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            # async_prologue: the prologue and its device -> host copy run on the input stream, AHEAD of whatever the current stream
            # still has queued (the previous step's backward and optimizer) -- the host gets the lengths without waiting for the GPU
            # to drain and keeps enqueueing a step ahead.  Outputs alternate between two persistent buffer sets: set k % 2 was last
            # read by step k - 2, which is complete once the current stream reaches the mark recorded at the start of step k - 1.
            main = torch.cuda.current_stream()
            k = self._pro_calls = getattr(self, "_pro_calls", 0) + 1
            marks = self.__dict__.setdefault("_pro_marks", {})
            if (k - 1) % 2 in marks:
                side.wait_event(marks[(k - 1) % 2])
            nf, ni = ops.prologue_sizes(lens, B, rowset)
            sets = self.__dict__.setdefault("_pro_bufs", {})
            bufs = sets.get(k % 2)
            if bufs is None or bufs[0].numel() < nf or bufs[1].numel() < ni or bufs[0].device != dev:
                bufs = sets[k % 2] = (torch.empty(nf, device=dev, dtype=torch.float32), torch.empty(ni, device=dev, dtype=torch.int32))
                side.wait_stream(main)                                   # (new buffers: ordered behind everything, once per shape)
            with torch.cuda.stream(side):
                pro = ops.prologue(segs, lens, B, labels, cfg.vocab_size, dev, bufs=bufs, rowset=rowset)
                nseq = pro.nseq
                host = torch.empty(nseq + 3, dtype=torch.int32, pin_memory=True)
                host.copy_(pro.words, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            mark = torch.cuda.Event()
            mark.record(main)
            marks[k % 2] = mark
            main.wait_event(ev)
        key_bias = pro.key_bias                                       # per-sequence padded layout, -1e30 = "no such key"
        self._last_valid_dev = pro.valid
        rows = (pro.idx, host[nseq:], ev) if want_rows else None
        # padded pair rows are masked-out keys at the tail of every joint sequence: the attention kernels skip them (exact)
        kv_len = pro.kv_len if getattr(self, "skip_masked_keys", True) else None
        # inference (no dropout, no autograd): the masked-out rows of a sequence are identical in every layer -> one stands for all
        infer = (not self.training) and (not torch.is_grad_enabled()) and getattr(self, "dedupe_masked_rows", True)
        self.last_backward_row_fraction = 1.0                            # (bookkeeping for bench.py: share of rows backward visits)
        pending = None
        if kv_len is not None:
            if infer:
                pending = self._request_lengths(plan, kv_len, labels, True, (pair_info, B, T))
            elif labels is not None and torch.is_grad_enabled() and getattr(self, "skip_padded_backward", True):
                pending = (host[:nseq], None, ev)                     # the prologue's valid[]: unmasked length, extended to the last label
        # ---- embeddings + encoder: one autograd node (_TrunkFn)
        if packed is not None:
            ids, tts = packed
        else:
            ids = torch.cat([p["ids"].reshape(-1).long() for p in passes])
            tts = torch.cat([(p["tt"].reshape(-1).long() if p.get("tt") is not None else torch.zeros(B * T, dtype=torch.long, device=dev)) for p in passes])
        if self.embed_ids_hook is not None and torch.is_grad_enabled():
            # data parallel: the ranks agree on the union of touched embedding rows now, on the input stream when the inputs are
            # complete (async_prologue) -- else on the current stream, where the prologue's host wait happens anyway
            if packed is not None or side is None:
                early_ids = ids                                    # (packed on the input stream when the prologue runs there)
            else:
                with torch.cuda.stream(side):
                    early_ids = torch.cat([p["ids"].reshape(-1).long() for p in passes])
            self.embed_ids_hook(early_ids, side)
        p_emb = cfg.hidden_dropout_prob if self.training else 0.0
        p_joint = je.dropout_prob if (self.training and je.training) else 0.0
        t = _Trunk()
        t.compact = None
        t.ids, t.tts, t.B, t.T, t.lens, t.pair_info, t.plan = ids, tts, B, T, lens, pair_info, plan
        t.feats = [None if info is None else tuple(_pair_features(f, dev) for f in info[0]) for info in pair_info]
        t.feat_versions = [None if fs is None else tuple(f._version for f in fs) for fs in t.feats]
        # round 6: the pair projections of the joint passes (relu(W.pair + b) into the pair rows of the joint pre-LayerNorm matrix: 17 + 28 us
        # at the headline shape) need nothing of this step but the features: queued NOW, one side stream per joint pass, beside the packing
        # and embedding launches that follow on the current stream (36 us of small dependent launches); _TrunkFn.forward joins in front
        # of the joint LayerNorm
        t.J_pre = None
        if getattr(self, "pairs_side_stream", True) and dev.type == "cuda" and any(info is not None for info in pair_info):
            w_ = self._w
            jtot = sum(B * lens[k] for k in range(len(lens)) if pair_info[k] is not None)
            J = torch.empty((jtot, cfg.hidden_size), device=dev, dtype=torch.bfloat16)
            evs, lo = [], 0
            main = torch.cuda.current_stream()
            for n, k in enumerate(k for k in range(len(lens)) if pair_info[k] is not None):
                S = lens[k]
                s_ = ops.side_stream("pairs%d" % n, dev)
                s_.wait_stream(main)
                with torch.cuda.stream(s_):
                    off = T
                    for feat, which in zip(t.feats[k], pair_info[k][1]):
                        ops.pair_proj_fwd(feat, w_[which + "_w"], w_[which + "_b"], J[lo:lo + B * S], T, seq_len=S, offset=off)
                        off += feat.shape[1]
                    evs.append(s_.record_event())
                lo += B * S
            t.J_pre = (J, evs)
        t.key_bias, t.kv_len, t.seed, t.infer = key_bias, kv_len, seed, infer
        t.d_emb = ops.make_drop(p_emb, seed, 1000)
        t.d_joint = [ops.make_drop(p_joint, seed, 1001 + k) for k in range(len(lens))]
        # rows of the top layer's output that can have a gradient (MLM-labelled rows + the [CLS] rows the heads read): known
        # when the caller is forward() / forward_fused() -- only they guarantee that nothing else is differentiated
        t.top_rows = (rows, plan["first"]) if (rows is not None and getattr(self, "sparse_top_layer_backward", True)) else None
        # the caller does not want the prediction scores (trainer.py never reads them): rows that nothing else reads are left out
        drop = (not infer) and labels is not None and not getattr(self, "return_scores", True)
        # The valid-first packing needs the per-sequence lengths on the host.  With the prologue on the input stream (async_prologue)
        # they are there already: the packing is built FIRST and the embedding kernels store their rows in packed order.  With the
        # prologue on the compute stream the host would wait for the GPU to drain: the embedding kernels are queued first (they keep
        # the GPU busy while the lengths travel and the host packs the layout), write the caller's order, and one gather packs.
        rank = pro.rank
        # (the device-built packing of the default training step needs no host wait at all: it always comes first)
        on_device = pending is not None and not infer and not drop and getattr(self, "device_split_layout", True) and len(lens) * B <= 1024
        t.late_split = side is None and not on_device
        t.split = None
        get_split = lambda: self._split_layout(plan, kv_len, pending, infer, drop, rank)
        if not t.late_split:
            t.split = get_split()
            get_split = None
        t.layout = get_split                                      # (late: _TrunkFn calls it once the embedding kernels are queued)
        y = _TrunkFn.apply(bert.embeddings.LayerNorm.weight, self, t)
        split = t.split
        self._last_trunk = t                                       # (forward() hands it to the MLM head: compact output gradient)
        if self.debug_hidden is not None:
            packed = self.debug_hidden.pop("_layers_packed", [])
            if split is not None:                   # back to the original row order (left-out rows of the drop form read as zeros)
                pad = getattr(split, "dropped", False)
                packed = [None if q is None else (torch.cat((q, q.new_zeros((1, q.shape[1])))) if pad else q).index_select(0, split.inv) for q in packed]
            self.debug_hidden["layers"] = [y.detach() if q is None else q for q in packed]
        return y, plan, lens, rows

    def _request_lengths(self, plan, kv_len, labels, infer=False, pairs=None):
        """Inference only (training takes the prologue's ``valid`` words): starts the device -> host copy of the per-sequence count
        of leading rows that keep a row of their own, plus one flag word.  Rows may share one representative only if their INPUTS
        are equal: that holds for masked-out PAIR rows whose features are all zero (no position embedding on pair rows) -- not for
        [PAD] text rows (positions differ), so text-only sequences keep every row and joint sequences keep at least their T text
        rows; the flag counts masked-out pair rows with a non-zero feature (they switch the short cut off)."""
        assert infer
        pair_info, B, T = pairs
        keep, bad = [], torch.zeros((), dtype=torch.int64, device=kv_len.device)
        for k, info in enumerate(pair_info):
            kv = kv_len[k * B:(k + 1) * B]
            if info is None:
                keep.append(torch.full_like(kv, T))
                continue
            kv = kv.clamp(min=T)
            keep.append(kv)
            off = T
            for f in info[0]:
                pos = off + torch.arange(f.shape[1], device=kv.device)[None, :]
                bad = bad + ((pos >= kv[:, None].long()) & (f.to(kv.device) != 0).any(-1)).sum()
                off += f.shape[1]
        kv_len = torch.cat(keep)
        host = torch.empty(kv_len.numel() + 1, dtype=torch.int32, pin_memory=True)
        host.copy_(torch.cat((kv_len, bad.to(torch.int32).view(1))), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host[:-1], host[-1:], ev

    def _split_layout(self, plan, kv_len, pending, infer=False, drop=False, rank=None):
        """Backward on the unmasked rows only.  A row behind its sequence's last unmasked key (a padded pair row; a [PAD] row of
        the text pass) that carries no MLM label has an exactly-zero gradient at the encoder output (the heads read [CLS] rows,
        the MLM loss ignores it), nothing flows into it through attention (as a key its probability is exactly 0, so dK = dV = 0;
        as a query dO = 0 gives dS = 0), and every other operator is row-wise -- by induction its gradient is zero in every layer
        and it adds nothing to any weight gradient.  Forward still computes those rows (the reference returns their prediction
        scores); they are packed BEHIND all other rows (ops.SplitLayout) so that backward is the same kernels on a shorter
        matrix.  Needs the lengths on the host: the one blocking wait of the step, on a copy requested before the embedding
        kernels were queued (_request_lengths).  A labelled row behind the last unmasked key keeps its sequence's rows up to it in
        the leading region (per sequence, not all-or-nothing).  Returns None when it does not apply (see there; also: under 3 %
        to save)."""
        if pending is None:
            return None
        lay = plan["layout"]
        valid_host, flag, ev = pending
        if not infer and not drop and getattr(self, "device_split_layout", True) and len(lay.lens) <= 1024:
            # the training step that returns its scores: every row is kept, so nothing about the packing has to be known on the HOST
            # before backward -- the maps and tile lists are built by two kernels from the prologue's device-side counts, the forward
            # pass has no host round trip (round 2 / early round 3: numpy on the prologue's words; async_prologue hid the wait)
            lay2 = ops.DeviceSplitLayout(lay, self._last_valid_dev, kv_len.device, rank=rank, words=(valid_host, ev))
            return lay2
        ev.synchronize()
        if flag is not None and int(flag[0]) != 0:
            return None
        valid = valid_host.numpy().copy()
        if rank is None and int(valid.sum()) > 0.97 * lay.tokens:          # (row-set mode: the key bias is in the packed order already)
            return None
        if infer:
            return ops.SplitLayout(lay, valid, kv_len.device, dedupe=True)
        self.last_backward_row_fraction = float(valid.sum()) / lay.tokens
        lay2 = ops.SplitLayout(lay, valid, kv_len.device, drop=drop, rank=rank)
        lay2.dropped = drop
        return lay2

class MMBertModel(_GpuModelBase):
    """REF:MMBertForPretraining.py:13-285.  Holds embeddings / encoder / pooler (+ jointEmbeddings)."""

    def __init__(self, config, _owner=None):
        super().__init__()
        self.config = config
        holder = _make_bert_holder(config)
        self.embeddings, self.encoder, self.pooler = holder.embeddings, holder.encoder, holder.pooler
        self._init_runtime()
        self._owner = _owner
        _hf_init(self, config.initializer_range)

    def set_joint_embeddings(self, dataset):
        self.dataset = dataset
        self.jointEmbeddings = JointEmbeddings(self.config.hidden_size, 0.5, dataset)      # REF:MMBertForPretraining.py:26
        dev = self.embeddings.word_embeddings.weight.device
        self.jointEmbeddings.to(dev)
        top = self._owner() if self._owner is not None else None
        if top is not None:
            import weakref
            self.jointEmbeddings._owner = weakref.ref(top)
            top._flat = None

    def _standalone_top(self, device):
        """``MMBertModel(config)`` constructed on its own (REF:MMBertForPretraining.py:13-22 allows it): the kernels read parameters
        from ONE flat storage that MMBertForPretraining lays out, so a standalone model adopts a private MMBertForPretraining around
        itself on first use (its heads are never run; the tied decoder aliases the word embedding, so they cost ~3 H^2 floats).  This
        module's parameters, ``state_dict()`` keys (no ``bert.`` prefix) and gradients stay its own."""
        import weakref
        top = self.__dict__.get("_private_top")
        if top is None:
            if not hasattr(self, "jointEmbeddings"):
                raise RuntimeError("call .set_joint_embeddings(dataset) first (REF:train.py:72)")
            top = MMBertForPretraining(self.config, _bert=self)
            top.to(device)
            top._seed, top._calls = self._seed, self._calls
            object.__setattr__(self, "_private_top", top)           # not a registered submodule: `top.bert` already is `self`
        self._owner = None                                          # (stays "standalone": a later .to() re-enters here)
        self.jointEmbeddings._owner = weakref.ref(top)
        return top

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        self.embeddings.word_embeddings = value

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, output_attentions=None,
                output_hidden_states=None, joint=False):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is not None:
            # REF:MMBertForPretraining.py:231-235 lets the argument through its checks, but REF :264 then calls ``input_ids.long()``
            # unconditionally: with inputs_embeds alone the REFERENCE raises AttributeError ('NoneType' object has no attribute 'long') --
            # the path does not work there, no caller passes it, and the embedding lookup is fused into the trunk here.  Same outcome
            # (an exception before any arithmetic), with a message that says why.
            raise NotImplementedError("MMBertModel: inputs_embeds without input_ids is not a working path of the reference either "
                                      "(REF:MMBertForPretraining.py:264 calls input_ids.long()); pass input_ids")
        if input_ids is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        top = self._owner() if self._owner is not None else self._standalone_top(
            (input_ids[0] if isinstance(input_ids, (tuple, list)) else input_ids).device)
        if top.training != self.training:
            top.train(self.training)
        if joint:
            text, pair = input_ids
            tmask, pmask = attention_mask
            p = dict(ids=text, tt=None, mask=tmask.to(text.device), pair=pair, pair_mask=pmask.to(text.device))   # token types forced to 0 (:223)
        else:
            text = input_ids
            tmask = attention_mask if attention_mask is not None else torch.ones(text.shape, device=text.device)
            p = dict(ids=text, tt=token_type_ids, mask=tmask)
        y, plan, lens, _ = top._encode([p])
        B = text.shape[0]
        seq = y.view(B, lens[0], -1).float()
        pooled = torch.tanh(F.linear(seq[:, 0], self.pooler.dense.weight, self.pooler.dense.bias))     # HF:457-463
        return (seq, pooled)


class MMBertPreTrainingHeads(nn.Module):
    """REF:MMBertForPretraining.py:287-302 (BertPreTrainingHeads + ``align``)."""

    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        pred = _Box()
        pred.transform = _Box()
        pred.transform.dense = nn.Linear(H, H)
        pred.transform.LayerNorm = nn.LayerNorm(H, eps=config.layer_norm_eps)
        pred.decoder = nn.Linear(H, config.vocab_size, bias=True)
        pred.bias = nn.Parameter(torch.zeros(config.vocab_size))
        pred.decoder.bias = pred.bias
        self.predictions = pred
        self.seq_relationship = nn.Linear(H, 2)
        self.align = nn.Linear(H, 2)
        self._owner = None

    def forward(self, sequence_output, pooled_output, joint=False):
        top = self._owner()
        B, S, H = sequence_output.shape
        y = sequence_output.reshape(B * S, H).to(torch.bfloat16)
        top._ensure_ready(y.device)
        labels = torch.full((B * S,), -100, dtype=torch.long, device=y.device)
        bounds = [0, B * S]
        _, logits, _ = _MLMHeadFn.apply(y, self.predictions.transform.LayerNorm.weight, top, labels, bounds, torch.tensor(bounds, dtype=torch.int32, device=y.device), True)
        scores = logits.view(B, S, -1)[:, :, :top.config.vocab_size]
        if joint:
            return scores, self.align(sequence_output[:, 0])
        return scores, self.seq_relationship(pooled_output)


class _HeadsStepFn(torch.autograd.Function):
    """The heads, one launch per dependency level (csrc/heads_coop.hip, round 6): ``mmbert_heads_step_fwd`` (ONE C call, seven launches) evaluates
    everything downstream of the [CLS] rows -- pooler, align / seq_relationship scores, gates, gated concatenation, classifier1_1 / 1_2, the
    three CPC terms, the losses and the joint loss -- and ``mmbert_heads_step_bwd`` (one call, six launches) its whole hand-derived backward
    (the gradient of the [CLS] rows, every head parameter's gradient accumulated into ``p.grad``, the gradient of the per-pass MLM losses).
    Same arithmetic and interface as ``_HeadsFn`` (19 launches + ~8 ATen launches around them: gather / cast of the rows, zero fills, cat),
    which stays as ``model.coop_heads = False`` and as this path's reference in the tests; no atomics, so no separate deterministic form.
    ``first``: fp32 [3B, H]; or, with ``src = (y bf16 [tokens, H], rows int64 [3B])``, a placeholder whose rows the kernel reads from y."""

    @staticmethod
    def _setup(top, B, H, dev, ap, sent, first=None, src=None):
        """The argument record of a forward pass (everything but the MLM losses) and its output tensors: (a, outs, keep)."""
        a = ops.heads_step_struct()
        a.B, a.H, a.tanh_lo = B, H, 1 if top.num_labels == 1 else 0
        a.alpha, a.beta = float(top.alpha), float(top.beta)
        if src is not None:
            y, rows = src
            assert y.dtype == torch.bfloat16 and y.stride(1) == 1 and rows.dtype == torch.int64 and rows.is_contiguous() and rows.numel() == 3 * B
            a.first, a.y, a.first_rows, a.ldy = None, y.data_ptr(), rows.data_ptr(), y.stride(0)
        else:
            assert first.dtype == torch.float32 and first.is_contiguous()
            a.first = first.data_ptr()
        pool, al, sr, at = top.bert.pooler.dense, top.cls.align, top.cls.seq_relationship, top.attn
        vs3 = (top.vt, top.vv, top.vs)
        c1, c2 = top.classifier1_1, top.classifier1_2
        qs = (top.cpc_zt.net, top.cpc_zv.net, top.cpc_za.net)
        assert c2.weight.shape[0] == 1 and at.weight.is_contiguous()
        a.Wp, a.bp, a.Wal, a.bal, a.Wsr, a.bsr = (t.data_ptr() for t in (pool.weight, pool.bias, al.weight, al.bias, sr.weight, sr.bias))
        a.Wat, a.bat, a.Wc1, a.bc1, a.Wc2, a.bc2 = (t.data_ptr() for t in (at.weight, at.bias, c1.weight, c1.bias, c2.weight, c2.bias))
        for m in range(3):
            a.vw[m], a.vb[m], a.Wq[m], a.bq[m] = vs3[m].weight.data_ptr(), vs3[m].bias.data_ptr(), qs[m].weight.data_ptr(), qs[m].bias.data_ptr()
        if isinstance(ap, tuple):                              # (visual labels [B], speech labels [B]): no concatenation launch
            ap = tuple(t.contiguous() for t in ap)
            assert all(t.dtype == torch.int64 and t.numel() == B for t in ap)
            a.ap, a.ap2 = ap[0].data_ptr(), ap[1].data_ptr()
        else:
            ap = ap.contiguous()
            assert ap.dtype == torch.int64 and ap.numel() == 2 * B
            a.ap = ap.data_ptr()
        sent = sent.contiguous()
        assert sent.dtype == torch.float32 and sent.numel() == B
        f32 = torch.float32
        loss, aux, out5 = torch.empty((), device=dev, dtype=f32), torch.empty(3, device=dev, dtype=f32), torch.empty(5, device=dev, dtype=f32)
        logits, t_rel, rel = torch.empty((B, 1), device=dev, dtype=f32), torch.empty((B, 2), device=dev, dtype=f32), torch.empty((2 * B, 2), device=dev, dtype=f32)
        ws = ops.heads_step_workspace(B, H, dev)
        a.sent = sent.data_ptr()
        a.loss, a.aux, a.out5, a.logits, a.t_rel, a.rel, a.ws = (t.data_ptr() for t in (loss, aux, out5, logits, t_rel, rel, ws))
        a.sync = ops.heads_step_sync(dev).data_ptr()
        return a, (loss, aux, logits, t_rel, rel), (ap, sent, ws, out5)

    @staticmethod
    def prelaunch(top, y, rows, ap, sent):
        """Round 6: forward levels 1 - 6 (everything but the losses) on the heads' side stream, queued BEFORE the MLM head's forward: they
        need the encoder output's [CLS] rows and nothing of the MLM head, whose launches (transform, the vocabulary GEMM, cross-entropy)
        then run beside them instead of in front of them.  ``forward`` picks the record up, joins, and runs level 7."""
        B, H, dev = rows.numel() // 3, y.shape[1], y.device
        a, outs, keep = _HeadsStepFn._setup(top, B, H, dev, ap, sent, src=(y, rows))
        s = ops.side_stream("heads", dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.heads_step_fwd(a, 1, 6)
            ev = s.record_event()
        top.__dict__["_heads_pre"] = (y.data_ptr(), rows.data_ptr(), a, outs, keep, ev)

    @staticmethod
    def forward(ctx, first, top, ap, sent, mlm=None, src=None):
        B, H = first.shape[0] // 3, first.shape[1]
        dev = first.device
        pre = top.__dict__.pop("_heads_pre", None)
        if pre is not None and not (src is not None and pre[0] == src[0].data_ptr() and pre[1] == src[1].data_ptr() and pre[2].B == B):
            pre = None                                          # (a record some other forward pass left behind)
        if pre is not None:
            _, _, a, outs, keep3, ev = pre
            torch.cuda.current_stream().wait_event(ev)
            lo = 7
        else:
            if src is None:
                first = first.contiguous()
            a, outs, keep3 = _HeadsStepFn._setup(top, B, H, dev, ap, sent, first=first, src=src)
            lo = 1
        keep = []
        if mlm is not None:
            mlm = mlm.detach().float().contiguous()
            a.mlm, a.nmlm = mlm.data_ptr(), mlm.numel()
            keep.append(mlm)
        ops.heads_step_fwd(a, lo, 7)
        loss, aux, logits, t_rel, rel = outs
        ctx.top, ctx.a, ctx.B, ctx.H, ctx.keep = top, a, B, H, (keep,) + tuple(keep3)
        ctx.side = src is not None          # (the gradient of the [CLS] rows then goes to _MLMHeadFn.backward and nowhere else: it joins the side stream)
        ctx.save_for_backward(*([] if src is not None else [first]))
        ctx.mark_non_differentiable(aux, logits, t_rel, rel)
        ctx.set_materialize_grads(False)
        return loss, aux, logits, t_rel, rel

    @staticmethod
    def backward(ctx, d, *_unused):
        if d is None:
            return None, None, None, None, None, None
        top, a, B, H = ctx.top, ctx.a, ctx.B, ctx.H
        dev = d.device
        d1 = d.reshape(1).float().contiguous()
        dfirst = torch.empty((3 * B, H), device=dev, dtype=torch.float32)
        dmlm = torch.empty(a.nmlm, device=dev, dtype=torch.float32) if a.nmlm else None
        pool, al, at = top.bert.pooler.dense, top.cls.align, top.attn
        vs3 = (top.vt, top.vv, top.vs)
        c1, c2 = top.classifier1_1, top.classifier1_2
        qs = (top.cpc_zt.net, top.cpc_zv.net, top.cpc_za.net)
        a.dloss, a.dfirst, a.dmlm = d1.data_ptr(), dfirst.data_ptr(), ops._ptr(dmlm)
        a.gWp, a.gbp, a.gWal, a.gbal = (t.grad.data_ptr() for t in (pool.weight, pool.bias, al.weight, al.bias))
        a.gWat, a.gbat, a.gWc1, a.gbc1, a.gWc2, a.gbc2 = (t.grad.data_ptr() for t in (at.weight, at.bias, c1.weight, c1.bias, c2.weight, c2.bias))
        for m in range(3):
            a.gvw[m], a.gvb[m], a.gWq[m], a.gbq[m] = (vs3[m].weight.grad.data_ptr(), vs3[m].bias.grad.data_ptr(), qs[m].weight.grad.data_ptr(),
                                                      qs[m].bias.grad.data_ptr())
        assert at.weight.grad.is_contiguous()
        a.sync = ops.heads_step_sync(dev).data_ptr()
        if ctx.side and dmlm is not None and getattr(top, "heads_side_stream", True):
            # round 6: the six levels on a side stream, beside the MLM head's sparse backward (which needs nothing of the heads' but the
            # gradient of its per-pass losses: one tiny launch here) -- two chains of small dependent launches that leave the chip empty,
            # 82 and 97 us, side by side.  _MLMHeadFn.backward joins before it reads dfirst (_join_heads) -- that is BEFORE the trunk's
            # backward starts, where a data-parallel wrapper's head_grad_hook reads the heads' gradients: the hook needs nothing extra
            ops.heads_step_dmlm(a)
            scratch = torch.empty_like(dmlm)
            a.dmlm = scratch.data_ptr()                            # (level 6 writes the same values: into its own words)
            s = ops.side_stream("heads", dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ops.heads_step_bwd(a)
                # (a list: a second graph's backward may fork before the first one's MLM head has joined -- one in-order stream, so the
                # LAST event covers all of them; every record keeps its operands alive until then)
                top.__dict__.setdefault("_heads_join", []).append((s.record_event(), scratch, d1, ctx.keep))
            a.dmlm = dmlm.data_ptr()
            return dfirst, None, None, None, dmlm, None
        ops.heads_step_bwd(a)
        return dfirst, None, None, None, dmlm, None


class _HeadsFn(torch.autograd.Function):
    """heads_loss = ap_loss + label_loss - beta * nce and the auxiliary outputs from the [CLS] rows, with a hand-written
    backward, entirely in csrc/heads.hip: the dense layers as lists of <= 128-row fp32 products (batch <= 32: the reference's default, REF:train.py:38; mmbert_skinny_mm / _wgrad, one
    launch per dependency level), the gates / CPC / losses and their gradients in between.  Same arithmetic as
    MMBertForPretraining._heads (the eager form, kept as the reference for tests and for configurations this path does not
    cover); 19 launches where the eager form needs ~250 (round 1: ~60, the dense layers through torch.addmm -> hipBLASLt).
    Parameter gradients are accumulated straight into ``p.grad`` (views of the flat gradient buffer); the auxiliary outputs are
    values (non-differentiable)."""

    @staticmethod
    def forward(ctx, first, top, ap, sent, mlm=None):
        """``mlm`` (fp32 vector of the per-pass MLM losses, optional): the returned loss is then the joint loss
        alpha * mean(mlm) + heads_loss (REF :427, :443) -- assembled by the loss kernel, one autograd node instead of seven."""
        B, H = first.shape[0] // 3, first.shape[1]
        pool, al, sr, at = top.bert.pooler.dense, top.cls.align, top.cls.seq_relationship, top.attn
        vs3 = (top.vt, top.vv, top.vs)
        c1, c2 = top.classifier1_1, top.classifier1_2
        qs = (top.cpc_zt.net, top.cpc_zv.net, top.cpc_za.net)
        first = first.contiguous()
        dev = first.device
        # the products are summed into zeroed outputs (fp32 atomics over the split inner dimension): ONE zero fill for all of them
        nl = c2.weight.shape[0]
        sizes = (3 * B * H, 2 * B, 4 * B, 3 * B * H, B * H, B * nl, 3 * B * H)
        buf = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        parts = torch.split(buf, sizes)
        P, t_rel, rel, Apre, T, lo, XP = (parts[0].view(3 * B, H), parts[1].view(B, 2), parts[2].view(2 * B, 2), parts[3].view(3 * B, H),
                                          parts[4].view(B, H), parts[5].view(B, nl), parts[6].view(3, B, H))
        ops.skinny_mm([(P, pool.bias, 0, False, [(first, pool.weight, 0, 0)])])
        ops.heads_tanh_(P)                                                                            # pooled = tanh(pooler(first))  HF:457-463
        ops.skinny_mm([(t_rel, sr.bias, 0, False, [(P[:B], sr.weight, 0, 0)]),                         # computed, never in a loss (:301)
                       (rel, al.bias, 0, False, [(first[B:], al.weight, 0, 0)]),                       # [2B, 2]: visual rows, speech rows (:297)
                       (Apre, at.bias, 0, False, [(P, at.weight[:, :H], 0, 0), (P, at.weight[:, H:], 0, 0)])])   # attn(cat(x, x)) = x (W1 + W2)^T + b
        g, Cc = ops.heads_gate_fwd(P, Apre, [v.weight for v in vs3], [v.bias for v in vs3], B)
        ops.skinny_mm([(T, c1.bias, 0, False, [(Cc, c1.weight, 0, 0)])])                              # :414
        ops.skinny_mm([(lo, c2.bias, 0, False, [(T, c2.weight, 0, 0)])]                               # :415
                      + [(XP[m], qs[m].bias, 0, False, [(T, qs[m].weight, 0, 0)]) for m in range(3)])  # REF:MMBertEmbedding.py:22
        tanh_lo = top.num_labels == 1
        if mlm is not None:
            mlm = mlm.detach().float().contiguous()
        out4, seeds, loss, aux = ops.heads_loss_fwd(P, XP, rel, ap, lo, sent, B, top.beta, tanh_lo, mlm=mlm, alpha=top.alpha)
        ctx.top, ctx.B = top, B
        ctx.nmlm, ctx.alpha = (0 if mlm is None else mlm.numel()), float(top.alpha)
        ctx.save_for_backward(first, P, Apre, g, Cc, T, seeds)
        logits_out = torch.tanh(lo) if tanh_lo else lo
        ctx.mark_non_differentiable(t_rel, rel, logits_out)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(aux)
        return loss, aux, logits_out, t_rel, rel

    @staticmethod
    def backward(ctx, d, *_unused):
        if d is None:
            return None, None, None, None, None
        first, P, Apre, g, Cc, T, seeds = ctx.saved_tensors
        top, B = ctx.top, ctx.B
        H = P.shape[1]
        pool, al, at = top.bert.pooler.dense, top.cls.align, top.attn
        vs3 = (top.vt, top.vv, top.vs)
        c1, c2 = top.classifier1_1, top.classifier1_2
        qs = (top.cpc_zt.net, top.cpc_zv.net, top.cpc_za.net)
        n = 3 * B * H
        dev = first.device
        d1 = d.reshape(1).float().contiguous()
        coef = (ctx.alpha / ctx.nmlm) if ctx.nmlm else 0.0
        # (the chain on a side stream beside the MLM head's sparse backward was measured -0.1 ... -0.3 % in round 4 -- two chains of 5-30-us
        # launches hardly overlap -- and is gone)
        dfirst, dmlm = _HeadsFn._backward_chain(ctx, top, first, P, Apre, g, Cc, T, seeds, d1, ctx.nmlm, coef)
        return dfirst, None, None, None, dmlm

    @staticmethod
    def _backward_chain(ctx, top, first, P, Apre, g, Cc, T, seeds, d1, nmlm, coef):
        B = ctx.B
        H = P.shape[1]
        pool, al, at = top.bert.pooler.dense, top.cls.align, top.attn
        vs3 = (top.vt, top.vv, top.vs)
        c1, c2 = top.classifier1_1, top.classifier1_2
        qs = (top.cpc_zt.net, top.cpc_zv.net, top.cpc_za.net)
        n = 3 * B * H
        # one launch: the seeds scaled by the upstream gradient (out of place: backward may run twice), the zero fill of dT | dC | dfirst
        # (the backward products are summed into them) and the gradient of the per-pass MLM losses
        seeds, zb, dmlm = ops.heads_seed(seeds, d1, 7 * B * H, nmlm, coef)
        dXP, dPc = seeds[:n].view(3, B, H), seeds[n:2 * n].view(3 * B, H)
        dlo, drel = seeds[2 * n:2 * n + B].view(B, 1), seeds[2 * n + B:].view(2 * B, 2)
        dT, dC, dfirst = zb[:B * H].view(B, H), zb[B * H:4 * B * H].view(B, 3 * H), zb[4 * B * H:].view(3 * B, H)
        ops.skinny_mm([(dT, None, 0, False, [(dXP[0], qs[0].weight, 1, 0), (dXP[1], qs[1].weight, 1, 0), (dXP[2], qs[2].weight, 1, 0),
                                            (dlo, c2.weight, 1, 0)])])
        ops.skinny_mm([(dC, None, 0, False, [(dT, c1.weight, 1, 0)])])
        dP, dA, E, dg = ops.heads_gate_bwd(dC, P, Apre, g, [v.weight for v in vs3], dPc, B)
        ops.skinny_mm([(dP, None, 0, True, [(dA, at.weight[:, :H], 1, 0), (dA, at.weight[:, H:], 1, 0)])])
        dpre = ops.heads_tanh_bwd(dP, P)
        ops.skinny_mm([(dfirst, None, 0, False, [(dpre, pool.weight, 1, 0), (drel, al.weight, 1, B)])])
        # every weight / bias gradient of the heads' dense layers in ONE launch
        ops.skinny_wgrad([(dXP[m], T, qs[m].weight.grad, qs[m].bias.grad) for m in range(3)]
                         + [(dlo, T, c2.weight.grad, c2.bias.grad), (dT, Cc, c1.weight.grad, c1.bias.grad),
                            (dA, P, at.weight.grad[:, :H], at.bias.grad), (dA, P, at.weight.grad[:, H:], None),
                            (dpre, first, pool.weight.grad, pool.bias.grad), (drel, first[B:], al.weight.grad, al.bias.grad)])
        E3, dg3 = E.view(3, B, H), dg.view(3, B, 1)
        ops.heads_colsum([(E3[m], vs3[m].weight.grad) for m in range(3)] + [(dg3[m], vs3[m].bias.grad) for m in range(3)])
        return dfirst, dmlm


class MMBertForPretraining(_GpuModelBase):
    """REF:MMBertForPretraining.py:304-449."""

    def __init__(self, config, _bert=None):
        super().__init__()
        import weakref
        self.config = config
        H = config.hidden_size
        # (_bert: a standalone MMBertModel adopting this object as its private owner -- MMBertModel._standalone_top)
        self.bert = _bert if _bert is not None else MMBertModel(config, _owner=weakref.ref(self))
        self.cls = MMBertPreTrainingHeads(config)
        self.cls._owner = weakref.ref(self)
        self.num_labels = 7
        self.classifier1_1 = nn.Linear(H * 3, H)
        self.classifier1_2 = nn.Linear(H, 1) if self.num_labels == 7 else nn.Linear(H, self.num_labels)
        self.attn = nn.Linear(H * 2, H)
        self.relu = nn.ReLU()
        self.vt, self.vs, self.vv = nn.Linear(H, 1), nn.Linear(H, 1), nn.Linear(H, 1)
        self.tanh, self.sigmoid = nn.Tanh(), nn.Sigmoid()
        self.dropout = nn.Dropout(0.38)                              # never used (REF :322)
        self.alpha, self.beta = 1, 1
        # x_size = hidden size: the reference hard-codes 1024 (:327-344) and only runs with bert-large
        self.cpc_zt, self.cpc_zv, self.cpc_za = (CPC(H, H, 1, "Tanh") for _ in range(3))
        self._init_runtime()
        self.return_scores = True
        # dtype of the returned prediction-score tensors (outputs[7], [9], [11]): zero-copy [B, S, vocab] views of the logits the
        # vocabulary GEMM writes.  DEVIATION from the reference (fp32 scores): the default stores bf16, the compute dtype of the path;
        # ``model.scores_dtype = torch.float32`` makes that GEMM store its fp32 accumulators instead (EPI_OUT_F32: +1.1 GB of writes,
        # +1-2 % on the headline step) for consumers that call ``.numpy()`` on the scores (REF:sampling.py-style readers).  Losses
        # and gradients are bit-identical either way (the CE kernels round fp32 logits to bf16 as they load them).  trainer.py
        # never reads the scores (``return_scores = False`` drops them altogether).
        self.scores_dtype = torch.bfloat16
        # heads through _HeadsFn (hand-written backward, csrc/heads.hip); False = the eager autograd form (_heads), in which
        # ap_loss / label_loss / nce and the relationship scores stay differentiable outputs
        self.fused_heads = os.environ.get("MMBERT_FUSED_HEADS", "1") != "0"
        # which fused form: True = one launch per dependency level (csrc/heads_coop.hip, _HeadsStepFn: 7 + 6 launches, up to 128 samples, no
        # atomics), False = the 19-launch form (csrc/heads.hip, _HeadsFn: up to 32 samples); beyond the limit the eager form runs, with a warning
        self.coop_heads = True
        # round 6: without a gradient hook, the weight gradients of the few rows that carry a loss (tied decoder, MLM transform, the sparse top
        # layer's sublayers) wait for the deferred multi-layer call at the end of backward and run on the CUs its last round leaves idle
        # (_late_wgrad); False = launched where they arise, on the serial tail of backward (the round-5 order)
        self.late_wgrads = True
        # ... and that call goes out on a side stream, beside the embedding stage's backward (_flush_late_wgrads); False = on the current stream
        self.wgrad_side_stream = True
        # the level-launch heads' backward on a side stream beside the MLM head's sparse backward (_HeadsStepFn.backward); False = in line
        self.heads_side_stream = True
        # the joint passes' pair projections on side streams beside the packing and embedding launches (_encode); False = in line
        self.pairs_side_stream = True
        _hf_init(self, config.initializer_range, skip=_bert)
        # weight tying (HF:728-731): decoder.weight IS the word embedding, decoder.bias IS predictions.bias
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    # ``model.deterministic = True``: every fp32 sum of the step is formed in an order that does not depend on how workgroups are
    # scheduled (slabs + ordered reduces, per-id run sums in ascending row order (mmbert_id_runs_sum_rows), a single adder per address) instead of fp32 atomics in arrival order:
    # the same seeded step gives bit-identical losses and gradients run to run, and two launch paths of the same function agree more
    # tightly (tests/test_train_gpu.py).  Process-global (the library's mmbert_set_deterministic; MMBERT_DETERMINISTIC=1 sets it at
    # import); costs ~ a dozen small launches per step.
    @property
    def deterministic(self) -> bool:
        return ops.deterministic()

    @deterministic.setter
    def deterministic(self, on: bool):
        ops.set_deterministic(bool(on))

    # ---- construction helpers ------------------------------------------------------------------
    @staticmethod
    def _checkpoint_keys(sd):
        """A HuggingFace BERT checkpoint's keys in this model's names -- what ``from_pretrained`` does on load in the reference
        flow (REF:train.py:70):
        * ``LayerNorm.gamma`` / ``LayerNorm.beta`` -> ``LayerNorm.weight`` / ``LayerNorm.bias``: the published ``bert-base-uncased`` /
          ``bert-large-uncased`` files still carry the TF names (HF ``conversion_mapping.py:1274-1283`` "legacy"; transformers 2.8
          did the same rename in ``modeling_utils.from_pretrained``);
        * a ``BertModel``-only checkpoint (keys ``embeddings.* / encoder.* / pooler.*`` without the ``bert.`` prefix) gets the prefix
          (HF ``base_model_prefix`` handling);
        * the transformers-4.x ``position_ids`` buffer is dropped."""
        out = {}
        for k, v in sd.items():
            if k.endswith("position_ids"):
                continue
            if k.endswith("LayerNorm.gamma"):
                k = k[:-len("gamma")] + "weight"
            elif k.endswith("LayerNorm.beta"):
                k = k[:-len("beta")] + "bias"
            out[k] = v
        if out and not any(k.startswith(("bert.", "cls.")) for k in out) and any(k.startswith(("embeddings.", "encoder.")) for k in out):
            out = {"bert." + k: v for k, v in out.items()}
        return out

    @classmethod
    def from_pretrained(cls, name_or_path, ignore_unexpected=True, **kw):
        """Loads ``config.json`` + ``pytorch_model.bin`` / ``model.safetensors`` from a LOCAL directory (there is no network on the
        target boxes): ``BertForPreTraining`` / ``BertForMaskedLM`` / ``BertModel`` key names, current or legacy (``_checkpoint_keys``).
        LOUD about what did not arrive: a missing ``bert.embeddings.*`` / ``bert.encoder.*`` tensor raises (a silently fresh encoder
        is never what the caller wants); a key of the file that matches nothing in the model is reported in a WARNING and in
        ``model.load_report["unexpected"]``, as HF's ``from_pretrained`` -- what the reference's flow calls, REF:train.py:70 -- does (a
        checkpoint with extra heads or buffers that loaded under the reference loads here too; ``ignore_unexpected=False`` makes it an
        error, round 3's default); heads the file does not have (``cls.*``, ``bert.pooler.*``) and the reference's own additions
        (jointEmbeddings, fusion head, CPC) keep their fresh initialisation, with a warning that names them -- HF's
        "newly initialized" message.  ``model.load_report`` = dict(missing=[...], unexpected=[...])."""
        import warnings
        if not os.path.isdir(name_or_path):
            raise OSError(f"{name_or_path}: from_pretrained needs a local checkpoint directory (no network access)")
        with open(os.path.join(name_or_path, "config.json")) as fh:
            cfg = MMBertConfig(**json.load(fh))
        model = cls(cfg)
        st = os.path.join(name_or_path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(name_or_path, "pytorch_model.bin"), map_location="cpu")
        sd = cls._checkpoint_keys(sd)
        res = model.load_state_dict(sd, strict=False)
        # tied pairs share storage (decoder.weight IS the word embedding, decoder.bias IS predictions.bias): one name of a pair suffices
        alias = {"cls.predictions.decoder.weight": "bert.embeddings.word_embeddings.weight", "cls.predictions.decoder.bias": "cls.predictions.bias"}
        alias.update({v: k for k, v in list(alias.items())})
        # (only names a HuggingFace BERT file can hold count as missing: cls.align, the fusion head, CPC are the reference's additions)
        hf_side = ("bert.embeddings.", "bert.encoder.", "bert.pooler.", "cls.predictions.", "cls.seq_relationship.")
        missing = [k for k in res.missing_keys if k.startswith(hf_side) and alias.get(k) not in sd]
        unexpected = list(res.unexpected_keys)
        model.load_report = dict(missing=missing, unexpected=unexpected)
        core = [k for k in missing if k.startswith(("bert.embeddings.", "bert.encoder."))]
        if core:
            raise ValueError(f"{name_or_path}: the checkpoint has no tensor for {len(core)} encoder parameter(s): {core[:8]}"
                             f"{' ...' if len(core) > 8 else ''} (key names after the legacy renames: see _checkpoint_keys)")
        if unexpected:
            msg = f"{name_or_path}: {len(unexpected)} checkpoint tensor(s) match no parameter of MMBertForPretraining: {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}"
            if not ignore_unexpected:
                raise ValueError(msg + " (ignore_unexpected=False was requested; the default loads the rest and warns)")
            warnings.warn(msg)
        fresh = [k for k in missing if k.startswith(("cls.", "bert.pooler."))]
        if fresh:
            warnings.warn(f"{name_or_path}: newly initialised (not in the checkpoint): {fresh}")
        return model

    def set_alpha_beta(self, alpha, beta):
        self.alpha, self.beta = alpha, beta

    def _apply(self, fn, *a, **k):
        self._flat = None                       # .cuda()/.to() re-creates parameter storage: re-flatten lazily
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("embeddings.position_ids")}   # transformers-4.x buffer
        return super().load_state_dict(sd, strict=strict, **kw)

    # ---- the hot path --------------------------------------------------------------------------
    def get_bert_output(self, input_ids, attention_mask, token_type_ids, joint=False):
        if joint:
            assert isinstance(input_ids, tuple)
        seq, pooled = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, joint=joint)
        return self.cls(seq, pooled, joint), pooled

    def _heads(self, first, ap_v, ap_s, sentiment):
        """Everything downstream of the [CLS] rows (REF:MMBertForPretraining.py:293-301, 399-443): returns
        (ap_loss + label_loss - beta * nce, ap_loss, label_loss, nce, logits_out, t_rel, v_rel, s_rel)."""
        B, H = first.shape[0] // 3, first.shape[1]
        pool = self.bert.pooler.dense
        pooled = torch.tanh(F.linear(first, pool.weight, pool.bias))
        pt = pooled[:B]
        with torch.no_grad():
            t_rel = self.cls.seq_relationship(pt)                                    # computed, never in a loss (:301, App. B-9)
        v_rel = self.cls.align(first[B:2 * B])                                       # :297-298
        s_rel = self.cls.align(first[2 * B:])
        v_ap = F.cross_entropy(v_rel.view(-1, 2), ap_v.view(-1).long())
        s_ap = F.cross_entropy(s_rel.view(-1, 2), ap_s.view(-1).long())

        # The three gates (:407-409, v(relu(attn(cat(x, x))))), the gated concat and the three CPC terms (REF:MMBertEmbedding.py:21-32)
        # are evaluated BATCHED over the modality axis: same arithmetic as the per-modality module calls of the reference,
        # a third of the [B,H]-sized kernel launches (each costs ~3 us of device time whatever its size).
        P3 = pooled.view(3, B, H)                                                     # pt, pv, ps
        a3 = self.relu(self.attn(torch.cat((pooled, pooled), dim=1))).view(3, B, H)
        Vw = torch.cat((self.vt.weight, self.vv.weight, self.vs.weight), dim=0)      # [3, H]
        Vb = torch.cat((self.vt.bias, self.vv.bias, self.vs.bias))                   # [3]
        g3 = (a3 * Vw[:, None, :]).sum(-1) + Vb[:, None]                             # [3, B]
        pooled_cat = (P3 * g3[:, :, None]).transpose(0, 1).reshape(B, 3 * H)
        temp = self.classifier1_1(pooled_cat)
        logits_out = self.classifier1_2(temp)
        Wc = torch.stack((self.cpc_zt.net.weight, self.cpc_zv.net.weight, self.cpc_za.net.weight))   # [3, H, H]
        bc = torch.stack((self.cpc_zt.net.bias, self.cpc_zv.net.bias, self.cpc_za.net.bias))[:, None, :]
        x_pred = torch.baddbmm(bc, temp.expand(3, B, H), Wc.transpose(1, 2))
        x_pred = x_pred / x_pred.norm(dim=2, keepdim=True)
        Xn = P3 / P3.norm(dim=2, keepdim=True)
        pos = torch.sum(Xn * x_pred, dim=-1)
        neg = torch.logsumexp(torch.bmm(Xn, x_pred.transpose(1, 2)), dim=-1)
        nce = -(pos - neg).mean(dim=1).sum()
        ap_loss = (v_ap + s_ap) / 2.0                                                # :428
        label_loss = None
        if sentiment is not None:
            if self.num_labels == 1 or self.num_labels == 7:
                if self.num_labels == 1:
                    logits_out = self.tanh(logits_out)
                label_loss = F.mse_loss(logits_out.view(-1), sentiment.view(-1).float())
            else:
                label_loss = F.cross_entropy(logits_out, sentiment)
                logits_out = torch.argmax(self.sigmoid(logits_out), dim=1)
        heads_loss = ap_loss + label_loss - self.beta * nce
        return heads_loss, ap_loss, label_loss, nce, logits_out, t_rel, v_rel, s_rel

    def _coop_heads_apply(self, sentiment, B) -> bool:
        """Whether the level-launch heads (_HeadsStepFn) will run this step: then the MLM head leaves the [CLS] rows in the encoder output for
        that kernel to read (no gather / cast launch)."""
        return bool(self.fused_heads and getattr(self, "coop_heads", True) and sentiment is not None and self.num_labels in (1, 7) and B <= 128
                    and self.config.hidden_size % 16 == 0
                    and all(q.grad is not None for q in (self.attn.weight, self.vt.weight, self.classifier1_1.weight)))

    def _run_heads(self, first, ap_v, ap_s, sentiment, dev, B, mlm=None):
        """The heads on the [3B, H] [CLS] rows: the fused kernels (csrc/heads.hip) where they apply, else the eager form.
        (A captured hipGraph of the eager [B,H]-sized glue -- forward and backward, ~200 dependent launches -- was built and
        measured in round 1: no gain; the device time of the tiny kernels, not their dispatch, is the cost.)"""
        src = self.__dict__.pop("_heads_src", None)
        grads_ok = all(q.grad is not None for q in (self.attn.weight, self.vt.weight, self.classifier1_1.weight))
        fused = self.fused_heads and sentiment is not None and first.is_cuda and self.num_labels in (1, 7) and grads_ok
        coop = fused and getattr(self, "coop_heads", True) and B <= 128 and first.shape[1] % 16 == 0
        if src is not None and not (coop and src[2] == first.data_ptr()):
            first = src[0].index_select(0, src[1]).float()      # (the rows were left to a path that does not run after all: gather them now)
            src = None
        if fused and not coop and B > 32:
            fused = False
            if not self.__dict__.get("_warned_heads_batch"):
                import warnings
                self.__dict__["_warned_heads_batch"] = True
                warnings.warn(f"msa_amd: per-GPU batch {B} is beyond the fused heads' limit (128 samples on the level-launch path, 32 on "
                              "the 19-launch path): the heads run in eager PyTorch (~250 small launches per step)")
        if fused:
            sent = sentiment.to(dev).view(-1).float()
            if coop:
                ap = (ap_v.to(dev).view(-1).long(), ap_s.to(dev).view(-1).long())
                loss, aux, logits_out, t_rel, rel = _HeadsStepFn.apply(first, self, ap, sent, mlm, None if src is None else (src[0], src[1]))
            else:
                ap = torch.cat((ap_v.to(dev).view(-1), ap_s.to(dev).view(-1))).long()
                loss, aux, logits_out, t_rel, rel = _HeadsFn.apply(first, self, ap, sent, mlm)
            return loss, aux[0], aux[1], aux[2], logits_out, t_rel, rel[:B], rel[B:]
        out = self._heads(first, ap_v.to(dev), ap_s.to(dev), None if sentiment is None else sentiment.to(dev))
        if mlm is not None:                                   # joint loss, eager form   (:427, :443)
            out = (self.alpha * mlm.mean() + out[0],) + tuple(out[1:])
        return out

    def forward(self, input_ids, token_type_ids, attention_mask, masked_labels, ap_label, sentiment):
        """REF:MMBertForPretraining.py:392-449, same arguments and the same 13-tuple + logits.  Deviations, all switchable:
        * outputs[7], [9], [11] (prediction scores) are ``self.scores_dtype`` views of the logits: bf16 by default (the compute dtype);
          ``model.scores_dtype = torch.float32`` gives the reference's fp32 tensors straight from the vocabulary GEMM's accumulators
          (+1-2 % step time); None with ``return_scores = False``;
        * outputs[4], [5], [6] (ap_loss, label_loss, nce) are returned as VALUES by the fused heads path -- the one differentiable
          output is outputs[0], which is what trainer.py differentiates (REF:trainer.py:83); ``model.fused_heads = False``
          (the eager heads) keeps them in the autograd graph like the reference."""
        self.outputs = ()
        text_ids, visual, speech, twv, tws = input_ids
        tt_t = token_type_ids[0]
        am_t, am_v, am_s = attention_mask
        lab_t, lab_v, lab_s = masked_labels
        ap_v, ap_s = ap_label
        dev = text_ids.device
        B, T = text_ids.shape
        passes = [dict(ids=text_ids, tt=tt_t, mask=am_t),
                  dict(ids=twv, tt=None, mask=am_v[0].to(dev), pair=visual, pair_mask=am_v[1].to(dev)),
                  dict(ids=tws, tt=None, mask=am_s[0].to(dev), pair=speech, pair_mask=am_s[1].to(dev))]
        H, V = self.config.hidden_size, self.config.vocab_size
        pk = self._pack_inputs(passes, (lab_t, lab_v, lab_s))
        if pk is not None:
            labels, pk = pk[2], pk[:2]
        else:
            side = self._prologue_stream(text_ids) if lab_t.is_cuda else None
            if side is None:
                labels = torch.cat((lab_t.reshape(-1), lab_v.reshape(-1), lab_s.reshape(-1))).to(device=dev, dtype=torch.long)
            else:                                   # async_prologue: the prologue's inputs must not queue behind the current stream
                with torch.cuda.stream(side):
                    labels = torch.cat((lab_t.reshape(-1), lab_v.reshape(-1), lab_s.reshape(-1))).to(device=dev, dtype=torch.long)
                labels.record_stream(torch.cuda.current_stream())
        if labels.numel() != B * (T + (T + visual.shape[1]) + (T + speech.shape[1])):
            raise ValueError("masked_labels must cover text (+ pair) positions of every pass")
        want_rows = torch.is_grad_enabled() and getattr(self, "sparse_mlm_backward", True) and labels.is_cuda
        y, plan, lens, rows = self._encode(passes, labels, want_rows, packed=pk)
        trunk, self._last_trunk = self._last_trunk, None
        self._heads_read_rows = self._coop_heads_apply(sentiment, B)
        if self._heads_read_rows and getattr(self, "heads_side_stream", True) and y.is_cuda and y.is_contiguous():
            # the heads' forward levels below the losses go out NOW, on the side stream, beside the MLM head's launches (_HeadsStepFn.prelaunch)
            _HeadsStepFn.prelaunch(self, y, plan["first"], (ap_v.to(dev).view(-1).long(), ap_s.to(dev).view(-1).long()),
                                   sentiment.to(dev).view(-1).float())
        # first = [3B, H]: the [CLS] rows of every sequence; joint_loss = alpha * (mlm_t + mlm_v + mlm_s) / 3 + heads_loss  (:427, :443)
        mlm, logits, first = _MLMHeadFn.apply(y, self.cls.predictions.transform.LayerNorm.weight, self, labels, plan["bounds"], plan["bounds_dev"],
                                              self.return_scores, rows, plan["first"], trunk)
        joint_loss, ap_loss, label_loss, nce, logits_out, t_rel, v_rel, s_rel = self._run_heads(first, ap_v, ap_s, sentiment, dev, B, mlm=mlm)
        scores = (None, None, None)
        if logits is not None:
            b = plan["bounds"]
            scores = tuple(logits[b[k]:b[k + 1]].view(B, lens[k], -1)[:, :, :V] for k in range(3))
            if self.scores_dtype != logits.dtype:
                scores = tuple(sc.to(self.scores_dtype) for sc in scores)
        self.outputs = (_scalar_loss(joint_loss), None, None, None, ap_loss, label_loss, nce,
                        scores[0], t_rel, scores[1], v_rel, scores[2], s_rel)
        return self.outputs, logits_out

    def forward_fused(self, input_ids, token_type_ids, attention_mask, masked_labels, ap_label, sentiment):
        """DECLARED EXTENSION, not in the reference (SURVEY S8(d) mode ``fused1050``; BASELINE.json quotes its metric on a "fused
        seq_len~1050" sequence that the reference never builds): text | visual | speech in ONE sequence of T + V + A tokens --
        JointEmbeddings applied to both projected modalities at once (cat -> LayerNorm -> dropout, REF:MMBertEmbedding.py:57-72
        extended by the second ``cat`` operand), ONE encoder pass, and the reference's objective (REF:MMBertForPretraining.py:
        392-449) evaluated with that pass standing in for all three: MLM over the fused sequence, alignment loss of its [CLS]
        row against both pair labels, fusion head / CPC with the one pooled vector in the three modality slots.

        input_ids=(text_ids[B,T], visual[B,V,Dv], speech[B,A,Ds]); token_type_ids: text's [B,T] or None;
        attention_mask=(text_mask[B,T], visual_mask[B,V,Dv], speech_mask[B,A,Ds]); masked_labels [B, T+V+A] (-100 = ignore).
        Returns ((joint_loss, None, None, None, ap_loss, label_loss, nce, scores[B,S,V] | None, rel[B,2]), logits[B,1]).
        Checked against oracle.fused_forward (the same extension of the CPU restatement)."""
        text_ids, visual, speech = input_ids
        am_t, am_v, am_s = attention_mask
        ap_v, ap_s = ap_label
        dev = text_ids.device
        B, T = text_ids.shape
        passes = [dict(ids=text_ids, tt=token_type_ids, mask=am_t, pair=(visual, speech), pair_mask=(am_v, am_s))]
        V = self.config.vocab_size
        pk = self._pack_inputs(passes, (masked_labels,))
        if pk is not None:
            labels, pk = pk[2], pk[:2]
        else:
            side = self._prologue_stream(text_ids) if masked_labels.is_cuda else None
            if side is None:
                labels = masked_labels.reshape(-1).to(device=dev, dtype=torch.long)
            else:                                   # async_prologue: see forward()
                with torch.cuda.stream(side):
                    labels = masked_labels.reshape(-1).to(device=dev, dtype=torch.long)
                labels.record_stream(torch.cuda.current_stream())
        if labels.numel() != B * (T + visual.shape[1] + speech.shape[1]):
            raise ValueError("masked_labels must cover the text and both pair blocks")
        want_rows = torch.is_grad_enabled() and getattr(self, "sparse_mlm_backward", True) and labels.is_cuda
        # the visual padding sits in the MIDDLE of the fused sequence: valid-first packing over the row set, not over a prefix
        y, plan, lens, rows = self._encode(passes, labels, want_rows, rowset=getattr(self, "fused_rowset_packing", True), packed=pk)
        trunk, self._last_trunk = self._last_trunk, None
        self._heads_read_rows = self._coop_heads_apply(sentiment, B)
        mlm, logits, first = _MLMHeadFn.apply(y, self.cls.predictions.transform.LayerNorm.weight, self, labels, plan["bounds"], plan["bounds_dev"],
                                              self.return_scores, rows, plan["first"].repeat(3), trunk)   # the one [CLS] row in the t / v / s slots
        joint_loss, ap_loss, label_loss, nce, logits_out, _t_rel, v_rel, _s_rel = self._run_heads(first, ap_v, ap_s, sentiment, dev, B, mlm=mlm)
        scores = None if logits is None else logits.view(B, lens[0], -1)[:, :, :V]
        if scores is not None and self.scores_dtype != scores.dtype:
            scores = scores.to(self.scores_dtype)
        self.outputs = (_scalar_loss(joint_loss), None, None, None, ap_loss, label_loss, nce, scores, v_rel)
        return self.outputs, logits_out
