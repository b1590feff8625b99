"""``torch.ops.mmbert.*``: the kernel families of include/mmbert_hip.h registered as PyTorch custom operators
(``torch.library``; dispatch key CUDA = the key torch-ROCm uses for HIP devices), with autograd formulas written
on the same operators -- SURVEY.md S8(b), last row / BASELINE north_star ("exposed to Python through PyTorch-ROCm
custom ops").  For users who compose their own modules; ``msa_amd.model`` drives the same C entry points through
``autograd.Function``s that keep more state between forward and backward (row maps, fused dropout, flat gradient
buffers) and does not pay the dispatcher per launch.

    import msa_amd.torch_ops                                   # registers the namespace
    y = torch.ops.mmbert.linear(x, w, b, "gelu")               # bf16 [M,K] x [N,K]^T (+bias, erf-GELU) -> bf16 [M,N]
    o = torch.ops.mmbert.layer_norm(x, g, b, 1e-12)
    c = torch.ops.mmbert.attention(qkv, key_bias, seq_lens, heads, 0.1, seed)
    e = torch.ops.mmbert.embed_ln(ids, token_types, word, type_, pos, g, b, 1e-12, 0.1, seed)          # BertEmbeddings
    j = torch.ops.mmbert.joint_embed(text_emb, pair_feats, W, bias, g, b, 1e-5, 0.5, seed)              # JointEmbeddings
    l = torch.ops.mmbert.mlm_head_ce(logits, labels, vocab)                                             # CE(ignore_index=-100), mean
    torch.ops.mmbert.adamw_multi_tensor(p, g, m, v, p_bf16, flags, lr, b1, b2, eps, wd, step, 1.0, "hf", True)   # in place
    labels = torch.ops.mmbert.mlm_mask_rng(ids, 0.15, seed, [101, 102], 103)                            # ids masked in place

One operator per kernel family of SURVEY.md S8(b): linear (gemm nt/tn + epilogues), layer_norm, attention, embed_ln, joint_embed,
mlm_head_ce, adamw_multi_tensor, mlm_mask_rng; the differentiable ones carry their autograd formulas.

No CPU implementation is registered: calling an operator with CPU tensors raises (there is no fallback path).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops

_lib = torch.library.Library("mmbert", "DEF")
_lib.define("linear(Tensor x, Tensor weight, Tensor? bias=None, str act='none') -> Tensor")
_lib.define("linear_bwd(Tensor dy, Tensor x, Tensor weight, Tensor? pre, bool has_bias) -> (Tensor, Tensor, Tensor)")
_lib.define("linear_pre(Tensor x, Tensor weight, Tensor? bias=None) -> (Tensor, Tensor)")
_lib.define("layer_norm(Tensor x, Tensor gamma, Tensor beta, float eps) -> Tensor")
_lib.define("layer_norm_fwd(Tensor x, Tensor gamma, Tensor beta, float eps) -> (Tensor, Tensor, Tensor)")
_lib.define("layer_norm_bwd(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor gamma) -> (Tensor, Tensor, Tensor)")
_lib.define("attention(Tensor qkv, Tensor key_bias, int[] seq_lens, int heads, float dropout_p=0.0, int seed=0) -> Tensor")
_lib.define("attention_fwd(Tensor qkv, Tensor key_bias, int[] seq_lens, int heads, float dropout_p=0.0, int seed=0) -> (Tensor, Tensor)")
_lib.define("attention_bwd(Tensor dctx, Tensor qkv, Tensor ctx, Tensor lse, Tensor key_bias, int[] seq_lens, int heads, float dropout_p=0.0, int seed=0) -> Tensor")

_layouts = {}


def _layout(seq_lens: List[int], heads: int, device) -> "ops.SeqLayout":
    key = (tuple(seq_lens), heads, str(device))
    lay = _layouts.get(key)
    if lay is None:
        lay = _layouts[key] = ops.SeqLayout(list(seq_lens), heads, device)
    return lay


def _bf(x):
    return x.contiguous() if x.dtype == torch.bfloat16 else x.to(torch.bfloat16).contiguous()


# ---- linear: y = act(x W^T + b) through mmbert_gemm_nt ----------------------------------------------------------------
def _linear(x, weight, bias=None, act="none"):
    if act not in ("none", "gelu"):
        raise ValueError("act must be 'none' or 'gelu'")
    return ops.gemm_nt(_bf(x), _bf(weight), bias=None if bias is None else bias.float(), gelu=(act == "gelu"))


def _linear_pre(x, weight, bias=None):
    """GELU form that also returns the pre-activation (what backward needs): one launch, two outputs."""
    xb = _bf(x)
    pre = torch.empty((xb.shape[0], weight.shape[0]), device=x.device, dtype=torch.bfloat16)
    y = ops.gemm_nt(xb, _bf(weight), bias=None if bias is None else bias.float(), gelu=True, aux=pre)
    return y, pre


def _linear_bwd(dy, x, weight, pre, has_bias):
    """dx = (dy * gelu'(pre)) W  (NT against W^T, GELU' fused in the epilogue when pre is given); dW = dy^T x (TN, fp32);
    db = column sums riding on the TN launch."""
    dyb, xb = _bf(dy), _bf(x)
    N, K = weight.shape
    wt = _bf(weight).t().contiguous()                                 # [K, N]: dgrad as an NT product
    if pre is not None:
        g = ops.gelu_bwd(dyb, pre.contiguous())                       # dy * gelu'(pre), bf16
    else:
        g = dyb
    dx = ops.gemm_nt(g, wt)
    dW = torch.zeros((N, K), device=x.device, dtype=torch.float32)
    db = torch.zeros(N, device=x.device, dtype=torch.float32)
    ops.gemm_tn(g, xb, dW, bias_out=db if has_bias else None)
    return dx, dW, db


# ---- layer norm --------------------------------------------------------------------------------------------------------
def _ln_fwd(x, gamma, beta, eps):
    return ops.ln_fwd(_bf(x), gamma.float().contiguous(), beta.float().contiguous(), eps)


def _ln(x, gamma, beta, eps):
    return _ln_fwd(x, gamma, beta, eps)[0]


def _ln_bwd(dy, x, mean, rstd, gamma):
    H = x.shape[1]
    dg = torch.zeros(H, device=x.device, dtype=torch.float32)
    db = torch.zeros(H, device=x.device, dtype=torch.float32)
    dx = ops.ln_bwd(_bf(dy), _bf(x), mean, rstd, gamma.float().contiguous(), dg, db)
    return dx, dg, db


# ---- attention over a packed variable-length batch ------------------------------------------------------------------------
def _drop(p, seed):
    return ops.make_drop(float(p), int(seed), 77) if p > 0.0 else None


def _attn_fwd(qkv, key_bias, seq_lens, heads, dropout_p=0.0, seed=0):
    lay = _layout(seq_lens, heads, qkv.device)
    return ops.attn_fwd(_bf(qkv), key_bias.float().contiguous(), lay, qkv.shape[1] // 3, drop=_drop(dropout_p, seed))


def _attn(qkv, key_bias, seq_lens, heads, dropout_p=0.0, seed=0):
    return _attn_fwd(qkv, key_bias, seq_lens, heads, dropout_p, seed)[0]


def _attn_bwd(dctx, qkv, ctx, lse, key_bias, seq_lens, heads, dropout_p=0.0, seed=0):
    lay = _layout(seq_lens, heads, qkv.device)
    return ops.attn_bwd(_bf(qkv), ctx, _bf(dctx), lse, key_bias.float().contiguous(), lay, qkv.shape[1] // 3, drop=_drop(dropout_p, seed))


for _name, _fn in (("linear", _linear), ("linear_pre", _linear_pre), ("linear_bwd", _linear_bwd), ("layer_norm", _ln),
                   ("layer_norm_fwd", _ln_fwd), ("layer_norm_bwd", _ln_bwd), ("attention", _attn), ("attention_fwd", _attn_fwd),
                   ("attention_bwd", _attn_bwd)):
    _lib.impl(_name, _fn, "CUDA")


# ---- autograd: the differentiable operators re-dispatch to their *_fwd / *_bwd companions -----------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        if act == "gelu":
            y, pre = torch.ops.mmbert.linear_pre(x, weight, bias)
        else:
            if act != "none":
                raise ValueError("act must be 'none' or 'gelu'")
            y, pre = _linear(x, weight, bias, "none"), None
        ctx.has_bias, ctx.has_pre = bias is not None, pre is not None
        ctx.save_for_backward(x, weight, *((pre,) if pre is not None else ()))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, *rest = ctx.saved_tensors
        dx, dW, db = torch.ops.mmbert.linear_bwd(dy.contiguous(), x, weight, rest[0] if ctx.has_pre else None, ctx.has_bias)
        return dx.to(x.dtype), dW.to(weight.dtype), (db if ctx.has_bias else None), None


class _LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = torch.ops.mmbert.layer_norm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        dx, dg, db = torch.ops.mmbert.layer_norm_bwd(dy.contiguous(), x, mean, rstd, gamma)
        return dx.to(x.dtype), dg.to(gamma.dtype), db.to(gamma.dtype), None


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, key_bias, seq_lens, heads, dropout_p, seed):
        c, lse = torch.ops.mmbert.attention_fwd(qkv, key_bias, seq_lens, heads, dropout_p, seed)
        ctx.args = (list(seq_lens), heads, dropout_p, seed)
        ctx.save_for_backward(qkv, c, lse, key_bias)
        return c

    @staticmethod
    def backward(ctx, dctx):
        qkv, c, lse, key_bias = ctx.saved_tensors
        seq_lens, heads, p, seed = ctx.args
        dqkv = torch.ops.mmbert.attention_bwd(dctx.contiguous(), qkv, c, lse, key_bias, seq_lens, heads, p, seed)
        return dqkv.to(qkv.dtype), None, None, None, None, None


_lib.impl("linear", lambda x, w, b=None, act="none": _LinearFn.apply(x, w, b, act), "AutogradCUDA")
_lib.impl("layer_norm", lambda x, g, b, eps: _LNFn.apply(x, g, b, eps), "AutogradCUDA")
_lib.impl("attention", lambda qkv, kb, lens, heads, p=0.0, seed=0: _AttnFn.apply(qkv, kb, lens, heads, p, seed), "AutogradCUDA")


# ======================================================================================================================
# the remaining kernel families of SURVEY.md S8(b): embed_ln, joint_embed, mlm_head_ce, adamw_multi_tensor, mlm_mask_rng
# ======================================================================================================================
_lib.define("embed_ln(Tensor ids, Tensor? token_types, Tensor word, Tensor type_emb, Tensor pos, Tensor gamma, Tensor beta, float eps, "
            "float dropout_p=0.0, int seed=0) -> Tensor")
_lib.define("embed_ln_fwd(Tensor ids, Tensor? token_types, Tensor word, Tensor type_emb, Tensor pos, Tensor gamma, Tensor beta, float eps, "
            "float dropout_p=0.0, int seed=0) -> (Tensor, Tensor, Tensor, Tensor)")
_lib.define("embed_ln_bwd(Tensor dy, Tensor ids, Tensor? token_types, Tensor e0, Tensor mean, Tensor rstd, Tensor gamma, int vocab, int types, "
            "int positions, float dropout_p=0.0, int seed=0) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("joint_embed(Tensor text_emb, Tensor pair, Tensor weight, Tensor bias, Tensor gamma, Tensor beta, float eps=1e-05, "
            "float dropout_p=0.0, int seed=0) -> Tensor")
_lib.define("joint_embed_fwd(Tensor text_emb, Tensor pair, Tensor weight, Tensor bias, Tensor gamma, Tensor beta, float eps=1e-05, "
            "float dropout_p=0.0, int seed=0) -> (Tensor, Tensor, Tensor, Tensor)")
_lib.define("joint_embed_bwd(Tensor dy, Tensor pair, Tensor j0, Tensor mean, Tensor rstd, Tensor gamma, int text_len, float dropout_p=0.0, "
            "int seed=0) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("mlm_head_ce(Tensor logits, Tensor labels, int vocab) -> Tensor")
_lib.define("mlm_head_ce_fwd(Tensor logits, Tensor labels, int vocab) -> (Tensor, Tensor, Tensor)")
_lib.define("mlm_head_ce_bwd(Tensor dloss, Tensor logits, Tensor labels, int vocab, Tensor inv_count, Tensor row_lse) -> Tensor")
_lib.define("adamw_multi_tensor(Tensor(a!) p, Tensor(b!) g, Tensor(c!) m, Tensor(d!) v, Tensor(e!)? p_bf16, Tensor flags, float lr, float beta1, "
            "float beta2, float eps, float weight_decay, int step, float grad_scale=1.0, str mode='hf', bool zero_grad=True) -> ()")
_lib.define("mlm_mask_rng(Tensor(a!) ids, float p, int seed, int[] special_ids, int mask_id=103) -> Tensor")

_DROP_SITE_EMB, _DROP_SITE_JOINT = 78, 79


def _emb_fwd(ids, token_types, word, type_emb, pos, gamma, beta, eps, dropout_p=0.0, seed=0):
    """BertEmbeddings (HF:53-108): word + type + position -> LayerNorm -> dropout.  ids [B,T] -> (y [B*T,H] bf16, e0, mean, rstd)."""
    B, T = ids.shape
    ids1 = ids.reshape(-1).long().contiguous()
    tt = None if token_types is None else token_types.reshape(-1).long().contiguous()
    e0 = ops.embed_gather(ids1, tt, word.float().contiguous(), type_emb.float().contiguous(), pos.float().contiguous(), T)
    drop = ops.make_drop(float(dropout_p), int(seed), _DROP_SITE_EMB) if dropout_p > 0.0 else None
    y, mean, rstd = ops.ln_fwd(e0, gamma.float().contiguous(), beta.float().contiguous(), eps, drop=drop)
    return y, e0, mean, rstd


def _emb(ids, token_types, word, type_emb, pos, gamma, beta, eps, dropout_p=0.0, seed=0):
    return _emb_fwd(ids, token_types, word, type_emb, pos, gamma, beta, eps, dropout_p, seed)[0]


def _emb_bwd(dy, ids, token_types, e0, mean, rstd, gamma, vocab, types, positions, dropout_p=0.0, seed=0):
    B, T = ids.shape
    H = e0.shape[1]
    dev = e0.device
    dword = torch.zeros((vocab, H), device=dev, dtype=torch.float32)
    dtype_ = torch.zeros((types, H), device=dev, dtype=torch.float32)
    dpos = torch.zeros((positions, H), device=dev, dtype=torch.float32)
    dg, db = torch.zeros(H, device=dev, dtype=torch.float32), torch.zeros(H, device=dev, dtype=torch.float32)
    drop = ops.make_drop(float(dropout_p), int(seed), _DROP_SITE_EMB) if dropout_p > 0.0 else None
    de0 = ops.ln_bwd(_bf(dy), e0, mean, rstd, gamma.float().contiguous(), dg, db, post_drop=drop)
    tt = None if token_types is None else token_types.reshape(-1).long().contiguous()
    ops.embed_scatter(ids.reshape(-1).long().contiguous(), tt, de0, T, dword, dtype_, dpos)
    return dword, dtype_, dpos, dg, db


def _pairf(pair):
    """fp32 or float64 features as they are (the kernels round float64 on load: REF:MMBertEmbedding.py:62,64's ``.float()``)."""
    return (pair if pair.dtype in (torch.float32, torch.float64) else pair.float()).contiguous()


def _joint_fwd(text_emb, pair, weight, bias, gamma, beta, eps=1e-5, dropout_p=0.0, seed=0):
    """JointEmbeddings (REF:MMBertEmbedding.py:57-72): cat(text_emb, relu(W pair + b)) -> LayerNorm -> dropout.
    text_emb [B,T,H], pair [B,P,D] -> (y [B,T+P,H] bf16, j0 (pre-LN) [B*(T+P),H], mean, rstd)."""
    B, T, H = text_emb.shape
    P = pair.shape[1]
    j0 = torch.empty((B * (T + P), H), device=text_emb.device, dtype=torch.bfloat16)
    j0.view(B, T + P, H)[:, :T].copy_(text_emb)
    ops.pair_proj_fwd(_pairf(pair), weight.float().contiguous(), bias.float().contiguous(), j0, T)
    drop = ops.make_drop(float(dropout_p), int(seed), _DROP_SITE_JOINT) if dropout_p > 0.0 else None
    y, mean, rstd = ops.ln_fwd(j0, gamma.float().contiguous(), beta.float().contiguous(), eps, drop=drop)
    return y.view(B, T + P, H), j0, mean, rstd


def _joint(text_emb, pair, weight, bias, gamma, beta, eps=1e-5, dropout_p=0.0, seed=0):
    return _joint_fwd(text_emb, pair, weight, bias, gamma, beta, eps, dropout_p, seed)[0]


def _joint_bwd(dy, pair, j0, mean, rstd, gamma, text_len, dropout_p=0.0, seed=0):
    B, P, D = pair.shape
    H = j0.shape[1]
    dev = j0.device
    dg, db = torch.zeros(H, device=dev, dtype=torch.float32), torch.zeros(H, device=dev, dtype=torch.float32)
    dW, dbias = torch.zeros((H, D), device=dev, dtype=torch.float32), torch.zeros(H, device=dev, dtype=torch.float32)
    drop = ops.make_drop(float(dropout_p), int(seed), _DROP_SITE_JOINT) if dropout_p > 0.0 else None
    dj0 = ops.ln_bwd(_bf(dy.reshape(-1, H)), j0, mean, rstd, gamma.float().contiguous(), dg, db, post_drop=drop)
    ops.pair_proj_bwd(_pairf(pair), j0, dj0, text_len, dW, dbias)
    dtext = dj0.view(B, text_len + P, H)[:, :text_len].contiguous()
    return dtext, dW, dbias, dg, db


def _ce_bounds(M, device):
    return torch.tensor([0, M], dtype=torch.int32, device=device)


def _ce_fwd(logits, labels, vocab):
    """Mean cross-entropy over the rows whose label is a vocabulary index (-100 = ignored): HF:466-496 head output -> loss.
    logits [M, ld >= vocab] bf16 (row stride a multiple of 8), labels int64 [M] -> (loss [], inv_count, row_lse)."""
    lg = _bf(logits)
    loss, inv, lse = ops.ce_fwd(lg, int(vocab), labels.long().contiguous(), _ce_bounds(lg.shape[0], lg.device), 1)
    return loss[0].clone(), inv, lse


def _ce(logits, labels, vocab):
    return _ce_fwd(logits, labels, vocab)[0]


def _ce_bwd(dloss, logits, labels, vocab, inv_count, row_lse):
    lg = _bf(logits)
    dl = torch.empty_like(lg)
    ops.ce_bwd(lg, int(vocab), labels.long().contiguous(), _ce_bounds(lg.shape[0], lg.device), 1, inv_count, dloss.reshape(1).float().contiguous(), row_lse, dl)
    return dl


def _adamw(p, g, m, v, p_bf16, flags, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, mode="hf", zero_grad=True):
    """Flat multi-tensor AdamW (REF:train.py:76-97): ONE launch over contiguous fp32 buffers; flags[i // 256] in {0 no decay, 1 decay,
    2 frozen}; also refreshes the bf16 working copy and zeroes g."""
    ops.adamw(p, g, m, v, p_bf16, flags, lr=lr, beta1=beta1, beta2=beta2, eps=eps, wd=weight_decay, step=step, gscale=grad_scale,
              mode={"hf": 0, "torch": 1}[mode], zero_grad=zero_grad)


def _mlm_mask(ids, p, seed, special_ids, mask_id=103):
    return ops.mlm_mask(ids, float(p), int(seed), special_ids=tuple(special_ids), mask_id=int(mask_id))


for _name, _fn in (("embed_ln", _emb), ("embed_ln_fwd", _emb_fwd), ("embed_ln_bwd", _emb_bwd), ("joint_embed", _joint),
                   ("joint_embed_fwd", _joint_fwd), ("joint_embed_bwd", _joint_bwd), ("mlm_head_ce", _ce), ("mlm_head_ce_fwd", _ce_fwd),
                   ("mlm_head_ce_bwd", _ce_bwd), ("adamw_multi_tensor", _adamw), ("mlm_mask_rng", _mlm_mask)):
    _lib.impl(_name, _fn, "CUDA")


class _EmbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, token_types, word, type_emb, pos, gamma, beta, eps, p, seed):
        y, e0, mean, rstd = torch.ops.mmbert.embed_ln_fwd(ids, token_types, word, type_emb, pos, gamma, beta, eps, p, seed)
        ctx.args = (token_types is not None, word.shape[0], type_emb.shape[0], pos.shape[0], p, seed)
        ctx.save_for_backward(ids, *((token_types,) if token_types is not None else ()), e0, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        has_tt, V, nt, npos, p, seed = ctx.args
        ids, *rest = ctx.saved_tensors
        tt = rest.pop(0) if has_tt else None
        e0, mean, rstd, gamma = rest
        dword, dtype_, dpos, dg, db = torch.ops.mmbert.embed_ln_bwd(dy.contiguous(), ids, tt, e0, mean, rstd, gamma, V, nt, npos, p, seed)
        return None, None, dword, dtype_, dpos, dg, db, None, None, None


class _JointOpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_emb, pair, weight, bias, gamma, beta, eps, p, seed):
        y, j0, mean, rstd = torch.ops.mmbert.joint_embed_fwd(text_emb, pair, weight, bias, gamma, beta, eps, p, seed)
        ctx.args = (text_emb.shape[1], p, seed, text_emb.dtype)
        ctx.save_for_backward(pair, j0, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        T, p, seed, dt = ctx.args
        pair, j0, mean, rstd, gamma = ctx.saved_tensors
        dtext, dW, dbias, dg, db = torch.ops.mmbert.joint_embed_bwd(dy.contiguous(), pair, j0, mean, rstd, gamma, T, p, seed)
        return dtext.to(dt), None, dW, dbias, dg, db, None, None, None


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, vocab):
        loss, inv, lse = torch.ops.mmbert.mlm_head_ce_fwd(logits, labels, vocab)
        ctx.vocab = vocab
        ctx.save_for_backward(logits, labels, inv, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, labels, inv, lse = ctx.saved_tensors
        return torch.ops.mmbert.mlm_head_ce_bwd(dloss, logits, labels, ctx.vocab, inv, lse).to(logits.dtype), None, None


_lib.impl("embed_ln", lambda ids, tt, w, t, p_, g, b, eps, p=0.0, seed=0: _EmbFn.apply(ids, tt, w, t, p_, g, b, eps, p, seed), "AutogradCUDA")
_lib.impl("joint_embed", lambda te, pr, w, b, g, be, eps=1e-5, p=0.0, seed=0: _JointOpFn.apply(te, pr, w, b, g, be, eps, p, seed), "AutogradCUDA")
_lib.impl("mlm_head_ce", lambda lg, lab, vocab: _CEFn.apply(lg, lab, vocab), "AutogradCUDA")
