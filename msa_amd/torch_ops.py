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
