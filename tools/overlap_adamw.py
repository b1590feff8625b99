#!/usr/bin/env python3
"""Does the AdamW kernel (HBM-bound, 29 VGPRs, no LDS) hide beside the backward pass's kernels (MFMA-bound, 2 waves of 214-239 VGPRs
per SIMD, so 32-80 registers per SIMD stay free)?  A: a backward-like stream of launches; B: AdamW over 116 M parameters in 12 slices
on a second stream; A || B against A + B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"; M = 13850; H = 768; I = 3072
torch.manual_seed(0)
x = torch.randn(M, H, device=dev).bfloat16(); u = torch.randn(M, I, device=dev).bfloat16(); dz = torch.randn(M, H, device=dev).bfloat16()
W1T = (torch.randn(H, I, device=dev) * 0.03).bfloat16(); W2T = (torch.randn(I, H, device=dev) * 0.03).bfloat16()
WoT = (torch.randn(H, H, device=dev) * 0.03).bfloat16(); WqkvT = (torch.randn(H, 3 * H, device=dev) * 0.03).bfloat16()
dqkv = torch.randn(M, 3 * H, device=dev).bfloat16()
gW1 = torch.zeros(I, H, device=dev); gW2 = torch.zeros(H, I, device=dev)
lens = [50] * 16 + [410] * 32
layout = ops.SeqLayout(lens, 12, dev); Ma = sum(lens)
qkv = torch.randn(Ma, 3 * H, device=dev).bfloat16(); dctx_a = torch.randn(Ma, H, device=dev).bfloat16(); bias = torch.zeros(Ma, device=dev)
drop = ops.make_drop(0.1, 1, 1)
ctx, lse = ops.attn_fwd(qkv, bias, layout, H, drop=drop)

def layer_backward():
    du = ops.gemm_nt(dz, W2T, gelu_bwd_u=u)                      # dgelu: N = 3072, K = 768
    dy1 = ops.gemm_nt(du, W1T, resid=dz)                         # N = 768, K = 3072
    dctx = ops.gemm_nt(dy1, WoT)
    ops.attn_bwd(qkv, ctx, dctx_a, lse, bias, layout, H, drop=drop)
    ops.gemm_nt(dqkv, WqkvT, resid=dy1)                          # N = 768, K = 2304
    ops.gemm_tn_grouped([(du, x, gW1, None), (dz, u, gW2, None)])

N = 116_185_154 // 256 * 256
p = torch.randn(N, device=dev); g = torch.randn(N, device=dev) * 1e-3; m = torch.zeros(N, device=dev); v = torch.zeros(N, device=dev)
ph = torch.empty(N, device=dev, dtype=torch.bfloat16); flags = torch.zeros(N // 256, device=dev, dtype=torch.uint8)
SL = 12
def adamw_slices():
    n = N // SL // 256 * 256
    for i in range(SL):
        s = slice(i * n, (i + 1) * n)
        ops.adamw(p[s], g[s], m[s], v[s], ph[s], flags[i * n // 256:(i + 1) * n // 256], lr=1e-5, step=3, zero_grad=True)

s2 = torch.cuda.Stream()
def timed(fa, fb, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0.record()
        if fb is not None:
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                fb()
        if fa is not None:
            fa()
        if fb is not None:
            torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3
def bwd4():
    for _ in range(4): layer_backward()
for _ in range(2): bwd4(); adamw_slices()
ta = timed(bwd4, None); tb = timed(None, adamw_slices); tab = timed(bwd4, adamw_slices)
print(f"A (4 layers of backward launches) {ta:7.1f} us   B (AdamW, 116 M parameters in {SL} slices) {tb:7.1f} us   A || B {tab:7.1f} us   "
      f"hidden {100 * (ta + tb - tab) / tb:5.1f} % of B")
def only(kind):
    def f():
        for _ in range(8):
            if kind == "nt":
                ops.gemm_nt(dz, W2T, gelu_bwd_u=u); ops.gemm_nt(u, W1T, resid=dz)
            elif kind == "tn":
                ops.gemm_tn_grouped([(u, x, gW1, None), (dz, u, gW2, None)])
            else:
                ops.attn_bwd(qkv, ctx, dctx_a, lse, bias, layout, H, drop=drop)
    return f
for kind in ("nt", "tn", "attn"):
    f = only(kind)
    f(); ta = timed(f, None); tab = timed(f, adamw_slices)
    print(f"  beside {kind:4s} launches only: A {ta:7.1f} us   A || B {tab:7.1f} us   hidden {100 * (ta + tb - tab) / tb:5.1f} % of B")
