#!/usr/bin/env python3
"""Attention kernel timing at the headline shapes (16x S=50, 32x S=550, 12 heads), random data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"; heads = 12; H = 768
lens = [50] * 16 + [550] * 32
M = sum(lens)
layout = ops.SeqLayout(lens, heads, dev)
qkv = torch.randn(M, 3 * H, device=dev).bfloat16(); dctx = torch.randn(M, H, device=dev).bfloat16()
bias = torch.zeros(M, device=dev)
fl = sum(4.0 * n * n * 64 * heads for n in lens)
for p in (0.0, 0.1):
    drop = ops.make_drop(p, 1, 1)
    for _ in range(3):
        ctx, lse = ops.attn_fwd(qkv, bias, layout, H, drop=drop); dq = ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop)
    def timeit(f, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = timeit(lambda: ops.attn_fwd(qkv, bias, layout, H, drop=drop))
    tb = timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop))
    print(f"p={p}: fwd {tf:7.1f} us ({fl/tf/1e6:6.1f} TF)   bwd {tb:7.1f} us ({2.5*fl/tb/1e6:6.1f} TF algorithmic)", flush=True)
