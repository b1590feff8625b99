#!/usr/bin/env python3
"""Same-process A/B of two builds of the library on the attention kernels at the headline shapes.
    LIB_B=tools/_ab/libother.so python tools/ab_attn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops, _lib
libA = _lib.load(); _lib._lib = None; _lib.LIB_PATH = os.path.abspath(os.environ["LIB_B"]); libB = _lib.load()
LIBS = {"A": libA, "B": libB}
dev = "cuda"; heads = 12; H = 768
lens = [50] * 16 + [550] * 32
M = sum(lens)
layout = ops.SeqLayout(lens, heads, dev)
qkv = torch.randn(M, 3 * H, device=dev).bfloat16(); dctx = torch.randn(M, H, device=dev).bfloat16()
bias = torch.zeros(M, device=dev)
drop = ops.make_drop(0.1, 1, 1)
def timeit(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {k: {"fwd": [], "bwd": []} for k in LIBS}
outs = {}
for k, lib in LIBS.items():
    _lib._lib = lib
    for _ in range(2):
        ctx, lse = ops.attn_fwd(qkv, bias, layout, H, drop=drop); dq = ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop)
    outs[k] = (ctx.float().clone(), dq.float().clone())
for r in range(int(os.environ.get("ROUNDS", 8))):
    # (the order alternates: the build timed second in a round reads up to 3 % faster whatever it is)
    for k, lib in (list(LIBS.items()) if r % 2 == 0 else list(LIBS.items())[::-1]):
        _lib._lib = lib
        res[k]["fwd"].append(timeit(lambda: ops.attn_fwd(qkv, bias, layout, H, drop=drop)))
        res[k]["bwd"].append(timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop)))
med = lambda x: sorted(x)[len(x) // 2]
for ph in ("fwd", "bwd"):
    a, b = med(res["A"][ph]), med(res["B"][ph])
    print(f"attention {ph}: A {a:7.1f} us  B {b:7.1f} us  A/B {a / b:5.3f}")
print("max |ctx A-B|", float((outs["A"][0] - outs["B"][0]).abs().max()), " max |dqkv A-B|", float((outs["A"][1] - outs["B"][1]).abs().max()))
