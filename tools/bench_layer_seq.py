#!/usr/bin/env python3
"""The encoder's forward GEMM sequence (qkv, Wo, FFN-up+GELU, FFN-down) as the train step issues it -- every layer on its OWN
activation buffers, kernels back to back -- against the same kernels repeated on one set of buffers (what a per-shape micro
benchmark measures).  Separates cache / Infinity-Cache residency effects from everything else: per-kernel HIP-event times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"
M, H, I, L = int(os.environ.get("M", 18400)), 768, 3072, 12
NSETS = int(os.environ.get("SETS", 12))
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).bfloat16()
W = dict(qkv=rb(3 * H, H), o=rb(H, H), w1=rb(I, H), w2=rb(H, I))
bias = dict(qkv=torch.zeros(3 * H, device=dev), o=torch.zeros(H, device=dev), w1=torch.zeros(I, device=dev), w2=torch.zeros(H, device=dev))
sets = [dict(x=rb(M, H), qkv=torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16), ctx=rb(M, H), z1=torch.empty(M, H, device=dev, dtype=torch.bfloat16),
             y1=rb(M, H), u=torch.empty(M, I, device=dev, dtype=torch.bfloat16), g=torch.empty(M, I, device=dev, dtype=torch.bfloat16),
             z2=torch.empty(M, H, device=dev, dtype=torch.bfloat16)) for _ in range(NSETS)]
drop = ops.make_drop(0.1, 7, 1)


def layer(s, ev=None):
    def rec(k):
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((k, e))
    rec("start")
    ops.gemm_nt(s["x"], W["qkv"], bias=bias["qkv"], out=s["qkv"]); rec("qkv")
    ops.gemm_nt(s["ctx"], W["o"], bias=bias["o"], resid=s["x"], drop=drop, out=s["z1"]); rec("o")
    ops.gemm_nt(s["y1"], W["w1"], bias=bias["w1"], gelu=True, aux=s["u"], out=s["g"]); rec("w1")
    ops.gemm_nt(s["g"], W["w2"], bias=bias["w2"], resid=s["y1"], drop=drop, out=s["z2"]); rec("w2")


for mode, pick in (("own buffers per layer", lambda i: sets[i % NSETS]), ("one buffer set", lambda i: sets[0])):
    for i in range(L):
        layer(pick(i))
    torch.cuda.synchronize()
    acc = {}
    for rep in range(5):
        ev = []
        for i in range(L):
            layer(pick(i), ev)
        torch.cuda.synchronize()
        for (k0, e0), (k1, e1) in zip(ev[:-1], ev[1:]):
            if k1 != "start":
                acc.setdefault(k1, []).append(e0.elapsed_time(e1) * 1e3)
    print(f"{mode:24s}: " + "  ".join(f"{k} {sorted(v)[len(v) // 2]:6.1f} us" for k, v in acc.items()), flush=True)
