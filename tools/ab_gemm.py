#!/usr/bin/env python3
"""Same-process A/B of two builds of the library on the NT GEMM shapes of the train step.

    LIB_B=tools/_ab/libold.so python tools/ab_gemm.py      # A = msa_amd/libmmbert_hip.so, B = the other build

Interleaved rounds on one box (box-to-box clocks differ by several percent, so only same-process ratios count)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops, _lib

libA = _lib.load()
_lib._lib = None
_lib.LIB_PATH = os.path.abspath(os.environ["LIB_B"])
libB = _lib.load()
LIBS = {"A": libA, "B": libB}
dev = "cuda"
M = int(os.environ.get("M", 18400))
shapes = [("qkv", M, 2304, 768, "bias"), ("o", M, 768, 768, "resid"), ("w1", M, 3072, 768, "gelu"), ("w2", M, 768, 3072, "resid"),
          ("dgelu", M, 3072, 768, "gelu_bwd"), ("dy1", M, 768, 3072, "resid0"), ("dx", M, 768, 2304, "resid0"), ("dctx", M, 768, 768, "plain"),
          ("vocab", M, 30592, 768, "bias")]
rounds = int(os.environ.get("ROUNDS", 6))
tot = {"A": 0.0, "B": 0.0}
for name, m, n, k, epi in shapes:
    A = torch.randn(m, k, device=dev).bfloat16(); B = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev); R = torch.randn(m, n, device=dev).bfloat16(); U = torch.randn(m, n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); aux = torch.empty_like(out)
    kw = dict(out=out)
    if epi == "bias": kw.update(bias=bias)
    elif epi == "resid": kw.update(bias=bias, resid=R, drop=(12345, libA.mmbert_dropout_thr16(0.1), 1.0 / 0.9))    # as in the train step
    elif epi == "resid0": kw.update(resid=R)
    elif epi == "gelu": kw.update(bias=bias, gelu=True, aux=aux)
    elif epi == "gelu_bwd": kw.update(gelu_bwd_u=U)
    outs = {}
    for key, lib in LIBS.items():
        _lib._lib = lib
        for _ in range(2): ops.gemm_nt(A, B, **kw)
        outs[key] = out.float().clone()
    ts = {key: [] for key in LIBS}
    for r in range(rounds):
        for key, lib in (list(LIBS.items()) if r % 2 == 0 else list(LIBS.items())[::-1]):     # alternating order: see ab_attn.py
            _lib._lib = lib
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_nt(A, B, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[key].append(e0.elapsed_time(e1) / 5)
    a, b = sorted(ts["A"])[len(ts["A"]) // 2] * 1e3, sorted(ts["B"])[len(ts["B"]) // 2] * 1e3
    tot["A"] += a; tot["B"] += b
    print(f"{name:6s} N={n:5d} K={k:5d} {epi:8s} A {a:7.1f} us   B {b:7.1f} us   A/B {a / b:5.3f}   max|A-B| {float((outs['A'] - outs['B']).abs().max()):.3g}", flush=True)
print(f"sum: A {tot['A']:.1f} us  B {tot['B']:.1f} us  A/B {tot['A'] / tot['B']:.3f}")
