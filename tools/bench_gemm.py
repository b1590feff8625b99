#!/usr/bin/env python3
"""Per-shape A/B of the GEMM kernels on the GPU (interleaved rounds in ONE process, random data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops, _lib

lib = _lib.load()
dev = "cuda"
M = int(os.environ.get("M", 18400))
shapes = [("qkv", M, 2304, 768, "bias"), ("o", M, 768, 768, "resid"), ("w1", M, 3072, 768, "gelu"), ("w2", M, 768, 3072, "resid"),
          ("dgelu", M, 3072, 768, "gelu_bwd"), ("dy1", M, 768, 3072, "resid0"), ("dx", M, 768, 2304, "resid0"), ("dctx", M, 768, 768, "plain"),
          ("vocab", M, 30592, 768, "bias"), ("dvocab", M, 768, 30592, "plain")]
if os.environ.get("SHAPESET") == "bert-large":            # the reference's default model (REF:train.py:28,32,38): H = 1024, I = 4096, M = 32 * 200
    M = int(os.environ.get("M", 6400))
    shapes = [("qkv", M, 3072, 1024, "bias"), ("o", M, 1024, 1024, "resid"), ("w1", M, 4096, 1024, "gelu"), ("w2", M, 1024, 4096, "resid"),
              ("dgelu", M, 4096, 1024, "gelu_bwd"), ("dy1", M, 1024, 4096, "resid0"), ("dx", M, 1024, 3072, "resid0"), ("dctx", M, 1024, 1024, "plain"),
              ("vocab", M, 30592, 1024, "bias")]
rounds = int(os.environ.get("ROUNDS", 5))
MODES = [int(x) for x in os.environ.get("MODES", "0,256").split(",")]        # mmbert_gemm_nt_force modes
NAMES = {0: "default", 1: "128sq", 8: "8phase", 128: "8ph128", 192: "8ph192", 224: "8ph224", 256: "8ph256"}
only = os.environ.get("SHAPES")
if only:
    shapes = [sh for sh in shapes if sh[0] in only.split(",")]
for name, m, n, k, epi in shapes:
    A = torch.randn(m, k, device=dev).bfloat16(); B = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev); R = torch.randn(m, n, device=dev).bfloat16(); U = torch.randn(m, n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); aux = torch.empty_like(out)
    kw = dict(out=out)
    if epi == "bias": kw.update(bias=bias)
    elif epi == "resid": kw.update(bias=bias, resid=R)
    elif epi == "resid0": kw.update(resid=R)
    elif epi == "gelu": kw.update(bias=bias, gelu=True, aux=aux)
    elif epi == "gelu_bwd": kw.update(gelu_bwd_u=U)
    res = {}
    for mode in MODES:
        lib.mmbert_gemm_nt_force(mode)
        for _ in range(2): ops.gemm_nt(A, B, **kw)
    ts = {m_: [] for m_ in MODES}
    for r in range(rounds):
        for mode in MODES:
            lib.mmbert_gemm_nt_force(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_nt(A, B, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[mode].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    s = f"{name:7s} M={m} N={n} K={k} {epi:9s}"
    for mode in MODES:
        t = sorted(ts[mode])[len(ts[mode]) // 2]
        s += f" | {NAMES[mode]}: {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF"
    print(s, flush=True)
lib.mmbert_gemm_nt_force(0)
