#!/usr/bin/env python3
"""the guide's 256 x 256 8-phase GEMM structure (tools/yardstick/gemm_256sq_8phase.hip, never linked into the
product) against the product's mmbert_gemm_nt on the headline shapes -- random data, COLD operands (a ring of buffer sets larger
than the 256-MiB Infinity Cache), interleaved rounds in one process, plain epilogue on both sides (the product's fused epilogues are
timed beside it as a third column).

    python tools/yardstick/run_yardstick.py --build     # here (hipcc only)
    python tools/yardstick/run_yardstick.py             # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, "libyardstick.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
                           "-shared", "-o", LIB, os.path.join(HERE, "gemm_256sq_8phase.hip")])
    print("built", LIB)
    sys.exit(0)

import torch
from msa_amd import ops, _lib

y = ctypes.CDLL(LIB)
y.yardstick_gemm_nt.restype = ctypes.c_int
y.yardstick_gemm_nt.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4
lib = _lib.load()
dev = "cuda"
stream = lambda: torch.cuda.current_stream().cuda_stream


def yard(A, B, C):
    rc = y.yardstick_gemm_nt(stream(), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), A.shape[0], B.shape[0], A.shape[1])
    assert rc == 0, rc


SHAPES = [("qkv", 18400, 2304, 768, "bias"), ("out-proj", 18400, 768, 768, "resid"), ("ffn-up", 18400, 3072, 768, "gelu"), ("ffn-down", 18400, 768, 3072, "resid"),
          ("vocab", 18400, 30592, 768, "bias"),
          ("dgrad qkv", 13850, 768, 2304, "resid0"), ("dgrad out", 13850, 768, 768, "plain"), ("dgelu", 13850, 3072, 768, "gelu_bwd"), ("dgrad up", 13850, 768, 3072, "resid0"),
          ("cube 4096", 4096, 4096, 4096, "plain"), ("cube 8192", 8192, 8192, 8192, "plain")]
only = os.environ.get("SHAPES")
rounds, reps = int(os.environ.get("ROUNDS", 7)), int(os.environ.get("REPS", 16))
print(f"# {torch.cuda.get_device_name(0)}; interleaved rounds={rounds} x {reps} launches each, every launch on its own operand set (cold); median per launch")
print(f"# {'shape':12s} {'M':>6s} {'N':>6s} {'K':>5s} | yardstick 256x256 8-phase | product, plain epilogue | product, its fused epilogue | yardstick/product(plain)")
for name, M, N, K, epi in SHAPES:
    if only and name not in only.split(","):
        continue
    per_set = 2 * (M * K + N * K + M * N)
    nsets = max(2, min(16, int(600e6 // per_set) + 1))                 # >= 600 MB of operands in rotation (Infinity Cache: 256 MiB)
    g = torch.Generator(device=dev).manual_seed(1)
    sets = []
    for i in range(nsets):
        A = torch.empty(M, K, device=dev, dtype=torch.bfloat16).uniform_(-1, 1, generator=g)
        B = torch.empty(N, K, device=dev, dtype=torch.bfloat16).uniform_(-1, 1, generator=g)
        sets.append((A, B, torch.empty(M, N, device=dev, dtype=torch.bfloat16)))
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).bfloat16()
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "gelu" else None
    kw = {"bias": dict(bias=bias), "resid": dict(bias=bias, resid=R, drop=ops.make_drop(0.1, 7, 3)), "resid0": dict(resid=R), "gelu": dict(bias=bias, gelu=True, aux=aux),
          "gelu_bwd": dict(gelu_bwd_u=R), "plain": {}}[epi]
    # correctness of the yardstick on this shape (fp32 reference on a row / column sample)
    A, B, C = sets[0]
    C.fill_(7.0)
    yard(A, B, C)
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (64,), device=dev)
    rows[:4] = torch.tensor([0, 1, M - 2, M - 1], device=dev)
    ref = A[rows].float() @ B.float().t()
    err = float((C[rows].float() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, (name, err)
    C2 = ops.gemm_nt(A, B)
    assert float((C2[rows].float() - ref).abs().max() / ref.abs().max()) < 1e-2
    fns = {"yard": lambda s: yard(*s), "plain": lambda s: ops.gemm_nt(s[0], s[1], out=s[2]), "fused": lambda s: ops.gemm_nt(s[0], s[1], out=s[2], **kw)}
    if epi == "resid":
        fns["fused"] = lambda s: ops.gemm_nt(s[0], s[1], out=s[2], bias=bias, resid=R, drop=kw["drop"])
    for f in fns.values():
        for i in range(4):
            f(sets[i % nsets])
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for r in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                f(sets[(i + r) % nsets])
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / reps)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    d = ops.gemm_nt_describe(M, N, K)
    print(f"{name:14s} {M:6d} {N:6d} {K:5d} | {med['yard'] * 1e3:8.1f} us {fl / med['yard'] / 1e9:7.1f} TF | {med['plain'] * 1e3:8.1f} us {fl / med['plain'] / 1e9:7.1f} TF | "
          f"{med['fused'] * 1e3:8.1f} us ({epi}) | x{med['yard'] / med['plain']:.3f}   [{d['kernel']} {d['tile']} rounds {d['rounds']}, yardstick rounds {((M + 255) // 256) * ((N + 255) // 256) / d['cus']:.2f}; yardstick err {err:.1e}]",
          flush=True)
    del sets
    torch.cuda.empty_cache()
