#!/usr/bin/env python3
"""Reference point only: the vendor library (torch.matmul -> hipBLASLt) on the step's GEMM shapes, no epilogue."""
import torch
dev = "cuda"; M = 18400
for name, n, k in (("qkv", 2304, 768), ("o", 768, 768), ("w1", 3072, 768), ("w2", 768, 3072), ("dx", 768, 2304), ("vocab", 30592, 768), ("dvocab", 768, 30592)):
    A = torch.randn(M, k, device=dev).bfloat16(); B = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(A, B.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.matmul(A, B.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{name:7s} M={M} N={n} K={k}: hipBLASLt (plain C = A.B^T) {t*1e3:7.1f} us {2.0*M*n*k/t/1e9:7.1f} TF")
for name, n, k in (("wgrad w1", 3072, 768), ("wgrad w2", 768, 3072)):
    A = torch.randn(M, n, device=dev).bfloat16(); B = torch.randn(M, k, device=dev).bfloat16()
    for _ in range(3): torch.matmul(A.t(), B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.matmul(A.t(), B)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{name:9s} N={n} K={k} over M={M}: hipBLASLt (A^T.B, bf16 out) {t*1e3:7.1f} us {2.0*M*n*k/t/1e9:7.1f} TF")
