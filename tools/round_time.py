#!/usr/bin/env python3
"""Time per ROUND of tiles of the persistent NT GEMM as a function of the number of rounds: the same N, K, tile and epilogue at growing M.
Separates what a launch pays once (dispatch of 256 large-LDS workgroups, cold first touches, the exposed last epilogue, the ragged last
round) from what every round of 224 x 256 tiles costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"
blk = torch.randn(8192, 8192, device=dev).bfloat16()
def timeit(fn, n=20):
    for _ in range(25): fn()                                  # (the first timed configuration of a process reads ~20 us high after only 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4): torch.mm(blk, blk)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, K, tag in ((2304, 768, "qkv + bias"), (3072, 768, "ffn-up + bias + gelu (two outputs)")):
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    print(f"--- N = {N}, K = {K}: {tag}")
    base = None
    for mult in (1, 2, 4, 8):
        M = 18400 * mult
        A = (torch.randn(M, K, device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if N == 3072 else None
        t = timeit(lambda: ops.gemm_nt(A, B, out=out, bias=bias, gelu=N == 3072, aux=aux))
        tiles = ((M + 223) // 224) * (N // 256)
        rounds = tiles / 256
        if base is None: base = (t, rounds)
        marg = (t - base[0]) / (rounds - base[1]) if rounds > base[1] else float("nan")
        print(f"M = {M:6d}: {tiles:5d} tiles = {rounds:5.2f} rounds  {t:7.1f} us  = {t / rounds:5.1f} us per round; marginal over the 18 400-row launch {marg:5.1f} us per round;  {2.0 * M * N * K / t / 1e6:6.0f} TF/s")
        del A, out, aux
