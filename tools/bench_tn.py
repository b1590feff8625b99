#!/usr/bin/env python3
"""A/B of the grouped weight-gradient GEMM (one encoder layer's four problems) over token-axis split counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
M = int(os.environ.get("M", 18400))
shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
probs = []
for N, K in shapes:
    probs.append(((torch.randn(M, N, device=dev) * 0.1).bfloat16(), torch.randn(M, K, device=dev).bfloat16(),
                  torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
fl = sum(2.0 * M * N * K for N, K in shapes)
splits = [int(x) for x in os.environ.get("SPLITS", "0,1,2,3,4,5").split(",")]
ts = {s: [] for s in splits}
for s in splits:
    lib.mmbert_gemm_tn_force_splits(s)
    for _ in range(2): ops.gemm_tn_grouped(probs)
for r in range(5):
    for s in splits:
        lib.mmbert_gemm_tn_force_splits(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm_tn_grouped(probs)
        e1.record(); torch.cuda.synchronize()
        ts[s].append(e0.elapsed_time(e1) / 5)
for s in splits:
    t = sorted(ts[s])[len(ts[s]) // 2]
    print(f"splits {s}: {t*1e3:7.1f} us  {fl/t/1e9:7.1f} TF (incl. slab reduce)")
lib.mmbert_gemm_tn_force_splits(0)
