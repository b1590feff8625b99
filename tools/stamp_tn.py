#!/usr/bin/env python3
"""Diagnostic (stamped build, see tools/stamp_gemm.py --build): cycles per 32-row stage of the grouped TN kernel."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from msa_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_stamp", "libmmbert_hip_stamps.so")
from msa_amd import ops
lib = _lib.load()
lib.mmbert_debug_set_stamps.restype = ctypes.c_int; lib.mmbert_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda"; M = 18400
buf = torch.zeros(8192 * 6, device=dev, dtype=torch.int64)
assert lib.mmbert_debug_set_stamps(buf.data_ptr()) == 0
shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
probs = [((torch.randn(M, N, device=dev) * 0.1).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)) for N, K in shapes]
for sp in (2, 4, 7):
    lib.mmbert_gemm_tn_force_splits(sp)
    for _ in range(2): ops.gemm_tn_grouped(probs)
    buf.zero_(); torch.cuda.synchronize()
    ops.gemm_tn_grouped(probs); torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 6); st = st[st[:, 3] == 1]
    loop, ns, epi, rt0, rt1 = st[:, 0].astype(float), st[:, 1].astype(float), st[:, 2].astype(float), st[:, 4], st[:, 5]
    life_us = (rt1 - rt0) / 100.0
    print(f"splits {sp}: wgs {len(st)} stages/wg {ns.mean():.0f} | loop clk/stage {(loop/ns).mean():.0f} | epilogue clk {epi.mean():.0f} | "
          f"clock {((loop+epi).sum()/life_us.sum()):.0f} MHz | WG life mean {life_us.mean():.1f} max {life_us.max():.1f} us | span {(rt1.max()-rt0.min())/100.0:.1f} us")
