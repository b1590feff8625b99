#!/usr/bin/env python3
"""Diagnostic (stamped build: python tools/stamp_gemm.py --build first): the weight-gradient kernel with 216 tiles (two layers: what the
step launches) against exactly 256 tiles (the same plus a 40-tile problem) -- cycles per 32-token stage and the clock each workgroup saw.
Answers: when the 40 idle CUs are filled, do the tiles get slower in CYCLES (memory / fabric contention) or in CLOCK (power)?"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from msa_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_stamp", "libmmbert_hip_stamps.so")
from msa_amd import ops
lib = _lib.load()
lib.mmbert_debug_set_stamps.restype = ctypes.c_int; lib.mmbert_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda"; M = 13850
buf = torch.zeros(8192 * 6, device=dev, dtype=torch.int64)
assert lib.mmbert_debug_set_stamps(buf.data_ptr()) == 0
def mk(shapes):
    return [((torch.randn(M, N, device=dev) * 0.1).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)) for N, K in shapes]
layer = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
cases = {"216 tiles (2 layers)": mk(layer * 2), "256 tiles (2 layers + 40)": mk(layer * 2 + [(2560, 1024)]), "128 tiles": mk(layer + [(1280, 1024)])}
for rep in range(2):
    for name, probs in cases.items():
        for _ in range(3): ops.gemm_tn_grouped(probs)
        buf.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_tn_grouped(probs); e1.record(); torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(-1, 6); st = st[st[:, 3] == 1]
        loop, ns, epi, rt0, rt1 = st[:, 0].astype(float), st[:, 1].astype(float), st[:, 2].astype(float), st[:, 4], st[:, 5]
        life_us = (rt1 - rt0) / 100.0
        fl = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in probs)
        print(f"{name:28s} wgs {len(st):4d} | loop clk/stage {(loop/ns).mean():7.0f} | clock {((loop+epi).sum()/life_us.sum()):5.0f} MHz | WG life mean {life_us.mean():6.1f} max {life_us.max():6.1f} us | "
              f"launch {e0.elapsed_time(e1)*1e3:6.1f} us = {fl/e0.elapsed_time(e1)/1e9:5.0f} TF/s (stamped build)", flush=True)
