#!/usr/bin/env python3
"""Same-process A/B of the weight-gradient kernel's two K loops (mmbert_gemm_tn_force_form: 0 = 4-slot ring of 32-token stages, 1 = 8-phase)
at the train step's launch shapes: two encoder layers per launch (216 tiles), one layer (108 tiles, token axis split), eleven layers
(whole rounds of 256 tiles).  Cold operands: a ring of buffer sets larger than the Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
M = int(os.environ.get("M", 13850))
HID, FF = int(os.environ.get("HID", 768)), int(os.environ.get("FF", 3072))
layer = [(FF, HID), (HID, FF), (3 * HID, HID), (HID, HID)]


def probs(nl, sets):
    out = []
    for _ in range(sets):
        one = []
        for l in range(nl):
            for N, K in layer:
                one.append((torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
        out.append(one)
    return out


def timed(ps, reps=6):
    for p in ps[:2]: ops.gemm_tn_grouped(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): ops.gemm_tn_grouped(ps[r % len(ps)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


fl = lambda nl: nl * sum(2.0 * M * N * K for N, K in layer)
cases = [("2 layers (216 tiles)", 2, 3), ("1 layer (split)", 1, 6), ("11 layers (rounds)", 11, 1)]
if HID != 768: cases = cases[:2]
for name, nl, sets in cases:
    ps = probs(nl, sets)
    res = {0: [], 1: []}
    for rnd in range(5):
        for form in (0, 1):
            lib.mmbert_gemm_tn_force_form(form)
            res[form].append(timed(ps))
    t0, t1 = sorted(res[0])[2], sorted(res[1])[2]
    print(f"M {M} {name:24s} ring {t0*1e3:7.1f} us {fl(nl)/t0/1e9:6.0f} TF/s | 8-phase {t1*1e3:7.1f} us {fl(nl)/t1/1e9:6.0f} TF/s | ratio {t1/t0:.3f}", flush=True)
    del ps
lib.mmbert_gemm_tn_force_form(1)
