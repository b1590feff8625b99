#!/usr/bin/env python3
"""Diagnostic: where a wave of the attention forward kernel spends its cycles (s_memtime stamps, -DMMB_STAMPS build of
tools/build_stamped.py).  Per wave and 64-key tile: wait (s_waitcnt vmcnt(0) + tile barrier), qk (staging issue, bias read,
S^T MFMAs, V read issue), softmax (max, exp2, pack, denominators, dropout), pv (LDS wait + O MFMAs).  MFMAs are asynchronous: a
phase's MFMA time shows up where the NEXT dependent instruction waits.  Only shares are read (the stamps fence the schedule)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from msa_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_stamp", "libmmbert_hip_stamps.so")
from msa_amd import ops
lib = _lib.load()
lib.mmbert_debug_set_attn_stamps.restype = ctypes.c_int
lib.mmbert_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda"; heads = 12; H = 768
lens = [50] * 16 + [550] * 32
M = sum(lens)
layout = ops.SeqLayout(lens, heads, dev)
qkv = torch.randn(M, 3 * H, device=dev).bfloat16()
bias = torch.zeros(M, device=dev)
ntiles = layout.nftiles
buf = torch.zeros(ntiles * heads * 4 * 8, device=dev, dtype=torch.int64)
assert lib.mmbert_debug_set_attn_stamps(buf.data_ptr()) == 0
for p in (0.0, 0.1):
    drop = ops.make_drop(p, 1, 1)
    for _ in range(3):
        ops.attn_fwd(qkv, bias, layout, H, drop=drop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attn_fwd(qkv, bias, layout, H, drop=drop); e1.record(); torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(heads, ntiles, 4, 8).astype(np.float64)
    act = st[..., 5] > 1                                   # waves of the S = 550 sequences
    w = st[act]
    nt = w[:, 5]
    per_tile = w[:, :4] / nt[:, None]
    total = w[:, 4]
    ok = st[..., 7] > 0
    span = st[..., 7][ok].max() - st[..., 6][ok].min()
    print("   s_memtime ticks per us of kernel time: %.0f" % (span / (e0.elapsed_time(e1) * 1e3)))
    print(f"p={p}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, stamp span {span:.0f} clk; {int(act.sum())} waves with {nt.mean():.1f} tiles")
    print("   per tile and wave [clk]: wait %.0f  qk %.0f  softmax %.0f  pv %.0f   sum %.0f;  wave lifetime %.0f clk (tiles %.0f %%)" % (
        *per_tile.mean(0), per_tile.sum(1).mean(), total.mean(), 100 * (w[:, :4].sum(1) / total).mean()))
    print("   wait: median %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(per_tile[:, 0], [50, 90, 100])))
