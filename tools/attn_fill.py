#!/usr/bin/env python3
"""How much of the attention kernels' time is work-item quantisation: the same kernels on layouts whose workgroup counts fill the chip's
768 forward slots (3 workgroups of 4 waves per CU) in whole and in fractional rounds.  Prints us and SIMD clocks per wave-tile (a wave's
32 query rows x one 64-key tile) at an assumed 2.1 GHz."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"; heads = 12; H = 768
CASES = {"headline 16x50+32x550": [50] * 16 + [550] * 32, "32x550": [550] * 32, "32x512": [512] * 32, "48x512": [512] * 48, "32x640": [640] * 32,
         "32x384": [384] * 32, "64x512": [512] * 64, "32x576": [576] * 32, "16x512": [512] * 16,
         "1x50": [50], "1x550": [550], "4x550": [550] * 4, "8x550": [550] * 8, "16x550": [550] * 16, "24x550": [550] * 24, "12x550": [550] * 12, "13x550 (780 wgs)": [550] * 13,
         "4x2048 (768 long workgroups)": [2048] * 4, "8x2048": [2048] * 8, "16x1024": [1024] * 16, "128x256": [256] * 128, "256x128": [128] * 256}
def timeit(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): f()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, lens in CASES.items():
    M = sum(lens)
    layout = ops.SeqLayout(lens, heads, dev)
    qkv = torch.randn(M, 3 * H, device=dev).bfloat16(); dctx = torch.randn(M, H, device=dev).bfloat16()
    bias = torch.zeros(M, device=dev)
    drop = ops.make_drop(0.1, 1, 1)
    ctx, lse = ops.attn_fwd(qkv, bias, layout, H, drop=drop)
    wgs = sum((n + 127) // 128 for n in lens) * heads
    wave_tiles = sum(((n + 31) // 32) * ((n + 63) // 64) for n in lens) * heads
    tf = timeit(lambda: ops.attn_fwd(qkv, bias, layout, H, drop=drop))
    tb = timeit(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop))
    print(f"{name:24s} {wgs:5d} workgroups = {wgs / 768:4.2f} rounds of 768  {wave_tiles:6d} wave-tiles   fwd {tf:6.1f} us = {tf * 2100 * 1024 / wave_tiles:5.0f} clk/wave-tile"
          f"   bwd {tb:6.1f} us = {tb * 2100 * 1024 / wave_tiles:5.0f}", flush=True)
