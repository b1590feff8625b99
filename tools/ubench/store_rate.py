#!/usr/bin/env python3
"""Builds and runs tools/ubench/store_rate.hip: time per 1-KiB wave store for 64-B / 128-B / 256-B / 1-KiB row segments."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "store_rate.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "store_rate.hip")])
    if "--build" in sys.argv: sys.exit(0)
lib = ctypes.CDLL(so)
lib.store_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
blocks, iters = 256, 512
for pitch in (1536, 6144):
    region = 64 * 8 * 16 * pitch                       # 64 steps of the tallest pattern
    buf = torch.empty(blocks * region, dtype=torch.uint8, device="cuda")
    clk = torch.zeros(blocks * 8, dtype=torch.int64, device="cuda")
    for pat, name in enumerate(["16 rows x 64 B", "8 rows x 128 B", "4 rows x 256 B", "1 KiB contiguous"]):
        for _ in range(2):
            lib.store_probe_launch(pat, pitch, iters, blocks, buf.data_ptr(), region, clk.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.store_probe_launch(pat, pitch, iters, blocks, buf.data_ptr(), region, clk.data_ptr(), torch.cuda.current_stream().cuda_stream); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        per = clk.float().mean().item() / (iters * 8)     # ticks per wave store per CU (8 waves issue concurrently)
        print(f"pitch {pitch:5d} B  {name:18s}: kernel {us:7.1f} us  = {blocks * 8 * iters * 1024 / us / 1e6:6.2f} TB/s   {per:6.1f} s_memtime ticks per 1-KiB store per CU")
