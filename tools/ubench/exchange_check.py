#!/usr/bin/env python3
"""Builds and runs tools/ubench/exchange_check.hip (see there): errors must be 0 / 0."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exchange_check.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exchange_check.hip")])
    if "--build" in sys.argv: sys.exit(0)
lib = ctypes.CDLL(so)
lib.exchange_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for blocks, per in ((256, 4096), (256, 64), (64, 1024)):
    ctr = torch.zeros(144, dtype=torch.int32, device="cuda"); data = torch.zeros(blocks * per, device="cuda"); err = torch.zeros(8, dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = lib.exchange_probe_launch(40, blocks, ctr.data_ptr(), data.data_ptr(), per, err.data_ptr(), torch.cuda.current_stream().cuda_stream); e1.record()
    torch.cuda.synchronize()
    print(f"{blocks} workgroups x {per} floats, 40 rounds: rc {rc}, errors (4-byte agent loads / sc1 16-byte buffer loads / 8-byte agent loads / sc0 sc1 16-byte buffer loads / nontemporal 16-byte loads) {err.tolist()[:5]}, {e0.elapsed_time(e1) * 1e3 / 80:.2f} us per phase")
