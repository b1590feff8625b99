#!/usr/bin/env python3
"""Builds and runs tools/ubench/barrier_rate.hip: microseconds per grid barrier (256 x 1024 threads) by coherence method, and whether the
exchanged data arrived (errors must be 0 for a usable method)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "barrier_rate.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "barrier_rate.hip")])
    if "--build" in sys.argv: sys.exit(0)
lib = ctypes.CDLL(so)
lib.barrier_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
names = ["__threadfence by every thread, both sides", "thread 0: release fence / acquire fence (agent)", "no fence, agent-scope atomic store / load of the data", "no fence, plain accesses (floor, not coherent)",
         "two-level arrival (8 group counters), atomic data"]
dirty = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
for blocks in (256, 128):
    for mode, name in enumerate(names):
        for dirty_l2 in (False, True):
            ctr = torch.zeros(16 * 9, dtype=torch.int32, device="cuda"); lines = torch.zeros(blocks * 32, device="cuda"); err = torch.zeros(1, dtype=torch.int32, device="cuda")
            iters = 50
            st = torch.cuda.current_stream().cuda_stream
            lib.barrier_probe_launch(mode, 2, blocks, ctr.data_ptr(), lines.data_ptr(), err.data_ptr(), st); ctr.zero_(); err.zero_()
            if dirty_l2: dirty.add_(1.0)                       # 256 MB of fresh dirty lines in the caches in front of the kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.barrier_probe_launch(mode, iters, blocks, ctr.data_ptr(), lines.data_ptr(), err.data_ptr(), st); e1.record()
            torch.cuda.synchronize()
            print(f"{blocks:4d} workgroups  {name:55s} dirty L2 {int(dirty_l2)}: {e0.elapsed_time(e1) * 1e3 / (2 * iters):7.2f} us per barrier, errors {int(err)}")
