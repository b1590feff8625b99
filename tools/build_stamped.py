#!/usr/bin/env python3
"""Diagnostic build (hipcc only, no GPU needed): the whole library with -DMMB_STAMPS into tools/_stamp/libmmbert_hip_stamps.so (git-ignored).
The stamp tools (stamp_attn.py) load THAT file; the product build (msa_amd/build.py) refuses every -DMMB_* define."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_stamp")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", "-DMMB_STAMPS"]
os.makedirs(OUT, exist_ok=True)
procs, objs = [], []
for f in ("gemm", "attention", "rowwise", "heads", "heads_coop", "layer"):
    o = os.path.join(OUT, f + "_stamps.o")
    objs.append(o)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, "-c", os.path.join(ROOT, "msa_amd", "csrc", f + ".hip"), "-o", o]))
for p in procs:
    assert p.wait() == 0
lib = os.path.join(OUT, "libmmbert_hip_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
print(lib)
