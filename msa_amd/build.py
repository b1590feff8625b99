"""Builds msa_amd/libmmbert_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

``python -m msa_amd.build`` or ``msa_amd.build.build()``.  The .so is git-ignored but travels to the
GPU box with the repo snapshot; it is rebuilt only when a source is newer than it.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmbert_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "rowwise.hip", "heads.hip", "heads_coop.hip", "layer.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]
EXTRA = os.environ.get("MMBERT_HIPCC_FLAGS", "").split()


def _check_defines(flags) -> None:
    """The product library knows no -DMMB_* switch at all (MMB_STAMPS belongs to tools/stamp_*.py's own builds, the round-3
    ablation switches to git tag r3-gemm-ablations): a typo must not yield a silently different library."""
    bad = [f for f in flags if f.startswith("-DMMB_")]
    if bad:
        raise RuntimeError(f"msa_amd.build: refusing diagnostic defines in a product build: {bad}")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    _check_defines([*FLAGS, *EXTRA, *os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", "").split()])
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, "_obj", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [HIPCC, *FLAGS, *EXTRA, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
