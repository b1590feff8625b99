#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the NT ring kernel spends its cycles (s_memtime stamps, -DMMB_STAMPS build).

    python tools/stamp_gemm.py --build      # here (hipcc only): builds tools/_stamp/libmmbert_hip_stamps.so
    python tools/stamp_gemm.py              # on the GPU box: per-shape shares of prologue / K loop / epilogue

The stamped build is never the product library and its run time is not a benchmark (the stamps fence the
schedule); only the SHARES are read."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_stamp")
VARIANT = os.environ.get("STAMP_VARIANT", "")            # e.g. "aloads1": -DMMB_EXP_ALOADS=1 (timing-only builds of round 3)
if VARIANT:
    # Round 4: the K-step ablation switches were taken out of the product kernel (a -D typo could yield a silently wrong library);
    # common.h now #errors on them.  The sources that honour them are the git tag r3-gemm-ablations:
    #     git worktree add /tmp/r3abl r3-gemm-ablations && (cd /tmp/r3abl && STAMP_VARIANT=... python tools/stamp_gemm.py --build)
    sys.exit("STAMP_VARIANT builds need the sources of git tag r3-gemm-ablations (see the comment above this line)")
LIB = os.path.join(OUT, f"libmmbert_hip_stamps{('_' + VARIANT) if VARIANT else ''}.so")
EXTRA = {"": [], "aloads1": ["-DMMB_EXP_ALOADS=1"], "aloads0": ["-DMMB_EXP_ALOADS=0"], "noloads": ["-DMMB_EXP_ALOADS=0", "-DMMB_EXP_NOBLOADS"],
         "nomfma": ["-DMMB_EXP_NOMFMA"], "nofrags": ["-DMMB_EXP_NOFRAGS"], "nofrags_noloads": ["-DMMB_EXP_NOFRAGS", "-DMMB_EXP_ALOADS=0", "-DMMB_EXP_NOBLOADS"],
         "nomfma_noloads": ["-DMMB_EXP_NOMFMA", "-DMMB_EXP_ALOADS=0", "-DMMB_EXP_NOBLOADS"],
         "wavea6": ["-DMMB_EXP_WAVEA=6"], "wavea4": ["-DMMB_EXP_WAVEA=4"], "exec0": ["-DMMB_EXP_EXEC0=1"],
         "onlymfma": ["-DMMB_EXP_NOFRAGS", "-DMMB_EXP_ALOADS=0", "-DMMB_EXP_NOBLOADS"]}[VARIANT]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]

if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for f in ("gemm", "attention", "rowwise", "heads"):
        o = os.path.join(OUT, f + VARIANT + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-DMMB_STAMPS", *EXTRA, "-c", os.path.join(ROOT, "msa_amd", "csrc", f + ".hip"), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print("built", LIB)
    sys.exit(0)

import ctypes
import numpy as np
import torch
from msa_amd import _lib
_lib.LIB_PATH = LIB
from msa_amd import ops

lib = _lib.load()
lib.mmbert_debug_set_stamps.restype = ctypes.c_int
lib.mmbert_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda"
M = int(os.environ.get("M", 18400))
mode = int(os.environ.get("MODE", 2))
shapes = [("qkv", M, 2304, 768, "bias"), ("o", M, 768, 768, "resid"), ("w1", M, 3072, 768, "gelu"), ("w2", M, 768, 3072, "resid"),
          ("dgelu", M, 3072, 768, "gelu_bwd"), ("dx", M, 768, 2304, "resid0"), ("dctx", M, 768, 768, "plain"), ("vocab", M, 30592, 768, "bias")]
buf = torch.zeros(16384 * 8 * 6, device=dev, dtype=torch.int64)
assert lib.mmbert_debug_set_stamps(buf.data_ptr()) == 0
if os.environ.get("DBG"):
    lib.mmbert_debug_set_nt_dbg.restype = ctypes.c_int
    lib.mmbert_debug_set_nt_dbg.argtypes = [ctypes.c_int]
    assert lib.mmbert_debug_set_nt_dbg(int(os.environ["DBG"])) == 0
    print("timing-only experiment", os.environ["DBG"], "(outputs are wrong by construction)")
lib.mmbert_gemm_nt_force(mode)
for name, m, n, k, epi in shapes:
    A = torch.randn(m, k, device=dev).bfloat16(); B = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev); R = torch.randn(m, n, device=dev).bfloat16(); U = torch.randn(m, n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); aux = torch.empty_like(out)
    kw = dict(out=out)
    if epi == "bias": kw.update(bias=bias)
    elif epi == "resid": kw.update(bias=bias, resid=R)
    elif epi == "resid0": kw.update(resid=R)
    elif epi == "gelu": kw.update(bias=bias, gelu=True, aux=aux)
    elif epi == "gelu_bwd": kw.update(gelu_bwd_u=U)
    for _ in range(3):
        ops.gemm_nt(A, B, **kw)
    buf.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(A, B, **kw); e1.record()
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 8, 6)
    used = st[:, 0, 0] != 0
    st = st[used]
    w0 = st[:, 0, :]                      # wave 0 of each workgroup
    rt0, rt3 = w0[:, 4], w0[:, 5]                      # 100 MHz constant clock, chip-wide
    if mode == 0 or mode >= 5:                         # persistent kernel: per-workgroup sums over its tiles
        nt_ = w0[:, 3].astype(np.float64)
        span_us = (rt3.max() - rt0.min()) / 100.0
        life_us = (rt3 - rt0) / 100.0
        tot = (w0[:, 0] + w0[:, 1] + w0[:, 2]).astype(np.float64)
        print(f"{name:6s} N={n:5d} K={k:5d} {epi:8s} wgs={len(w0):4d} tiles/wg {nt_.mean():4.2f} kernel {e0.elapsed_time(e1)*1e3:7.1f} us, span {span_us:7.1f} us | "
              f"per-tile clk: stage-0 wait {(w0[:,0]/nt_).mean():6.0f}  K-loop {(w0[:,1]/nt_).mean():7.0f}  epilogue {(w0[:,2]/nt_).mean():6.0f} | "
              f"shares {w0[:,0].sum()/tot.sum():.2f}/{w0[:,1].sum()/tot.sum():.2f}/{w0[:,2].sum()/tot.sum():.2f} | clock {tot.sum()/life_us.sum():5.0f} MHz (stamped part) | "
              f"WG life mean {life_us.mean():6.1f} max {life_us.max():6.1f} us", flush=True)
        continue
    pro, loop, epi_c = (w0[:, 1] - w0[:, 0]), (w0[:, 2] - w0[:, 1]), (w0[:, 3] - w0[:, 2])
    life = (w0[:, 3] - w0[:, 0])
    span_us = (rt3.max() - rt0.min()) / 100.0
    life_us = (rt3 - rt0) / 100.0
    mhz = life.sum() / life_us.sum()
    print(f"{name:6s} N={n:5d} K={k:5d} {epi:8s} wgs={len(w0):5d} kernel {e0.elapsed_time(e1)*1e3:7.1f} us, first start..last end {span_us:7.1f} us | "
          f"clock {mhz:6.0f} MHz | per-WG clk: prologue {pro.mean():6.0f}  K-loop {loop.mean():7.0f}  epilogue {epi_c.mean():6.0f}  total {life.mean():7.0f} "
          f"({life_us.mean():5.1f} us) | shares {pro.mean()/life.mean():.2f}/{loop.mean()/life.mean():.2f}/{epi_c.mean()/life.mean():.2f} | "
          f"CU occupancy {life_us.sum()/(256*span_us):.2f} | first-round start spread {np.sort(rt0)[min(255,len(rt0)-1)]/100.0-rt0.min()/100.0:5.1f} us", flush=True)
lib.mmbert_gemm_nt_force(0)
