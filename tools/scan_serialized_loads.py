"""Scan the ISA of every kernel for "one global load, then s_waitcnt vmcnt(0)" -- the signature of loads that hipcc serialized (DESIGN 3.3:
a select that became a branch per load, a ring rotated by register copies, scalar row pointers spilled to VGPR lanes).  Compiles each
msa_amd/csrc/*.hip to assembly with the product's flags (no GPU needed) and prints, per kernel, how many loads it has, how many full
drains, and how many of those drains follow a single load.  LDS-DMA loads (global_load_lds / buffer_load ... lds) are not counted: their
double-buffered loops drain by design.

    python tools/scan_serialized_loads.py [min_serialized=2]

tests/test_isa_cpu.py pins the kernels that round 4 fixed (``scan_source``)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msa_amd.build import HIPCC, FLAGS  # noqa: E402

SRC_DIR = os.path.join(ROOT, "msa_amd", "csrc")


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out)) if len(out) >= len(names) else {n: n for n in names}
    except OSError:
        return {n: n for n in names}


def scan_source(name):
    """{demangled kernel name: (loads, full drains, drains that follow a single load)} for msa_amd/csrc/<name>."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, name + ".s")
        flags = [f for f in FLAGS if f != "-fPIC"]
        subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", os.path.join(SRC_DIR, name), "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    kern, res, since = None, {}, 0
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, since = m.group(1), 0
            res[kern] = [0, 0, 0]
            continue
        if kern is None:
            continue
        t = l.strip()
        if (t.startswith("global_load") or t.startswith("buffer_load")) and "lds" not in t:
            res[kern][0] += 1
            since += 1
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            res[kern][1] += 1
            if since == 1:
                res[kern][2] += 1
            since = 0
    dem = _demangle(list(res))
    return {dem[k]: tuple(v) for k, v in res.items()}


if __name__ == "__main__":
    thr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for name in sorted(os.listdir(SRC_DIR)):
        if not name.endswith(".hip"):
            continue
        for k, v in scan_source(name).items():
            if v[2] >= thr:
                print(f"{name:14s} loads {v[0]:3d}  vmcnt(0) {v[1]:3d}  single-load-then-drain {v[2]:3d}  {k[:110]}")
