#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into per-kernel
HBM/fabric bytes per launch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a
wide coalesced stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Units of both counters: KiB.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv > profiles/...
"""
import csv, sys, collections, re, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_digest
def load(path, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing")
print("# csrc_sha256: " + csrc_digest())      # bench.py refuses this summary once the kernel sources differ
try:
    print("# git_sha: " + subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip())
except Exception:
    pass
print("# per launch (mean over the run's launches of that kernel); fetch_MB = 2*FETCH_SIZE[KiB]/1024 (gfx950: 64 B counted per 128-B request), write_MB = WRITE_SIZE[KiB]/1024")
print("kernel,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,fetch_MB,write_MB,total_MB")
rows = []
for k in f:
    n, fs = f[k]; ws = w.get(k, [n, 0.0])[1] / max(w.get(k, [n, 0.0])[0], 1); fs /= n
    rows.append((2 * fs / 1024 + ws / 1024, k, n, fs, ws))
for tot, k, n, fs, ws in sorted(rows, reverse=True)[:24]:
    print(f"\"{k}\",{n},{fs:.1f},{ws:.1f},{2*fs/1024:.1f},{ws/1024:.1f},{tot:.1f}")
