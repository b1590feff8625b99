#!/usr/bin/env python3
"""Summarise the four rocprofv3 --pmc passes of tools/pmc_mfma.sh (one counter per pass, same bench command) into per-kernel
MFMA utilisation.  Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES = shader cycles, summed over the 1024 SIMDs;
GRBM_GUI_ACTIVE = cycles summed over the 8 XCDs; SQ_INSTS_VALU_MFMA_MOPS_BF16 = units of 512 FLOP (checked: the plain
M=18400, N=K=768 product counts 4.284e7 = 83 tiles x 224 x 768 x 768 x 2 / 512).

    python tools/pmc_mfma.py gpurun_out/pmc4_ > profiles/r1_pmc_mfma_util.csv
"""
import csv, collections, re, sys
pre = sys.argv[1]
import os, sys as _s
_s.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
def load(c):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f"{pre}{c}/pmc_counter_collection.csv")):
        if r["Counter_Name"] != c: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"]); acc[k][2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc
m, g, o = load("SQ_VALU_MFMA_BUSY_CYCLES"), load("GRBM_GUI_ACTIVE"), load("SQ_INSTS_VALU_MFMA_MOPS_BF16")
print("# csrc_sha256: " + csrc_digest())
print("# rocprofv3 --pmc <counter> --kernel-trace, one counter per pass -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fused")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (128 SIMDs per XCD x GRBM_GUI_ACTIVE); executed_TFLOP_s = MOPS x 512 / duration (duration of the MOPS pass)")
print("# (GRBM_GUI_ACTIVE / 8 / duration reads ~2.2-2.3 GHz on these 30-180 us dispatches: the guide notes the quotient reads high below 0.3 ms)")
print("kernel,launches,mfma_busy_cycles_per_launch,gui_active_per_launch,mfma_util,mops_bf16_per_launch,avg_duration_us,executed_TFLOP_s")
for k in sorted(m, key=lambda k: -m[k][1]):
    n = m[k][0]
    if m[k][1] / n < 1e6 or k not in g or k not in o: continue
    busy, gui, mops, dur = m[k][1] / n, g[k][1] / g[k][0], o[k][1] / o[k][0], o[k][2] / o[k][0] / 1e3
    print(f"\"{k}\",{n},{busy:.4g},{gui:.4g},{busy / (128 * gui):.3f},{mops:.4g},{dur:.1f},{mops * 512 / (dur * 1e-6) / 1e12:.0f}")
