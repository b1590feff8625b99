#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass: per kernel and launch, L2 hits / misses (128-B lines, summed over the 8 XCDs'
L2s) and the hit rate TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2).

    python tools/pmc_l2.py gpurun_out/pmc_r5_TCC/pmc_counter_collection.csv > profiles/r5_pmc_l2_hit.csv"""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 2 --warmup 1 ...; per launch (mean)")
print("kernel,launches,TCC_HIT,TCC_MISS,hit_rate,miss_MB_at_128B")
rows = []
for k, c in acc.items():
    n = max(c["TCC_HIT_sum"][0], 1)
    h, m = c["TCC_HIT_sum"][1] / n, c["TCC_MISS_sum"][1] / max(c["TCC_MISS_sum"][0], 1)
    rows.append((h + m, k, n, h, m))
for tot, k, n, h, m in sorted(rows, reverse=True)[:24]:
    print(f"\"{k}\",{n},{h:.4g},{m:.4g},{h / max(h + m, 1):.3f},{m * 128 / 1e6:.1f}")
