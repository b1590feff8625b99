#!/usr/bin/env python3
"""One train step out of a rocprofv3 --kernel-trace CSV: busy time, idle gaps, and the kernels in timeline order.
    python tools/trace_gaps.py gpurun_out/prof_x/x_kernel_trace.csv [--list]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw")]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, kernel time {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms, {len(step)} launches")
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = short(r["Kernel_Name"])
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
print("per-step kernel time by name (us):")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {t:9.1f}  {n:4d}x  {k}")
gaps = []
for p, q in zip(step[:-1], step[1:]):
    gaps.append((int(q["Start_Timestamp"]) - int(p["End_Timestamp"]), short(p["Kernel_Name"]), short(q["Kernel_Name"])))
print("idle by gap size: ", {f">{lo}us": round(sum(g for g, _, _ in gaps if g > lo * 1000) / 1e3) for lo in (0, 2, 5, 10, 20, 50)}, "us")
if "--list" in sys.argv:
    for r, (g, _, _) in zip(step, [(0, "", "")] + gaps):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} gap {g / 1e3:6.1f}  {short(r['Kernel_Name'])}")
