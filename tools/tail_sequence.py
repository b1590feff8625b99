#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: the launch sequence of ONE step between the forward cross-entropy kernel and the first dense
attention backward (the serialized start of backward): name, duration, gap to the previous kernel.
    python tools/tail_sequence.py gpurun_out/prof_X/X_kernel_trace.csv [--step]"""
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]))
rows.sort()
# the LAST complete step: from the last "ce_row_kernel<0" to the following attn_bwd_dq launch whose duration is above 30 us (dense);
# with --step: a whole step, from the optimizer launch of the step before to this step's
if "--step" in sys.argv:
    idx = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
    start, end = idx[-2] + 1, idx[-1]
else:
    idx = [i for i, r in enumerate(rows) if r[2].startswith("ce_row_kernel<0")]
    start = idx[-2] if len(idx) > 1 else idx[-1]
    end = next(i for i in range(start, len(rows)) if rows[i][2].startswith("attn_bwd_dq") and rows[i][1] - rows[i][0] > 30000)
t0 = rows[start][0]
tot_k = 0
print(f"{'offset_us':>9} {'dur_us':>7} {'gap_us':>7}  kernel")
for i in range(start, end + 1):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3 if i > start else 0.0
    tot_k += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {n}")
print(f"launches {end - start + 1}, kernel time {tot_k:.0f} us, wall {(rows[end][0] - t0) / 1e3:.0f} us (profiled: gaps include the profiler's host cost)")
