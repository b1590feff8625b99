#!/usr/bin/env python3
"""One train step out of a rocprofv3 --kernel-trace CSV: busy time, idle gaps, and the kernels in timeline order.
    python tools/trace_gaps.py gpurun_out/prof_x/x_kernel_trace.csv [--list] [--gaps US]
--gaps US (default 20): every idle gap longer than US microseconds with the kernel that ends before it and the one that starts after
it, for the cut-out step and -- as a count / sum per (before, after) pair -- over the last 10 steps of the trace.
Kernels of different streams overlap (the step prologue on the input stream runs under the previous step's AdamW): busy time is the
UNION of the kernels' intervals and a gap is measured from the latest end so far (a first version summed durations and took gaps
between neighbours in start order: the overlap showed up as a 300-500 us "gap" behind the prologue's copy and was subtracted from the
idle time elsewhere)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw")]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
def union_and_gaps(seq, before=None):
    """(busy ns of the union of the intervals, [(gap ns, row before = the one that ended last, row after)]); ``before`` = the kernel
    in front of the cut (the previous step's AdamW: the prologue of this step starts under it)"""
    busy, gaps, end, last = 0, [], None, None
    if before is not None:
        end, last = int(before["End_Timestamp"]), before
        first = seq[0]
        gaps.append((max(0, int(first["Start_Timestamp"]) - end), last, first))
        busy += max(0, int(first["End_Timestamp"]) - max(end, int(first["Start_Timestamp"])))
        if int(first["End_Timestamp"]) >= end:
            end, last = int(first["End_Timestamp"]), first
        seq = seq[1:]
        gaps.pop()                                           # (the caller's gap list is aligned with seq[1:])
    for r in seq:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if end is None:
            busy += e_ - s_; end, last = e_, r
            continue
        if s_ >= end:
            gaps.append((s_ - end, last, r)); busy += e_ - s_
        else:
            gaps.append((0, last, r)); busy += max(0, e_ - end)
        if e_ >= end:
            end, last = e_, r
    return busy, gaps
busy, step_gaps = union_and_gaps(step, rows[a])
t0 = max(t0, int(rows[a]["End_Timestamp"]))
t1 = max(int(r["End_Timestamp"]) for r in step)
ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, kernel time {1e-6 * busy:.3f} ms (union over streams; sum of durations {1e-6 * ksum:.3f}), idle {1e-6 * (t1 - t0 - busy):.3f} ms, {len(step)} launches")
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = short(r["Kernel_Name"])
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
print("per-step kernel time by name (us):")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {t:9.1f}  {n:4d}x  {k}")
gaps = [(g, short(p["Kernel_Name"]), short(q["Kernel_Name"])) for g, p, q in step_gaps]
print("idle by gap size: ", {f">{lo}us": round(sum(g for g, _, _ in gaps if g > lo * 1000) / 1e3) for lo in (0, 2, 5, 10, 20, 50)}, "us")
thr = 20.0
if "--gaps" in sys.argv:
    thr = float(sys.argv[sys.argv.index("--gaps") + 1])
print(f"gaps > {thr:g} us in this step (offset in step, gap, kernel before -> kernel after):")
for (g, kb, ka), q in zip(gaps, step[1:]):
    if g > thr * 1000:
        print(f"  {(int(q['Start_Timestamp']) - t0) / 1e3:9.1f}  {g / 1e3:7.1f} us   {kb[:48]}  ->  {ka[:48]}")
# the same over the last 10 steps (which gaps are systematic)
if len(idx) >= 12:
    lo, hi = idx[-12], idx[-2]
    pairs = collections.defaultdict(lambda: [0, 0.0])
    seq = rows[lo + 1:hi + 1]
    busy10, seq_gaps = union_and_gaps(seq, rows[lo])
    for g, p, q in seq_gaps:
        if g > thr * 1000:
            k = (short(p["Kernel_Name"])[:44], short(q["Kernel_Name"])[:44])
            pairs[k][0] += 1; pairs[k][1] += g / 1e3
    wall = (max(int(r["End_Timestamp"]) for r in seq) - max(int(seq[0]["Start_Timestamp"]), int(rows[lo]["End_Timestamp"]))) / 1e3
    busy10 = busy10 / 1e3
    print(f"last 10 steps: wall {wall / 10:.1f} us/step, kernels {busy10 / 10:.1f} us/step, idle {(wall - busy10) / 10:.1f} us/step, {len(seq) / 10:.1f} launches/step; gaps > {thr:g} us by (before -> after), per step:")
    for (kb, ka), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / 10:8.1f} us  {n / 10:5.1f}x  {kb}  ->  {ka}")
if "--list" in sys.argv:
    for r, (g, _, _) in zip(step, [(0, "", "")] + gaps):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} gap {g / 1e3:6.1f}  {short(r['Kernel_Name'])}")
