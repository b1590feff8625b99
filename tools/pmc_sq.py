#!/usr/bin/env python3
"""Per-kernel SQ counters of one rocprofv3 --pmc pass (tools/profile_round.sh): wave-cycle split (parked / issue-stalled / issuing), VMEM
and LDS instruction counts, LDS bank conflicts -- for the GEMM families.   python tools/pmc_sq.py <pmc_counter_collection.csv>"""
import csv, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
    if not k.startswith(("gemm_", "attn_")): continue
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
print("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- python3 bench.py --steps 2 --warmup 1 (per launch, mean)")
print("# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked at s_waitcnt / s_barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY; quad-cycles summed over waves")
print("kernel,launches," + ",".join(names) + ",wait_any_share,wait_inst_share,active_share,lds_conflict_per_active")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 0])[1]):
    v = {c: acc[k][c][1] / max(acc[k][c][0], 1) for c in names}
    n = max(a[0] for a in acc[k].values())
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    la = v.get("SQ_LDS_IDX_ACTIVE", 0) or 1
    print(f"\"{k}\",{n}," + ",".join(f"{v[c]:.4g}" for c in names) + f",{v.get('SQ_WAIT_ANY', 0) / wc:.3f},{v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f},{v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f},{v.get('SQ_LDS_BANK_CONFLICT', 0) / la:.4f}")
