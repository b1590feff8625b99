#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_attn.sh: per kernel and launch, every collected counter, plus derived shares.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots):
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.   python tools/pmc_attn.py gpurun_out/pmc_attn_g > profiles/..."""
import csv, collections, glob, re, sys
pre = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(pre + "*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if "attn" not in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
print("# rocprofv3 --pmc (3 passes, tools/pmc_attn.sh) -- python3 tools/bench_attn.py (16 x S=50 + 32 x S=550, 12 heads; p = 0 and p = 0.1 launches pooled)")
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join(f"{acc[k][c][1] / max(acc[k][c][0], 1):.4g}" for c in names))
print("# derived (per launch)")
for k in sorted(acc):
    v = {c: acc[k][c][1] / max(acc[k][c][0], 1) for c in names}
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    out = [f"wait_any {v.get('SQ_WAIT_ANY', 0) / wc:.3f}", f"wait_inst_any {v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}", f"active_inst_any {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}",
           f"active_valu {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}", f"active_lds {v.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}", f"wait_inst_lds {v.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}"]
    if v.get("SQ_LDS_IDX_ACTIVE"): out.append(f"lds_bank_conflict/idx_active {v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']:.3f}")
    if v.get("SQ_INSTS_VALU") and v.get("SQ_INSTS_MFMA"): out.append(f"valu_per_mfma {(v['SQ_INSTS_VALU'] - v['SQ_INSTS_MFMA']) / v['SQ_INSTS_MFMA']:.2f}")
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"): out.append(f"mfma_util {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * v['GRBM_GUI_ACTIVE']):.3f}")
    print(f"# {k}: " + "  ".join(out))
