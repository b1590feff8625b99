#!/bin/bash
# Per-kernel time of the headline train step in TWO trees on one box (rocprofv3 --kernel-trace --stats, 20 steps each), largest differences first.
#   bash tools/prof_compare.sh tools/_ab/r5_tree > gpurun_out/r6_prof_compare.txt
OTHER=${1:-tools/_ab/r5_tree}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --no-reference-default --no-scores-fp32 --no-deterministic --no-dp-reference-legs --steps 20 --warmup 5"
cd /tmp
rm -rf /tmp/pc_A /tmp/pc_B
(cd "$HERE/$OTHER" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_A -o a -- python3 bench.py $FLAGS > /tmp/pc_A.log 2>&1)
(cd "$HERE" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_B -o b -- python3 bench.py $FLAGS --no-box-probe > /tmp/pc_B.log 2>&1)
python3 - <<'PY'
import csv, glob, re
def load(d):
    f = [x for x in glob.glob(d + "/**/*.csv", recursive=True) if "kernel_stats" in x][0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"])[:70]
        out[name] = out.get(name, 0.0) + float(r["TotalDurationNs"]) / 25.0 / 1e3      # us per step (20 timed + 5 warm-up steps)
    return out
A, B = load("/tmp/pc_A"), load("/tmp/pc_B")
rows = sorted(((B.get(k, 0.0) - A.get(k, 0.0), k) for k in set(A) | set(B)), key=lambda x: -abs(x[0]))
print(f"kernel time per step: A {sum(A.values()):9.1f} us   B {sum(B.values()):9.1f} us   B - A {sum(B.values()) - sum(A.values()):+8.1f} us")
for d, k in rows[:40]:
    print(f"{d:+9.1f} us   A {A.get(k, 0.0):8.1f}   B {B.get(k, 0.0):8.1f}   {k}")
PY
