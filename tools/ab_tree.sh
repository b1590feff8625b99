#!/bin/bash
# Whole-tree A/B on ONE box, alternating processes ("alternating-process A/B of the r5-final tree against the r6-final tree").
#   bash tools/ab_tree.sh tools/_ab/r5_tree 3 40 > gpurun_out/r6_ab_tree.log
# A = the other tree (its own msa_amd, bench.py and library), B = this tree.  Each process runs the headline leg only.
OTHER=${1:-tools/_ab/r5_tree}; ROUNDS=${2:-3}; STEPS=${3:-40}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
FLAGS="--no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --no-reference-default --no-scores-fp32 --no-deterministic --no-dp-reference-legs --steps $STEPS --warmup 8"
for r in $(seq 1 $ROUNDS); do
  for side in A B; do
    if [ "$side" = "A" ]; then dir="$HERE/$OTHER"; extra=""; else dir="$HERE"; extra="--no-box-probe"; fi
    out=$(cd "$dir" && python3 bench.py $FLAGS $extra 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['value'])")
    echo "round $r $side $out"
  done
done | tee /tmp/_ab_tree.txt
python3 - <<'PY'
import statistics
a=[float(l.split()[3]) for l in open('/tmp/_ab_tree.txt') if ' A ' in l]; b=[float(l.split()[3]) for l in open('/tmp/_ab_tree.txt') if ' B ' in l]
print(f"A (other tree) median {statistics.median(a):.3f} ms  B (this tree) median {statistics.median(b):.3f} ms  B/A {statistics.median(b)/statistics.median(a):.4f}")
PY
