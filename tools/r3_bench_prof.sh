#!/bin/bash
# bench (no CPU leg) twice + one profiled run with the step timeline: tools/r3_bench_prof.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; T=${1:-x}
python bench.py --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err
python bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only > $O/bench_${T}_2.json 2>> $O/bench_$T.err
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- $BENCH --steps 20 --warmup 5 > $O/prof_$T.log 2>&1
python3 tools/trace_gaps.py $O/prof_$T/${T}_kernel_trace.csv > $O/${T}_step_timeline.txt
python3 - <<PY
import json
for f in ("$O/bench_$T.json", "$O/bench_${T}_2.json"):
    r = json.load(open(f))
    print(f, r["value"], r["ms_per_step"], r.get("ms_per_step_instrumented"), {k: r[k]["value"] for k in ("dense_backward_reference", "train_only", "fused1050") if k in r}, r.get("roofline", {}).get("frac"))
PY
head -3 $O/${T}_step_timeline.txt; grep "last 10 steps" $O/${T}_step_timeline.txt
