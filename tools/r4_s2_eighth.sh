#!/bin/bash
# round 4, session 2, eighth GPU call: fewer launches at the start of backward -- full suite, bench, region time, timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r4s2_pytest8.log 2>&1; echo "rc $?" >> $O/r4s2_pytest8.log; tail -12 $O/r4s2_pytest8.log | cut -c1-400
python3 tools/tail_region.py 2>&1 | grep -v amdgpu > $O/r4s2_tail_region.txt; cat $O/r4s2_tail_region.txt
python bench.py --no-cpu-baseline > $O/r4s2_bench8.json 2> $O/r4s2_bench8.err; cut -c1-300 $O/r4s2_bench8.json
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --no-reference-default"
rm -rf $O/prof_r4s2; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4s2 -o r4s2 -- $BENCH --steps 20 --warmup 5 > $O/prof_r4s2.log 2>&1
python3 tools/trace_gaps.py $O/prof_r4s2/r4s2_kernel_trace.csv > $O/r4s2_step_timeline.txt; head -3 $O/r4s2_step_timeline.txt; grep "last 10 steps" $O/r4s2_step_timeline.txt
