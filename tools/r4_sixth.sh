#!/bin/bash
# round 4, sixth GPU call: sustained headline run (>= 20 s), reference-default kernel trace, forced-DP record
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing > $O/r4_bench_sustained.json 2> $O/r4_bench_sustained.err; cut -c1-260 $O/r4_bench_sustained.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4_refdef -o refdef -- python3 bench.py --preset reference-default --no-kernel-timing --steps 16 --warmup 3 > $O/prof_r4_refdef.log 2>&1
python3 tools/trace_gaps.py $O/prof_r4_refdef/refdef_kernel_trace.csv > $O/r4_refdef_timeline.txt; head -50 $O/r4_refdef_timeline.txt | cut -c1-160
python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default > $O/r4_bench_forcedp.json 2> $O/r4_bench_forcedp.err; cut -c1-200 $O/r4_bench_forcedp.json; grep -c "RCCL summary" $O/r4_bench_forcedp.err
