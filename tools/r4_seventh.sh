#!/bin/bash
# round 4, seventh GPU call: fused heads at batch <= 32, reference-default leg again, forced DP with the RCCL summary
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -k "fused_heads or bert_large or standalone or grouped_tile_walk" > $O/r4_pytest7.log 2>&1; echo "rc $?" >> $O/r4_pytest7.log; tail -8 $O/r4_pytest7.log | cut -c1-300
python bench.py --preset reference-default > $O/r4_refdef_b.json 2> $O/r4_refdef_b.err; cut -c1-400 $O/r4_refdef_b.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4_refdef2 -o refdef -- python3 bench.py --preset reference-default --no-kernel-timing --steps 16 --warmup 3 > $O/prof_r4_refdef2.log 2>&1
python3 tools/trace_gaps.py $O/prof_r4_refdef2/refdef_kernel_trace.csv > $O/r4_refdef_timeline2.txt; head -30 $O/r4_refdef_timeline2.txt | cut -c1-160
python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default > $O/r4_bench_forcedp.json 2> $O/r4_bench_forcedp.err; cut -c1-200 $O/r4_bench_forcedp.json; grep -c "RCCL summary" $O/r4_bench_forcedp.err; true
