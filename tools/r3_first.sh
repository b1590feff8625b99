#!/bin/bash
# round 3, first GPU call: full GPU test suite, bench with / without per-launch events, host enqueue time, step timelines
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
free -g > $O/r3_box.txt; nproc >> $O/r3_box.txt; rocm-smi --showclocks >> $O/r3_box.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r3_pytest.log 2>&1; echo "pytest rc $?" >> $O/r3_pytest.log
tail -5 $O/r3_pytest.log
python bench.py --no-cpu-baseline > $O/r3_bench_a.json 2> $O/r3_bench_a.err
python bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only > $O/r3_bench_noev.json 2>> $O/r3_bench_a.err
python bench.py --no-cpu-baseline --no-fused --no-dense-reference --no-train-only > $O/r3_bench_ev.json 2>> $O/r3_bench_a.err
python bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only > $O/r3_bench_noev2.json 2>> $O/r3_bench_a.err
python tools/host_time.py > $O/r3_host_time.txt 2>&1
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3a -o r3a -- $BENCH --steps 20 --warmup 5 > $O/prof_r3a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3a_sync -o r3a_sync -- $BENCH --steps 20 --warmup 5 --sync-prologue > $O/prof_r3a_sync.log 2>&1
python3 tools/trace_gaps.py $O/prof_r3a/r3a_kernel_trace.csv > $O/r3a_step_timeline_async.txt
python3 tools/trace_gaps.py $O/prof_r3a_sync/r3a_sync_kernel_trace.csv > $O/r3a_step_timeline_sync.txt
cat $O/r3_bench_*.json | cut -c1-400
