#!/bin/bash
# one rocprofv3 kernel-trace pass over the default bench workload -> per-step kernel table + timeline (gpurun_out/quick_*)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_quick -o quick -- $BENCH --steps 20 --warmup 5 > gpurun_out/prof_quick.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_quick/quick_kernel_trace.csv > gpurun_out/quick_step_timeline.txt
head -50 gpurun_out/quick_step_timeline.txt
tail -2 gpurun_out/prof_quick.log
