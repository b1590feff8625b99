#!/bin/bash
# kernel trace of the bench command with the data-parallel wrapper forced on one GPU (world size 1 over RCCL): what the wrapper adds per step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dp -o dp -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --force-dp --steps 20 --warmup 5 > gpurun_out/prof_dp.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_dp/dp_kernel_trace.csv > gpurun_out/dp_step_timeline.txt
head -60 gpurun_out/dp_step_timeline.txt | cut -c1-150
grep "last 10" gpurun_out/dp_step_timeline.txt
