#!/bin/bash
# rocprofv3 kernel-trace passes over the default bench workload, prologue on the input stream and on the compute stream, on ONE box
# -> per-step kernel tables + timelines (gpurun_out/quick_*).  (The profiler slows the host by tens of us per launch: how much of the
# step then shows up as idle depends on the box's host; tools/host_time.py gives the un-profiled host cost of a step.)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_quick -o quick -- $BENCH --steps 20 --warmup 5 > gpurun_out/prof_quick.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_quick/quick_kernel_trace.csv > gpurun_out/quick_step_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_quick_sync -o quick -- $BENCH --steps 20 --warmup 5 --sync-prologue > gpurun_out/prof_quick_sync.log 2>&1
python3 tools/trace_gaps.py gpurun_out/prof_quick_sync/quick_kernel_trace.csv > gpurun_out/quick_step_timeline_sync.txt
python3 tools/host_time.py 2>&1 | grep -i "enqueue\|host cost" | tail -2 > gpurun_out/quick_host_time.txt
grep "last 10" gpurun_out/quick_step_timeline.txt gpurun_out/quick_step_timeline_sync.txt; cat gpurun_out/quick_host_time.txt
