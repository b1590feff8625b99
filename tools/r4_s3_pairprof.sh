#!/bin/bash
# true kernel durations of the pair projection kernels (the stand-alone tool's event times include the host's launch latency)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pp; mkdir -p gpurun_out/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp -o pp -- python3 tools/bench_pair_proj.py > gpurun_out/pp/run.log 2>&1
grep "fwd\|bwd" gpurun_out/pp/run.log
f=$(find gpurun_out/pp -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -i "pair\|Name" "$f" | cut -c1-200; else echo "no stats file"; fi
