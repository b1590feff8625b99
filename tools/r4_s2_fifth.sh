#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt or gemm_tn" > $O/r4s2_pytest5.log 2>&1; echo "rc $?" >> $O/r4s2_pytest5.log; tail -4 $O/r4s2_pytest5.log | cut -c1-300
ROUNDS=7 STEPS=8 timeout 900 python tools/ab_step.py dropped: live:MMBERT_NT8_DEADLIVE=1 > $O/r4s2_ab_deadloads.log 2>&1; cat $O/r4s2_ab_deadloads.log
