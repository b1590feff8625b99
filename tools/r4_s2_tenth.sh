#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "side_stream or lazy or adamw or trajectory" > $O/r4s2_pytest10.log 2>&1; echo "rc $?" >> $O/r4s2_pytest10.log; tail -12 $O/r4s2_pytest10.log | cut -c1-300
ROUNDS=7 STEPS=8 timeout 900 python tools/ab_step.py inline: overlap:attr.overlap_heads_backward=True > $O/r4s2_ab_heads_bwd_overlap.log 2>&1; grep -v amdgpu $O/r4s2_ab_heads_bwd_overlap.log
