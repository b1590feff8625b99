#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py -m gpu -q -x > $O/r4s2_pytest7.log 2>&1; echo "rc $?" >> $O/r4s2_pytest7.log; tail -25 $O/r4s2_pytest7.log | cut -c1-400
