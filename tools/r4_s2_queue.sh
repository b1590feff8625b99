#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > $O/r4s2_pytest_queue.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_queue.log; tail -8 $O/r4s2_pytest_queue.log | cut -c1-300
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "data_parallel or rccl or dp or hook" > $O/r4s2_pytest_queue2.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_queue2.log; tail -4 $O/r4s2_pytest_queue2.log | cut -c1-300
for i in 1 2; do python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forcedp q8', r['value'], r['ms_per_step'])"; MMBERT_NT_8PHASE_QUEUE=0 python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forcedp ring', r['value'], r['ms_per_step'])"; done
ROUNDS=4 STEPS=40 timeout 600 python tools/ab_step.py base: 2>&1 | grep -v amdgpu
