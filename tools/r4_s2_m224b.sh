#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > $O/r4s2_pytest_m224.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_m224.log; tail -8 $O/r4s2_pytest_m224.log | cut -c1-300
ROUNDS=5 STEPS=8 timeout 900 python tools/ab_step.py m224: ring:MMBERT_NT_8PHASE_M224=0 "noqkv:MMBERT_NT_8PHASE_M224_SKIP=2304:768:1" "noup:MMBERT_NT_8PHASE_M224_SKIP=3072:768:3" "nodgelu:MMBERT_NT_8PHASE_M224_SKIP=3072:768:8" > $O/r4s2_ab_m224_shapes.log 2>&1; grep -v amdgpu $O/r4s2_ab_m224_shapes.log
