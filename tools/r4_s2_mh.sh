#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > $O/r4s2_pytest_mh.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_mh.log; tail -8 $O/r4s2_pytest_mh.log | cut -c1-300
ROUNDS=4 STEPS=20 timeout 900 python tools/ab_refdef.py base: mh224:MMBERT_NT_8PHASE_MH=224 2>&1 | grep -v amdgpu
ROUNDS=3 STEPS=40 timeout 600 python tools/ab_step.py base: mh224:MMBERT_NT_8PHASE_MH=224 2>&1 | grep -v amdgpu
