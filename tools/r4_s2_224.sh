#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > $O/r4s2_pytest_224.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_224.log; tail -12 $O/r4s2_pytest_224.log | cut -c1-300
ROUNDS=7 STEPS=8 timeout 900 python tools/ab_step.py bm224: bm256:MMBERT_NT_8PHASE_BM224=0 > $O/r4s2_ab_bm224.log 2>&1; grep -v amdgpu $O/r4s2_ab_bm224.log
