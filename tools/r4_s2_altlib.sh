#!/bin/bash
# alternating 600-step processes: the library in the tree against another build of it (tools/_ab/libmmbert_prev.so), same Python
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > $O/r4s2_pytest_bias.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_bias.log; tail -4 $O/r4s2_pytest_bias.log | cut -c1-300
B="python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in new prev; do
    case $v in new) E="";; prev) E="MMBERT_LIB_PATH=$PWD/tools/_ab/libmmbert_prev.so";; esac
    env $E $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['ms_per_step'])"
  done
done
