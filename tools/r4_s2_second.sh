#!/bin/bash
# round 4, session 2, second GPU call: the 8-phase TN kernel -- exactness against the ring form, stand-alone A/B, in-step A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_tn" > $O/r4s2_pytest2.log 2>&1; echo "rc $?" >> $O/r4s2_pytest2.log; tail -15 $O/r4s2_pytest2.log | cut -c1-400
timeout 600 python tools/bench_tn_forms.py > $O/r4s2_tn_forms.log 2>&1; cat $O/r4s2_tn_forms.log
ROUNDS=5 STEPS=8 timeout 900 python tools/ab_step.py tn8: ring:MMBERT_TN_8PHASE=0 > $O/r4s2_ab_tn8.log 2>&1; cat $O/r4s2_ab_tn8.log
