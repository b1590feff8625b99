#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r4s2_pytest_final2.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_final2.log; tail -5 $O/r4s2_pytest_final2.log | cut -c1-300
ROUNDS=4 STEPS=40 timeout 900 python tools/ab_step.py base: vocab:MMBERT_NT_8PHASE_M224=2 noqkv:MMBERT_NT_8PHASE_M224_SKIP=2304:768:1 noup:MMBERT_NT_8PHASE_M224_SKIP=3072:768:3 nodgelu:MMBERT_NT_8PHASE_M224_SKIP=3072:768:8 2>&1 | grep -v amdgpu | tee $O/r4s2_ab_m224_final.log
ROUNDS=4 STEPS=20 timeout 900 python tools/ab_refdef.py base: ring:MMBERT_TN_8PHASE=0 nom224:MMBERT_NT_8PHASE_M224=0 paired:attr.defer_wgrads=False 2>&1 | grep -v amdgpu | tee $O/r4s2_ab_refdef_final.log
