#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
ROUNDS=4 STEPS=40 timeout 900 python tools/ab_step.py base: m224:MMBERT_NT_8PHASE_M224=1 lvl2:MMBERT_NT_8PHASE=2 wgradside:attr.overlap_wgrad=True,attr.defer_wgrads=False overlap:attr.overlap_heads_backward=True ring:MMBERT_TN_8PHASE=0 paired:attr.defer_wgrads=False 2>&1 | grep -v amdgpu
