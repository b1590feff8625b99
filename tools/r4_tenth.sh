#!/bin/bash
# round 4, tenth GPU call: the multi-tile 8-phase form per shape
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
ROUNDS=7 STEPS=8 python tools/ab_step.py base: qkv:MMBERT_NT_8PHASE_MULTI=2304:768:1 up:MMBERT_NT_8PHASE_MULTI=3072:768:3 dgelu:MMBERT_NT_8PHASE_MULTI=3072:768:8 "vocab:MMBERT_NT_8PHASE_MULTI=30592:768:1" > $O/r4_ab_8phase_multi_per_shape.log 2>&1; cat $O/r4_ab_8phase_multi_per_shape.log
