#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in bm224 base; do
    case $v in base) E="";; bm224) E="MMBERT_NT_8PHASE_BM224=1";; esac
    env $E $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['ms_per_step'])"
  done
done
ROUNDS=5 STEPS=8 timeout 600 python tools/ab_step.py base: bm224:MMBERT_NT_8PHASE_BM224=1 2>&1 | grep -v amdgpu
ROUNDS=5 STEPS=40 timeout 600 python tools/ab_step.py base: bm224:MMBERT_NT_8PHASE_BM224=1 2>&1 | grep -v amdgpu
