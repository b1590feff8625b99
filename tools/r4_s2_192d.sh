#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in 1 0; do
    MMBERT_NT_8PHASE_BM192=$v $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BM192=$v', r['value'], r['ms_per_step'])"
  done
done
