#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in base m224 lvl2; do
    case $v in base) E="";; m224) E="MMBERT_NT_8PHASE_M224=1";; lvl2) E="MMBERT_NT_8PHASE=2";; esac
    env $E $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['ms_per_step'])"
  done
done
# and the driver's own form: default flags, 20 steps after 5 warm-up steps, per process
for rep in 1 2; do
  for v in base m224; do
    case $v in base) E="";; m224) E="MMBERT_NT_8PHASE_M224=1";; esac
    env $E python bench.py --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short $v', r['value'], r['ms_per_step'])"
  done
done
