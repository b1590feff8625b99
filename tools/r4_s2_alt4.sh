#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in base tn8 defer both; do
    case $v in base) E="";; tn8) E="MMBERT_TN_8PHASE=1";; defer) E="MMBERT_DEFER_WGRADS=1";; both) E="MMBERT_TN_8PHASE=1 MMBERT_DEFER_WGRADS=1";; esac
    env $E $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['ms_per_step'])"
  done
done
