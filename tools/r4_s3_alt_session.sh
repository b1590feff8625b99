#!/bin/bash
# alternating 600-step processes: the tree against the tree at the start of the round's last session (worktree _old, commit 2f17867), same box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing"
for rep in 1 2 3; do
  for v in new old; do
    case $v in new) D=.;; old) D=_old;; esac
    (cd $D && timeout 300 python $B 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['value'], r['ms_per_step'])")
  done
done
