#!/bin/bash
# round 4, session 2, fourth GPU call: the 8-phase TN kernel on the reference's default model (bert-large), and per-kernel durations in the step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
  for f in 1 0; do
    MMBERT_TN_8PHASE=$f python bench.py --preset reference-default --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('refdef TN_8PHASE=$f', r.get('value'), r.get('ms_per_step'))" | tee -a $O/r4s2_refdef_tn8.log
  done
done
HID=1024 FF=4096 M=6400 timeout 300 python tools/bench_tn_forms.py 2>&1 | grep -v amdgpu | tee -a $O/r4s2_refdef_tn8.log
cd /tmp
for f in 1 0; do
  rm -rf /tmp/prof_tn$f
  MMBERT_TN_8PHASE=$f rocprofv3 --kernel-trace --stats -d /tmp/prof_tn$f -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --no-reference-default > /dev/null 2>&1
  python3 - <<PY | tee -a $GRAFT_REPO_ROOT/$O/r4s2_tn8_instep_kernels.log
import csv, glob
f = glob.glob('/tmp/prof_tn$f/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('TN_8PHASE=$f total kernel ms per step', tot / 13 / 1e6)
for r in rows[:12]:
    print('   ', r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs']) / 13 / 1e3, 1), 'us/step', round(float(r['AverageNs']) / 1e3, 1), 'us avg')
PY
done
