#!/bin/bash
# round 4, eleventh GPU call: deferred weight gradients (one call, whole rounds)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -k "gemm_tn or deferred or replayed_masks or cfg1_matches or unmasked_rows or sparse_backward or hook_fires or trainer_loop" > $O/r4_pytest11.log 2>&1; echo "rc $?" >> $O/r4_pytest11.log; tail -8 $O/r4_pytest11.log | cut -c1-300
ROUNDS=7 STEPS=8 python tools/ab_step.py deferred: paired:attr.defer_wgrads=False > $O/r4_ab_deferred_wgrads.log 2>&1; cat $O/r4_ab_deferred_wgrads.log
python bench.py --no-cpu-baseline > $O/r4_bench_e.json 2> $O/r4_bench_e.err; cut -c1-300 $O/r4_bench_e.json
