#!/bin/bash
# round 4, eighth GPU call: attention backward query limit (top layer)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -k "attention or sparse_backward or replayed_masks or unmasked_rows or cfg1_matches or hook_fires" > $O/r4_pytest8.log 2>&1; echo "rc $?" >> $O/r4_pytest8.log; tail -8 $O/r4_pytest8.log | cut -c1-300
ROUNDS=7 STEPS=8 python tools/ab_step.py qlimit: dense_top_attn:attr.top_layer_query_limit=False > $O/r4_ab_query_limit.log 2>&1; cat $O/r4_ab_query_limit.log
