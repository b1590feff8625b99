#!/bin/bash
# Round-3 profile refresh after the LayerNorm / paired-wgrad changes: kernel stats + timeline, PMC traffic / MFMA util / SQ (tools/profile_round.sh),
# attention counters, the sync-prologue timeline.  (The NT K-step ablation log profiles/r3_stamp_nt_ablation.log is unchanged: gemm_ntp untouched.)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
bash tools/profile_round.sh r3 > $O/profile_round_r3.log 2>&1
bash tools/pmc_attn.sh > $O/pmc_attn_r3.log 2>&1
python3 tools/pmc_attn.py $O/pmc_attn_g > $O/r3_pmc_attention.csv 2>&1
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3_sync -o r3_sync -- $BENCH --steps 20 --warmup 5 --sync-prologue > $O/prof_r3_sync.log 2>&1
python3 tools/trace_gaps.py $O/prof_r3_sync/r3_sync_kernel_trace.csv > $O/r3_step_timeline_sync_prologue.txt
tail -3 $O/profile_round_r3.log; head -3 $O/r3_step_timeline.txt; grep "last 10" $O/r3_step_timeline.txt $O/r3_step_timeline_sync_prologue.txt
