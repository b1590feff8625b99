#!/bin/bash
# Round-3 profile set: tools/profile_round.sh (kernel stats, timeline, PMC traffic / MFMA util / SQ), attention counters, the sync-prologue
# timeline, and the NT K-step ablation log (stamped builds: run tools/stamp_gemm.py --build for every STAMP_VARIANT first, here).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
bash tools/profile_round.sh r3 > $O/profile_round_r3.log 2>&1
bash tools/pmc_attn.sh > $O/pmc_attn_r3.log 2>&1
python3 tools/pmc_attn.py $O/pmc_attn_g > $O/r3_pmc_attention.csv 2>&1
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3_sync -o r3_sync -- $BENCH --steps 20 --warmup 5 --sync-prologue > $O/prof_r3_sync.log 2>&1
python3 tools/trace_gaps.py $O/prof_r3_sync/r3_sync_kernel_trace.csv > $O/r3_step_timeline_sync_prologue.txt
{
  echo "# tools/stamp_gemm.py, MODE=7 (persistent NT kernel, 224-row tile), M = 18400, one MI355X, round 3: compile-time ablations of the K step"
  echo "# (STAMP_VARIANT builds; timing only, outputs wrong by construction).  Per-tile clocks: wait for stage 0 / K loop / epilogue."
  for v in "" aloads1 aloads0 noloads nomfma nofrags nofrags_noloads nomfma_noloads; do
    for d in 0 1 32; do
      if [ "$d" != "0" ] && [ "$v" != "" ] && [ "$v" != "nomfma" ]; then continue; fi
      echo "== STAMP_VARIANT=${v:-product-K-step} DBG=$d   (DBG 1: zero-record descriptors, every load dropped; 32: whole-cache-line source pattern)"
      STAMP_VARIANT=$v DBG=$d MODE=7 python3 tools/stamp_gemm.py 2>&1 | grep -v "amdgpu.ids\|timing-only" | cut -c1-230
    done
  done
} > $O/r3_stamp_nt_ablation.log 2>&1
tail -3 $O/profile_round_r3.log; head -3 $O/r3_step_timeline.txt; grep "last 10" $O/r3_step_timeline.txt $O/r3_step_timeline_sync_prologue.txt; wc -l $O/r3_stamp_nt_ablation.log
