#!/bin/bash
# Round profile set (run on the GPU box): kernel stats + step timeline, HBM-traffic PMC passes, L2 hit / miss, MFMA-utilisation passes, NT-family
# SQ counters, the attention kernels' counters.  Every rocprofv3 --pmc pass is its own run with --kernel-trace only (no other trace domain).
#   bash tools/profile_round.sh r5     -> gpurun_out/prof_r5/*, summaries gpurun_out/r5_*.{csv,txt} (copy the ones to be judged to profiles/)
cd "$(dirname "$0")/.."
R=${1:-rX}
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only --no-reference-default --no-scores-fp32 --no-deterministic --no-dp-reference-legs --no-box-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o $R -- $BENCH --steps 20 --warmup 5 > gpurun_out/prof_$R.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${R}_$c -o pmc -- $BENCH --steps 2 --warmup 1 > gpurun_out/pmc_${R}_$c.log 2>&1
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_${R}_TCC -o pmc -- $BENCH --steps 2 --warmup 1 > gpurun_out/pmc_${R}_TCC.log 2>&1
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcm_${R}_$c -o pmc -- $BENCH --steps 2 --warmup 1 > gpurun_out/pmcm_${R}_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv \
  -d gpurun_out/pmcs_$R -o pmc -- $BENCH --steps 2 --warmup 1 > gpurun_out/pmcs_$R.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_${R}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${R}_WRITE_SIZE/pmc_counter_collection.csv > gpurun_out/${R}_pmc_hbm_traffic.csv
python3 tools/pmc_l2.py gpurun_out/pmc_${R}_TCC/pmc_counter_collection.csv > gpurun_out/${R}_pmc_l2_hit.csv
python3 tools/pmc_mfma.py gpurun_out/pmcm_${R}_ > gpurun_out/${R}_pmc_mfma_util.csv
python3 tools/pmc_sq.py gpurun_out/pmcs_$R/pmc_counter_collection.csv > gpurun_out/${R}_pmc_sq_gemm.csv
python3 tools/trace_gaps.py gpurun_out/prof_$R/${R}_kernel_trace.csv > gpurun_out/${R}_step_timeline.txt
python3 tools/tail_sequence.py gpurun_out/prof_$R/${R}_kernel_trace.csv > gpurun_out/${R}_tail_sequence.txt
python3 tools/tail_sequence.py gpurun_out/prof_$R/${R}_kernel_trace.csv --step > gpurun_out/${R}_step_sequence.txt
cp gpurun_out/prof_$R/${R}_kernel_stats.csv gpurun_out/${R}_kernel_stats.csv
bash tools/pmc_attn.sh > gpurun_out/pmc_attn_$R.log 2>&1
python3 tools/pmc_attn.py gpurun_out/pmc_attn_g > gpurun_out/${R}_pmc_attention.csv
# (raw traces are large: keep the summaries only)
rm -rf gpurun_out/prof_$R/*_kernel_trace.csv gpurun_out/pmc_${R}_*/pmc_kernel_trace.csv gpurun_out/pmcm_${R}_*/pmc_kernel_trace.csv gpurun_out/pmcs_$R/pmc_kernel_trace.csv
ls gpurun_out/${R}_*
