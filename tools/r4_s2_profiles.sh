#!/bin/bash
# round 4, session 2: the final profile set of the round (kernel stats, timeline, PMC traffic / MFMA / SQ, attention counters, host time, region) + bench records
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r4s2_pytest_final.log 2>&1; echo "rc $?" >> $O/r4s2_pytest_final.log; tail -4 $O/r4s2_pytest_final.log | cut -c1-300
bash tools/profile_round.sh r4 > $O/profile_round_r4.log 2>&1
bash tools/pmc_attn.sh > $O/pmc_attn_r4.log 2>&1
python3 tools/pmc_attn.py $O/pmc_attn_g > $O/r4_pmc_attention.csv 2>&1
python3 tools/host_time.py > $O/r4_host_time.txt 2>&1
python3 tools/tail_region.py > $O/r4_tail_region.txt 2>&1
python bench.py > $O/r4_bench_final.json 2> $O/r4_bench_final.err; cut -c1-260 $O/r4_bench_final.json
python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default --no-kernel-timing > $O/r4_bench_sustained.json 2> $O/r4_bench_sustained.err; cut -c1-260 $O/r4_bench_sustained.json
python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only --no-reference-default > $O/r4_bench_forcedp.json 2> $O/r4_bench_forcedp.err; cut -c1-200 $O/r4_bench_forcedp.json
python bench.py --preset reference-default --no-cpu-baseline > $O/r4_bench_reference_default.json 2> $O/r4_bench_reference_default.err; cut -c1-300 $O/r4_bench_reference_default.json
head -4 $O/r4_step_timeline.txt; tail -3 $O/r4_host_time.txt | head -2; tail -1 $O/r4_tail_region.txt
