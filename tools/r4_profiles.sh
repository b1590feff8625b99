#!/bin/bash
# round 4 profile set: kernel stats + timeline + PMC (traffic, MFMA util, SQ), attention counters, host enqueue time, start-of-backward region
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
bash tools/profile_round.sh r4 > $O/profile_round_r4.log 2>&1
bash tools/pmc_attn.sh > $O/pmc_attn_r4.log 2>&1
python3 tools/pmc_attn.py $O/pmc_attn_g > $O/r4_pmc_attention.csv 2>&1
python3 tools/host_time.py > $O/r4_host_time.txt 2>&1
python3 tools/tail_region.py > $O/r4_tail_region.txt 2>&1
tail -3 $O/profile_round_r4.log; head -40 $O/r4_step_timeline.txt | cut -c1-200; cat $O/r4_host_time.txt | tail -5; tail -3 $O/r4_tail_region.txt
