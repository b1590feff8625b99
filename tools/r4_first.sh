#!/bin/bash
# round 4, first GPU call: the GPU suite with the new parity tests, the bench with the reference-default leg, the forced-DP diagnostics,
# per-shape group_m A/B of the persistent NT kernel in the train step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
free -g > $O/r4_box.txt; nproc >> $O/r4_box.txt
timeout 3000 python -m pytest tests -m gpu -x -q > $O/r4_pytest.log 2>&1; echo "pytest rc $?" >> $O/r4_pytest.log
tail -15 $O/r4_pytest.log
python bench.py > $O/r4_bench_a.json 2> $O/r4_bench_a.err; tail -3 $O/r4_bench_a.err
python bench.py --force-dp --rank-report --no-cpu-baseline --no-fused --no-dense-reference --no-train-only > $O/r4_bench_forcedp.json 2> $O/r4_bench_forcedp.err; tail -5 $O/r4_bench_forcedp.err
ROUNDS=5 STEPS=8 python tools/ab_step.py base: \
  qkv1:MMBERT_NT_GM_TABLE=2304:768:1=1 qkv4:MMBERT_NT_GM_TABLE=2304:768:1=4 qkv6:MMBERT_NT_GM_TABLE=2304:768:1=6 qkv8:MMBERT_NT_GM_TABLE=2304:768:1=8 qkv16:MMBERT_NT_GM_TABLE=2304:768:1=16 \
  up4:MMBERT_NT_GM_TABLE=3072:768:3=4 up6:MMBERT_NT_GM_TABLE=3072:768:3=6 up8:MMBERT_NT_GM_TABLE=3072:768:3=8 up16:MMBERT_NT_GM_TABLE=3072:768:3=16 \
  dg1:MMBERT_NT_GM_TABLE=3072:768:8=1 dg4:MMBERT_NT_GM_TABLE=3072:768:8=4 dg6:MMBERT_NT_GM_TABLE=3072:768:8=6 dg12:MMBERT_NT_GM_TABLE=3072:768:8=12 \
  > $O/r4_ab_group_m.log 2>&1
cat $O/r4_ab_group_m.log
cut -c1-600 $O/r4_bench_a.json
