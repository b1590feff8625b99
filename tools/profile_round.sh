#!/bin/bash
# Round-end evidence on the GPU box: kernel trace + stats of the bench command, two HBM-traffic PMC passes, four MFMA PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-fused --no-dense-reference --no-train-only"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o final --output-format csv -- $BENCH --steps 20 --warmup 5 > $R/gpurun_out/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmcf_$c -o pmc --output-format csv -- $BENCH --steps 2 --warmup 1 > $R/gpurun_out/pmcf_$c.log 2>&1
  echo "$c rc=$?"
done
python3 $R/bench.py --steps 30 --warmup 8 > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
tail -c 600 $R/gpurun_out/bench_final.json
ls $R/gpurun_out/prof_final
