cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16; do
  rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc4_$c -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fused --no-box-probe > $GRAFT_REPO_ROOT/gpurun_out/pmc4_$c.log 2>&1
  echo "$c rc=$?"; ls $GRAFT_REPO_ROOT/gpurun_out/pmc4_$c | head -3
done
