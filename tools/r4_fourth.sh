#!/bin/bash
# round 4, fourth GPU call: persistent 8-phase kernel + the seam-wait fix of the ring-persistent kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/r4_pytest4a.log 2>&1; echo "rc $?" >> $O/r4_pytest4a.log; tail -8 $O/r4_pytest4a.log | cut -c1-300
ROUNDS=7 STEPS=8 python tools/ab_step.py l1: l0:MMBERT_NT_8PHASE=0 l2:MMBERT_NT_8PHASE=2 l3:MMBERT_NT_8PHASE=3 > $O/r4_ab_8phase_levels.log 2>&1; cat $O/r4_ab_8phase_levels.log
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_kernels_gpu.py > $O/r4_pytest4b.log 2>&1; echo "rc $?" >> $O/r4_pytest4b.log; tail -8 $O/r4_pytest4b.log | cut -c1-300
python bench.py --no-cpu-baseline > $O/r4_bench_c.json 2> $O/r4_bench_c.err; cut -c1-300 $O/r4_bench_c.json
