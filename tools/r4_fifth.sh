#!/bin/bash
# round 4, fifth GPU call: single-tile vs multi-tile form of the 8-phase kernel, weight gradients on a side stream again, full GPU suite
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > $O/r4_pytest5a.log 2>&1; echo "rc $?" >> $O/r4_pytest5a.log; tail -6 $O/r4_pytest5a.log | cut -c1-300
ROUNDS=7 STEPS=8 python tools/ab_step.py single: multi:MMBERT_NT_8PHASE_FORM=multi wgrad_side:attr.overlap_wgrad=True l0:MMBERT_NT_8PHASE=0 > $O/r4_ab_8phase_form.log 2>&1; cat $O/r4_ab_8phase_form.log
python bench.py > $O/r4_bench_d.json 2> $O/r4_bench_d.err; cut -c1-300 $O/r4_bench_d.json
timeout 2400 python -m pytest tests -m gpu -q > $O/r4_pytest5b.log 2>&1; echo "rc $?" >> $O/r4_pytest5b.log; tail -6 $O/r4_pytest5b.log | cut -c1-300
