#!/bin/bash
# round 4, third GPU call: the 8-phase product kernel -- kernel tests, the previously failing parity tests, in-situ A/B, bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or dropout_hash" > $O/r4_pytest3a.log 2>&1; echo "rc $?" >> $O/r4_pytest3a.log; tail -12 $O/r4_pytest3a.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -k "bert_large or batch8 or trained_state or hook_fires or rccl_world1 or replayed_masks or standalone or cfg1_matches" > $O/r4_pytest3b.log 2>&1; echo "rc $?" >> $O/r4_pytest3b.log; tail -15 $O/r4_pytest3b.log | cut -c1-300
ROUNDS=7 STEPS=8 python tools/ab_step.py pers:MMBERT_NT_8PHASE=0 eight: > $O/r4_ab_8phase.log 2>&1; cat $O/r4_ab_8phase.log
python bench.py --no-cpu-baseline > $O/r4_bench_b.json 2> $O/r4_bench_b.err; tail -2 $O/r4_bench_b.err; cut -c1-300 $O/r4_bench_b.json
SHAPESET=bert-large MODES=0,1,8 timeout 600 python tools/bench_gemm.py > $O/r4_bert_large_gemm_modes2.log 2>&1; cat $O/r4_bert_large_gemm_modes2.log
