#!/bin/bash
# round 4, second GPU call: the whole GPU suite (no -x), the NT yardstick, the bert-large shapes on the 128x128 kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > $O/r4_pytest2.log 2>&1; echo "pytest rc $?" >> $O/r4_pytest2.log
tail -25 $O/r4_pytest2.log | cut -c1-300
timeout 900 python tools/yardstick/run_yardstick.py > $O/r4_nt_yardstick.log 2>&1; cat $O/r4_nt_yardstick.log | cut -c1-400
SHAPESET=bert-large MODES=0,1 timeout 600 python tools/bench_gemm.py > $O/r4_bert_large_gemm_modes.log 2>&1; cat $O/r4_bert_large_gemm_modes.log
