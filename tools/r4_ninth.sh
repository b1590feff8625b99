#!/bin/bash
# round 4, ninth GPU call: 8-phase kernel on 128-row tiles (reference-default shapes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > $O/r4_pytest9.log 2>&1; echo "rc $?" >> $O/r4_pytest9.log; tail -6 $O/r4_pytest9.log | cut -c1-300
SHAPESET=bert-large MODES=0,1,8 SHAPES=o,w2,dy1,dx,dctx timeout 600 python tools/bench_gemm.py > $O/r4_bert_large_gemm_modes3.log 2>&1; cat $O/r4_bert_large_gemm_modes3.log
python bench.py --preset reference-default > $O/r4_refdef_c.json 2> $O/r4_refdef_c.err; cut -c1-200 $O/r4_refdef_c.json
MMBERT_NT_8PHASE_BM128=0 python bench.py --preset reference-default > $O/r4_refdef_c0.json 2> $O/r4_refdef_c0.err; cut -c1-200 $O/r4_refdef_c0.json
python bench.py --preset reference-default > $O/r4_refdef_c2.json 2> $O/r4_refdef_c2.err; cut -c1-200 $O/r4_refdef_c2.json
