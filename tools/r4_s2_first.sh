#!/bin/bash
# round 4, session 2, first GPU call: full -m gpu suite and the default bench line at HEAD
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r4s2_pytest1.log 2>&1; echo "rc $?" >> $O/r4s2_pytest1.log; tail -8 $O/r4s2_pytest1.log | cut -c1-300
python bench.py > $O/r4s2_bench1.json 2> $O/r4s2_bench1.err; cut -c1-600 $O/r4s2_bench1.json
