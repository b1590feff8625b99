python -m pytest tests -q -m gpu 2>&1 | tail -2
SPLITS=0,2,3 python tools/bench_tn.py 2>&1 | tail -3
for m in 0 ; do MMBERT_NT_MODE=$m python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500; done
