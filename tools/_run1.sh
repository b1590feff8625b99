timeout 900 python -m pytest tests -q -m gpu -k "model or train" 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
