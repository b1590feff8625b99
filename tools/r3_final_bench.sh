#!/bin/bash
# Round-3 bench records: the contract run (with the CPU leg), the data-parallel wrapper forced on one GPU, the torchrun launch the driver
# uses at world size 1, and BASELINE configs[3] (A = V = 1375, batch 4) in its 3-pass and fused forms.
cd "$(dirname "$0")/.."
O=gpurun_out
python bench.py > $O/r3_bench_final.json 2> $O/r3_bench_final.err
python bench.py --no-cpu-baseline --force-dp > $O/r3_bench_forcedp.json 2> $O/r3_bench_forcedp.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/r3_bench_torchrun_world1.json 2> $O/r3_bench_torchrun_world1.err
python bench.py --no-cpu-baseline --no-kernel-timing --batch 4 --pair 1375 --no-dense-reference --no-train-only > $O/r3_bench_longfusion.json 2> $O/r3_bench_longfusion.err
python - <<PY
import json
for n in ("final", "forcedp", "torchrun_world1", "longfusion"):
    try:
        r = json.load(open("$O/r3_bench_%s.json" % n))
        print(n, r["value"], r["ms_per_step"], r.get("ms_per_step_instrumented"), {k: r[k]["value"] for k in ("dense_backward_reference", "train_only", "fused1050") if k in r},
              r.get("roofline", {}).get("frac"), r.get("roofline", {}).get("traffic"), (r.get("cpu_baseline") or {}).get("value"), (r.get("cpu_baseline") or {}).get("cores"))
    except Exception as e:
        print(n, "FAILED", e)
PY
