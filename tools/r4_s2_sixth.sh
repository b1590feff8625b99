#!/bin/bash
# round 4, session 2, sixth GPU call: lazy zero of the dense weights' gradients -- tests, then the same-process A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py -m gpu -q -x > $O/r4s2_pytest6.log 2>&1; echo "rc $?" >> $O/r4s2_pytest6.log; tail -25 $O/r4s2_pytest6.log | cut -c1-400
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "adamw" >> $O/r4s2_pytest6.log 2>&1; tail -3 $O/r4s2_pytest6.log
python - <<'PY' > $O/r4s2_ab_lazy_zero.log 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
model._ensure_ready(dev)
opts = {}
for lazy in (True, False):
    opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
    opt.lazy_zero = lazy
    opts[lazy] = (opt, sched)
def step(i, opt, sched):
    out, _ = model(**pool[i % 4]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()
res = {True: [], False: []}
for lazy in (True, False):
    model._flat.settle()
    for i in range(6): step(i, *opts[lazy])
torch.cuda.synchronize()
for rnd in range(7):
    for lazy in (True, False):
        model._flat.settle()
        step(0, *opts[lazy]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8): step(i, *opts[lazy])
        torch.cuda.synchronize()
        res[lazy].append((time.perf_counter() - t0) / 8 * 1e3)
for lazy in (True, False):
    r = sorted(res[lazy]); print(f"lazy_zero={lazy}: median {r[len(r)//2]:.3f} ms/step (min {r[0]:.3f} max {r[-1]:.3f})")
PY
cat $O/r4s2_ab_lazy_zero.log
