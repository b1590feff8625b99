#!/usr/bin/env python3
"""How far ahead of the GPU the host runs in the train step: time for 8 step() calls to RETURN (enqueue) vs time until the GPU is done."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
model.async_prologue = os.environ.get("ASYNC_PROLOGUE", "1") != "0"
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
def step(i):
    out, _ = model(**pool[i % 4]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()
for i in range(5): step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(8): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / 8:.2f} ms/step, GPU done {1e3 * (t2 - t0) / 8:.2f} ms/step")
# one step at a time from an empty queue: the host's own cost of a step (no back-pressure from a full queue), forward and backward apart
import statistics
fw, bw, op_ = [], [], []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, _ = model(**pool[i % 4]); loss = out[0].mean()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(); sched.step(); opt.zero_grad()
    t3 = time.perf_counter()
    fw.append(t1 - t0); bw.append(t2 - t1); op_.append(t3 - t2)
print(f"host cost of one step from an empty queue: forward {1e3 * statistics.median(fw):.2f} ms, backward {1e3 * statistics.median(bw):.2f} ms, "
      f"optimizer {1e3 * statistics.median(op_):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(4): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
