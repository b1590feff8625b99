#!/usr/bin/env python3
"""Which source lines of msa_amd launch the remaining ATen kernels of the train step (torch.profiler, one step): count and device time."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(2)]
def step(i):
    out, _ = model(**pool[i % 2]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()
for i in range(4): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(0); torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0) or 0
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:110]))
tot = 0.0
for t, n, name, shp in sorted(rows, reverse=True)[:60]:
    print(f"{t:8.1f} us {n:3d}x  {name:20s} {shp}")
    tot += t
print(f"aten self device time listed: {tot:.0f} us")
