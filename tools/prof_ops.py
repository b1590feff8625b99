#!/usr/bin/env python3
"""torch.profiler view of one train step: which ATen ops (glue outside the HIP library) cost device time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from msa_amd import parallel
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
cfg = MMBertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
model = MMBertForPretraining(cfg); model.bert.set_joint_embeddings("mosei"); model.set_alpha_beta(1.0, 1.0); model.to(dev); model.train(); model.manual_seed(1)
model.return_scores = True
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), num_train_optimization_steps=100)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, vocab=30522, seed=1 + i), dev) for i in range(2)]
def step(i):
    out, _ = model(**pool[i % 2]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(3): step(i)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0: rows.append((t / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"device time per step {tot/1e3:.2f} ms")
for t, c, k, sh in rows[:70]:
    print(f"{t:9.1f} us {c:6.1f}x  {k[:44]:44s} {sh}")
