#!/usr/bin/env python3
"""GPU time of the step's "small-kernel" region WITHOUT a profiler: from the cross-entropy launch (right behind the vocabulary GEMM) to the
first dense attention backward (the heads forward + backward, the MLM head's sparse backward, the top layer's sparse backward), measured
with two events per step.  The kernel-trace sum of the same region's launches (profiles) says how much of it is gaps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
marks = {}
def wrap(name, key):
    orig = getattr(ops, name)
    def f(*a, **k):
        if key not in marks:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks[key] = e
        return orig(*a, **k)
    setattr(ops, name, f)
wrap(sys.argv[1] if len(sys.argv) > 1 else "ce_fwd", "a")
wrap("attn_bwd", "b")
def step(i):
    out, _ = model(**pool[i % 4]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()
for i in range(6): step(i); marks.clear()
torch.cuda.synchronize()
ts, tot = [], []
for i in range(16):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); step(i); s1.record()
    torch.cuda.synchronize()
    ts.append(marks["a"].elapsed_time(marks["b"]) * 1e3); tot.append(s0.elapsed_time(s1) * 1e3); marks.clear()
ts.sort(); tot.sort()
print(f"CE launch -> first attention backward: median {ts[len(ts) // 2]:.0f} us (min {ts[0]:.0f}, max {ts[-1]:.0f}) of a {tot[len(tot) // 2]:.0f}-us step")
