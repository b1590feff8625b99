#!/usr/bin/env python3
"""Same-process, interleaved A/B of the reference's DEFAULT model (bert-large, T = P = 40, batch 32: REF:train.py:28,32,38) under model
attribute / environment toggles -- tools/ab_step.py for the other shape.
    python tools/ab_refdef.py paired: deferred:attr.defer_wgrads=True"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
variants = []
for a in sys.argv[1:]:
    name, _, envs = a.partition(":")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
steps, rounds = int(os.environ.get("STEPS", 20)), int(os.environ.get("ROUNDS", 4))      # long windows: see tools/ab_step.py
dev = torch.device("cuda", 0)
torch.manual_seed(0)
L, H, heads, I, V, T, B = 24, 1024, 16, 4096, 30522, 40, 32
model = MMBertForPretraining(MMBertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=I))
model.bert.set_joint_embeddings("mosei"); model.set_alpha_beta(1.0, 1.0); model.to(dev).train(); model.manual_seed(4321)
model.async_prologue = True
opt, sched = build_optimizer(model, default_args(train_batch_size=B, learning_rate=5e-5), 100000)
pool = [batch_to(synthetic_batch(B, T, T, T, vocab=V, seed=50 + i), dev) for i in range(4)]
all_keys = {k for _, env in variants for k in env}
attr_defaults = {k[5:]: getattr(model, k[5:], None) for k in all_keys if k.startswith("attr.")}


def step(i):
    out, _ = model(**pool[i % 4]); out[0].mean().backward(); opt.step(); sched.step(); opt.zero_grad()


def setenv(env):
    for k in all_keys:
        os.environ.pop(k, None)
    for k, v in attr_defaults.items():
        setattr(model, k, v)
    for k, v in env.items():
        if k.startswith("attr."):
            setattr(model, k[5:], eval(v))
        else:
            os.environ[k] = v


ts = {n: [] for n, _ in variants}
for n, env in variants:
    setenv(env)
    for i in range(3): step(i)
torch.cuda.synchronize()
for r in range(rounds):
    for n, env in variants:
        setenv(env); step(0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps): step(i)
        torch.cuda.synchronize()
        ts[n].append((time.perf_counter() - t0) / steps * 1e3)
base = None
for n, _ in variants:
    r = sorted(ts[n]); med = r[len(r) // 2]
    base = base or med
    print(f"{n:24s} median {med:8.3f} ms/step  (min {r[0]:.3f} max {r[-1]:.3f})  {B / med * 1e3:8.1f} samples/s   x{med / base:.4f}")
