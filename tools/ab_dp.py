#!/usr/bin/env python3
"""Same-process A/B of the data-parallel wrapper's own cost on ONE GPU (RCCL world of 1): plain step / dynamic tile queue only /
hooks without collectives / the full wrapper."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
from msa_amd import parallel
parallel.init_from_env(force=True)
import torch
from msa_amd import ops
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
model.async_prologue = True
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
dp = parallel.DataParallel(model, opt, bucket_mb=float(os.environ.get("BUCKET_MB", 32)), force_dynamic_queue=True)
hook = model.grad_hook
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
def step(i, mode):
    ops.dynamic_tile_queue = mode != "plain"
    model.grad_hook = hook if mode in ("hooks", "full") else None
    dp.bucketer.enabled = mode == "full"
    out, _ = model(**pool[i % 4]); out[0].mean().backward()
    if mode in ("hooks", "full"): dp.finish_backward()
    opt.step(); sched.step(); opt.zero_grad()
modes = ["plain", "queue", "hooks", "full"]
for m in modes:
    for i in range(3): step(i, m)
torch.cuda.synchronize()
ts = {m: [] for m in modes}
for r in range(5):
    for m in modes:
        step(0, m); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8): step(i, m)
        torch.cuda.synchronize(); ts[m].append((time.perf_counter() - t0) / 8 * 1e3)
base = sorted(ts["plain"])[2]
for m in modes:
    t = sorted(ts[m]); print(f"{m:6s} median {t[2]:7.3f} ms/step (min {t[0]:.3f} max {t[-1]:.3f})  x{t[2] / base:.4f}", flush=True)
torch.distributed.destroy_process_group()
