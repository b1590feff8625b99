#!/usr/bin/env python3
"""Round 6: the side streams and the deferred few-row weight gradients at the HEADLINE shape (12 layers, d = 768, batch 16, T = 50, A = V = 500), deterministic
mode: N seeded train steps with everything on against the same run with every stream off (late weight gradients stay on in both: their
top-layer QKV gradient sums in another order otherwise) -- losses and every parameter must be bit-identical; repeated REPS times with allocator
churn in between (a missing wait shows up as a rare difference).
    python tools/dbg_streams.py [steps=6] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]


def run(streams: bool):
    torch.manual_seed(0)
    m = MMBertForPretraining(MMBertConfig())
    m.bert.set_joint_embeddings("mosei")
    m.to(dev).train()
    m.manual_seed(1234)
    m.async_prologue = True
    m.wgrad_side_stream = m.heads_side_stream = m.pairs_side_stream = streams
    ops.SIDE_TRANSPOSES = streams
    opt, sched = build_optimizer(m, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
    losses = []
    for i in range(steps):
        out, _ = m(**pool[i % 4])
        out[0].mean().backward()
        opt.step(); sched.step(); opt.zero_grad()
        losses.append(out[0].detach())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), {n: q.detach().clone().cpu() for n, q in m.named_parameters()}


ops.set_deterministic(True)
bad = 0
ref = run(False)
for rep in range(reps):
    got = run(True)
    same = torch.equal(got[0], ref[0]) and all(torch.equal(got[1][n], ref[1][n]) for n in ref[1])
    if not same:
        bad += 1
        worst = max(((got[1][n].double() - ref[1][n].double()).abs().max().item(), n) for n in ref[1])
        print("rep", rep, "DIFFERENT: losses", got[0].tolist(), ref[0].tolist(), "worst parameter", worst)
    junk = [torch.randn(1 + 37 * rep, 1000 + rep, device="cuda") for _ in range(3)]
    del junk
print("losses", ref[0].tolist())
print("different runs", bad, "of", reps)
