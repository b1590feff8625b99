#!/usr/bin/env python3
"""Debug: what q_limit does the top layer get on the bench batches, and what does it save in the top layer's attention backward?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig())
model.bert.set_joint_embeddings("mosei")
model.to(dev).train()
model.manual_seed(1)
batch = batch_to(synthetic_batch(16, 50, 500, 500, seed=2), dev)
seen = []
orig_q, orig_b = ops.attn_q_limit, ops.attn_bwd
def q(rows, layout):
    out = orig_q(rows, layout)
    seen.append(("qlim", out.clone(), layout.kv_len.clone() if hasattr(layout, "kv_len") else None, rows.numel()))
    return out
times = []
def b(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_b(*a, **k); e1.record()
    times.append((k.get("q_limit") is not None, e0, e1))
    return r
ops.attn_q_limit, ops.attn_bwd = q, b
for it in range(3):
    times.clear(); seen.clear()
    out, _ = model(**batch)
    out[0].mean().backward()
    torch.cuda.synchronize()
ql = seen[0][1].cpu()
print("rows in R:", seen[0][3], "q_limit min/mean/max:", int(ql.min()), float(ql.float().mean()), int(ql.max()), "kv_len mean", float(seen[0][2].float().mean()) if seen[0][2] is not None else None)
print("per-layer attn_bwd ms (top layer first):", [(lim, round(e0.elapsed_time(e1), 4)) for lim, e0, e1 in times])
model.top_layer_query_limit = False
times.clear()
out, _ = model(**batch)
out[0].mean().backward()
torch.cuda.synchronize()
print("without the limit:", [(lim, round(e0.elapsed_time(e1), 4)) for lim, e0, e1 in times][:3])
