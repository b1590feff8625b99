#!/usr/bin/env python3
"""Host time of one train step from an empty queue, by autograd node (forward / backward of the trunk, the MLM head, the heads) and the
encoder's backward loop -- wall time of the Python bodies (they only enqueue), median over steps.  python tools/host_breakdown.py [composite=1]"""
import os, sys, time, statistics, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import model as MM
from msa_amd.data import synthetic_batch, batch_to
from msa_amd.model import MMBertConfig, MMBertForPretraining
from msa_amd.trainer import build_optimizer, default_args
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = MMBertForPretraining(MMBertConfig()); model.bert.set_joint_embeddings("mosei"); model.to(dev).train(); model.manual_seed(1234)
model.async_prologue = True
model.composite_layers = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
opt, sched = build_optimizer(model, default_args(train_batch_size=16, learning_rate=5e-5), 1000)
pool = [batch_to(synthetic_batch(16, 50, 500, 500, seed=1 + i), dev) for i in range(4)]
acc = collections.defaultdict(list)
cur = collections.defaultdict(float)


def timed(name, fn):
    def inner(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            cur[name] += time.perf_counter() - t0
    return inner


for cls, tag in ((MM._TrunkFn, "trunk"), (MM._MLMHeadFn, "mlm_head"), (MM._HeadsFn, "heads")):
    cls.forward = staticmethod(timed(tag + ".forward", cls.forward))
    cls.backward = staticmethod(timed(tag + ".backward", cls.backward))
MM._EncoderFn.run_forward = staticmethod(timed("  encoder.run_forward", MM._EncoderFn.run_forward))
MM._EncoderFn.run_backward = staticmethod(timed("  encoder.run_backward", MM._EncoderFn.run_backward))
MM._EncoderFn._last_layer_sparse = staticmethod(timed("    top layer sparse", MM._EncoderFn._last_layer_sparse))
for i in range(16):
    torch.cuda.synchronize()
    cur.clear()
    t0 = time.perf_counter()
    out, _ = model(**pool[i % 4]); loss = out[0].mean()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(); sched.step(); opt.zero_grad()
    t3 = time.perf_counter()
    if i >= 4:
        acc["forward total"].append(t1 - t0); acc["backward total"].append(t2 - t1); acc["optimizer"].append(t3 - t2)
        for k, v in cur.items():
            acc[k].append(v)
print(f"composite_layers = {model.composite_layers}")
for k in sorted(acc, key=lambda s: s.strip()):
    print(f"{k:28s} {1e3 * statistics.median(acc[k]):7.3f} ms")
