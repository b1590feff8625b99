#!/usr/bin/env python3
"""Where does a deterministic-mode training trajectory first differ between two identical runs?  Two fresh models, same seeds, the same
micro-batches (the trained-state test's configuration); after every backward the gradient buffers and after every step the parameters
are compared bit for bit; prints the first mismatching micro-batch and the parameters that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mmbert_oracle as O       # (a debugging tool: the oracle only provides the seeded initial weights)
from msa_amd.data import synthetic_batch, batch_to
from msa_amd import trainer as T, ops
from msa_amd.model import MMBertConfig, MMBertForPretraining
DEV = "cuda"
cfg = dict(hidden=768, layers=int(os.environ.get("LAYERS", 2)), heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
B = int(os.environ.get("B", 8)); n_micro = int(os.environ.get("N", 52)); REPS = int(os.environ.get("REPS", 4))
def build():
    c = MMBertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                     intermediate_size=cfg["intermediate"], hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    m = MMBertForPretraining(c); m.bert.set_joint_embeddings("mosei"); m.bert.jointEmbeddings.dropout_prob = 0.5
    m.load_state_dict(O.seeded_params(cfg), strict=False)
    return m.to(DEV)
pool = [batch_to(synthetic_batch(B, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i in range(4)]
ops.set_deterministic(os.environ.get("DET", "1") == "1")
def run():
    m = build(); m.train(); m.manual_seed(3)
    args = T.default_args(train_batch_size=B, learning_rate=5e-4, mlm=True)
    opt, sched = T.build_optimizer(m, args, n_micro // 2, mode="hf")
    trace = []
    for step in range(n_micro):
        out, _ = m(**pool[step % 4]); out[0].mean().backward()
        torch.cuda.synchronize()
        trace.append(("grad", step, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, float(out[0])))
        if T.should_step(step, args.gradient_accumulation_step, True):
            opt.step(); sched.step(); opt.zero_grad()
            torch.cuda.synchronize()
            trace.append(("param", step, {n: p.detach().clone() for n, p in m.named_parameters()}, 0.0))
    return trace
ref = run()
for rep in range(1, REPS):
    cur = run()
    first = None
    for (k0, s0, d0, l0), (k1, s1, d1, l1) in zip(ref, cur):
        bad = [n for n in d0 if not torch.equal(d0[n], d1[n])]
        if bad or l0 != l1:
            first = (k0, s0, bad, l0, l1); break
    if first is None:
        print(f"repetition {rep}: {len(ref)} checkpoints bit-identical (final loss {ref[-2][3] if ref[-1][0]=='param' else ref[-1][3]:.6f})")
    else:
        k, s, bad, l0, l1 = first
        print(f"repetition {rep}: first difference at micro-batch {s} ({k}); loss {l0!r} vs {l1!r}; {len(bad)} tensors differ:")
        for n in bad[:12]:
            d = (ref[[i for i, t in enumerate(ref) if t[0] == k and t[1] == s][0]][2][n].float() - cur[[i for i, t in enumerate(cur) if t[0] == k and t[1] == s][0]][2][n].float()).abs()
            print(f"    {n:60s} max |diff| {float(d.max()):.3e}  elements {int((d > 0).sum())}")
