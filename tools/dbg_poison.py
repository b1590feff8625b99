"""Diagnostic: the parameters after NM micro-batches of trainer.train_epoch (MLM masking on, dropout on), twice on ordinary memory and once
with every torch.empty poisoned (tools/poison_empty.py): a kernel that reads memory it never wrote would move the poisoned run away from
the clean ones by more than they differ from each other (they do differ: the trajectory is chaotic in the fp32 atomics' order)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import test_train_gpu as TT
import poison_empty
from msa_amd.data import synthetic_batch, batch_to
from msa_amd import trainer as T
DEV = "cuda"
cfg = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
B = 4
NM = int(os.environ.get("NM", 8))
def run():
    m = TT.build(cfg, dropout=0.1)
    m.train(); m.manual_seed(3)
    args = T.default_args(train_batch_size=B, learning_rate=5e-4, mlm=True)
    opt, sched = T.build_optimizer(m, args, NM // 2, mode="hf")
    pool = [batch_to(synthetic_batch(B, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i in range(4)]
    ret = T.train_epoch(args, m, None, opt, sched, device=DEV, batches=(pool[i % 4] for i in range(NM)))
    torch.cuda.synchronize()
    return m, m._flat.params.clone(), ret
m0, p0, r0 = run()
m0b, p0b, r0b = run()
real = (torch.empty, torch.empty_like, torch.Tensor.new_empty)
poison_empty.install()
m1, p1, r1 = run()
torch.empty, torch.empty_like, torch.Tensor.new_empty = real
flat = m0._flat
for tag, a, c in (("clean vs clean", p0, p0b), ("clean vs poison", p0, p1)):
    bad = []
    for n in flat.order:
        o, kk = flat.offset[n], flat.numel[n]
        x, y = a[o:o+kk], c[o:o+kk]
        d = float((x - y).norm()); nx = float(x.norm())
        if not (d == d) or (nx > 0 and d / nx > 1e-4): bad.append((d / max(nx, 1e-30), n))
    bad.sort(reverse=True)
    print(tag, "params total rel", float((a - c).norm() / a.norm()), [(f"{r:.2e}", n[-40:]) for r, n in bad[:6]], flush=True)
print("returns", r0, r0b, r1)
