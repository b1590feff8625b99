"""Diagnostic: trainer.train_epoch for NM micro-batches with and without the lazy zero of the dense weights' gradients (AdamW.lazy_zero),
three runs each, all pairwise parameter distances: lazy against zero-filled must look like zero-filled against zero-filled."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_train_gpu as TT
from msa_amd.data import synthetic_batch, batch_to
from msa_amd import trainer as T
DEV = "cuda"
cfg = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
B = 4
NM = int(os.environ.get("NM", 8))
def run(lazy):
    m = TT.build(cfg, dropout=0.1)
    m.train(); m.manual_seed(3)
    args = T.default_args(train_batch_size=B, learning_rate=5e-4, mlm=True)
    opt, sched = T.build_optimizer(m, args, NM // 2, mode="hf")
    opt.lazy_zero = lazy
    pool = [batch_to(synthetic_batch(B, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i in range(4)]
    ret = T.train_epoch(args, m, None, opt, sched, device=DEV, batches=(pool[i % 4] for i in range(NM)))
    torch.cuda.synchronize()
    return m._flat.params.clone(), ret
rel = lambda a, c: float((a - c).norm() / a.norm())
runs = [(lz, run(lz)) for lz in (False, True, False, True, False, True)]
for i, (la, (pa, ra)) in enumerate(runs):
    for j, (lb, (pb, rb)) in enumerate(runs):
        if j > i:
            print(f"lazy {la} vs {lb}: params rel {rel(pa, pb):.2e}", flush=True)
for lz, (p, r) in runs:
    print("lazy", lz, "returns", [round(x, 4) for x in r])
