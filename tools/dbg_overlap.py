"""Diagnostic: the gradients of three batches with the heads' backward in line / on the side stream (model.overlap_heads_backward), per
pair the parameters that differ most -- and, with INLINE_ONLY=1, in line against in line: the run-to-run differences are bimodal (one bf16
rounding where the [CLS] rows' gradient joins the encoder's), which is what the bound of test_heads_backward_on_a_side_stream_... allows for."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_train_gpu as TT
from msa_amd.data import synthetic_batch, batch_to
m = TT.build(); m.eval()
DEV = "cuda"
batches = [batch_to(synthetic_batch(2, 16, 40, 24, vocab=TT.CFG["vocab"], seed=50 + i), DEV) for i in range(3)]
m._ensure_ready(torch.device(DEV, 0)); flat = m._flat
got = {}
INL = bool(os.environ.get("INLINE_ONLY"))
for rnd, overlap in enumerate((False, True, False, True)):
    key = overlap
    overlap = False if INL else overlap
    m.overlap_heads_backward = overlap
    gs = []
    for b in batches:
        flat.grads.zero_(); flat.stale.clear()
        out, _ = m(**b); out[0].mean().backward(); torch.cuda.synchronize()
        gs.append(flat.grads.clone())
    got.setdefault(key, []).append(gs)
def worst(a, c):
    bad = []
    for n in flat.order:
        o, k = flat.offset[n], flat.numel[n]
        x, y = a[o:o+k], c[o:o+k]
        d = float((x - y).norm()); nx = float(x.norm())
        if nx > 0: bad.append((d, d / nx, n))
    bad.sort(reverse=True)
    return [(f"{d:.2e}", f"{r:.2e}", n[-40:]) for d, r, n in bad[:4]]
for i, (a, c) in enumerate(zip(got[False][0] + got[False][1], got[True][0] + got[True][1])):
    print("pair", i, "rel", float((a - c).norm() / a.norm()), worst(a, c))
for i, (a, c) in enumerate(zip(got[False][0], got[False][1])):
    print("inline pair", i, "rel", float((a - c).norm() / a.norm()), worst(a, c))
