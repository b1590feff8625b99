"""Stand-alone timing of mmbert_pair_proj_fwd / mmbert_pair_proj_bwd at the headline shapes (16 x 500 rows, D = 35 / 74, H = 768)
+ a check against torch (fp32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msa_amd import ops
dev = "cuda"
B, P, T, H = 16, 500, 50, 768


def med(fn, n=20):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for D in (35, 74):
    g = torch.Generator().manual_seed(D)
    feat = torch.randn(B, P, D, generator=g).to(dev); W = (torch.randn(H, D, generator=g) * 0.2).to(dev); b = (torch.randn(H, generator=g) * 0.1).to(dev)
    out = torch.zeros(B * (T + P), H, device=dev, dtype=torch.bfloat16)
    ops.pair_proj_fwd(feat, W, b, out, T)
    ref = torch.relu(feat @ W.t() + b).bfloat16()
    got = out.view(B, T + P, H)[:, T:]
    err = float((got.float() - ref.float()).abs().max())
    t = med(lambda: ops.pair_proj_fwd(feat, W, b, out, T))
    print(f"fwd D {D}: {t:.1f} us   max |err| vs torch {err:.3e} (scale {float(ref.float().abs().max()):.2f})")
    dJ = torch.randn(B * (T + P), H, generator=g).bfloat16().to(dev)
    dW, db = torch.zeros(H, D, device=dev), torch.zeros(H, device=dev)
    ops.pair_proj_bwd(feat, out, dJ, T, dW, db)
    gy = (dJ.float().view(B, T + P, H)[:, T:] * (got.float() > 0)).double()
    rW = torch.einsum("bph,bpk->hk", gy, feat.double()); rb = gy.sum((0, 1))
    eW = float((dW.double() - rW).abs().max()); eb = float((db.double() - rb).abs().max())
    dW2, db2 = torch.zeros_like(dW), torch.zeros_like(db)
    ops.pair_proj_bwd(feat, out, dJ, T, dW2, db2)
    same = bool(torch.equal(dW, dW2) and torch.equal(db, db2))
    t = med(lambda: ops.pair_proj_bwd(feat, out, dJ, T, dW, db))
    print(f"bwd D {D}: {t:.1f} us   max |err| vs float64: dW {eW:.3e} (scale {float(rW.abs().max()):.1f}), db {eb:.3e} (scale {float(rb.abs().max()):.1f}); "
          f"two runs bit-identical: {same}")
