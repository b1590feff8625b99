"""Stand-alone timing of mmbert_pair_proj_fwd at the headline shapes (16 x 500 rows, D = 35 / 74, H = 768) + a check against torch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msa_amd import ops
dev = "cuda"
B, P, T, H = 16, 500, 50, 768
for D in (35, 74):
    g = torch.Generator().manual_seed(D)
    feat = torch.randn(B, P, D, generator=g).to(dev); W = (torch.randn(H, D, generator=g) * 0.2).to(dev); b = (torch.randn(H, generator=g) * 0.1).to(dev)
    out = torch.zeros(B * (T + P), H, device=dev, dtype=torch.bfloat16)
    ops.pair_proj_fwd(feat, W, b, out, T)
    ref = torch.relu(feat @ W.t() + b).bfloat16()
    got = out.view(B, T + P, H)[:, T:]
    err = float((got.float() - ref.float()).abs().max())
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.pair_proj_fwd(feat, W, b, out, T); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"D {D}: {ts[len(ts)//2]:.1f} us   max |err| vs torch {err:.3e} (scale {float(ref.float().abs().max()):.2f})")
