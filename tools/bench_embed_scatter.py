import os, sys, torch
sys.path.insert(0, os.getcwd())
from msa_amd import ops
dev = "cuda"
B, T, H, V = 48, 50, 768, 30522
n = B * T
g = torch.Generator(device="cpu").manual_seed(0)
ids = torch.randint(1, V, (n,), generator=g).to(dev)
tts = torch.zeros(n, dtype=torch.int64, device=dev)
d = torch.randn(n, H, generator=g).bfloat16().to(dev)
gw, gt, gp = torch.zeros(V, H, device=dev), torch.zeros(2, H, device=dev), torch.zeros(512, H, device=dev)
ref_w = torch.zeros(V, H, device=dev).index_add_(0, ids, d.float())
for sl in (1, 2, 3, 4, 6, 8):
    os.environ["MMBERT_EMBED_SLICES"] = str(sl)
    gw.zero_(); gt.zero_(); gp.zero_()
    ops.embed_scatter(ids, tts, d, T, gw, gt, gp)
    torch.cuda.synchronize()
    ew = float((gw - ref_w).abs().max()); et = float((gt[0] - d.float().sum(0)).abs().max()); ep = float((gp[:T] - d.float().view(B, T, H).sum(0)).abs().max())
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.embed_scatter(ids, tts, d, T, gw, gt, gp); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"slices {sl}: {ts[len(ts)//2]:.1f} us   max err word {ew:.2e} type {et:.2e} pos {ep:.2e}")
