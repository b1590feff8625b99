#!/usr/bin/env python3
"""LayerNorm kernels at the headline shape: time and effective HBM rate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev = "cuda"; M, H = 18400, 768
x = torch.randn(M, H, device=dev).bfloat16(); dy = torch.randn(M, H, device=dev).bfloat16()
g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev); dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); d2 = torch.zeros(H, device=dev)
drop = ops.make_drop(0.1, 1, 2)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
y, mean, rstd = ops.ln_fwd(x, g, b, 1e-12)
t = timeit(lambda: ops.ln_fwd(x, g, b, 1e-12, out=y)); print(f"ln_fwd            {t:6.1f} us  {2*M*H*2/t/1e6:5.2f} TB/s")
t = timeit(lambda: ops.ln_fwd(x, g, b, 1e-12, out=y, drop=drop)); print(f"ln_fwd + dropout  {t:6.1f} us  {2*M*H*2/t/1e6:5.2f} TB/s")
dx = torch.empty_like(x); dx2 = torch.empty_like(x)
t = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, g, dg, db, dx=dx)); print(f"ln_bwd            {t:6.1f} us  {3*M*H*2/t/1e6:5.2f} TB/s")
t = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, g, dg, db, dx=dx, dx2=dx2, pre_drop=drop, post_drop=drop, dbias2=d2)); print(f"ln_bwd + branches {t:6.1f} us  {4*M*H*2/t/1e6:5.2f} TB/s")
