"""Diagnostic: one epilogue of the multi-tile 8-phase NT kernel on the device tile queue against the static walk (argv: epi N)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import msa_amd.ops as O
from msa_amd import ops
dev = "cuda"
epi, N = sys.argv[1], int(sys.argv[2])
M, K = 18400 - 37, 768
g = torch.Generator().manual_seed(1)
A = (torch.randn(M, K, generator=g) * 0.1).bfloat16().to(dev); B = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
bias = torch.randn(N, generator=g).to(dev); R = torch.randn(M, N, generator=g).bfloat16().to(dev)
kw = {"plain": {}, "bias": dict(bias=bias), "gelu": dict(bias=bias, gelu=True), "resid": dict(resid=R), "gelu_bwd": dict(gelu_bwd_u=R)}[epi]
print(ops.gemm_nt_describe(M, N, K, epi={"plain": 0, "bias": 1, "gelu": 3, "resid": 4, "gelu_bwd": 8}[epi], with_queue=True), flush=True)
O.dynamic_tile_queue = False
ref = ops.gemm_nt(A, B, **kw); torch.cuda.synchronize(); print("static ok", flush=True)
O.dynamic_tile_queue = True
for rep in range(3):
    out = ops.gemm_nt(A, B, **kw); torch.cuda.synchronize()
    print("queue rep", rep, "equal", bool(torch.equal(out, ref)), flush=True)
