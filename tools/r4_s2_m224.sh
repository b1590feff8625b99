#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, torch
sys.path.insert(0, os.getcwd())
from msa_amd import ops
dev = "cuda"
for (M, N, K, kw) in ((18400, 2304, 768, "bias"), (18400, 3072, 768, "gelu"), (13850, 3072, 768, "gelu_bwd")):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); U = torch.randn(M, N, device=dev).bfloat16()
    args = dict(bias=bias) if kw == "bias" else (dict(bias=bias, gelu=True, aux=torch.empty(M, N, device=dev, dtype=torch.bfloat16)) if kw == "gelu" else dict(gelu_bwd_u=U))
    os.environ.pop("MMBERT_NT_8PHASE_M224", None)
    ref = ops.gemm_nt(A, B, **args).float()
    d0 = ops.gemm_nt_describe(M, N, K)
    os.environ["MMBERT_NT_8PHASE_M224"] = "1"
    got = ops.gemm_nt(A, B, **args).float()
    d1 = ops.gemm_nt_describe(M, N, K)
    print(kw, d0["kernel"], d0["tile"], "->", d1["kernel"], d1["tile"], d1["tiles"], "max diff", float((got - ref).abs().max()), "scale", float(ref.abs().max()))
    for env in (None, "1", None, "1"):
        if env: os.environ["MMBERT_NT_8PHASE_M224"] = env
        else: os.environ.pop("MMBERT_NT_8PHASE_M224", None)
        for _ in range(3): ops.gemm_nt(A, B, **args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(A, B, **args)
        e1.record(); torch.cuda.synchronize()
        print("   ", "m224" if env else "ring", f"{e0.elapsed_time(e1) * 100:.1f} us")
PY
ROUNDS=5 STEPS=8 timeout 900 python tools/ab_step.py ring: m224:MMBERT_NT_8PHASE_M224=1 > $O/r4s2_ab_m224.log 2>&1; grep -v amdgpu $O/r4s2_ab_m224.log
