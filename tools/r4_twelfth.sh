#!/bin/bash
# round 4, twelfth GPU call: per-launch durations of the weight-gradient kernel, paired (216 tiles) against deferred (rounds of 256 tiles)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
cat > /tmp/tn_rounds.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from msa_amd import ops
dev = "cuda"
M = 13850
def probs(nl):
    out = []
    for l in range(nl):
        for N, K in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
            out.append((torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
    return out
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
p2, p11, p1 = probs(2), probs(11), probs(1)
for rep in range(2):
    t2 = timed(lambda: ops.gemm_tn_grouped(p2))
    t11 = timed(lambda: ops.gemm_tn_grouped(p11))
    t1 = timed(lambda: ops.gemm_tn_grouped(p1))
    fl = lambda nl: nl * 2.0 * M * (3072 * 768 * 2 + 2304 * 768 + 768 * 768)
    print(f"2 layers (216 tiles, 1 launch): {t2*1e3:.1f} us = {fl(2)/t2/1e9:.0f} TF/s | 11 layers (1188 tiles, 5 launches of <= 256): {t11*1e3:.1f} us = {fl(11)/t11/1e9:.0f} TF/s "
          f"= {t11/4.64*1e3:.1f} us per full round | 1 layer (108 tiles x 2 token splits + reduce): {t1*1e3:.1f} us = {fl(1)/t1/1e9:.0f} TF/s")
PY
python /tmp/tn_rounds.py > $O/r4_tn_rounds.log 2>&1; cat $O/r4_tn_rounds.log
