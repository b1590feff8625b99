#!/usr/bin/env python3
"""LayerNorm launches against a plain streaming kernel of the SAME bytes, operands cold (a ring of buffer sets larger than the
256 MB Infinity Cache), at the train step's shapes: forward rows = all 18 400, backward rows = the ~14 000 the valid-first packing
keeps.  Answers: how far is ln_fwd / ln_bwd from what ANY kernel reaches in a 15-20 us launch of these sizes?"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "stream_rate.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "stream_rate.hip")])
    if "--build" in sys.argv: sys.exit(0)
from msa_amd import ops
lib = ctypes.CDLL(so)
lib.stream_launch.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.row_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev, H, NSET = "cuda", 768, 16

_blk = torch.randn(8192, 8192, device=dev).bfloat16()

def timeit(fn, n=48):
    """GPU time per launch: the launches are queued behind a ~10 ms blocker so that the host's per-call cost does not enter."""
    for k in range(NSET): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8): torch.mm(_blk, _blk)
    e0.record()
    for k in range(n): fn(k % NSET)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, tag in ((18400, "forward rows"), (14000, "backward rows")):
    sets = [[torch.randn(M, H, device=dev).bfloat16() for _ in range(4)] for _ in range(NSET)]
    g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev); dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
    n16 = M * H * 2 // 16
    s = torch.cuda.current_stream().cuda_stream
    print(f"--- M = {M} ({tag}), H = {H}, bf16, {NSET} buffer sets = {NSET * 4 * M * H * 2 / 1e6:.0f} MB ---")
    for nr in (1, 2, 3):
        best = None
        for blocks in (512, 1024, 2048, 4096, 8192):
            t = timeit(lambda k: lib.stream_launch(nr, blocks, sets[k][0].data_ptr(), sets[k][1].data_ptr(), sets[k][2].data_ptr(), sets[k][3].data_ptr(), n16, s))
            if best is None or t < best[0]: best = (t, blocks)
        t, blocks = best
        print(f"stream {nr} in + 1 out : {t:6.1f} us  {(nr + 1) * M * H * 2 / t / 1e6:5.2f} TB/s   (best grid {blocks})")
    for level, name in enumerate(["row copy (8 B per lane)", "+ fp32 unpack / repack, gamma", "+ row sum before the store", "+ variance reduction", "+ dropout hash"]):
        res = []
        for blocks in (512, 768, 1024, 1536, 2048, 4096):
            res.append((timeit(lambda k: lib.row_launch(level, blocks, sets[k][0].data_ptr(), sets[k][3].data_ptr(), M, g.data_ptr(), s)), blocks))
        t, blocks = min(res)
        print(f"row kernel level {level} {name:32s}: {t:6.1f} us  {2 * M * H * 2 / t / 1e6:5.2f} TB/s  (best grid {blocks}; " + " ".join(f"{b}:{tt:.1f}" for tt, b in res) + ")")
    drop = ops.make_drop(0.1, 1, 2)
    st = [ops.ln_fwd(q[0], g, b, 1e-12) for q in sets]
    if "--sweep" in sys.argv:
        for rows in (1,):
            for cap in (512, 768, 1024, 1536, 2048):
                os.environ["MMBERT_LN_ROWS"], os.environ["MMBERT_LN_FWD_BLOCKS"] = str(rows), str(cap)
                t = timeit(lambda k: ops.ln_fwd(sets[k][0], g, b, 1e-12, out=sets[k][3], drop=drop))
                print(f"   ln_fwd + dropout  rows/wave {rows} blocks <= {cap:5d}: {t:6.1f} us")
        os.environ.pop("MMBERT_LN_ROWS"); os.environ.pop("MMBERT_LN_FWD_BLOCKS")
        d2 = torch.zeros(H, device=dev)
        lnd = ops.LnDeferred(8)            # as the encoder's backward: the partial sums of 8 calls folded by one launch
        for rows in (4, 8, 16):
            for cap in (128, 256, 384, 512, 768, 1024, 1536):
                os.environ["MMBERT_LN_BWD_WPB"], os.environ["MMBERT_LN_BWD_BLOCKS"] = str(rows), str(cap)
                t = timeit(lambda k: ops.ln_bwd(sets[k][1], sets[k][0], st[k][1], st[k][2], g, dg, db, dx=sets[k][3], dx2=sets[k][2], pre_drop=drop, dbias2=d2, deferred=lnd))
                lnd.flush()
                print(f"   ln_bwd encoder form waves/block {rows} blocks <= {cap:5d}: {t:6.1f} us")
        os.environ.pop("MMBERT_LN_BWD_WPB"); os.environ.pop("MMBERT_LN_BWD_BLOCKS")
    t = timeit(lambda k: ops.ln_fwd(sets[k][0], g, b, 1e-12, out=sets[k][3])); print(f"ln_fwd (1 in 1 out)          {t:6.1f} us  {2 * M * H * 2 / t / 1e6:5.2f} TB/s")
    t = timeit(lambda k: ops.ln_fwd(sets[k][0], g, b, 1e-12, out=sets[k][3], drop=drop)); print(f"ln_fwd + dropout            {t:6.1f} us  {2 * M * H * 2 / t / 1e6:5.2f} TB/s")
    lnd = ops.LnDeferred(8)
    t = timeit(lambda k: ops.ln_bwd(sets[k][1], sets[k][0], st[k][1], st[k][2], g, dg, db, dx=sets[k][3], deferred=lnd)); lnd.flush(); print(f"ln_bwd (2 in 1 out)          {t:6.1f} us  {3 * M * H * 2 / t / 1e6:5.2f} TB/s")
    t = timeit(lambda k: ops.ln_bwd(sets[k][1], sets[k][0], st[k][1], st[k][2], g, dg, db, dx=sets[k][3], dx2=sets[k][2], pre_drop=drop, dbias2=dg, deferred=lnd))
    lnd.flush()
    print(f"ln_bwd encoder form (2 in 2 out){t:6.1f} us  {4 * M * H * 2 / t / 1e6:5.2f} TB/s")
    del sets, st
