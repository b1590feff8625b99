"""Tensor-level wrappers over the C ABI (include/mmbert_hip.h).  torch is used for device memory
and the current stream only; every computation below is a hand-written gfx950 kernel.

All tensors must live on the GPU; bf16 operands are ``torch.bfloat16``; leading dimensions are
taken from ``stride(0)`` so row-slices of larger buffers can be passed without copies.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

EPI_BIAS, EPI_GELU, EPI_RESID, EPI_GELU_BWD, EPI_OUT_F32 = 1, 2, 4, 8, 16

# round 6: the optimizer's transposed-copy launch on a side stream (flat.FlatParams.refresh_transposes); False = on the current stream (A/B)
SIDE_TRANSPOSES = True
# priority of the side streams: "low" = the lowest the device offers (the current stream's small launches get their CUs first), "default"
SIDE_STREAM_PRIORITY = "low"
_side_streams = {}


def side_stream(tag: str, device) -> "torch.cuda.Stream":
    """One cached side stream per (purpose, device, priority setting)."""
    key = (tag, str(device), SIDE_STREAM_PRIORITY)
    s = _side_streams.get(key)
    if s is None:
        prio = 0
        if SIDE_STREAM_PRIORITY == "low":
            try:
                prio = max(torch.cuda.Stream.priority_range())
            except Exception:
                prio = 0
        s = _side_streams[key] = torch.cuda.Stream(device=device, priority=prio)
    return s

Drop = Optional[Tuple[int, int, float]]      # (rng stream, thr16, scale)
NO_DROP = (0, 0, 1.0)


# The current HIP stream of the calling thread as a raw handle.  torch.cuda.current_stream() builds a Python Stream object per call
# (~9 us, ~110 calls per train step = 1 ms of host time on the critical enqueue path); the two C accessors below return the same
# handle in well under a microsecond.  They are private torch API: the public path is the fallback.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_rng_fn = None


def rng_stream(seed: int, site: int) -> int:
    global _rng_fn
    if _rng_fn is None:
        _rng_fn = _lib.load().mmbert_rng_stream
    return _rng_fn(seed & 0xFFFFFFFFFFFFFFFF, site & 0xFFFFFFFF)


_thr_cache = {}


def make_drop(p: float, seed: int, site: int) -> Tuple[int, int, float]:
    """(stream, thr16, scale) of a dropout site; the effective drop probability is thr16/65536."""
    if p <= 0.0:
        return NO_DROP
    t = _thr_cache.get(p)
    if t is None:                                        # (the threshold of a probability never changes: one C call per distinct p)
        thr = _lib.load().mmbert_dropout_thr16(p)
        t = _thr_cache[p] = (thr, 1.0 / (1.0 - thr / 65536.0))
    return (rng_stream(seed, site), t[0], t[1])


_tile_queues = {}
# the persistent NT GEMM draws its tiles from a device-side queue (mmbert_gemm_nt's tile_queue) instead of the static b, b+G, ...
# schedule: what data-parallel runs need while RCCL's channel kernels hold CUs (parallel.DataParallel turns it on).  On an
# otherwise idle GPU the static schedule is the faster one -- round 2, same-process A/B of the train step, after the grouped walk
# of the vocabulary projection went in: static 15.99 ms, per-XCD queue 16.06 (+0.4 %), ONE global counter 16.22 (+1.5 %: any CU
# takes the next tile, the XCD's L2 loses its panels -- that kernel's fetch 1.27 -> 4.2 GB, 750 -> 940 us) -- so it is off by
# default there.  (Round 3: the queue's fetch no longer drains the load queue -- csrc/gemm.hip, queue_fetch -- and the same A/B reads
# static 14.47 / per-XCD queue 14.47 ms: the queue is free now; the default stays static.)  MMBERT_NT_DYNAMIC=1 turns it on everywhere.
dynamic_tile_queue = bool(int(__import__("os").environ.get("MMBERT_NT_DYNAMIC", "0")))


def _tile_queue(device) -> Optional[int]:
    """The 64-byte tile queue (8 per-XCD fetch counters + exit counter) of the persistent NT GEMM for the current stream (mmbert_gemm_nt's ``tile_queue``), or None for the
    static schedule.  One zeroed buffer per (device, stream): the kernel leaves it zeroed, launches of one stream are ordered."""
    if not dynamic_tile_queue:
        return None
    key = (device, _stream())
    q = _tile_queues.get(key)
    if q is None:
        q = _tile_queues[key] = torch.zeros(16, device=device, dtype=torch.int32)
    return q.data_ptr()


def gemm_nt(A, B, *, out=None, bias=None, gelu=False, aux=None, resid=None, gelu_bwd_u=None, alpha=1.0,
            alpha_dev=None, drop: Drop = None, out_f32=False):
    """out[M,N] = epi(alpha * A[M,K] @ B[N,K]^T)  (see mmbert_gemm_nt)."""
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and A.stride(1) == 1 and B.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    epi = 0
    if bias is not None:
        epi |= EPI_BIAS
    if gelu:
        epi |= EPI_GELU
    if resid is not None:
        epi |= EPI_RESID
    if gelu_bwd_u is not None:
        epi |= EPI_GELU_BWD
    if out_f32:
        epi |= EPI_OUT_F32
    d = drop or NO_DROP
    _lib.check(lib.mmbert_gemm_nt(_stream(), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0),
                                  M, N, K, epi, _ptr(bias), _ptr(resid), resid.stride(0) if resid is not None else 0,
                                  _ptr(aux), aux.stride(0) if aux is not None else 0,
                                  _ptr(gelu_bwd_u), gelu_bwd_u.stride(0) if gelu_bwd_u is not None else 0,
                                  float(alpha), _ptr(alpha_dev), d[0], d[1], d[2], _tile_queue(A.device)), "mmbert_gemm_nt")
    return out


def gemm_nt_describe(M: int, N: int, K: int, epi: int = 0, with_queue: bool = False) -> dict:
    """Which kernel / tile / tile walk ``gemm_nt`` uses for a shape on the current device (mmbert_gemm_nt_describe; launches nothing)."""
    import ctypes
    out = (ctypes.c_int * 8)()
    _lib.check(_lib.load().mmbert_gemm_nt_describe(int(M), int(N), int(K), int(epi), int(bool(with_queue)), out), "mmbert_gemm_nt_describe")
    kern = {0: "128x128", 1: "ring", 2: "persistent", 3: "8phase"}[out[0]]
    return dict(kernel=kern, tile=f"{out[1]}x{out[2]}", tiles=out[3], workgroups=out[4], rounds=out[5] / 100.0, group_m=out[6], cus=out[7])


_slab_cache = {}


def _slab(nbytes: int, device) -> Optional[torch.Tensor]:
    if nbytes == 0:
        return None
    key = (device, _stream())
    t = _slab_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, device=device, dtype=torch.uint8)
        _slab_cache[key] = t
    return t


_ws_cache = {}


def _ws_f32(nfloats: int, device) -> torch.Tensor:
    key = (device, _stream())
    t = _ws_cache.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 1), device=device, dtype=torch.float32)
        _ws_cache[key] = t
    return t


def gemm_nt_splitk(A, B, out=None, resid=None):
    """out[M,N] (bf16) = A[M,K] @ B[N,K]^T (+ resid[M,N] bf16) for long K and few output tiles (see mmbert_gemm_nt_splitk)."""
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.bfloat16)
    if resid is not None:
        assert resid.shape == (M, N) and resid.dtype == torch.bfloat16 and resid.stride(1) == 1
    ws = _ws_f32((lib.mmbert_gemm_nt_splitk_workspace(M, N, K) + 3) // 4, A.device)
    _lib.check(lib.mmbert_gemm_nt_splitk(_stream(), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0),
                                         M, N, K, ws.data_ptr(), _ptr(resid), 0 if resid is None else resid.stride(0)), "mmbert_gemm_nt_splitk")
    return out


def gemm_tn(A, B, W, *, accumulate=True, alpha=1.0, alpha_dev=None, bias_out=None):
    """W[N,K] (fp32) (+)= alpha * A[M,N]^T @ B[M,K]; bias_out[N] += alpha * colsum(A)  (see mmbert_gemm_tn)."""
    lib = _lib.load()
    M, N = A.shape
    K = B.shape[1]
    assert B.shape[0] == M and W.shape == (N, K) and W.dtype == torch.float32 and W.is_contiguous()
    need = lib.mmbert_gemm_tn_workspace(M, N, K, None)
    slab = _slab(need, A.device)
    _lib.check(lib.mmbert_gemm_tn(_stream(), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), W.data_ptr(), W.stride(0),
                                  M, N, K, 1 if accumulate else 0, float(alpha), _ptr(alpha_dev), _ptr(slab), _ptr(bias_out)), "mmbert_gemm_tn")
    return W


TN_MAX_PROBLEMS = 52                      # (gemm.hip: TN_MAXP)
_PtrArr = {n: (ctypes.c_void_p * n) for n in range(1, TN_MAX_PROBLEMS + 1)}
_IntArr = {n: (ctypes.c_int * n) for n in range(1, TN_MAX_PROBLEMS + 1)}


def gemm_tn_grouped(problems, *, accumulate=True, alpha=1.0):
    """problems: up to 52 tuples (A[M,N] bf16, B[M,K] bf16, W[N,K] fp32, bias[N] fp32 or None) -- the deferred weight gradients of up to 12
    layers (round 4), and behind them (round 6) problems of FEWER rows, whose tiles ride in the call's last launch: one call computes every
    W (+)= A^T @ B and bias += colsum(A)  (see mmbert_gemm_tn_grouped_rows).  Problems are taken longest first (a stable sort by rows).
    ``accumulate``: one flag, or one per problem."""
    lib = _lib.load()
    n = len(problems)
    each = None
    if isinstance(accumulate, (list, tuple)):                 # one flag per problem
        each = [1 if a else 0 for a in accumulate]
        accumulate = all(each)
        if accumulate or not any(each):
            each = None
    rows = [p[0].shape[0] for p in problems]
    if any(rows[i] < rows[i + 1] for i in range(n - 1)):
        order = sorted(range(n), key=lambda i: -rows[i])
        problems = [problems[i] for i in order]
        rows = [rows[i] for i in order]
        if each is not None:
            each = [each[i] for i in order]
    M = rows[0]
    PA, IA = _PtrArr[n], _IntArr[n]
    Ns = IA(*[p[0].shape[1] for p in problems])
    Ks = IA(*[p[1].shape[1] for p in problems])
    mixed = rows[-1] != M
    slab = None
    if not mixed:
        need = lib.mmbert_gemm_tn_grouped_workspace(n, Ns, Ks, M, None)
        slab = _slab(need, problems[0][0].device)
    _lib.check(lib.mmbert_gemm_tn_grouped_rows(
        _stream(), n, PA(*[p[0].data_ptr() for p in problems]), IA(*[p[0].stride(0) for p in problems]),
        PA(*[p[1].data_ptr() for p in problems]), IA(*[p[1].stride(0) for p in problems]),
        PA(*[p[2].data_ptr() for p in problems]), PA(*[(p[3].data_ptr() if p[3] is not None else None) for p in problems]),
        Ns, Ks, IA(*rows), 1 if accumulate else 0, IA(*each) if each is not None else None, float(alpha), None, _ptr(slab)), "mmbert_gemm_tn_grouped_rows")


def colsum(X, out, *, alpha=1.0, alpha_dev=None):
    lib = _lib.load()
    M, N = X.shape
    _lib.check(lib.mmbert_colsum(_stream(), X.data_ptr(), X.stride(0), M, N, out.data_ptr(), float(alpha), _ptr(alpha_dev)), "mmbert_colsum")
    return out


def dropout_mask(n: int, drop: Tuple[int, int, float], device) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty(n, device=device, dtype=torch.uint8)
    _lib.check(lib.mmbert_dropout_mask(_stream(), out.data_ptr(), n, drop[0], drop[1]), "mmbert_dropout_mask")
    return out


def mlm_mask(ids, p_select: float, seed: int, *, special_ids=(101, 102), mask_id=103, p_replace=0.8, site=4242):
    """In-place MLM masking of an int64 id tensor on the GPU (mmbert_mlm_mask); returns the labels (-100 = not selected)."""
    assert ids.dtype == torch.int64 and ids.is_contiguous() and len(special_ids) <= 3
    sp = list(special_ids) + [special_ids[0] if special_ids else -1] * (3 - len(special_ids))
    labels = torch.empty_like(ids)
    _lib.check(_lib.load().mmbert_mlm_mask(_stream(), ids.data_ptr(), labels.data_ptr(), ids.numel(), rng_stream(seed, site),
                                           int(round(p_select * 65536)), int(round(p_replace * 65536)), sp[0], sp[1], sp[2], mask_id), "mmbert_mlm_mask")
    return labels


def ln_fwd(x, gamma, beta, eps, *, M=None, out=None, in_rows=None, out_rows=None, drop: Drop = None, stats=True, drop_row0=0):
    """``drop_row0``: row i of this launch draws the dropout mask of row i + drop_row0 of the site (a slice of the site's rows)."""
    lib = _lib.load()
    H = x.shape[1]
    if M is None:
        M = in_rows.numel() if in_rows is not None else x.shape[0]
    if out is None:
        out = torch.empty((M, H), device=x.device, dtype=torch.bfloat16)
    if isinstance(stats, tuple):                             # the caller's (mean, rstd) buffers (slices of larger ones)
        mean, rstd = stats
        assert mean.numel() == M and rstd.numel() == M and mean.dtype == torch.float32 and mean.is_contiguous() and rstd.is_contiguous()
    else:
        mean = torch.empty(M, device=x.device, dtype=torch.float32) if stats else None
        rstd = torch.empty(M, device=x.device, dtype=torch.float32) if stats else None
    d = drop or NO_DROP
    _lib.check(lib.mmbert_ln_fwd(_stream(), x.data_ptr(), x.stride(0), _ptr(in_rows), out.data_ptr(), out.stride(0), _ptr(out_rows),
                                 M, H, gamma.data_ptr(), beta.data_ptr(), float(eps), _ptr(mean), _ptr(rstd), d[0], d[1], d[2], int(drop_row0)), "mmbert_ln_fwd")
    return out, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, *, M=None, dx=None, dx2=None, dy_rows=None, x_rows=None, dx_rows=None,
           post_drop: Drop = None, pre_drop: Drop = None, dbias2=None, drop_rows=None, deferred: "LnDeferred" = None, dy_row_limit=0):
    """``deferred``: an LnDeferred collector -- the gamma / beta (/ bias) partial sums of this call stay in a workspace slice of
    the collector and are folded into the gradients by ITS one reduce launch (``deferred.flush()``) instead of one per call.
    ``dy_row_limit`` (> 0, with ``dy_rows``): mapped dy rows at or past the limit do not exist -- their gradient is zero."""
    lib = _lib.load()
    H = x.shape[1]
    if M is None:
        M = mean.numel()
    if dx is None:
        dx = torch.empty((M, H), device=x.device, dtype=torch.bfloat16)
    po, pr = post_drop or NO_DROP, pre_drop or NO_DROP
    if deferred is not None:
        ws_ptr = deferred.slot(M, H, x.device, dgamma, dbeta, dbias2)
    else:
        ws_ptr = _ws_f32(lib.mmbert_ln_bwd_workspace(M, H), x.device).data_ptr()
    _lib.check(lib.mmbert_ln_bwd(_stream(), dy.data_ptr(), dy.stride(0), _ptr(dy_rows), x.data_ptr(), x.stride(0), _ptr(x_rows),
                                 mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), M, H,
                                 dx.data_ptr(), dx.stride(0), _ptr(dx_rows), _ptr(dx2), dx2.stride(0) if dx2 is not None else 0,
                                 _ptr(dgamma), _ptr(dbeta), _ptr(dbias2), po[0], po[1], po[2], pr[0], pr[1], pr[2], ws_ptr, _ptr(drop_rows),
                                 1 if deferred is not None else 0, int(dy_row_limit)), "mmbert_ln_bwd")
    return dx


_lnd_cache = {}


class LnDeferred:
    """Collects the partial-sum workspaces of several ``ln_bwd(..., deferred=self)`` calls that share H -- an encoder's 2 per layer,
    and (round 4) calls of other row counts too: the MLM head's and the sparse top layer's few hundred rows, the embedding stage --
    and folds them into their gradients with ONE ``mmbert_ln_bwd_reduce_rows`` launch per ``slots`` calls: ``flush()``.
    ``slots``: how many calls the caller expects before its flush (at most 32 per launch).  The workspace is ONE persistent buffer
    per (device, stream) that grows on demand -- not a fresh allocation per backward (300 MB at the headline shape where 24 slots
    are used, and churn in the caching allocator when M changes with the data)."""

    def __init__(self, slots: int = 32):
        self.items, self.H, self.ws, self.used = [], None, None, 0
        self.slots = max(1, min(32, int(slots)))

    def slot(self, M, H, device, dgamma, dbeta, dbias2) -> int:
        if self.H is not None and (H != self.H or len(self.items) == self.slots):
            self.flush()
        self.H = H
        per = _lib.load().mmbert_ln_bwd_workspace(M, H)
        ck = (device, _stream())
        if not self.items:
            self.ws, self.used = _lnd_cache.get(ck), 0
        if self.ws is None or self.ws.numel() < self.used + per:
            # (a buffer in use by queued launches is replaced only once its reduce is queued behind them: same stream, so the caching
            # allocator cannot hand it out earlier)
            self.flush()
            self.H = H
            want = max(self.slots * per, 2 * (self.ws.numel() if self.ws is not None else 0))
            self.ws, self.used = torch.empty(want, device=device, dtype=torch.float32), 0
            _lnd_cache[ck] = self.ws
        ptr = self.ws.data_ptr() + 4 * self.used
        self.used += per
        self.items.append((ptr, dgamma.data_ptr(), dbeta.data_ptr(), dbias2.data_ptr() if dbias2 is not None else None, int(M)))
        return ptr

    def reserve(self, n, M, H, device):
        """Make the next ``n`` ``slot(M, H, ...)`` calls flush-free.  A caller that takes several slots BEFORE launching their LayerNorm'
        calls (the composite layer call takes two) must not have the second slot() fold the first one's partial sums before they exist:
        whatever would make slot() flush within those calls -- another H, the slot count, a workspace that has to grow -- happens here,
        while every slot taken so far has been launched."""
        per = _lib.load().mmbert_ln_bwd_workspace(M, H)
        self.slots = min(32, max(self.slots, int(n)))            # (n > slots would make the n-th slot() flush the first ones: ADVICE r5)
        assert n <= self.slots, "LnDeferred.reserve: at most 32 slots per reduce launch"
        if self.H is not None and (H != self.H or len(self.items) + n > self.slots):
            self.flush()
        ck = (device, _stream())
        if not self.items:
            self.ws, self.used = _lnd_cache.get(ck), 0
        if self.ws is None or self.ws.numel() < self.used + n * per:
            self.flush()
            want = max(max(self.slots, n) * per, 2 * (self.ws.numel() if self.ws is not None else 0))
            self.ws, self.used = torch.empty(want, device=device, dtype=torch.float32), 0
            _lnd_cache[ck] = self.ws

    def drop(self):
        """Forget collected calls without folding them (the leftovers of a backward pass that raised half-way)."""
        self.items, self.H, self.used = [], None, 0

    def flush(self):
        n = len(self.items)
        if n:
            PA = ctypes.c_void_p * n
            cols = list(zip(*self.items))
            Ms = (ctypes.c_int * n)(*cols[4])
            _lib.check(_lib.load().mmbert_ln_bwd_reduce_rows(_stream(), n, PA(*cols[0]), PA(*cols[1]), PA(*cols[2]), PA(*cols[3]), Ms, self.H),
                       "mmbert_ln_bwd_reduce_rows")
        self.items, self.H, self.used = [], None, 0


def embed_gather(ids, tts, word, type_, pos, T, out=None):
    lib = _lib.load()
    n = ids.numel()
    V, H = word.shape
    if out is None:
        out = torch.empty((n, H), device=word.device, dtype=torch.bfloat16)
    _lib.check(lib.mmbert_embed_gather(_stream(), ids.data_ptr(), _ptr(tts), word.data_ptr(), type_.data_ptr(), pos.data_ptr(),
                                       n, T, H, V, out.data_ptr(), out.stride(0)), "mmbert_embed_gather")
    return out


def set_deterministic(on: bool) -> None:
    """Process-global (mmbert_set_deterministic): ordered sums instead of fp32 atomics everywhere in the library; the scatter-adds with
    data-dependent collisions (embed_scatter's word / token-type rows, rows_to_block) go through mmbert_id_runs_sum_rows here (no sort: the first row of an id
    collects the id's rows in ascending order and adds them once; O(n^2) id compares per launch, n <= 8192)."""
    _lib.load().mmbert_set_deterministic(1 if on else 0)


def deterministic() -> bool:
    return bool(_lib.load().mmbert_get_deterministic())


if os.environ.get("MMBERT_DETERMINISTIC", "0") not in ("", "0"):         # read once at import; model.deterministic = True does the same
    set_deterministic(True)


def scatter_add_rows_ordered(keys, src, dst, vocab, union=None):
    """dst[row_of(key_i)] += src[i] for every row i, DETERMINISTICALLY (mmbert_id_runs_sum_rows): the rows of one key are summed in a fixed
    association (ascending i) and land with one add per destination element.  row_of(key) = key, or its index in the ascending ``union``;
    keys outside (0, vocab) or not in the union are skipped.  One launch, nothing is sorted or read back by the host."""
    keys = keys.reshape(-1).long().contiguous()
    fn, bf = _lib.load().mmbert_id_runs_sum_rows, 1 if src.dtype == torch.bfloat16 else 0
    # (the kernel keeps an id's row list in LDS: 8192 rows per launch; longer batches go out in consecutive launches -- still ordered: the
    # launches of a stream run one after the other, each adds its part of a run once)
    for off in range(0, keys.numel(), 8192):
        n = min(8192, keys.numel() - off)
        _lib.check(fn(_stream(), src[off:off + n].data_ptr(), bf, src.stride(0), keys[off:off + n].data_ptr(), n, src.shape[1], int(vocab),
                      _ptr(union), 0 if union is None else union.numel(), dst.data_ptr(), dst.stride(0)), "mmbert_id_runs_sum_rows")
    return dst


def rows_to_block(ids, rows, union, vocab, block):
    """block[pos(ids[i])] += rows[i] with pos = the index of ids[i] in the ascending int64 list ``union`` (mmbert_rows_to_block): the
    local side of the data-parallel compact row exchange.  ids outside (0, vocab) or not in the list are skipped."""
    ids = ids.reshape(-1).long().contiguous()
    union = union.long().contiguous()
    if deterministic():
        if union.numel():
            scatter_add_rows_ordered(ids, rows, block, vocab, union=union)
        return block
    _lib.check(_lib.load().mmbert_rows_to_block(_stream(), ids.data_ptr(), rows.data_ptr(), 1 if rows.dtype == torch.bfloat16 else 0, rows.stride(0),
                                                ids.numel(), rows.shape[1], union.data_ptr(), union.numel(), int(vocab), block.data_ptr()),
               "mmbert_rows_to_block")
    return block


def embed_scatter(ids, tts, d, T, gword, gtype, gpos, vocab=None):
    """``gword`` None: position and token-type gradients only (the word rows are exchanged in compact form: parallel.DataParallel)."""
    lib = _lib.load()
    n = ids.numel()
    V, H = gword.shape if gword is not None else (int(vocab), d.shape[1])
    slab = None
    if deterministic():
        # position sums: a single adder per address in the kernel; token-type sums: per-position partials in a slab, folded in position
        # order; word rows (a word at several positions would be atomics in arrival order): per-id run sums (mmbert_id_runs_sum_rows)
        slab = _ws_f32(2 * T * H, d.device)
        if gword is not None:
            scatter_add_rows_ordered(ids, d, gword, V)
            gword = None
    _lib.check(lib.mmbert_embed_scatter(_stream(), ids.data_ptr(), _ptr(tts), d.data_ptr(), d.stride(0), n, T, H, V,
                                        _ptr(gword), gtype.data_ptr(), gpos.data_ptr(), _ptr(slab)), "mmbert_embed_scatter")


def _pair_rows(T, Pn, ld, seq_len, offset):
    """The kernels address the pair rows of sample b as ``b*(T+P) + T + p``.  A pair block at ``offset`` inside sequences
    of ``seq_len`` rows (several modalities in one sequence) is the same pattern with T' = seq_len - P seen from a base
    pointer moved by ``offset - T'`` rows: returns (T', byte shift for bf16 rows of ``ld`` elements)."""
    if seq_len is None:
        return T, 0
    Tp = seq_len - Pn
    return Tp, (offset - Tp) * ld * 2


def pair_proj_fwd(feat, W, bias, out, T, *, seq_len=None, offset=None):
    """feat fp32 or float64 [B,P,D] contiguous; writes relu(W.feat+b) into rows b*(T+P)+T+p of ``out`` (bf16 [B*(T+P), H]); with ``seq_len`` /
    ``offset``: into rows b*seq_len + offset + p."""
    lib = _lib.load()
    B, Pn, D = feat.shape
    assert feat.dtype in (torch.float32, torch.float64) and feat.is_contiguous()
    H = W.shape[0]
    Tp, shift = _pair_rows(T, Pn, out.stride(0), seq_len, offset)
    _lib.check(lib.mmbert_pair_proj_fwd(_stream(), feat.data_ptr(), 1 if feat.dtype == torch.float64 else 0, B, Pn, D, W.data_ptr(), bias.data_ptr(), H,
                                        out.data_ptr() + shift, out.stride(0), Tp), "mmbert_pair_proj_fwd")


def pair_proj_bwd(feat, J, dJ, T, dW, db, *, seq_len=None, offset=None):
    lib = _lib.load()
    B, Pn, D = feat.shape
    H = dW.shape[0]
    assert J.stride(0) == dJ.stride(0) and feat.dtype in (torch.float32, torch.float64) and feat.is_contiguous()
    Tp, shift = _pair_rows(T, Pn, J.stride(0), seq_len, offset)
    ws = _ws_f32((lib.mmbert_pair_proj_bwd_workspace(B, Pn, D, H) + 3) // 4, feat.device)
    _lib.check(lib.mmbert_pair_proj_bwd(_stream(), feat.data_ptr(), 1 if feat.dtype == torch.float64 else 0, B, Pn, D, J.data_ptr() + shift, dJ.data_ptr() + shift, J.stride(0), Tp,
                                        dW.data_ptr(), db.data_ptr(), H, ws.data_ptr()), "mmbert_pair_proj_bwd")


NO_KEY = -1.0e30        # key-bias value of the padding slots ("no such key")


class SeqLayout:
    """Device-side description of the packed variable-length token matrix: ``lens[i]`` tokens per
    sequence, sequences back to back.  Cached per shape (built on the host, no device sync)."""

    def __init__(self, lens, heads, device):
        lib = _lib.load()
        rows_f, rows_b = lib.mmbert_attn_tile_rows(0), lib.mmbert_attn_tile_rows(1)
        starts, tiles_seq, tiles_r0, ftiles_seq, ftiles_r0, bases = [], [], [], [], [], []
        bstarts, bidx = [], []
        s = 0
        e = 0
        bp = 0
        for i, n in enumerate(lens):
            bstarts.append(bp)
            bidx.extend(range(bp, bp + n))
            bp += (n + 127) // 128 * 128
            starts.append(s)
            s += n
            bases.append(e)
            e += heads * n * ((n + 3) // 4 * 4)
            e = (e + 3) // 4 * 4
            for r0 in range(0, n, rows_b):
                tiles_seq.append(i)
                tiles_r0.append(r0)
            for r0 in range(0, n, rows_f):
                ftiles_seq.append(i)
                ftiles_r0.append(r0)
        if e >= 2 ** 32:
            raise ValueError("attention dropout index space exceeds 2^32 elements")
        mk = lambda x, dt: torch.tensor(x, dtype=dt, device=device)
        self.lens = list(lens)
        self.tokens = s
        self.heads = heads
        self.seq_start = mk(starts, torch.int32)
        self.seq_len = mk(list(lens), torch.int32)
        self.elem_base = mk(bases, torch.int64).to(torch.int32)      # bit pattern of uint32
        self.elem_base_host = bases
        self.tile_seq = mk(tiles_seq, torch.int32)
        self.tile_r0 = mk(tiles_r0, torch.int32)
        self.ntiles = len(tiles_seq)
        self.bias_start = mk(bstarts, torch.int32)        # padded key-bias layout (ceil128 entries per sequence)
        self.bias_index = mk(bidx, torch.int64)           # token -> slot of the padded array
        self.bias_len = bp
        self.ftile_seq = mk(ftiles_seq, torch.int32)
        self.ftile_r0 = mk(ftiles_r0, torch.int32)
        self.nftiles = len(ftiles_seq)
        self._rows_f = rows_f
        self._row_dev = None
        self._row_seq = np.repeat(np.arange(len(lens)), lens)                      # host: sequence of every packed row
        self._row_pos = np.concatenate([np.arange(n) for n in lens]) if len(lens) else np.zeros(0, np.int64)


def _seq_row_tables(self, device):
    """(sequence, position) of every packed row as device int64 tensors (cached)."""
    if self._row_dev is None:
        self._row_dev = (torch.from_numpy(self._row_seq).to(device), torch.from_numpy(self._row_pos).to(device))
    return self._row_dev


SeqLayout.row_tables = _seq_row_tables


class SplitLayout:
    """"Valid-first" packing of the same sequences: the first ``valid[s]`` rows of every sequence back to back (region A,
    ``rows_a`` rows in all), then the remaining (masked-out) rows of all sequences (region B).  Keys/values of a sequence are its
    region-A rows; forward visits the queries of both regions, backward only region A -- rows whose keys are masked out and
    that carry no label have exactly-zero gradients everywhere (model._encode).  ``perm[new] = old`` row, ``inv[old] = new``.
    Dropout indices, key-bias slots and ``seq_len`` stay those of the ORIGINAL sequences, so masks do not depend on the packing.
    Built per batch on the host (numpy) and shipped in one copy."""

    def __init__(self, base: SeqLayout, valid, device, dedupe=False, drop=False, rank=None):
        """``dedupe`` (inference without dropout only): the masked-out rows of a sequence all have the same input and see the same
        keys, hence the same hidden states in every layer -- region B keeps ONE of them per sequence and ``inv`` maps all of them
        to it (``perm`` is then shorter than ``inv``: not a permutation, forward only).
        ``drop`` (training when the caller does not ask for the prediction scores): region B is left out altogether -- nothing
        but the returned scores ever reads those rows -- and ``inv`` sends them to row ``rows_a``, one past the packed matrix
        (callers append a zero row before gathering).
        ``rank`` (device int32 per row, from prologue(rowset=True)): every sequence's own valid-first order -- ``valid[s]`` then
        counts its active rows, which need not be a prefix of the original sequence (GPU only)."""
        assert rank is None or (torch.device(device).type == "cuda" and not dedupe)
        lens = np.asarray(base.lens, dtype=np.int64)
        v = np.minimum(np.asarray(valid, dtype=np.int64), lens)
        pad = lens - v
        start_a = np.concatenate(([0], np.cumsum(v)[:-1]))
        self.rows_a = int(v.sum())
        self.valid_host = [int(x) for x in v]
        on_gpu = torch.device(device).type == "cuda"         # the [tokens]-sized row maps come from a kernel there (mmbert_split_rows)
        mode = 2 if drop else 1 if dedupe else 0
        if drop:
            pad = np.zeros_like(pad)
            start_b = np.full_like(start_a, self.rows_a)
            lens = v
        elif dedupe:
            pad = np.minimum(pad, 1)
            start_b = self.rows_a + np.concatenate(([0], np.cumsum(pad)[:-1]))
            lens = v + pad                                   # the sequences as the attention tiles see them
        else:
            start_b = self.rows_a + np.concatenate(([0], np.cumsum(pad)[:-1]))
        n_packed = self.rows_a + int(pad.sum())
        if not on_gpu:                                       # host form of the same maps (CPU tests of the packing)
            rs, rp = base._row_seq, base._row_pos
            is_valid = rp < v[rs]
            if drop:
                inv = np.where(is_valid, start_a[rs] + rp, self.rows_a)
                perm = np.nonzero(is_valid)[0]               # valid rows keep their relative order: perm[inv[valid]] = valid
            elif dedupe:
                inv = np.where(is_valid, start_a[rs] + rp, start_b[rs])
                keep = is_valid | (rp == v[rs])
                perm = np.empty(n_packed, dtype=inv.dtype)
                perm[inv[keep]] = np.nonzero(keep)[0]
            else:
                inv = np.where(is_valid, start_a[rs] + rp, start_b[rs] + rp - v[rs])
                perm = np.empty_like(inv)
                perm[inv] = np.arange(inv.size)
        rows = base._rows_f
        # tile lists (numpy, no per-sequence Python loop: this runs on the critical path of every step): region A tiles of all
        # sequences, then region B tiles; backward uses the region A part only
        ns_ = len(lens)
        seq_ids = np.arange(ns_)

        lpt = xcd_group = True     # longest-first tile lists, sequences grouped per XCD (round 2: -1.3 % / -0.2 % of the step; the switches are gone)

        def tiles(count_rows, first_row, shift, end):
            nt = (count_rows + rows - 1) // rows
            sq = np.repeat(seq_ids, nt)
            k = np.arange(int(nt.sum())) - np.repeat(np.cumsum(nt) - nt, nt)
            r0 = first_row[sq] + k * rows
            if lpt and len(sq):
                # longest work first: a workgroup's time is its key (or query) loop, i.e. the sequence's unmasked length; the
                # launch is 2-3 rounds of workgroups, and short tiles at the END fill the last round instead of trailing it.
                # The kernels launch grid (heads, tiles) and workgroups go to the 8 XCDs round-robin, so tile t of head h lands
                # on XCD (heads * t + h) % 8: sequences are interleaved in groups of `xs` = 8 / gcd(heads, 8), which puts the
                # tiles of ONE (sequence, head) -- they all read the same K / V (dK/dV: the same Q / dO) -- `xs` list entries
                # apart, i.e. on ONE XCD and its L2.
                xs = 8 // int(np.gcd(base.heads, 8)) if xcd_group else 1
                rank = np.empty(ns_, dtype=np.int64)
                rank[np.argsort(-v, kind="stable")] = np.arange(ns_)
                order = np.lexsort((rank[sq] % xs, k, rank[sq] // xs))
                sq, r0 = sq[order], r0[order]
            return sq, r0, shift[sq], end[sq]
        qa = tiles(v, np.zeros_like(v), start_a, v)
        qb = tiles(lens - v, v, start_b - v, lens)
        q_seq, q_r0, q_sh, q_end = qa
        f_seq, f_r0, f_sh, f_end = (np.concatenate((x, y)) for x, y in zip(qa, qb))
        nf, nq, ns = len(f_seq), len(q_seq), len(lens)
        ints = np.concatenate([np.asarray(x, dtype=np.int32) for x in (f_seq, f_r0, f_sh, f_end, q_seq, q_r0, q_sh, q_end, start_a, v, start_b)])
        if on_gpu:
            # through pinned memory: a copy from pageable memory makes the host wait until the stream has drained (hipMemcpyAsync
            # stages it synchronously), i.e. for the whole previous step when the host runs ahead (model.async_prologue)
            pin = torch.empty(ints.size, dtype=torch.int32, pin_memory=True)
            pin.numpy()[:] = ints
            dev_i = pin.to(device, non_blocking=True)
        else:
            dev_i = torch.from_numpy(ints).to(device, non_blocking=True)
        cut = np.cumsum([0, nf, nf, nf, nf, nq, nq, nq, nq, ns, ns, ns])
        part = [dev_i[cut[k]:cut[k + 1]] for k in range(11)]
        (self.ftile_seq, self.ftile_r0, self.ftile_qshift, self.ftile_qend, self.tile_seq, self.tile_r0, self.qtile_qshift,
         self.qtile_qend, self.seq_start, self.kv_len, start_b_dev) = part
        self.nftiles, self.ntiles = nf, nq
        M = base.tokens
        if on_gpu:
            rs_d, rp_d = base.row_tables(device)
            dev_p = torch.empty(n_packed + M, dtype=torch.int64, device=device)
            dev_p32 = torch.empty(n_packed + M, dtype=torch.int32, device=device)
            _lib.check(_lib.load().mmbert_split_rows(_stream(), rs_d.data_ptr(), rp_d.data_ptr(), self.seq_start.data_ptr(), start_b_dev.data_ptr(),
                                                     self.kv_len.data_ptr(), mode, M, self.rows_a, dev_p.data_ptr(), dev_p.data_ptr() + 8 * n_packed,
                                                     _ptr(rank), dev_p32.data_ptr(), dev_p32.data_ptr() + 4 * n_packed),
                       "mmbert_split_rows")
        else:
            dev_p = torch.from_numpy(np.concatenate((perm, inv))).to(device)
            dev_p32 = dev_p.to(torch.int32)
        self.perm, self.inv = dev_p[:n_packed], dev_p[n_packed:]
        self.perm32, self.inv32 = dev_p32[:n_packed], dev_p32[n_packed:]       # the same maps as int32 row lists (ln_fwd / ln_bwd / gather_rows)
        self.rows_packed = n_packed
        self.base, self.heads, self.tokens, self.lens = base, base.heads, base.tokens, base.lens
        self.seq_len, self.elem_base, self.bias_start, self.bias_len = base.seq_len, base.elem_base, base.bias_start, base.bias_len
        self.split = True


class DeviceSplitLayout:
    """The same valid-first packing as SplitLayout (mode 0: every row kept), but built on the DEVICE from the prologue's ``valid``
    counts (mmbert_split_layout + mmbert_split_rows): the forward pass needs no host round trip.  The tile lists are sized for the
    worst case (unused entries carry sequence -1: the attention kernels leave at once), ``nftiles`` is that static size.  What only
    BACKWARD needs on the host -- ``rows_a`` (its launch sizes), ``ntiles``, ``valid_host`` -- is read lazily from the prologue's
    pinned words (``words`` = (host int32 view, event)): by then they arrived a whole forward pass ago."""

    split = True
    dropped = False

    def __init__(self, base: SeqLayout, valid_dev, device, rank=None, words=None):
        lib = _lib.load()
        ns = len(base.lens)
        rows = base._rows_f
        nq_max = sum((n + rows - 1) // rows for n in base.lens)
        nf_max = nq_max + ns
        xs = 8 // int(np.gcd(base.heads, 8))
        buf = torch.empty(4 * nf_max + 4 * nq_max + 3 * ns + 4, device=device, dtype=torch.int32)
        _lib.check(lib.mmbert_split_layout(_stream(), base.seq_len.data_ptr(), valid_dev.data_ptr(), ns, rows, xs, nf_max, nq_max, buf.data_ptr()),
                   "mmbert_split_layout")
        cut = np.cumsum([0, nf_max, nf_max, nf_max, nf_max, nq_max, nq_max, nq_max, nq_max, ns, ns, ns, 4])
        part = [buf[cut[k]:cut[k + 1]] for k in range(12)]
        (self.ftile_seq, self.ftile_r0, self.ftile_qshift, self.ftile_qend, self.tile_seq, self.tile_r0, self.qtile_qshift,
         self.qtile_qend, self.seq_start, self.kv_len, start_b_dev, self.counts) = part
        self.nftiles = nf_max
        M = base.tokens
        rs_d, rp_d = base.row_tables(device)
        dev_p = torch.empty(2 * M, dtype=torch.int64, device=device)
        dev_p32 = torch.empty(2 * M, dtype=torch.int32, device=device)
        _lib.check(lib.mmbert_split_rows(_stream(), rs_d.data_ptr(), rp_d.data_ptr(), self.seq_start.data_ptr(), start_b_dev.data_ptr(),
                                         self.kv_len.data_ptr(), 0, M, 0, dev_p.data_ptr(), dev_p.data_ptr() + 8 * M, _ptr(rank),
                                         dev_p32.data_ptr(), dev_p32.data_ptr() + 4 * M), "mmbert_split_rows")
        self.perm, self.inv = dev_p[:M], dev_p[M:]
        self.perm32, self.inv32 = dev_p32[:M], dev_p32[M:]
        self.rows_packed = M
        self.base, self.heads, self.tokens, self.lens = base, base.heads, base.tokens, base.lens
        self.seq_len, self.elem_base, self.bias_start, self.bias_len = base.seq_len, base.elem_base, base.bias_start, base.bias_len
        self._words, self._host = words, None

    def _resolve(self):
        if self._host is None:
            if self._words is not None:
                host, ev = self._words
                ev.synchronize()
                v = np.minimum(host.numpy().astype(np.int64), np.asarray(self.lens, dtype=np.int64))
            else:                                            # (no pinned words handed over: one blocking read)
                v = self.kv_len.cpu().numpy().astype(np.int64)
            rows = self.base._rows_f
            self._host = (int(v.sum()), int(((v + rows - 1) // rows).sum()), [int(x) for x in v])
        return self._host

    rows_a = property(lambda self: self._resolve()[0])
    ntiles = property(lambda self: self._resolve()[1])
    valid_host = property(lambda self: self._resolve()[2])


def pad_key_bias(key_bias, layout: "SeqLayout"):
    """[tokens] additive key bias -> the padded per-sequence layout the attention kernels read."""
    out = torch.full((layout.bias_len,), NO_KEY, device=key_bias.device, dtype=torch.float32)
    out.index_copy_(0, layout.bias_index, key_bias.float())
    return out


_MASK_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int64: 2, torch.int32: 3, torch.bfloat16: 4, torch.uint8: 5, torch.bool: 5, torch.float16: 6}


class Prologue:
    """Result of ``prologue()``: padded key bias, kv_len / valid per sequence, the labelled-row list and the host words' device copy."""
    __slots__ = ("key_bias", "kv_len", "valid", "idx", "words", "nseq", "rank")


def prologue_sizes(pass_lens, B, rowset=False):
    """(floats of the padded key bias, ints of everything else) that ``prologue`` writes for these shapes (row-set mode: the key
    bias twice -- original and valid-first order -- and one rank per row)."""
    nseq = len(pass_lens) * B
    tokens = sum(B * S for S in pass_lens)
    nf = sum(B * ((S + 127) // 128 * 128) for S in pass_lens)
    ni = 2 * nseq + 3 * nseq + max(tokens, 1) + nseq + 3
    return (2 * nf, ni + max(tokens, 1)) if rowset else (nf, ni)


def prologue(segs, pass_lens, B, labels, vocab, device, bufs=None, rowset=False) -> Prologue:
    """The step prologue in two launches (mmbert_prologue): ``segs`` = [(mask2d [B, len] -- any stride, any mask dtype --, pass index,
    first position)], ``pass_lens`` = positions per sequence of every pass, ``labels`` = int64 [tokens] in packed order or None.
    ``bufs`` = (fp32, int32) output buffers of ``prologue_sizes`` elements (the caller's persistent ones) instead of fresh tensors.
    ``rowset``: row-set mode (see mmbert_prologue) -- ``valid`` counts the active rows, ``rank`` orders every sequence valid-first
    and ``key_bias`` comes back in THAT order (for SplitLayout(..., rank=...))."""
    lib = _lib.load()
    n, npass = len(segs), len(pass_lens)
    nseq = npass * B
    tokens = sum(B * S for S in pass_lens)
    bias_len, nints = prologue_sizes(pass_lens, B)
    nf_all, ni_all = prologue_sizes(pass_lens, B, rowset)
    out = Prologue()
    out.nseq = nseq
    out.rank = None
    if bufs is not None:
        fl, ints_all = bufs[0][:nf_all], bufs[1][:ni_all]
        assert fl.numel() == nf_all and ints_all.numel() == ni_all and ints_all.dtype == torch.int32 and fl.dtype == torch.float32
    else:
        fl = torch.empty(nf_all, device=device, dtype=torch.float32)
        ints_all = torch.empty(ni_all, device=device, dtype=torch.int32)
    out.key_bias, ints = fl[:bias_len], ints_all[:nints]
    bias_perm = None
    if rowset:
        out.rank, bias_perm = ints_all[nints:], fl[bias_len:]
    out.kv_len, out.valid = ints[:nseq], ints[nseq:2 * nseq]
    seq_cnt = ints[2 * nseq:5 * nseq]
    out.idx = ints[5 * nseq:5 * nseq + max(tokens, 1)]
    out.words = ints[5 * nseq + max(tokens, 1):]
    PA, LA, IA = ctypes.c_void_p * max(n, 1), ctypes.c_longlong * max(n, 1), ctypes.c_int * max(n, 1)
    for m, _, _ in segs:
        assert m.dim() == 2 and m.shape[0] == B and m.device.type == "cuda" and m.dtype in _MASK_DTYPES, (m.shape, m.dtype, m.device)
    es = [m.element_size() for m, _, _ in segs]
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == tokens
    _lib.check(lib.mmbert_prologue(_stream(), n, PA(*[m.data_ptr() for m, _, _ in segs]), LA(*[m.stride(0) * e for (m, _, _), e in zip(segs, es)]),
                                   LA(*[m.stride(1) * e for (m, _, _), e in zip(segs, es)]), IA(*[_MASK_DTYPES[m.dtype] for m, _, _ in segs]),
                                   IA(*[p for _, p, _ in segs]), IA(*[o for _, _, o in segs]), IA(*[m.shape[1] for m, _, _ in segs]),
                                   npass, (ctypes.c_int * npass)(*pass_lens), B, _ptr(labels), int(vocab),
                                   out.key_bias.data_ptr(), out.kv_len.data_ptr(), out.valid.data_ptr(), seq_cnt.data_ptr(), out.idx.data_ptr(),
                                   out.words.data_ptr(), _ptr(out.rank), _ptr(bias_perm)), "mmbert_prologue")
    if rowset:
        out.key_bias = bias_perm
    return out


def attn_kv_len(key_bias_padded, layout: SeqLayout):
    """int32 [sequences]: per sequence the count of leading keys behind which every key is masked out (see
    mmbert_attn_kv_len); pass it to attn_fwd / attn_bwd as ``kv_len`` and they skip those keys (exact)."""
    assert key_bias_padded.numel() == layout.bias_len and key_bias_padded.dtype == torch.float32
    out = torch.empty(layout.seq_len.numel(), device=key_bias_padded.device, dtype=torch.int32)
    _lib.check(_lib.load().mmbert_attn_kv_len(_stream(), key_bias_padded.data_ptr(), layout.bias_start.data_ptr(), layout.seq_len.data_ptr(),
                                              out.numel(), out.data_ptr()), "mmbert_attn_kv_len")
    return out


def attn_fwd(qkv, key_bias, layout: SeqLayout, H, *, drop: Drop = None, ctx=None, lse=None, kv_len=None):
    """``key_bias``: padded layout (pad_key_bias); a [tokens] vector is padded on the fly."""
    lib = _lib.load()
    M = qkv.shape[0]
    if key_bias.numel() != layout.bias_len or layout.bias_len == M:
        key_bias = pad_key_bias(key_bias, layout) if key_bias.numel() == M else key_bias
    assert qkv.shape[1] == 3 * H and qkv.is_contiguous()
    if getattr(layout, "split", False):
        kv_len = layout.kv_len                           # keys of a sequence = its region-A rows
    if ctx is None:
        ctx = torch.empty((M, H), device=qkv.device, dtype=torch.bfloat16)
    if lse is None:
        lse = torch.empty((M, layout.heads), device=qkv.device, dtype=torch.float32)
    d = drop or NO_DROP
    _lib.check(lib.mmbert_attn_fwd(_stream(), qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), key_bias.data_ptr(), layout.bias_start.data_ptr(), H, layout.heads,
                                   layout.seq_start.data_ptr(), layout.seq_len.data_ptr(), layout.elem_base.data_ptr(),
                                   layout.ftile_seq.data_ptr(), layout.ftile_r0.data_ptr(), layout.nftiles, d[0], d[1], d[2], _ptr(kv_len),
                                   _ptr(getattr(layout, "ftile_qshift", None)), _ptr(getattr(layout, "ftile_qend", None))), "mmbert_attn_fwd")
    return ctx, lse


def attn_q_limit(rows32, layout):
    """Per sequence of ``layout``: 1 + the largest query index among the packed rows ``rows32`` (int32) -- what ``attn_bwd(q_limit=...)``
    takes when only those rows have a non-zero output gradient (mmbert_attn_q_limit)."""
    ns = layout.seq_len.numel()
    out = torch.empty(ns, device=rows32.device, dtype=torch.int32)
    _lib.check(_lib.load().mmbert_attn_q_limit(_stream(), rows32.data_ptr(), rows32.numel(), layout.seq_start.data_ptr(), ns, out.data_ptr()),
               "mmbert_attn_q_limit")
    return out


def attn_bwd(qkv, ctx, dctx, lse, key_bias, layout: SeqLayout, H, *, drop: Drop = None, dqkv=None, kv_len=None, q_limit=None):
    lib = _lib.load()
    M = qkv.shape[0]
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    delta = torch.empty((M, layout.heads), device=qkv.device, dtype=torch.float32)
    d = drop or NO_DROP
    if key_bias.numel() != layout.bias_len or layout.bias_len == M:
        key_bias = pad_key_bias(key_bias, layout) if key_bias.numel() == M else key_bias
    assert dctx.is_contiguous() and ctx.is_contiguous()
    split = getattr(layout, "split", False)
    if split:                      # backward covers region A only: its query tiles double as the key tiles (same 128-row grid)
        q_seq, q_r0, nq = layout.tile_seq, layout.tile_r0, layout.ntiles
        kv_len = layout.kv_len
    else:
        q_seq, q_r0, nq = layout.ftile_seq, layout.ftile_r0, layout.nftiles
    _lib.check(lib.mmbert_attn_bwd(_stream(), qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                   key_bias.data_ptr(), layout.bias_start.data_ptr(), H, layout.heads, layout.seq_start.data_ptr(), layout.seq_len.data_ptr(),
                                   layout.elem_base.data_ptr(), q_seq.data_ptr(), q_r0.data_ptr(), nq,
                                   layout.tile_seq.data_ptr(), layout.tile_r0.data_ptr(), layout.ntiles,
                                   d[0], d[1], d[2], _ptr(kv_len), _ptr(getattr(layout, "qtile_qshift", None)), _ptr(getattr(layout, "qtile_qend", None)),
                                   1 if split else 0, _ptr(q_limit)), "mmbert_attn_bwd")
    return dqkv


def attn_dropout_mask(S, elem_base, head, drop, device):
    lib = _lib.load()
    out = torch.empty((S, S), device=device, dtype=torch.uint8)
    _lib.check(lib.mmbert_attn_dropout_mask(_stream(), out.data_ptr(), S, elem_base, head, drop[0], drop[1]), "mmbert_attn_dropout_mask")
    return out


def ce_fwd(logits, V, labels, seg_bounds, nseg):
    """Returns (loss_mean_per_segment[nseg], inv_count[4], row_lse[M])  (see mmbert_ce_fwd)."""
    lib = _lib.load()
    M = logits.shape[0]
    inv = torch.empty(4, device=logits.device, dtype=torch.float32)
    loss = torch.empty(nseg, device=logits.device, dtype=torch.float32)   # exactly nseg (<= 4) entries are written: returned as it is, not as a view
    lse = torch.empty(M, device=logits.device, dtype=torch.float32)
    assert logits.dtype in (torch.bfloat16, torch.float32)
    row_loss = _ws_f32(M, logits.device) if deterministic() else None      # (deterministic mode: the rows' loss terms, summed in a fixed order)
    _lib.check(lib.mmbert_ce_fwd(_stream(), logits.data_ptr(), logits.stride(0), V, labels.data_ptr(), M, seg_bounds.data_ptr(), nseg,
                                 inv.data_ptr(), loss.data_ptr(), lse.data_ptr(), 1 if logits.dtype == torch.float32 else 0, _ptr(row_loss)),
               "mmbert_ce_fwd")
    return loss, inv, lse


def ce_bwd(logits, V, labels, seg_bounds, nseg, inv, gscale, lse, dlogits, rows=None):
    """dlogits (may be ``logits`` itself) = d(sum_s gscale[s] * loss_s) / d(logits)  (see mmbert_ce_bwd).
    ``rows`` (int32 row list): compact output, dlogits[j] = gradient of row rows[j]."""
    lib = _lib.load()
    M = logits.shape[0]
    _lib.check(lib.mmbert_ce_bwd(_stream(), logits.data_ptr(), logits.stride(0), V, labels.data_ptr(), M, seg_bounds.data_ptr(), nseg,
                                 inv.data_ptr(), gscale.data_ptr(), lse.data_ptr(), dlogits.data_ptr(), dlogits.stride(0),
                                 _ptr(rows), 0 if rows is None else rows.numel(), 1 if logits.dtype == torch.float32 else 0), "mmbert_ce_bwd")
    return dlogits


def active_rows(labels, V):
    """(idx int32 [M], count int32 [1]) on the device: rows with a label in [0, V), ascending (see mmbert_active_rows)."""
    lib = _lib.load()
    M = labels.numel()
    idx = torch.empty(max(M, 1), device=labels.device, dtype=torch.int32)
    count = torch.empty(1, device=labels.device, dtype=torch.int32)
    _lib.check(lib.mmbert_active_rows(_stream(), labels.data_ptr(), M, V, idx.data_ptr(), count.data_ptr()), "mmbert_active_rows")
    return idx, count


def adamw(p, g, m, v, p_bf16, flags, *, lr, beta1=0.9, beta2=0.999, eps=1e-6, wd=0.01, step=1, gscale=1.0, mode=0, zero_grad=True):
    lib = _lib.load()
    _lib.check(lib.mmbert_adamw(_stream(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), flags.data_ptr(), p.numel(),
                                float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), float(gscale), int(mode),
                                1 if zero_grad else 0), "mmbert_adamw")


def gelu_bwd(dy, u, du=None):
    if du is None:
        du = torch.empty_like(dy)
    assert dy.is_contiguous() and u.is_contiguous() and du.is_contiguous()
    _lib.check(_lib.load().mmbert_gelu_bwd(_stream(), dy.data_ptr(), u.data_ptr(), du.data_ptr(), dy.numel()), "mmbert_gelu_bwd")
    return du


def cast_f32_bf16(x, y):
    _lib.check(_lib.load().mmbert_cast_f32_bf16(_stream(), x.data_ptr(), y.data_ptr(), x.numel()), "mmbert_cast_f32_bf16")
    return y


def cast_bf16_f32(x, y):
    _lib.check(_lib.load().mmbert_cast_bf16_f32(_stream(), x.data_ptr(), y.data_ptr(), x.numel()), "mmbert_cast_bf16_f32")
    return y


def transpose_cast(src_flat, dst_flat, descs_dev, ndesc, total_tiles):
    """Batched transposes into bf16; ``src_flat`` fp32, or bf16 with the same element offsets."""
    fn = "mmbert_transpose_cast" if src_flat.dtype == torch.float32 else "mmbert_transpose_bf16"
    assert src_flat.dtype in (torch.float32, torch.bfloat16)
    _lib.check(getattr(_lib.load(), fn)(_stream(), src_flat.data_ptr(), dst_flat.data_ptr(), descs_dev.data_ptr(), ndesc, total_tiles), fn)


# ------------------------------------------------------------------------------------ pretraining heads (csrc/heads.hip)
def _ptr3(ts):
    return _PtrArr[3](*[t.data_ptr() for t in ts])


def heads_gate_fwd(P, Apre, vws, vbs, B):
    H = P.shape[1]
    g = torch.empty(3 * B, device=P.device, dtype=torch.float32)
    Cc = torch.empty((B, 3 * H), device=P.device, dtype=torch.float32)
    _lib.check(_lib.load().mmbert_heads_gate_fwd(_stream(), P.data_ptr(), Apre.data_ptr(), _ptr3(vws), _ptr3(vbs), B, H, g.data_ptr(), Cc.data_ptr()),
               "mmbert_heads_gate_fwd")
    return g, Cc


def heads_loss_fwd(P, XP, rel, ap, lo, sent, B, beta, tanh_lo, mlm=None, alpha=1.0):
    """returns (out5, seeds, loss, aux) with out5 = [ap_loss, label_loss, nce, heads_loss, alpha * mean(mlm) + heads_loss],
    seeds = [dXP (3BH) | dPc (3BH) | dlo (B) | drel (4B)] for an upstream gradient of 1, loss = out5[4] (0-dim) and aux = out5[:3] in
    buffers of their own (no views: what an autograd function may return)."""
    H = P.shape[1]
    n = 3 * B * H
    seeds = torch.empty(2 * n + 5 * B, device=P.device, dtype=torch.float32)
    out4 = torch.empty(5, device=P.device, dtype=torch.float32)
    if mlm is not None:
        assert mlm.dtype == torch.float32 and mlm.is_contiguous()
    part = torch.empty(3, device=P.device, dtype=torch.float32)
    loss = torch.empty((), device=P.device, dtype=torch.float32)
    aux = torch.empty(3, device=P.device, dtype=torch.float32)
    base = seeds.data_ptr()
    _lib.check(_lib.load().mmbert_heads_loss_fwd(_stream(), P.data_ptr(), XP.data_ptr(), rel.data_ptr(), ap.data_ptr(), lo.data_ptr(), sent.data_ptr(),
                                                 B, H, float(beta), int(tanh_lo), out4.data_ptr(), base, base + 4 * n, base + 4 * (2 * n + B), base + 4 * 2 * n,
                                                 part.data_ptr(), _ptr(mlm), 0 if mlm is None else mlm.numel(), float(alpha),
                                                 loss.data_ptr(), aux.data_ptr()), "mmbert_heads_loss_fwd")
    return out4, seeds, loss, aux


def heads_scale(x, s):
    _lib.check(_lib.load().mmbert_heads_scale(_stream(), x.data_ptr(), x.numel(), s.data_ptr()), "mmbert_heads_scale")


def heads_seed(seeds, d, nzero, nmlm, coef):
    """(seeds * d, a zero-filled fp32 buffer of nzero elements, d * coef expanded to nmlm elements or None) in one launch (mmbert_heads_seed)."""
    buf = torch.empty(seeds.numel() + nzero + nmlm, device=seeds.device, dtype=torch.float32)
    n = seeds.numel()
    base = buf.data_ptr()
    _lib.check(_lib.load().mmbert_heads_seed(_stream(), seeds.data_ptr(), n, d.data_ptr(), base, base + 4 * n, nzero, base + 4 * (n + nzero), nmlm, float(coef)),
               "mmbert_heads_seed")
    return buf[:n], buf[n:n + nzero], (buf[n + nzero:] if nmlm else None)


def heads_gate_bwd(dC, P, Apre, g, vws, dPc, B):
    """returns dP, dApre, E (= dg * relu(Apre)) [3B,H] and dg [3B]."""
    H = P.shape[1]
    dP, dA, E = torch.empty_like(P), torch.empty_like(P), torch.empty_like(P)
    dg = torch.empty(3 * B, device=P.device, dtype=torch.float32)
    _lib.check(_lib.load().mmbert_heads_gate_bwd(_stream(), dC.data_ptr(), P.data_ptr(), Apre.data_ptr(), g.data_ptr(), _ptr3(vws), dPc.data_ptr(), B, H,
                                                 dP.data_ptr(), dA.data_ptr(), E.data_ptr(), dg.data_ptr()), "mmbert_heads_gate_bwd")
    return dP, dA, E, dg


def heads_tanh_(x):
    _lib.check(_lib.load().mmbert_heads_tanh(_stream(), x.data_ptr(), x.numel()), "mmbert_heads_tanh")
    return x


def heads_tanh_bwd(dP, P):
    dpre = torch.empty_like(P)
    _lib.check(_lib.load().mmbert_heads_tanh_bwd(_stream(), dP.data_ptr(), P.data_ptr(), dpre.data_ptr(), P.numel()), "mmbert_heads_tanh_bwd")
    return dpre


class RowInverse:
    """The stamped inverse of a row list over ``nrows`` rows (mmbert_compact_rows_inv / mmbert_scatter_rows_zero): one persistent int64
    array, zero-filled once, and a stamp that ``next()`` advances per list -- entries of earlier lists simply stop counting."""

    def __init__(self, nrows, device):
        self.table = torch.zeros(nrows, device=device, dtype=torch.int64)
        self.stamp, self.nlist = 0, 0

    def next(self, nlist):
        self.stamp = self.stamp % 0xFFFFFFF0 + 1                 # never 0 (the zero fill); a wrap after 4e9 lists meets no live entry
        self.nlist = nlist
        return self.stamp


def compact_rows(rows, extra, row_map=None, inverse: "RowInverse" = None):
    """(int64, int32) lists ``map[cat(rows, extra)]`` in one launch (mmbert_compact_rows): rows int32 or int64, extra int64, row_map int64
    or None.  ``inverse``: a RowInverse that receives the list's inverse (for ``scatter_rows_zero``)."""
    n, ne = rows.numel(), extra.numel()
    out64 = torch.empty(n + ne, device=extra.device, dtype=torch.int64)
    out32 = torch.empty(n + ne, device=extra.device, dtype=torch.int32)
    assert rows.dtype in (torch.int32, torch.int64) and extra.dtype == torch.int64 and rows.is_contiguous() and extra.is_contiguous()
    assert row_map is None or (row_map.dtype == torch.int64 and row_map.is_contiguous())
    r32, r64 = (rows.data_ptr(), None) if rows.dtype == torch.int32 else (None, rows.data_ptr())
    if inverse is None:
        _lib.check(_lib.load().mmbert_compact_rows(_stream(), r32, r64, n, extra.data_ptr(), ne, _ptr(row_map), out64.data_ptr(), out32.data_ptr()), "mmbert_compact_rows")
    else:
        _lib.check(_lib.load().mmbert_compact_rows_inv(_stream(), r32, r64, n, extra.data_ptr(), ne, _ptr(row_map), out64.data_ptr(), out32.data_ptr(),
                                                       inverse.table.data_ptr(), inverse.next(n + ne)), "mmbert_compact_rows_inv")
    return out64, out32


def scatter_rows_zero(srcs, inverse: "RowInverse", nrows):
    """[zeros(nrows, ...).index_copy_(0, list, t) for t in srcs] in ONE launch (mmbert_scatter_rows_zero): ``srcs`` = up to 4 2-D tensors
    (unit inner stride, rows a multiple of 16 bytes) whose row i belongs at row list[i] of the output, ``inverse`` the RowInverse that
    ``compact_rows`` filled for that list."""
    n = len(srcs)
    assert 0 < n <= 4 and nrows <= inverse.table.numel() and inverse.stamp != 0
    outs = [torch.empty((nrows, t.shape[1]), device=t.device, dtype=t.dtype) for t in srcs]
    if nrows == 0:
        return outs
    for t in srcs:
        assert t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == inverse.nlist
    LA = ctypes.c_longlong * n
    rb = [t.shape[1] * t.element_size() for t in srcs]
    _lib.check(_lib.load().mmbert_scatter_rows_zero(_stream(), n, _PtrArr[n](*[t.data_ptr() for t in srcs]), _PtrArr[n](*[o.data_ptr() for o in outs]),
                                                    LA(*[t.stride(0) * t.element_size() for t in srcs]), LA(*rb), _IntArr[n](*rb),
                                                    inverse.table.data_ptr(), inverse.stamp, inverse.nlist, nrows), "mmbert_scatter_rows_zero")
    return outs


def gather_rows(srcs, idx32):
    """[t.index_select(0, idx) for t in srcs] in ONE launch (mmbert_gather_rows): ``srcs`` = up to 12 tensors (1-D or 2-D, unit inner
    stride, element size a multiple of 4 bytes per row) sharing the int32 row list ``idx32``."""
    n = len(srcs)
    assert 0 < n <= 12 and idx32.dtype == torch.int32 and idx32.is_contiguous()
    rows = idx32.numel()
    outs = [torch.empty((rows,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype) for t in srcs]
    if rows == 0:
        return outs
    for t in srcs:
        assert t.dim() in (1, 2) and (t.dim() == 1 or t.stride(1) == 1)
    LA = ctypes.c_longlong * n
    rb = [(t.shape[1] if t.dim() == 2 else 1) * t.element_size() for t in srcs]
    _lib.check(_lib.load().mmbert_gather_rows(_stream(), n, _PtrArr[n](*[t.data_ptr() for t in srcs]), _PtrArr[n](*[o.data_ptr() for o in outs]),
                                              LA(*[t.stride(0) * t.element_size() for t in srcs]), LA(*[b for b in rb]),
                                              _IntArr[n](*rb), idx32.data_ptr(), rows), "mmbert_gather_rows")
    return outs


def pack_i64(segments, total=None):
    """One int64 tensor holding ``segments`` back to back in ONE launch (mmbert_pack_i64): a segment is an int64 tensor (flattened;
    contiguous) or ``(count, fill)`` for a constant run.  Returns (packed, [start offsets])."""
    LA = ctypes.c_longlong * len(segments)
    srcs, offs, cnts, fills, keep = [], [], [], [], []
    off = 0
    dev = None
    for sg in segments:
        if torch.is_tensor(sg):
            assert sg.dtype == torch.int64 and sg.is_cuda
            t = sg.reshape(-1)
            if not t.is_contiguous():
                t = t.contiguous()
            keep.append(t)
            srcs.append(t.data_ptr()); cnts.append(t.numel()); fills.append(0)
            dev = t.device
        else:
            srcs.append(None); cnts.append(int(sg[0])); fills.append(int(sg[1]))
        offs.append(off)
        off += cnts[-1]
    out = torch.empty(off if total is None else total, device=dev, dtype=torch.int64)
    n = len(segments)
    _lib.check(_lib.load().mmbert_pack_i64(_stream(), n, _PtrArr[n](*srcs), LA(*offs), LA(*cnts), LA(*fills), out.data_ptr()), "mmbert_pack_i64")
    return out, offs


def heads_colsum(pairs):
    """pairs: list of (src [rows, cols] fp32 with unit column stride, dst [cols] fp32): dst += column sums, one launch."""
    n = len(pairs)
    PA, IA = _PtrArr[n], _IntArr[n]
    _lib.check(_lib.load().mmbert_heads_colsum(_stream(), n, PA(*[s.data_ptr() for s, _ in pairs]), PA(*[d.data_ptr() for _, d in pairs]),
                                               IA(*[s.shape[0] for s, _ in pairs]), IA(*[s.shape[1] for s, _ in pairs]),
                                               IA(*[s.stride(0) for s, _ in pairs])), "mmbert_heads_colsum")


# ------------------------------------------------------------------------------------ the heads, one launch per dependency level (csrc/heads_coop.hip)
_VP, _VP3 = ctypes.c_void_p, ctypes.c_void_p * 3


class _HeadsStep(ctypes.Structure):
    """Mirror of ``mmbert_heads_step`` (include/mmbert_hip.h), field for field."""
    _fields_ = ([("B", ctypes.c_int), ("H", ctypes.c_int), ("tanh_lo", ctypes.c_int), ("nmlm", ctypes.c_int), ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
                 ("first", _VP), ("y", _VP), ("first_rows", _VP), ("ldy", ctypes.c_int), ("pad0_", ctypes.c_int), ("ap", _VP), ("ap2", _VP), ("sent", _VP), ("mlm", _VP)]
                + [(n, _VP) for n in ("Wp", "bp", "Wal", "bal", "Wsr", "bsr", "Wat", "bat")] + [("vw", _VP3), ("vb", _VP3)]
                + [(n, _VP) for n in ("Wc1", "bc1", "Wc2", "bc2")] + [("Wq", _VP3), ("bq", _VP3)]
                + [(n, _VP) for n in ("loss", "aux", "out5", "logits", "t_rel", "rel", "ws", "dloss", "dfirst", "dmlm")]
                + [(n, _VP) for n in ("gWp", "gbp", "gWal", "gbal", "gWat", "gbat")] + [("gvw", _VP3), ("gvb", _VP3)]
                + [(n, _VP) for n in ("gWc1", "gbc1", "gWc2", "gbc2")] + [("gWq", _VP3), ("gbq", _VP3)] + [("sync", _VP)])


_heads_sync = {}
_heads_struct_checked = False


def heads_step_struct() -> "_HeadsStep":
    global _heads_struct_checked
    if not _heads_struct_checked:
        n = _lib.load().mmbert_heads_step_struct_size()
        if n != ctypes.sizeof(_HeadsStep):
            raise RuntimeError(f"mmbert_heads_step: the library's struct has {n} bytes, the binding's {ctypes.sizeof(_HeadsStep)}")
        _heads_struct_checked = True
    return _HeadsStep()


def heads_step_sync(device) -> torch.Tensor:
    """The four zeroed counter words of the heads' loss level for the current stream (the kernel leaves them zeroed)."""
    key = (device, _stream())
    t = _heads_sync.get(key)
    if t is None:
        t = _heads_sync[key] = torch.zeros(4, device=device, dtype=torch.int32)
    return t


def heads_step_workspace(B: int, H: int, device) -> torch.Tensor:
    return torch.empty(_lib.load().mmbert_heads_step_workspace(int(B), int(H)) // 4, device=device, dtype=torch.float32)


def heads_step_fwd(a: "_HeadsStep", lo: int = 1, hi: int = 7):
    _lib.check(_lib.load().mmbert_heads_step_fwd_levels(_stream(), ctypes.addressof(a), lo, hi), "mmbert_heads_step_fwd_levels")


def heads_step_dmlm(a: "_HeadsStep"):
    _lib.check(_lib.load().mmbert_heads_step_dmlm(_stream(), ctypes.addressof(a)), "mmbert_heads_step_dmlm")


def heads_step_bwd(a: "_HeadsStep", lo: int = 1, hi: int = 6):
    _lib.check(_lib.load().mmbert_heads_step_bwd_levels(_stream(), ctypes.addressof(a), lo, hi), "mmbert_heads_step_bwd_levels")


# ------------------------------------------------------------------------------------ the heads' dense layers (skinny fp32 products)
class _SkSrc(ctypes.Structure):
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("ldx", ctypes.c_int), ("ldw", ctypes.c_int), ("inner", ctypes.c_int),
                ("row0", ctypes.c_int), ("rows", ctypes.c_int), ("w_inner_major", ctypes.c_int)]


class _SkOp(ctypes.Structure):
    _fields_ = [("Y", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("ldy", ctypes.c_int), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("nsrc", ctypes.c_int), ("act", ctypes.c_int), ("accumulate", ctypes.c_int), ("src", _SkSrc * 4)]


class _SkWOp(ctypes.Structure):
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("ldy", ctypes.c_int),
                ("ldx", ctypes.c_int), ("ldw", ctypes.c_int), ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int)]


def _f32_2d(t):
    assert t.dtype == torch.float32 and t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1) and t.is_cuda, (t.dtype, t.shape, t.stride())
    return t


def skinny_mm(oplist):
    """One launch of mmbert_skinny_mm: Y += bias + sum of the sources' products (fp32 atomics: Y must be zeroed, or hold what to add
    to).  Each op: (Y [M,N], bias [N] | None, 0, accumulate (informational), sources) with sources =
    [(X [rows, inner], W, w_inner_major, row0)]: W is [N, inner] (y = x W^T) for w_inner_major = 0, [inner, N] (y = x W) for 1;
    the source feeds the output rows row0 .. row0 + rows.  fp32, unit column stride, any row stride."""
    n = len(oplist)
    arr = (_SkOp * n)()
    keep = []
    for o, (Y, bias, act, accumulate, sources) in zip(arr, oplist):
        _f32_2d(Y)
        o.Y, o.bias, o.ldy, o.M, o.N = Y.data_ptr(), _ptr(bias), Y.stride(0), Y.shape[0], Y.shape[1]
        o.nsrc, o.act, o.accumulate = len(sources), int(act), 1 if accumulate else 0
        for sc, (X, W, wim, row0) in zip(o.src, sources):
            _f32_2d(X); _f32_2d(W)
            inner = X.shape[1]
            assert (W.shape == (inner, Y.shape[1])) if wim else (W.shape == (Y.shape[1], inner)), (W.shape, inner, Y.shape, wim)
            sc.X, sc.W, sc.ldx, sc.ldw, sc.inner, sc.row0, sc.rows, sc.w_inner_major = X.data_ptr(), W.data_ptr(), X.stride(0), W.stride(0), inner, row0, X.shape[0], 1 if wim else 0
            keep.append((X, W))
    lib = _lib.load()
    if deterministic():
        ws = _ws_f32((lib.mmbert_skinny_mm_workspace(n, ctypes.cast(arr, ctypes.c_void_p)) + 3) // 4, oplist[0][0].device)
        _lib.check(lib.mmbert_skinny_mm_ordered(_stream(), n, ctypes.cast(arr, ctypes.c_void_p), ws.data_ptr()), "mmbert_skinny_mm_ordered")
        return
    _lib.check(lib.mmbert_skinny_mm(_stream(), n, ctypes.cast(arr, ctypes.c_void_p)), "mmbert_skinny_mm")


def skinny_wgrad(oplist):
    """One launch of mmbert_skinny_wgrad.  Each op: (dY [M,N], X [M,K], dW [N,K] (+=), db [N] | None (+= column sums of dY))."""
    n = len(oplist)
    arr = (_SkWOp * n)()
    for o, (dY, X, dW, db) in zip(arr, oplist):
        _f32_2d(dY); _f32_2d(X); _f32_2d(dW)
        assert dY.shape[0] == X.shape[0] and dW.shape == (dY.shape[1], X.shape[1]), (dY.shape, X.shape, dW.shape)
        o.dY, o.X, o.dW, o.db = dY.data_ptr(), X.data_ptr(), dW.data_ptr(), _ptr(db)
        o.ldy, o.ldx, o.ldw, o.M, o.N, o.K = dY.stride(0), X.stride(0), dW.stride(0), dY.shape[0], dY.shape[1], X.shape[1]
    _lib.check(_lib.load().mmbert_skinny_wgrad(_stream(), n, ctypes.cast(arr, ctypes.c_void_p)), "mmbert_skinny_wgrad")


# ------------------------------------------------------------------------------------ composite encoder-layer calls (mmbert_layer_fwd / _bwd)
class _Drop(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_uint32), ("thr16", ctypes.c_uint32), ("scale", ctypes.c_float)]


_VP, _CI = ctypes.c_void_p, ctypes.c_int


class _AttnLayout(ctypes.Structure):
    _fields_ = [(n, _VP) for n in ("key_bias", "bias_start", "seq_start", "seq_len", "elem_base", "ftile_seq", "ftile_r0", "ftile_qshift", "ftile_qend",
                                    "qtile_seq", "qtile_r0", "qtile_qshift", "qtile_qend", "tile_seq", "tile_r0", "kv_len")] + \
               [(n, _CI) for n in ("nftiles", "nqtiles", "ntiles", "split", "heads", "pad_")]


class _LayerFwd(ctypes.Structure):
    _fields_ = [(n, _VP) for n in ("x", "Wqkv", "Wo", "W1", "W2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                                    "qkv", "actx", "lse", "z1", "y1", "m1", "r1", "u", "g", "z2", "y2", "m2", "r2", "y2_rows", "tile_queue")] + \
               [("att", _Drop), ("h1", _Drop), ("h2", _Drop)] + [(n, _CI) for n in ("rows", "H", "I", "ldx", "ldy2")] + [("ln_eps", ctypes.c_float)]


class _LayerBwd(ctypes.Structure):
    _fields_ = [(n, _VP) for n in ("dy", "dy_rows", "z2", "m2", "r2", "ln2_g", "g_ln2_g", "g_ln2_b", "ln2_ws", "z1", "m1", "r1", "ln1_g", "g_ln1_g", "g_ln1_b", "ln1_ws",
                                    "u", "qkv", "actx", "lse", "W2T", "W1T", "WoT", "WqkvT",
                                    "dz2", "dz2d", "du", "dy1", "dz1", "dz1d", "dctx", "dqkv", "delta", "dx", "tile_queue")] + \
               [("att", _Drop), ("h1", _Drop), ("h2", _Drop)] + [(n, _CI) for n in ("rows", "H", "I", "lddy")]


_layer_structs_checked = False


def _check_layer_structs():
    global _layer_structs_checked
    if not _layer_structs_checked:
        sz = (ctypes.c_int * 3)()
        _lib.check(_lib.load().mmbert_layer_struct_sizes(sz), "mmbert_layer_struct_sizes")
        got = (ctypes.sizeof(_AttnLayout), ctypes.sizeof(_LayerFwd), ctypes.sizeof(_LayerBwd))
        if tuple(sz) != got:
            raise RuntimeError(f"msa_amd.ops: structure layout differs from include/mmbert_hip.h: C {tuple(sz)} != ctypes {got}")
        _layer_structs_checked = True


def attn_layout_struct(key_bias, layout, backward: bool):
    """The attention kernels' view of ``layout`` as the C structure of the composite layer calls (the same fields attn_fwd / attn_bwd pass
    one by one; ``backward`` reads the split layout's lazily resolved tile counts)."""
    _check_layer_structs()
    if key_bias.numel() != layout.bias_len or layout.bias_len == layout.tokens:
        key_bias = pad_key_bias(key_bias, layout) if key_bias.numel() == layout.tokens else key_bias
    split = bool(getattr(layout, "split", False))
    L = _AttnLayout()
    L.key_bias, L.bias_start, L.seq_start, L.seq_len, L.elem_base = (key_bias.data_ptr(), layout.bias_start.data_ptr(), layout.seq_start.data_ptr(),
                                                                     layout.seq_len.data_ptr(), layout.elem_base.data_ptr())
    L.ftile_seq, L.ftile_r0, L.nftiles = layout.ftile_seq.data_ptr(), layout.ftile_r0.data_ptr(), layout.nftiles
    L.ftile_qshift, L.ftile_qend = _ptr(getattr(layout, "ftile_qshift", None)), _ptr(getattr(layout, "ftile_qend", None))
    L.kv_len = _ptr(layout.kv_len) if split else None
    L.split, L.heads = 1 if split else 0, layout.heads
    if backward:
        if split:                    # backward covers region A only: its query tiles double as the key tiles (attn_bwd)
            L.qtile_seq, L.qtile_r0, L.nqtiles = layout.tile_seq.data_ptr(), layout.tile_r0.data_ptr(), layout.ntiles
        else:
            L.qtile_seq, L.qtile_r0, L.nqtiles = layout.ftile_seq.data_ptr(), layout.ftile_r0.data_ptr(), layout.nftiles
        L.tile_seq, L.tile_r0, L.ntiles = layout.tile_seq.data_ptr(), layout.tile_r0.data_ptr(), layout.ntiles
        L.qtile_qshift, L.qtile_qend = _ptr(getattr(layout, "qtile_qshift", None)), _ptr(getattr(layout, "qtile_qend", None))
    L._keep = (key_bias, layout)                                   # (the padded bias must outlive the launches)
    return L


def _set_drop(d: _Drop, drop: Drop):
    d.stream, d.thr16, d.scale = drop or NO_DROP


def layer_fwd(L: _AttnLayout, a: _LayerFwd):
    _lib.check(_lib.load().mmbert_layer_fwd(_stream(), ctypes.addressof(L), ctypes.addressof(a)), "mmbert_layer_fwd")


def layer_bwd(L: _AttnLayout, a: _LayerBwd):
    _lib.check(_lib.load().mmbert_layer_bwd(_stream(), ctypes.addressof(L), ctypes.addressof(a)), "mmbert_layer_bwd")


# the per-launch wrappers as defined here: model.py takes the composite path only while nobody has wrapped them (bench.py's per-launch
# event timing, tests that spy on launches)
_UNWRAPPED = dict(gemm_nt=gemm_nt, attn_fwd=attn_fwd, attn_bwd=attn_bwd, ln_fwd=ln_fwd, ln_bwd=ln_bwd)


def launches_unwrapped() -> bool:
    g = globals()
    return all(g[k] is v for k, v in _UNWRAPPED.items())
