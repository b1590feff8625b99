"""Data-parallel training over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference is single-GPU (REF:train.py:22,75); this is the north-star addition.
Design for xGMI (7 point-to-point links per GPU, per-link-bound rings): few LARGE collectives on
contiguous slices of the model's flat fp32 gradient buffer instead of one call per tensor --

* the flat layout is in gradient-completion order (flat.py), so when backward finishes encoder
  layer i the slice [done, end_of_layer_i) is final and its all-reduce (SUM) is issued at once,
  overlapping the remaining backward compute; the tail (embeddings + tied word embedding) is
  reduced by ``finish()``;
* consecutive finished layers are merged until a bucket holds >= ``bucket_mb`` MiB;
* the 1/world_size average is folded into the AdamW kernel's gradient scale (no extra pass);
* never-differentiated parameters hold zeros and are frozen in the optimizer:
  their slices are cut out of the buckets (``GradBucketer(skip=...)``); on accumulation micro-steps (the reference steps every 2nd
  micro-batch, REF:trainer.py:96) ``no_sync()`` skips the exchange.

Parity definition: DP(N ranks, local batch b) == mean over ranks of the single-rank gradients of
the N shards (CPC negatives and CE means are per-rank, as in any DDP run).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradBucketer:
    """All-reduces slices of one flat gradient tensor as they become final.

    ``boundaries``: ascending element offsets b_0=0 < b_1 < ... < b_n=len(flat); slice k =
    [b_k, b_{k+1}) becomes final when ``ready(k)`` is called (k must be called in order).  Pure
    torch.distributed: works on CPU tensors with gloo (tests) and on GPU tensors with RCCL.

    ``wire_dtype`` (opt-in, e.g. torch.bfloat16): the exchange moves that dtype instead of fp32 -- half the bytes on the links.  Not
    an all-reduce in the low precision (which would also ACCUMULATE in it): every rank's slice is cast into a staging buffer, an
    all-to-all hands rank r the r-th chunk of every rank (on xGMI: all seven links at once, no ring), rank r adds the chunks in fp32
    in rank order, rounds the sum once, and an all-gather returns the sums -- every rank ends up with bit-identical values, each
    rank's contribution is rounded once on the way in and the sum once on the way out.  Stage two of a bucket is issued at a FIXED point
    of the program -- right before the next bucket's stage one, or in ``finish()`` -- so that every rank issues its collectives in the
    same order (a first version issued it as soon as a later ``ready()`` FOUND stage one complete: ranks that see the completion at
    different calls interleave all-gathers and all-to-alls differently on one communicator -- a deadlock, met as a flaky gloo test);
    by then stage one has had a whole bucket of backward to complete, so both stages still overlap the rest of backward."""

    def __init__(self, flat_grads: torch.Tensor, boundaries: Sequence[int], group=None, bucket_mb: float = 32.0,
                 skip: Sequence[Tuple[int, int]] = (), wire_dtype: Optional[torch.dtype] = None):
        """``skip``: ascending, disjoint element ranges [a, b) that are never exchanged (parameters the reference never
        differentiates: their gradient slots hold zeros on every rank) -- a bucket is cut around them."""
        assert boundaries[0] == 0 and boundaries[-1] == flat_grads.numel() and list(boundaries) == sorted(boundaries)
        self.flat, self.bounds, self.group = flat_grads, list(boundaries), group
        self.min_elems = int(bucket_mb * (1 << 20) / flat_grads.element_size())
        self.skip = [(int(a), int(b)) for a, b in skip if b > a]
        assert all(self.skip[i][1] <= self.skip[i + 1][0] for i in range(len(self.skip) - 1))
        self.wire_dtype = wire_dtype
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handles: List = []
        self.stage1: List = []   # wire mode: (handle, lo, hi, recv buffer, padded chunk) of buckets whose all-to-all is in flight
        self.stage2: List = []   # wire mode: (handle, lo, hi, gathered sums)
        self.done = 0            # elements already handed to the collective
        self.next_slice = 0
        self.enabled = True
        # world 1 short-circuits the wire-dtype exchange to a plain all-reduce; tests set this to run the all-to-all + fp32 sum +
        # all-gather path over RCCL on the one GPU a test box has (tests/test_train_gpu.py)
        self.force_wire_path = False
        self.calls = 0           # number of collectives issued since reset (observable in tests)
        self.elems = 0           # elements exchanged since reset

    def reset(self):
        self.handles, self.done, self.next_slice, self.calls, self.elems = [], 0, 0, 0, 0
        self.stage1, self.stage2 = [], []

    def _segments(self, lo: int, hi: int):
        """[lo, hi) minus the skip ranges."""
        for a, b in self.skip:
            if b <= lo or a >= hi:
                continue
            if a > lo:
                yield lo, a
            lo = max(lo, b)
        if lo < hi:
            yield lo, hi

    def reduce_range(self, lo: int, hi: int):
        """Starts the exchange of [lo, hi) (minus the skip ranges) now, whatever the slice bookkeeping says -- for a range that is
        final out of layout order (DataParallel: the tied decoder's dense gradient, final right after the MLM head's backward)."""
        if not self.enabled:
            return
        for a, b in self._segments(lo, hi):
            self._start(a, b)

    def _start(self, lo: int, hi: int):
        self.calls += 1
        self.elems += hi - lo
        if self.wire_dtype is None or (self.world == 1 and not self.force_wire_path):
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        self._advance()                                           # stage two of the buckets before this one: same point on every rank
        W, n = self.world, hi - lo
        chunk = (n + W - 1) // W
        send = torch.zeros(W * chunk, device=self.flat.device, dtype=self.wire_dtype)
        send[:n].copy_(self.flat[lo:hi])                          # one cast pass over the slice
        recv = torch.empty_like(send)
        h = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        self.stage1.append((h, lo, hi, recv, chunk, send))

    def _advance(self):
        """Stage two (fp32 sum of the received chunks, all-gather of the rounded sums) of every bucket in stage one, in issue order.
        ``wait()`` blocks the host with gloo; with RCCL it only orders the current stream behind the all-to-all."""
        while self.stage1:
            h, lo, hi, recv, chunk, send = self.stage1[0]
            h.wait()
            self.stage1.pop(0)
            W = self.world
            total = recv.view(W, chunk).to(torch.float32).sum(0).to(self.wire_dtype)      # rank order, fp32: the same bits on every run
            out = torch.empty(W * chunk, device=recv.device, dtype=self.wire_dtype)
            h2 = dist.all_gather_into_tensor(out, total, group=self.group, async_op=True)
            self.stage2.append((h2, lo, hi, out))

    def _issue(self, end: int):
        if end > self.done and self.enabled:
            for lo, hi in self._segments(self.done, end):
                self._start(lo, hi)
        self.done = end

    def ready(self, k: int):
        assert k == self.next_slice, "slices must become ready in layout order"
        self.next_slice = k + 1
        end = self.bounds[k + 1]
        if end - self.done >= self.min_elems:
            self._issue(end)

    def finish(self, upto: Optional[int] = None):
        """Reduce whatever is left (up to element ``upto``: default everything) and wait for every outstanding collective."""
        self._issue(self.bounds[-1] if upto is None else upto)
        self._advance()
        for h in self.handles:
            h.wait()
        for h2, lo, hi, out in self.stage2:
            h2.wait()
            self.flat[lo:hi].copy_(out[:hi - lo])                 # back to fp32 in place
        self.calls_per_step = self.calls
        self.reset()


class DataParallel:
    """Wraps an ``MMBertForPretraining``: broadcast of the initial weights, bucketed gradient
    all-reduce overlapped with backward, gradient averaging folded into the optimizer.

    The tail (``early_word_embedding``, default on).  The flat layout ends with the word-embedding table (94 MB of the 440 MB at the
    headline size), whose gradient has two parts: the tied MLM decoder's DENSE gradient, final right after the MLM head's backward --
    the first thing backward does -- and the embedding lookup's <= passes x B x T scattered rows, final only when backward ends.
    Reduced as one slice it could not start before backward was over (the exposed tail of an 8-GPU step).  So: the table's slice
    is all-reduced as soon as the MLM head's backward is done (``model.head_grad_hook``), the lookup keeps its rows OUT of the table
    (``model.defer_embed_rows``: the trunk hands over ids and row gradients), and ``finish_backward`` exchanges them in compact
    form -- all-gather of the ids, the same sorted union on every rank, the local rows summed into a [union, H] block, ONE
    all-reduce of that block (a few MB), and a scatter of the reduced rows into the table (unique rows: deterministic, every
    rank bit-identical)."""

    def __init__(self, model, optimizer=None, group=None, bucket_mb: float = 32.0, force_dynamic_queue: bool = False,
                 wire_dtype: Optional[torch.dtype] = None, early_word_embedding: bool = True, equal_batch_shapes: bool = False):
        """``equal_batch_shapes``: the caller guarantees that every rank's token-id tensor has the same element count in every step
        (DeviceBatchBuilder / bench.py batches do) -- the id exchange then skips its size agreement (gather_union)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model, self.group = model, group
        self.world = dist.get_world_size(group)
        dev = next(model.parameters()).device
        model._ensure_ready(dev)
        flat = model._flat
        dist.broadcast(flat.params, src=0, group=group)               # REF has one process; ranks must start equal
        flat.refresh()
        if dev.type == "cuda" and (self.world > 1 or force_dynamic_queue):
            # RCCL's channel kernels hold CUs while the all-reduce of finished layers overlaps the rest of backward: let the
            # persistent GEMM draw its tiles from a queue, so that workgroups that start late do not strand a static share
            from . import ops
            ops.dynamic_tile_queue = True
        L = model.config.num_hidden_layers
        # slice k (k = 0..L-1) ends with encoder layer L-1-k; slice L is the embedding tail
        bounds = [0]
        for i in reversed(range(L)):
            last = f"bert.encoder.layer.{i}.attention.self.value.bias"
            bounds.append(flat.offset[last] + flat.numel[last])
        bounds.append(flat.total)
        # frozen 256-element blocks (flat.FROZEN: W_cv / W_cs / seq_relationship -- no gradient in the reference, zeros here) are left out
        fl = (flat.flags == 2).cpu().numpy().astype("int8")
        edges = (fl[1:] != fl[:-1]).nonzero()[0] + 1
        cuts = [0, *[int(e) for e in edges], len(fl)]
        skip = [(256 * a, 256 * b) for a, b in zip(cuts[:-1], cuts[1:]) if fl[a] == 1]
        self.bucketer = GradBucketer(flat.grads, bounds, group, bucket_mb, skip=skip, wire_dtype=wire_dtype)
        self._L = L
        model.grad_hook = self._on_layer_done
        # the word-embedding table: the last tensor of the layout (flat.py)
        wname = "bert.embeddings.word_embeddings.weight"
        self._word = (flat.offset[wname], flat.offset[wname] + flat.vpad * model.config.hidden_size)
        self.early_word = bool(early_word_embedding) and self._word[1] <= flat.total and flat.order[-1] == wname
        self.equal_batch_shapes = bool(equal_batch_shapes)
        self._word_started = False
        self._unions = []            # one entry per differentiated forward since the last finish_backward(): (union, event) or None
        self._unions_cut = False     # entries were dropped (bounded list, _on_ids)
        # the id exchange of step k + 1 must not queue behind the gradient collectives of step k (one communicator runs its
        # collectives in order): it gets a communicator of its own
        self._ids_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None) if self.early_word else None
        if self.early_word:
            model.head_grad_hook = self._on_heads_done
            model.defer_embed_rows = True
            model.embed_ids_hook = self._on_ids
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        self.optimizer = optimizer

    def _on_layer_done(self, i: int):
        self.bucketer.ready(self._L - 1 - i)

    def _on_ids(self, ids: torch.Tensor, stream=None):
        """Start of a forward pass that will be differentiated: the token ids of the step (inputs) -> the union of touched table rows
        over all ranks, long before backward needs it (gather_union reads a size back: here the host is ahead of the GPU)."""
        # A differentiated forward whose backward never comes (a validation pass with grad enabled) would leave its union -- a GPU tensor
        # -- here for ever: the list is bounded by the backwards that can still claim an entry.  Dropping an entry is safe:
        # finish_backward() rebuilds the union when it finds fewer of them than pending backwards (every rank runs the same program,
        # so every rank drops the same entries).
        cap = 4 + 2 * len(self.model.__dict__.get("_deferred_embed_rows") or ())
        if len(self._unions) >= cap:
            del self._unions[:len(self._unions) - cap + 1]
            self._unions_cut = True            # the list no longer holds every forward: finish_backward() rebuilds the union
        if not self.bucketer.enabled:
            self._unions.append(None)          # (a forward under no_sync(): if its backward is exchanged after all, the union is rebuilt)
            return
        if stream is not None:
            with torch.cuda.stream(stream):
                union = gather_union(ids, self.model.config.vocab_size, self._ids_group, self.equal_batch_shapes)
            union.record_stream(torch.cuda.current_stream())
            ev = torch.cuda.Event()
            ev.record(stream)
        else:
            union, ev = gather_union(ids, self.model.config.vocab_size, self._ids_group, self.equal_batch_shapes), None
        self._unions.append((union, ev))

    def _on_heads_done(self):
        """The MLM head's backward is complete: the tied decoder's dense gradient of the word-embedding table is final (the lookup's
        rows stay out of the table: model.defer_embed_rows)."""
        if not self.bucketer.enabled:
            return
        if self._word_started:
            # The table's all-reduce of this step is already in flight: a second MLM-head backward would add into the very slice a
            # collective is reading and writing (and its sum would be reduced a second time).  One differentiated heads backward per
            # finish_backward(); gradient accumulation goes through no_sync(), which keeps every micro-step but the last local.
            raise RuntimeError("DataParallel: a second heads backward before finish_backward() -- wrap accumulation micro-steps in "
                               "no_sync(), or construct DataParallel(early_word_embedding=False)")
        self.bucketer.reduce_range(*self._word)
        self._word_started = True

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient-accumulation micro-step: keep local sums, exchange nothing."""
        self.bucketer.enabled = False
        try:
            yield
        finally:
            self.bucketer.enabled = True
            self.bucketer.reset()
            self._fold_rows_locally()

    def _fold_rows_locally(self):
        """A micro-step without exchange: the lookup's rows go into the local table like any other gradient."""
        for ids, rows in self.model.__dict__.pop("_deferred_embed_rows", None) or ():
            _scatter_rows(self.model._w["g_word"], ids, rows)
        self._unions, self._unions_cut = [], False

    def finish_backward(self):
        """Call after ``loss.backward()`` and before ``optimizer.step()``."""
        bk = self.bucketer
        pend = self.model.__dict__.pop("_deferred_embed_rows", None) or []
        unions, self._unions = self._unions, []
        cut, self._unions_cut = self._unions_cut, False
        started, self._word_started = self._word_started, False
        if not (self.early_word and started):
            for ids, rows in pend:                                 # (no early exchange happened: rows into the table, then the usual tail)
                _scatter_rows(self.model._w["g_word"], ids, rows)
            bk.finish()
            return
        # compact exchange of the lookup's rows, while the table's own all-reduce (started long ago) and the tail are in flight
        block = None
        if pend:
            # every trunk backward since the last call handed its rows over (one in the reference's loop; several when a caller
            # differentiates more than one forward pass per step): ONE exchange of all of them
            ids = pend[0][0].reshape(-1) if len(pend) == 1 else torch.cat([i.reshape(-1) for i, _ in pend])
            rows = pend[0][1] if len(pend) == 1 else torch.cat([r for _, r in pend])
            # the union must cover every id above.  It does when every one of those forwards registered its ids while the exchange
            # was enabled (each union is identical on all ranks, so their merge is too); a forward under no_sync() whose backward is
            # exchanged after all, or a backward without a registered forward, rebuilds it here (all ranks take the same branch: SPMD)
            union = None
            if unions and not cut and all(u is not None for u in unions) and len(unions) >= len(pend):
                for u, ev in unions:
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                union = unions[0][0] if len(unions) == 1 else torch.unique(torch.cat([u for u, _ in unions]))
            block = exchange_rows(ids, rows, self.model.config.vocab_size, self.group, union=union)
        bk.finish(upto=self._word[0])                               # the tail in front of the table + every outstanding collective
        bk.done = 0
        if block is not None:
            union, summed = block
            self.model._w["g_word"].index_add_(0, union, summed)    # unique rows: one add per element, the same bits on every rank

    def __call__(self, *a, **k):
        return self.model(*a, **k)


def _ordered_scatter(rows: torch.Tensor) -> bool:
    """Deterministic mode (model.deterministic / mmbert_set_deterministic) on GPU rows: duplicate ids must not go through
    ``index_add_`` (fp32 atomics in arrival order on HIP) -- the ordered kernel (mmbert_id_runs_sum_rows) takes them."""
    if not rows.is_cuda:
        return False
    from . import ops
    if not ops.deterministic():
        return False
    if rows.dtype not in (torch.bfloat16, torch.float32) or rows.stride(1) != 1:
        raise RuntimeError("deterministic mode: the row gradients must be contiguous bf16 / fp32 rows for the ordered scatter")
    return True


def _scatter_rows(table: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor):
    """table[ids[i]] += rows[i] for ids in (0, V) -- row 0 is the padding row of the lookup (HF:58, no gradient)."""
    ids = ids.reshape(-1)
    if _ordered_scatter(rows):
        from . import ops
        ops.scatter_add_rows_ordered(ids, rows, table, table.shape[0])
        return
    keep = (ids > 0) & (ids < table.shape[0])
    table.index_add_(0, ids.clamp(0, table.shape[0] - 1), rows.to(table.dtype) * keep[:, None].to(table.dtype))


def gather_union(ids: torch.Tensor, vocab: int, group=None, equal_sizes: bool = False) -> torch.Tensor:
    """The sorted union over all ranks of the table rows ``ids`` touches (ids outside (0, vocab) dropped): all-gather + unique, the
    same list on every rank.  ``torch.unique`` reads its output size back to the host, so call this where the host is AHEAD of the
    GPU anyway -- the ids are inputs of the step: DataParallel does it on the input stream at the start of forward, not at the
    end of backward where it would drain the queue."""
    W = dist.get_world_size(group)
    ids = ids.reshape(-1).long().contiguous()
    n = ids.numel()
    if W > 1 and not equal_sizes:
        # all_gather_into_tensor needs the same element count on every rank; a plain DataLoader pads each rank's batch to ITS longest
        # text (pad_sequence) and may end on a short batch.  Agree on the maximum first and pad with id 0 (the padding row: dropped
        # below).  One tiny MAX all-reduce + read-back, at a point where the host is ahead of the GPU anyway (torch.unique reads its
        # size back too); DeviceBatchBuilder batches are equal by construction (equal_sizes=True skips it).
        nmax = torch.tensor([n], dtype=torch.long, device=ids.device)
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=group)
        nmax = int(nmax.item())
        if nmax != n:
            ids = torch.cat((ids, ids.new_zeros(nmax - n)))
            n = nmax
    allids = torch.empty(W * n, dtype=torch.long, device=ids.device)
    dist.all_gather_into_tensor(allids, ids, group=group)
    return torch.unique(allids[(allids > 0) & (allids < vocab)])


def exchange_rows(ids: torch.Tensor, rows: torch.Tensor, vocab: int, group=None, union: Optional[torch.Tensor] = None):
    """Sum over ranks of sparse row gradients: every rank holds n rows ``rows[i]`` for table row ``ids[i]`` (duplicates allowed, ids
    outside (0, vocab) carry no gradient).  The same sorted union of touched rows on every rank (``union``, or gather_union() here)
    -> local rows summed into a [union, H] fp32 block -> ONE all-reduce of the block.  Returns (union ids int64, summed rows fp32):
    what a dense all-reduce of the scattered table would hold in those rows (tests/test_host_cpu.py: equal to it).  With ``union``
    given nothing here reads anything back to the host."""
    ids = ids.reshape(-1).long()
    if union is None:
        union = gather_union(ids, vocab, group)
    block = torch.zeros((union.numel(), rows.shape[1]), device=rows.device, dtype=torch.float32)
    if union.numel():
        if rows.is_cuda and rows.dtype in (torch.bfloat16, torch.float32) and rows.stride(1) == 1:
            from . import ops                                  # one kernel (mmbert_rows_to_block) for the eleven launches of the torch form
            ops.rows_to_block(ids, rows, union, vocab, block)
        else:
            _ordered_scatter(rows)                             # deterministic mode: GPU rows the ordered kernel cannot take raise here
            mine = (ids > 0) & (ids < vocab)
            pos = torch.searchsorted(union, ids.clamp(0, vocab - 1)).clamp(max=union.numel() - 1)
            block.index_add_(0, pos, rows.to(torch.float32) * mine[:, None].to(torch.float32))
        dist.all_reduce(block, op=dist.ReduceOp.SUM, group=group)
    return union, block


def init_from_env(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group when world > 1 -- or, with
    ``force``, also for a single process (``bench.py --force-dp``: the RCCL path exercised on one GPU).  Call it BEFORE the
    first GPU call of the process."""
    import os
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
