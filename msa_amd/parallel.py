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
    torch.distributed: works on CPU tensors with gloo (tests) and on GPU tensors with RCCL."""

    def __init__(self, flat_grads: torch.Tensor, boundaries: Sequence[int], group=None, bucket_mb: float = 32.0,
                 skip: Sequence[Tuple[int, int]] = ()):
        """``skip``: ascending, disjoint element ranges [a, b) that are never exchanged (parameters the reference never
        differentiates: their gradient slots hold zeros on every rank) -- a bucket is cut around them."""
        assert boundaries[0] == 0 and boundaries[-1] == flat_grads.numel() and list(boundaries) == sorted(boundaries)
        self.flat, self.bounds, self.group = flat_grads, list(boundaries), group
        self.min_elems = int(bucket_mb * (1 << 20) / flat_grads.element_size())
        self.skip = [(int(a), int(b)) for a, b in skip if b > a]
        assert all(self.skip[i][1] <= self.skip[i + 1][0] for i in range(len(self.skip) - 1))
        self.handles: List = []
        self.done = 0            # elements already handed to the collective
        self.next_slice = 0
        self.enabled = True
        self.calls = 0           # number of collectives issued since reset (observable in tests)
        self.elems = 0           # elements exchanged since reset

    def reset(self):
        self.handles, self.done, self.next_slice, self.calls, self.elems = [], 0, 0, 0, 0

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

    def _issue(self, end: int):
        if end > self.done and self.enabled:
            for lo, hi in self._segments(self.done, end):
                self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.calls += 1
                self.elems += hi - lo
        self.done = end

    def ready(self, k: int):
        assert k == self.next_slice, "slices must become ready in layout order"
        self.next_slice = k + 1
        end = self.bounds[k + 1]
        if end - self.done >= self.min_elems:
            self._issue(end)

    def finish(self):
        """Reduce whatever is left and wait for every outstanding collective."""
        self._issue(self.bounds[-1])
        for h in self.handles:
            h.wait()
        self.calls_per_step = self.calls
        self.reset()


class DataParallel:
    """Wraps an ``MMBertForPretraining``: broadcast of the initial weights, bucketed gradient
    all-reduce overlapped with backward, gradient averaging folded into the optimizer."""

    def __init__(self, model, optimizer=None, group=None, bucket_mb: float = 32.0, force_dynamic_queue: bool = False):
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
        self.bucketer = GradBucketer(flat.grads, bounds, group, bucket_mb, skip=skip)
        self._L = L
        model.grad_hook = self._on_layer_done
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        self.optimizer = optimizer

    def _on_layer_done(self, i: int):
        self.bucketer.ready(self._L - 1 - i)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient-accumulation micro-step: keep local sums, exchange nothing."""
        self.bucketer.enabled = False
        try:
            yield
        finally:
            self.bucketer.enabled = True
            self.bucketer.reset()

    def finish_backward(self):
        """Call after ``loss.backward()`` and before ``optimizer.step()``."""
        self.bucketer.finish()

    def __call__(self, *a, **k):
        return self.model(*a, **k)


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
