"""Flat parameter storage for the MI355X MMBert model.

All parameters live in ONE fp32 buffer (``params``), their gradients in a second (``grads``), and a
bf16 working copy in a third (``half``) at the same element offsets; the GEMM weights additionally
have pre-transposed bf16 copies (``halfT``) so that input gradients are NT GEMMs too.  Consequences:

* AdamW is a single kernel launch over the flat buffers (mmbert_adamw) that also refreshes ``half``
  and zeroes ``grads``;
* the data-parallel gradient exchange is a handful of large RCCL all-reduces on contiguous slices
  of ``grads`` (parallel.py) -- sized for xGMI, not one call per tensor;
* q/k/v weights (and biases) are adjacent, so the packed [3H,H] projection needs no copy.

``nn.Parameter.data`` / ``.grad`` are views into these buffers, so ``state_dict()``,
``load_state_dict()``, ``named_parameters()`` keep the reference's names and shapes.
The layout order follows the order in which backward finishes gradients (heads, layer L-1..0,
embeddings; the tied word embedding last) so buckets can be reduced while backward still runs.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops

ALIGN = 256
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")                 # REF:train.py:78
# parameters the reference never differentiates: optimizer must skip them
FROZEN = ("bert.jointEmbeddings.W_cv.", "bert.jointEmbeddings.W_cs.", "cls.seq_relationship.")


def _round_up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


class FlatParams:
    def __init__(self, model: torch.nn.Module, cfg, device):
        self.device = device
        H, L, I, V = cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.vocab_size
        self.vpad = _round_up(V, 128)
        named = dict(model.named_parameters())               # tied aliases already de-duplicated
        groups: List[List[str]] = []
        used = set()

        def add(*names):
            g = [n for n in names if n in named and n not in used]
            if g:
                groups.append(g)
                used.update(g)

        # ---- layout order = gradient completion order in backward ----
        add("cls.predictions.bias")
        add("cls.predictions.transform.dense.weight")
        add("cls.predictions.transform.dense.bias")
        add("cls.predictions.transform.LayerNorm.weight", "cls.predictions.transform.LayerNorm.bias")
        for n in sorted(named):
            if not n.startswith("bert.") and not n.startswith("cls.predictions."):
                add(n)                                       # fusion head, CPC, align, seq_relationship
        add("bert.pooler.dense.weight")
        add("bert.pooler.dense.bias")
        for i in reversed(range(L)):
            p = f"bert.encoder.layer.{i}."
            add(p + "output.LayerNorm.weight", p + "output.LayerNorm.bias")
            add(p + "output.dense.weight")
            add(p + "output.dense.bias")
            add(p + "intermediate.dense.weight")
            add(p + "intermediate.dense.bias")
            add(p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias")
            add(p + "attention.output.dense.weight")
            add(p + "attention.output.dense.bias")
            add(p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight")
            add(p + "attention.self.query.bias", p + "attention.self.key.bias", p + "attention.self.value.bias")
        for n in sorted(named):
            if n.startswith("bert.jointEmbeddings."):
                add(n)
        add("bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias")
        add("bert.embeddings.token_type_embeddings.weight")
        add("bert.embeddings.position_embeddings.weight")
        add("bert.embeddings.word_embeddings.weight")
        for n in sorted(named):
            add(n)                                           # anything unforeseen
        assert used == set(named)

        self.offset: Dict[str, int] = {}
        self.numel: Dict[str, int] = {}
        flags: List[int] = []
        off = 0
        for g in groups:
            off = _round_up(off, ALIGN)
            start = off
            frozen = any(g[0].startswith(f) for f in FROZEN)
            decay = not any(nd in g[0] for nd in NO_DECAY)
            for n in g:
                assert (not any(nd in n for nd in NO_DECAY)) == decay, "mixed decay flags inside a packed group"
                self.offset[n] = off
                self.numel[n] = named[n].numel()
                off += named[n].numel()
                assert off % 4 == 0 or n == g[-1], n          # packed neighbours must stay 16-byte aligned
            if g[0] == "bert.embeddings.word_embeddings.weight":
                off = start + self.vpad * H                   # zero rows V..Vpad: the MLM decoder's padded N
            if g[0] == "cls.predictions.bias":
                off = start + self.vpad
            end = _round_up(off, ALIGN)
            flags += [2 if frozen else (1 if decay else 0)] * ((end - start) // ALIGN)
            off = end
        self.total = off
        self.params = torch.zeros(self.total, device=device, dtype=torch.float32)
        self.grads = torch.zeros(self.total, device=device, dtype=torch.float32)
        self.half = torch.zeros(self.total, device=device, dtype=torch.bfloat16)
        self.flags = torch.tensor(flags, dtype=torch.uint8, device=device)
        self.named = named
        self.order = [n for g in groups for n in g]
        with torch.no_grad():
            for n, p in named.items():
                o, k = self.offset[n], self.numel[n]
                self.params[o:o + k].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                p.data = self.params[o:o + k].view(p.shape)
                p.grad = self.grads[o:o + k].view(p.shape)
                p._mmb_flat = (self, o)

        # ---- transposed bf16 copies of the GEMM weights: name -> (src_off, rows, cols, dst_off, dst_ld) ----
        self.t_off: Dict[str, Tuple[int, int, int]] = {}
        descs = []
        toff = 0
        tile0 = 0

        def add_t(key, src_off, rows, cols, dst_ld=None):
            nonlocal toff, tile0
            dst_ld = dst_ld or rows
            self.t_off[key] = (toff, cols, dst_ld)
            descs.append((src_off, toff, rows, cols, dst_ld, tile0))
            tile0 += ((rows + 63) // 64) * ((cols + 63) // 64)
            toff += _round_up(cols * dst_ld, 8)

        for i in range(L):
            p = f"bert.encoder.layer.{i}."
            add_t(p + "qkv", self.offset[p + "attention.self.query.weight"], 3 * H, H)
            add_t(p + "o", self.offset[p + "attention.output.dense.weight"], H, H)
            add_t(p + "w1", self.offset[p + "intermediate.dense.weight"], I, H)
            add_t(p + "w2", self.offset[p + "output.dense.weight"], H, I)
        add_t("transform", self.offset["cls.predictions.transform.dense.weight"], H, H)
        add_t("word", self.offset["bert.embeddings.word_embeddings.weight"], V, H, self.vpad)
        self.halfT = torch.zeros(max(toff, 8), device=device, dtype=torch.bfloat16)
        raw = np.zeros((len(descs), 4), dtype=np.int64)
        for j, (so, do, r, c, ld, t0) in enumerate(descs):
            raw[j] = (so, do, np.int64(r) | (np.int64(c) << 32), np.int64(ld) | (np.int64(t0) << 32))
        self._descs = torch.from_numpy(raw).to(device)
        self._ndesc, self._ntiles = len(descs), tile0
        self._version = -1
        self.grads_dirty = False             # set by backward, cleared by the fused zero_grad in AdamW
        # ---- lazy zero (round 4): gradients that ONE weight-gradient launch per backward produces whole (the encoder's dense weights,
        # the tied word-embedding table) need no zero fill between optimizer steps -- the first launch of the next backward overwrites
        # them (accumulate = 0) instead of reading zeros and adding.  What torch does with zero_grad(set_to_none=True): the old gradient
        # is simply dropped.  ``lazy`` = the registered gradient views by id; ``stale`` = the ids whose buffer holds a DROPPED gradient.
        self.lazy: Dict[int, Tuple[torch.Tensor, List[str]]] = {}
        self.stale: set = set()
        self.lazy_detached = False
        self.refresh()

    # ------------------------------------------------------------------------------------------
    def register_lazy(self, gview: torch.Tensor, names: List[str]):
        """``gview``: a contiguous view of ``grads`` covering whole 256-element blocks, written whole by one weight-gradient launch per
        backward; ``names``: the parameters inside it (their .grad is None while the buffer is stale)."""
        o = (gview.data_ptr() - self.grads.data_ptr()) // 4
        if not (gview.is_contiguous() and o % ALIGN == 0 and gview.numel() % ALIGN == 0):
            return                                               # (odd widths: not whole blocks -- stays with the zero fill)
        self.lazy[id(gview)] = (gview, list(names))

    def lazy_block_mask(self) -> torch.Tensor:
        """uint8 per 256-element block: 4 where the optimizer's fused zero_grad must leave the gradient alone."""
        m = torch.zeros(self.total // ALIGN, dtype=torch.uint8)
        for g, _ in self.lazy.values():
            o = (g.data_ptr() - self.grads.data_ptr()) // 4
            m[o // ALIGN:(o + g.numel()) // ALIGN] = 4
        return m

    def take_accumulate(self, gviews) -> bool:
        """The ``accumulate`` flag of a weight-gradient launch that writes ``gviews`` whole: False iff all of them hold a dropped
        gradient (the launch overwrites them); a mixed set is settled first (zero fill of the stale ones)."""
        if not self.stale:
            return True
        ids = [id(g) for g in gviews]
        st = [i in self.stale for i in ids]
        if not any(st):
            return True
        if not all(st):
            for g, s_ in zip(gviews, st):
                if s_:
                    g.zero_()
        self.stale.difference_update(ids)
        return not all(st)

    def settle(self, gviews=None):
        """Zero-fills the gradients that are still stale (all, or those of ``gviews``): before anything but an overwriting launch
        reads or adds to them -- the optimizer itself when no backward wrote them, the embedding scatter when no MLM loss ran."""
        if not self.stale:
            return
        ids = list(self.stale) if gviews is None else [id(g) for g in gviews if id(g) in self.stale]
        for i in ids:
            self.lazy[i][0].zero_()
            self.stale.discard(i)

    def drop_lazy(self):
        """After the optimizer's fused zero_grad: the lazy gradients are dropped, not zeroed."""
        self.stale = set(self.lazy)

    def detach_lazy(self):
        """zero_grad(): .grad of the parameters whose buffer is stale reads None (torch's set_to_none), until the next backward."""
        if self.lazy_detached or not self.stale:
            return
        for i in self.stale:
            for n in self.lazy[i][1]:
                self.named[n].grad = None
        self.lazy_detached = True

    def attach_lazy(self):
        if not self.lazy_detached:
            return
        for _, names in self.lazy.values():
            for n in names:
                p = self.named[n]
                if p.grad is None:
                    o = self.offset[n]
                    p.grad = self.grads[o:o + self.numel[n]].view(p.shape)
        self.lazy_detached = False

    def view32(self, name: str, shape=None) -> torch.Tensor:
        o, k = self.offset[name], self.numel[name]
        t = self.params[o:o + k]
        return t.view(shape) if shape is not None else t.view(self.named[name].shape)

    def span(self, buf: torch.Tensor, first: str, numel: int, shape) -> torch.Tensor:
        o = self.offset[first]
        return buf[o:o + numel].view(shape)

    def tview(self, key: str) -> torch.Tensor:
        o, rows, ld = self.t_off[key]
        return self.halfT[o:o + rows * ld].view(rows, ld)

    def refresh(self):
        """fp32 masters -> bf16 copy and transposed copies (after any out-of-band parameter change)."""
        self.wait_transposes()                # (a side-stream transposed-copy launch may still read the bf16 copy rewritten here)
        ops.cast_f32_bf16(self.params, self.half)
        self.refresh_transposes()
        self._version = self._param_versions()

    def refresh_transposes(self, side: bool = False):
        """``side`` (the optimizer's call, round 6): on the storage's own side stream, behind the work queued on the current stream so far.
        The transposed copies are read by BACKWARD only (the input gradients' W^T operands), the launch is a 0.4-GB copy at HBM rate
        (92 us) and what follows an optimizer step on the current stream is the next step's prologue -- ten launches of 5-30 us that
        leave the chip empty: the two run side by side.  Readers call wait_transposes() (model._MLMHeadFn / _EncoderFn backward; the
        next optimizer step, which rewrites the bf16 copy this launch reads)."""
        if side and ops.SIDE_TRANSPOSES and self.halfT.is_cuda:
            self.wait_transposes()
            s = ops.side_stream("transposes", self.halfT.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ops.transpose_cast(self.half, self.halfT, self._descs, self._ndesc, self._ntiles)
                self.__dict__["_t_event"] = s.record_event()
            return
        self.wait_transposes()
        ops.transpose_cast(self.half, self.halfT, self._descs, self._ndesc, self._ntiles)      # from the bf16 copy: half the bytes, same values

    def wait_transposes(self):
        """The current stream waits for a transposed-copy launch still running on the side stream (no-op otherwise)."""
        ev = self.__dict__.pop("_t_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def mark_synced(self):
        """The bf16 copies were just refreshed by the optimizer kernel itself."""
        self._version = self._param_versions()

    def name_at(self, offset: int) -> str:
        if not hasattr(self, "_by_offset"):
            self._by_offset = {o: n for n, o in self.offset.items()}
        return self._by_offset[offset]

    def owns_any(self) -> bool:
        n = self.order[0]
        return self.named[n].data_ptr() == self.params.data_ptr() + 4 * self.offset[n]

    def _param_versions(self) -> int:
        return sum(p._version for p in self.named.values())

    def maybe_refresh(self):
        # in-place torch ops on a parameter (foreign optimizer, load_state_dict, init) bump its
        # version counter; our own AdamW kernel writes through raw pointers and refreshes itself
        if self._param_versions() != self._version:
            self.refresh()

    def owns(self, model: torch.nn.Module) -> bool:
        """Cheap per-forward check (three sentinel parameters) that the module's parameters still
        alias this storage; re-attaches ``.grad`` views dropped by ``zero_grad(set_to_none=True)``."""
        if not hasattr(self, "_sentinels"):
            lazy_names = {n for _, names in self.lazy.values() for n in names}          # (their .grad is None between zero_grad and backward)
            cand = [n for n in self.order if n not in lazy_names]
            self._sentinels = (cand[0], cand[len(cand) // 2], cand[-1])
        sentinels = self._sentinels
        for n in sentinels:
            p = self.named[n]
            if p.data_ptr() != self.params.data_ptr() + 4 * self.offset[n]:
                return False
        if any(self.named[n].grad is None for n in sentinels):
            for n, p in self.named.items():
                o = self.offset[n]
                if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                    p.grad = self.grads[o:o + self.numel[n]].view(p.shape)
            self.lazy_detached = False
        return True
