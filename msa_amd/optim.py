"""AdamW + linear warm-up schedule with the reference's construction API (REF:train.py:76-97),
executed as ONE HIP kernel over the model's flat fp32 buffers (mmbert_adamw): update, weight decay,
bf16 working-copy refresh and zero_grad in a single pass over 28 B/parameter.

``mode="hf"`` (default) is transformers-2.8 ``optimization.AdamW`` -- the class the reference
imports (eps 1e-6, bias correction, decay applied after the update); ``mode="torch"`` is
``torch.optim.AdamW``, the optimizer the golden fixture G8 was generated with.
"""
from __future__ import annotations

from typing import Iterable, List, Union

import torch

from . import ops
from .flat import FROZEN


class AdamW:
    def __init__(self, params: Union[Iterable[torch.nn.Parameter], List[dict]], lr=1e-3, betas=(0.9, 0.999), eps=1e-6,
                 weight_decay=0.0, correct_bias=True, mode="hf"):
        params = list(params)
        if params and isinstance(params[0], dict):
            self.param_groups = [dict(g, params=list(g["params"])) for g in params]
        else:
            self.param_groups = [dict(params=params, weight_decay=weight_decay)]
        for g in self.param_groups:
            g.setdefault("weight_decay", weight_decay)
            g.setdefault("lr", lr)
            g["initial_lr"] = g["lr"]
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is not used by the reference")
        self.betas, self.eps, self.mode = betas, eps, {"hf": 0, "torch": 1}[mode]
        self.grad_scale = 1.0               # set to 1/world_size by parallel.DataParallel
        self._lazy_zero = True
        self._flat = None
        self._steps = 0

    # The dense weights' gradients (3/4 of the parameters) are not zero-filled between steps: the next backward's weight-gradient
    # launches overwrite them (flat.FlatParams lazy zero; torch's zero_grad(set_to_none=True) semantics).  False: fill with zeros.
    # A plain switch at ANY time (as an attribute it was baked into the kernel's flag array at the first step, and turning it
    # off later left the fused zero_grad skipping blocks whose gradients were then accumulated into for ever): the setter rebuilds the
    # flag array, and only blocks this optimizer owns ever carry the "leave it to the next backward" bit.
    @property
    def lazy_zero(self) -> bool:
        return self._lazy_zero

    @lazy_zero.setter
    def lazy_zero(self, on: bool):
        on = bool(on)
        if on != self._lazy_zero:
            self._lazy_zero = on
            if self._flat is not None:
                self._build_flags()

    def _build_flags(self):
        flags = self._base_flags.clone()
        if self._lazy_zero:
            lazy = self._flat.lazy_block_mask().to(flags.dtype)
            lazy[flags == 2] = 0                 # frozen / not owned by this optimizer: never stepped, so never dropped either
            flags |= lazy                        # + 4: the fused zero_grad leaves these blocks to the next backward's overwriting launches
        self._flags = flags.to(self._flat.device)

    # the scheduler scales every group's lr by the same factor; the kernel takes one lr
    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    def _bind(self):
        flat = None
        for g in self.param_groups:
            for p in g["params"]:
                ref = getattr(p, "_mmb_flat", None)
                if ref is None:
                    raise RuntimeError("AdamW: parameter is not backed by msa_amd flat storage -- run one forward "
                                       "pass (or model._ensure_ready) on the GPU before the first optimizer step")
                if flat is None:
                    flat = ref[0]
                elif ref[0] is not flat:
                    raise RuntimeError("AdamW: parameters belong to different flat storages")
        if not flat.owns_any():
            raise RuntimeError("AdamW: the model was re-materialised (e.g. moved) after this storage was created")
        wds = sorted({float(g["weight_decay"]) for g in self.param_groups if g["weight_decay"] > 0})
        if len(wds) > 1:
            raise NotImplementedError("one non-zero weight_decay value is supported (the reference uses 0.01 / 0.0)")
        self.wd = wds[0] if wds else 0.0
        flags = flat.flags.clone().cpu()
        flags[flags != 2] = 3                # 3 = not owned by this optimizer -> treated as frozen below
        for g in self.param_groups:
            f = 1 if g["weight_decay"] > 0 else 0
            for p in g["params"]:
                _, off = p._mmb_flat
                name = flat.name_at(off)
                if any(name.startswith(fr) for fr in FROZEN):
                    continue                 # grad is always None in the reference -> optimizer skips it
                b0, b1 = off // 256, (off + p.numel() + 255) // 256
                if ((flags[b0:b1] != 3) & (flags[b0:b1] != f)).any():
                    raise NotImplementedError(f"{name}: packed neighbours must share one weight-decay setting")
                flags[b0:b1] = f
        flags[flags == 3] = 2
        self._base_flags = flags
        self._m = torch.zeros_like(flat.params)
        self._v = torch.zeros_like(flat.params)
        self._flat = flat
        self._build_flags()

    def step(self):
        if self._flat is None:
            self._bind()
        flat = self._flat
        self._steps += 1
        flat.wait_transposes()               # (a step without a backward in between: the last step's transposed copies still read the bf16 copy)
        flat.settle()                        # a lazy gradient no backward has written since the last step counts as zero
        flat.attach_lazy()
        ops.adamw(flat.params, flat.grads, self._m, self._v, flat.half, self._flags, lr=self.lr, beta1=self.betas[0],
                  beta2=self.betas[1], eps=self.eps, wd=self.wd, step=self._steps, gscale=self.grad_scale, mode=self.mode,
                  zero_grad=True)
        flat.refresh_transposes(side=True)
        flat.mark_synced()
        flat.grads_dirty = False
        if self.lazy_zero:
            flat.drop_lazy()

    def zero_grad(self, set_to_none: bool = False):
        flat = self._flat
        if flat is None:
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        p.grad.zero_()
            return
        if flat.grads_dirty:                 # step() already zeroed the buffer in the same kernel
            flat.grads.zero_()
            flat.grads_dirty = False
            flat.stale.clear()
            flat.attach_lazy()
        else:
            flat.detach_lazy()               # the dropped gradients read None (torch: set_to_none) until the next backward writes them

    def state_dict(self):
        return dict(steps=self._steps, m=None if self._flat is None else self._m, v=None if self._flat is None else self._v,
                    lrs=[g["lr"] for g in self.param_groups])

    def load_state_dict(self, sd):
        if self._flat is None:
            self._bind()
        self._steps = sd["steps"]
        if sd["m"] is not None:
            self._m.copy_(sd["m"])
            self._v.copy_(sd["v"])
        for g, lr in zip(self.param_groups, sd["lrs"]):
            g["lr"] = lr


class LinearWarmupSchedule:
    """``transformers.get_linear_schedule_with_warmup`` (LambdaLR semantics: lr = initial_lr * lambda(step),
    lambda(0) applied at construction).  The reference calls it with warmup == total (REF:train.py:93-97)."""

    def __init__(self, optimizer: AdamW, num_warmup_steps, num_training_steps):
        self.opt, self.warmup, self.total = optimizer, num_warmup_steps, num_training_steps
        self.last_step = 0
        self._apply()

    def _lambda(self, step):
        if step < self.warmup:
            return float(step) / float(max(1, self.warmup))
        return max(0.0, float(self.total - step) / float(max(1, self.total - self.warmup)))

    def _apply(self):
        lam = self._lambda(self.last_step)
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * lam

    def step(self):
        self.last_step += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps):
    return LinearWarmupSchedule(optimizer, num_warmup_steps, num_training_steps)
