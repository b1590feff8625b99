"""Synthetic MMBertDataset batches that reproduce the reference's input contract.

The reference's data path (``MMBertDataset.__getitem__`` -> ``model_utils.collate`` ->
``model_utils.mask_tokens`` -> the packing in ``trainer.train_epoch``, REF:trainer.py:41-64) needs
the CMU pickles and a tokenizer, neither of which exists offline.  This module builds batches with
the same tuple structure, dtypes and quirks directly:

* ``text / twv / tws``: int64 ``[B,T]``, ``[CLS]=101, tokens, [SEP]=102, PAD=0...``; the three are
  the same sentence with independently re-drawn MLM masking (REF:trainer.py:45-47).
* ``visual`` / ``speech``: float64 ``[B,P,D]`` features whose trailing rows are exactly 0 (padding).
* text mask float64 with 0 at PAD (REF:model_utils.py:118-120); pair masks ``feature != 0`` with the
  features' shape (float64 for visual, int64 for speech, REF:model_utils.py:124-133); the
  text-with-pair masks are all ones (the ``==`` bug, REF:model_utils.py:128,136).
* labels: -100 where not selected, 80 % of the selected inputs -> 103 (REF:model_utils.py:27-32); [CLS]/[SEP] are never
  selected, [PAD] positions are (the reference's PAD exclusion is a discarded ``masked_fill``, see ``_mask_tokens``);
  pair-position labels are a copy of the text labels when P == T (REF:trainer.py:50,53) and -100
  otherwise (the reference cannot express P != T through ``collate``).

Determinism: numpy PCG64 only, so a (seed, shape) pair means the same batch on every box.
"""
from __future__ import annotations

import numpy as np
import torch

PAD, CLS, SEP, MASK = 0, 101, 102, 103
MODALITY_DIMS = {"mosi": (47, 74), "mosei": (35, 74), "ur_funny": (371, 81)}    # REF:config.py:13-17


def _mask_tokens(rng, ids, p_mlm, pad_selectable=True):
    """REF:model_utils.py:6-39 on numpy arrays.  Never selected: [CLS] and [SEP] -- what transformers-2.8
    ``BertTokenizer.get_special_tokens_mask(already_has_special_tokens=True)`` flags (the pin of REF:requirement.txt).
    [PAD] positions ARE selectable in the reference: its PAD branch (:24-26) is a non-in-place ``masked_fill`` whose
    result is dropped, so 15 % of the padding gets label 0 and 80 % of those become [MASK].  ``pad_selectable=False``
    is the opt-in deviation (PAD never selected: what the branch was meant to do)."""
    special = (ids == CLS) | (ids == SEP)
    if not pad_selectable:
        special |= ids == PAD
    select = (rng.random(ids.shape) < p_mlm) & ~special
    replace = (rng.random(ids.shape) < 0.8) & select
    labels = np.where(select, ids, -100)
    out = np.where(replace, MASK, ids)
    return out.astype(np.int64), labels.astype(np.int64)


def synthetic_batch(batch: int, text_len: int, visual_len: int, speech_len: int, *, dataset="mosei",
                    vocab=30522, seed=1, mlm_probability=0.15, full_length=False, pad_selectable=True):
    """Returns the six keyword arguments of ``MMBertForPretraining.forward`` as CPU tensors:
    ``dict(input_ids=..., token_type_ids=..., attention_mask=..., masked_labels=..., ap_label=...,
    sentiment=...)``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vd, sd = MODALITY_DIMS[dataset]
    B, T = batch, text_len
    lo = min(1000, vocab - 1)
    text = np.zeros((B, T), np.int64)
    for b in range(B):
        n = T - 2 if full_length else int(rng.integers(max(1, T // 2), T - 1))
        text[b, 0] = CLS
        text[b, 1:1 + n] = rng.integers(lo, vocab, n)
        text[b, 1 + n] = SEP

    def feats(P, D):
        x = rng.standard_normal((B, P, D))
        x[x == 0.0] = 1e-3
        for b in range(B):
            n = P if full_length else int(rng.integers(max(1, P // 2), P + 1))
            x[b, n:] = 0.0
        return x
    visual, speech = feats(visual_len, vd), feats(speech_len, sd)

    t_in, t_lab = _mask_tokens(rng, text, mlm_probability, pad_selectable)
    v_in, v_lab = _mask_tokens(rng, text, mlm_probability, pad_selectable)
    s_in, s_lab = _mask_tokens(rng, text, mlm_probability, pad_selectable)

    def pair_labels(lab, P):
        if P == T:
            return np.concatenate((lab, lab), axis=-1)                    # REF:trainer.py:50,53
        return np.concatenate((lab, np.full((B, P), -100, np.int64)), axis=-1)

    text_mask = (text != PAD).astype(np.float64)
    tt = torch.from_numpy
    input_ids = (tt(t_in), tt(visual), tt(speech), tt(v_in), tt(s_in))
    token_type_ids = (
        torch.zeros(B, T, dtype=torch.int64),
        tt(np.concatenate((np.zeros((B, T)), np.ones((B, visual_len))), axis=1)),
        tt(np.concatenate((np.zeros((B, T)), np.ones((B, speech_len))), axis=1)),
    )
    attention_mask = (
        tt(text_mask),
        (torch.ones(B, T, dtype=torch.float64), tt((visual != 0).astype(np.float64))),
        (torch.ones(B, T, dtype=torch.int64), tt((speech != 0).astype(np.int64))),
    )
    masked_labels = (tt(t_lab), tt(pair_labels(v_lab, visual_len)), tt(pair_labels(s_lab, speech_len)))
    ap_label = (tt(rng.integers(0, 2, B).astype(np.int64)), tt(rng.integers(0, 2, B).astype(np.int64)))
    sentiment = tt(rng.uniform(-3, 3, B).astype(np.float32))
    return dict(input_ids=input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask,
                masked_labels=masked_labels, ap_label=ap_label, sentiment=sentiment)


def batch_to(batch: dict, device):
    """Moves every tensor of a (nested-tuple) batch to ``device`` keeping dtypes, like the
    ``.to(DEVICE)`` calls in REF:trainer.py:49-64."""
    def mv(x):
        if isinstance(x, tuple):
            return tuple(mv(y) for y in x)
        return x.to(device)
    return {k: mv(v) for k, v in batch.items()}


def to_fused(batch: dict) -> dict:
    """The keyword arguments of ``MMBertForPretraining.forward_fused`` (text | visual | speech in ONE sequence; a declared
    extension, see its docstring) from a three-pass batch: the text pass's ids / mask / labels, both feature blocks and
    their masks, -100 labels on the pair positions."""
    text_ids, visual, speech, _twv, _tws = batch["input_ids"]
    am_t, am_v, am_s = batch["attention_mask"]
    lab_t = batch["masked_labels"][0]
    B = text_ids.shape[0]
    pad = torch.full((B, visual.shape[1] + speech.shape[1]), -100, dtype=lab_t.dtype, device=lab_t.device)
    return dict(input_ids=(text_ids, visual, speech), token_type_ids=batch["token_type_ids"][0],
                attention_mask=(am_t, am_v[1], am_s[1]), masked_labels=torch.cat((lab_t, pad), dim=1),
                ap_label=batch["ap_label"], sentiment=batch["sentiment"])
