"""Counterpart of the reference's ``MMBertDataset`` (REF:MMBertDataset.py:12-202) and a device-resident batch builder
(the input pipeline on the device).

``MMBertDataset`` mirrors the reference class item by item -- same constructor, same 16-tuple from ``__getitem__``, same
pair-selection rule driven by Python's ``random`` in the same call order (so a seeded ``random`` reproduces the
reference's pairing), same quirks:

* the alignment label is **1 for the matching pair** and 0 for a random other sample (REF:MMBertDataset.py:143-155; the
  docstring there says the opposite), and the last item is always paired with itself (:137-141);
* token types are float64 (``np.zeros`` / ``np.ones``, :160-166), the text label is always 0 (:176);
* ``sentiment_selection`` returns ``None`` for (dataset, task, mode) combinations it does not know (:63-99).

``DeviceBatchBuilder`` keeps every item in HBM (ids int64, features converted to fp32 ONCE -- the reference ships float64
features to the device on every step, REF:trainer.py:49-64) and builds a batch with index gathers on the device: the
same tuple ``model_utils.collate`` returns (REF:model_utils.py:51-143, dtypes and mask quirks included) for the same
``random`` state, ready for ``trainer.pack_step_inputs``.  Only the pair draws (2 x B host random numbers) stay on the host.
"""
from __future__ import annotations

import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .data import MODALITY_DIMS

emotions = ["sentiment", "happy", "sad", "anger", "surprise", "disgust", "fear"]      # REF:MMBertDataset.py:10


class MMBertDataset(Dataset):
    """REF:MMBertDataset.py:12-202.  ``features``: list of ``((input_ids, visual, speech, input_mask), label, segment, words)``
    (REF:train.py:191-195)."""

    def __init__(self, tokenizer, features, dataset, task, num_labels, device="cpu"):
        self.tokenizer = tokenizer
        self.items = features
        self.total_item = self.count()
        self.dataset = dataset
        self.task = task
        self.num_labels = num_labels
        self.device = torch.device(device)                     # the reference's module-level ``cudas`` (:8)
        if dataset in MODALITY_DIMS:
            self.VISUALDIM, self.SPEECHDIM = MODALITY_DIMS[dataset]       # :52-60

    def sentiment_selection(self, sentiment, mode):
        """REF:MMBertDataset.py:62-99."""
        if self.dataset == "mosei":
            if self.task == "sentiment":
                if mode == "2":
                    return torch.tensor([1]) if sentiment[0] >= 0 else torch.tensor([0])
                if mode == "7":
                    return torch.tensor(sentiment[0])
                if mode == "1":
                    return torch.tensor(sentiment[0]) / 3
            else:
                if mode == "2":
                    return torch.tensor([1]) if torch.tensor(sentiment[emotions.index(self.task)]) != 0 else torch.tensor([0])
                if mode == "6":
                    return torch.argmax(torch.tensor(sentiment[1:]))
        elif self.dataset == "mosi":
            if mode == "2":
                return torch.tensor([1]) if sentiment >= 0 else torch.tensor([0])
            if mode == "7":
                return torch.tensor(sentiment)
            if mode == "1":
                return torch.tensor(sentiment) / 3
        elif self.dataset == "ur_funny":
            if mode == "2":
                return torch.tensor([1]) if sentiment == 1 else torch.tensor([0])
        return None

    def draw_pair(self, i):
        """The pair-selection rule of ``create_concat_joint_sentence`` (REF:MMBertDataset.py:137-155) alone: returns
        (second index, label); consumes ``random`` exactly as the reference does."""
        n = len(self.items)
        if i == n - 1:
            return i, 1
        if random.uniform(0, 1) > 0.5:
            return i, 1
        second = random.choice(range(n))
        while second == i:
            second = random.choice(range(n))
        return second, 0

    def create_concat_joint_sentence(self, i, mode):
        """REF:MMBertDataset.py:102-167."""
        pair_index = {"visual": 1, "speech": 2}.get(mode, -1)
        assert pair_index != -1
        sentiment = self.sentiment_selection(self.items[i][1][0], str(self.num_labels))
        second, label = self.draw_pair(i)
        text_sentence = list(self.items[i][0][0])
        pair_sentence = self.items[second][0][pair_index]
        tti = torch.cat((torch.tensor(np.zeros(len(text_sentence)), device=self.device),
                         torch.tensor(np.ones(len(pair_sentence)), device=self.device)))
        return text_sentence, pair_sentence, torch.tensor(label, dtype=torch.int64, device=self.device), tti, sentiment

    def create_text_sentence(self, i):
        """REF:MMBertDataset.py:169-181."""
        first = self.items[i][0][0]
        sentiment = self.sentiment_selection(self.items[i][1][0], str(self.num_labels))
        return (torch.tensor(first), torch.tensor(0, dtype=torch.int64, device=self.device),
                torch.tensor(np.zeros(len(first)), device=self.device), sentiment, self.items[i][-2], self.items[i][-1])

    def count(self):
        return len(self.items)

    def __len__(self):
        return self.total_item

    def __getitem__(self, i):
        """REF:MMBertDataset.py:194-202: text item, then the visual draw, then the speech draw."""
        t_sent, t_label, t_tti, t_sentiment, segment, raw = self.create_text_sentence(i)
        t2, v_sent, v_label, v_tti, v_sentiment = self.create_concat_joint_sentence(i, "visual")
        t3, s_sent, s_label, s_tti, s_sentiment = self.create_concat_joint_sentence(i, "speech")
        return (t_sent, t_label, t_tti, t_sentiment, t2, v_sent, v_label, v_tti, v_sentiment,
                t3, s_sent, s_label, s_tti, s_sentiment, segment, raw)


class DeviceBatchBuilder:
    """All items of an ``MMBertDataset`` resident on ``device``; ``batch(indices)`` returns what
    ``collate([dataset[i] for i in indices])`` returns -- same tuple, dtypes, mask quirks and, from the same ``random``
    state, the same pairs -- built by gathers on the device.  Deliberate difference: the feature tensors are fp32 (converted
    once at construction) instead of float64; the model casts them to fp32 anyway (REF:MMBertEmbedding.py:62,64 ``.float()``).
    Requires items of one common length (the reference pads every item to ``max_seq_length``, REF:train.py:101-133)."""

    def __init__(self, dataset: MMBertDataset, device):
        self.ds = dataset
        self.device = torch.device(device)
        items = dataset.items
        L = len(items[0][0][0])
        for it in items:
            assert len(it[0][0]) == len(it[0][1]) == len(it[0][2]) == L, "items must share one padded length"
        self.L = L
        self.ids = torch.tensor(np.asarray([it[0][0] for it in items]), dtype=torch.int64, device=self.device)
        self.visual = torch.tensor(np.stack([np.asarray(it[0][1]) for it in items]), dtype=torch.float32, device=self.device)
        self.speech = torch.tensor(np.stack([np.asarray(it[0][2]) for it in items]), dtype=torch.float32, device=self.device)
        sent = [dataset.sentiment_selection(it[1][0], str(dataset.num_labels)) for it in items]
        if any(s is None for s in sent):
            raise ValueError("sentiment_selection has no rule for this (dataset, task, num_labels)")
        # collate() turns every sentiment into one python number per item: float for floating tensors, int for integer ones
        self.sent_is_int = not torch.as_tensor(sent[0]).is_floating_point()
        flat = [float(torch.as_tensor(s).reshape(-1)[0]) for s in sent]
        self.sent = torch.tensor(flat, dtype=torch.int64 if self.sent_is_int else torch.float32, device=self.device)
        self.segments = [it[-2] for it in items]
        self.raw = [it[-1] for it in items]
        # constant pieces of every batch
        self._tti_text = torch.zeros(L, dtype=torch.int64, device=self.device)
        self._tti_pair = torch.cat((torch.zeros(L, dtype=torch.float64), torch.ones(L, dtype=torch.float64))).to(self.device)

    def draw(self, indices):
        """Host side of a batch: the reference's pair draws in ITS order (per item: visual, then speech)."""
        v_idx, v_lab, s_idx, s_lab = [], [], [], []
        for i in indices:
            j, l = self.ds.draw_pair(int(i)); v_idx.append(j); v_lab.append(l)
            j, l = self.ds.draw_pair(int(i)); s_idx.append(j); s_lab.append(l)
        return v_idx, v_lab, s_idx, s_lab

    def batch(self, indices):
        dev = self.device
        v_idx, v_lab, s_idx, s_lab = self.draw(indices)
        B = len(v_idx)
        idx = torch.as_tensor([int(i) for i in indices], dtype=torch.int64, device=dev)
        text = self.ids.index_select(0, idx)
        vis = self.visual.index_select(0, torch.as_tensor(v_idx, dtype=torch.int64, device=dev))
        sp = self.speech.index_select(0, torch.as_tensor(s_idx, dtype=torch.int64, device=dev))
        sent = self.sent.index_select(0, idx)
        text_mask = (text != 0).to(torch.float64)                              # REF:model_utils.py:118-120
        vis_mask = (vis != 0).to(torch.float64)                                # :124-125
        sp_mask = (sp != 0).to(torch.int64)                                    # :132-133
        ones_f = torch.ones((B, self.L), dtype=torch.float64, device=dev)      # the `==` quirk: never zeroed (:128,136)
        ones_i = torch.ones((B, self.L), dtype=torch.int64, device=dev)
        lab0 = torch.zeros(B, dtype=torch.int64, device=dev)
        text_batch = (text, lab0, self._tti_text.expand(B, -1).contiguous(), text_mask, sent)
        visual_batch = (text.clone(), vis, torch.as_tensor(v_lab, dtype=torch.int64, device=dev),
                        self._tti_pair.expand(B, -1).contiguous(), vis_mask, sent.clone())
        speech_batch = (text.clone(), sp, torch.as_tensor(s_lab, dtype=torch.int64, device=dev),
                        self._tti_pair.expand(B, -1).contiguous(), sp_mask, sent.clone())
        segs = [self.segments[int(i)] for i in indices]
        raws = [self.raw[int(i)] for i in indices]
        return text_batch, visual_batch, speech_batch, (ones_f, ones_i), segs, raws

    def epoch(self, args, *, batch_size=None, shuffle=True, generator=None, mask_generator=None, rank=0, world=1):
        """One pass over the data as ready keyword-argument dicts for the model (``trainer.train_epoch(batches=...)``): a random
        permutation (the reference's ``RandomSampler``, REF:trainer.py:28; last batch kept short like its ``DataLoader``), batches
        built on the device and packed with the trainer's MLM masking.
        Data parallel (``world`` > 1): every rank must draw the SAME permutation, so a ``generator`` seeded
        identically on all ranks is required when shuffling; the permutation is padded by wrapping around to a multiple of
        ``world`` (torch's ``DistributedSampler`` rule) and sharded ``perm[rank::world]`` -- every rank then yields the same
        number of batches of the same sizes, which the gradient all-reduce of ``parallel.DataParallel`` needs (a rank with one
        batch more would wait in a collective nobody else joins)."""
        from .trainer import pack_step_inputs
        n = len(self.ds)
        bs = int(batch_size or args.train_batch_size)
        if world > 1 and shuffle and generator is None:
            raise ValueError("DeviceBatchBuilder.epoch: with world > 1 pass a torch.Generator seeded identically on every rank "
                             "(each process has its own global RNG: the shards would overlap)")
        order = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
        if world > 1:
            per_rank = -(-n // world)
            order = (order * (1 + (per_rank * world - 1) // max(n, 1)))[:per_rank * world]
            order = order[rank::world]
        for k in range(0, len(order), bs):
            yield pack_step_inputs(self.batch(order[k:k + bs]), args, self.device, mask_generator)
