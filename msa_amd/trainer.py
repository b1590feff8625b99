"""Counterpart of the reference's training loop (REF:trainer.py:13-101) and of the input producers it
calls (REF:model_utils.py:6-143): same packing, same stepping rule, same return value.

Differences, all deliberate and documented in DESIGN.md:
* tensors are built on / moved to the GPU once per step and loss bookkeeping stays on the device
  (the reference calls ``.item()`` >= 3x per step, REF:trainer.py:85-93: a device sync each);
* ``quirk_step=True`` keeps ``(step+1) & gradient_accumulation_step == 0`` (REF:trainer.py:96 -- at
  gas=1 the optimizer steps every SECOND micro-batch); ``False`` gives the intended modulo;
* with a ``parallel.DataParallel`` wrapper the gradient exchange happens only on stepping micro-batches.
"""
from __future__ import annotations

import types
from typing import Optional

import torch
from torch.nn.utils.rnn import pad_sequence

PAD, CLS, SEP, MASK = 0, 101, 102, 103
_mask_calls = 0          # fallback call counter (only when the CUDA generator's offset cannot be read)


def _device_mask_seed(device) -> int:
    """The per-call seed of the device-side masking kernel, taken from torch's default CUDA generator of ``device``: its seed and
    its philox OFFSET, which this call advances by 4 (what one ``torch.bernoulli`` launch would consume).  So ``torch.manual_seed(s)``
    (seed := s, offset := 0) reproduces the masks from that point on, ``get_rng_state`` / ``set_rng_state`` capture them, and other
    CUDA draws in between shift them -- like a torch draw.  The data-parallel rank is folded in: ranks seeded alike still draw
    different selection patterns.  NOT the bits torch.bernoulli would produce (counter RNG of csrc/common.h; selection / replacement
    probabilities are quantised to 1/65536)."""
    global _mask_calls
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    try:
        off = gen.get_offset()
        gen.set_offset(off + 4)
    except (AttributeError, RuntimeError):                    # (older torch: no offset accessors -> process-wide call counter)
        _mask_calls += 1
        off = 4 * _mask_calls
    rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    return (gen.initial_seed() * 1000003 + (off // 4) * 8191 + rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def mask_tokens(inputs: torch.Tensor, args, generator: Optional[torch.Generator] = None, special_ids=(CLS, SEP), mask_id=MASK):
    """REF:model_utils.py:6-39 on the tensor's own device: Bernoulli(mlm_probability) selection with
    special tokens excluded (:17-23), labels -100 elsewhere (:28), 80 % of the selected positions
    -> [MASK] (:30-32, in place like the reference); the 10 % random-token branch is commented out
    in the reference (:34-37) and therefore absent.
    ``special_ids``: [CLS] and [SEP] -- what ``get_special_tokens_mask(already_has_special_tokens=True)`` of the pinned
    transformers 2.8 flags.  [PAD] is NOT excluded: the reference's PAD branch (:24-26) calls the non-in-place
    ``masked_fill`` and drops the result, so padding positions are selected at the same 15 % (label 0, 80 % -> [MASK]);
    pass ``special_ids=(PAD, CLS, SEP)`` for the intended rule (a deviation from the reference).  With ``generator=None``
    on CPU the draws are the reference's own (same two ``torch.bernoulli`` calls in the same order on the global RNG:
    pinned by tests/golden/mask_tokens.npz)."""
    if inputs.is_cuda and generator is None and inputs.dtype == torch.int64 and len(special_ids) <= 3:
        # on the GPU: one launch of the library's counter-RNG masking kernel (mmbert_mlm_mask) instead of ~10 element-wise
        # torch kernels; the per-call seed is drawn from torch's default CUDA generator (seed + offset, see _device_mask_seed), so
        # torch.manual_seed() / the CUDA RNG state govern it; pass a ``generator`` for torch's own bernoulli draws
        from . import ops
        x = inputs if inputs.is_contiguous() else inputs.contiguous()
        labels = ops.mlm_mask(x, float(args.mlm_probability), _device_mask_seed(inputs.device),
                              special_ids=tuple(special_ids), mask_id=mask_id)
        if x is not inputs:
            inputs.copy_(x)
        return inputs, labels
    labels = inputs.clone()
    prob = torch.full(labels.shape, float(args.mlm_probability), device=inputs.device)
    special = torch.zeros_like(inputs, dtype=torch.bool)
    for s in special_ids:
        special |= inputs == s
    prob.masked_fill_(special, 0.0)
    masked = torch.bernoulli(prob, generator=generator).bool()
    labels[~masked] = -100
    replaced = torch.bernoulli(torch.full(labels.shape, 0.8, device=inputs.device), generator=generator).bool() & masked
    inputs[replaced] = mask_id
    return inputs, labels


def collate(examples):
    """REF:model_utils.py:51-143 output contract: (text_batch, visual_batch, speech_batch,
    attention_batch, segments, rawData) with the reference's dtypes and quirks -- text mask 0 at
    PAD (:118-120); pair masks ``feature != 0`` (float64 visual :124-125, int64 speech :132-133);
    text-with-pair masks stay all ones (``==`` instead of ``=``, :128,136)."""
    te, tl, tti, ts, twv, ve, vl, vti, vs, tws, se, sl, sti, ss, seg, raw = zip(*examples)
    for a, b, c, d, e in zip(te, ve, se, twv, tws):
        assert len(a) == len(b) == len(c) == len(d) == len(e)                # :92
    sent_dtype = torch.long if torch.as_tensor(ts[0]).dtype == torch.int64 else torch.float
    mk_sent = lambda x: torch.tensor([float(v) if sent_dtype == torch.float else int(v) for v in x], dtype=sent_dtype)
    text = pad_sequence([torch.as_tensor(t) for t in te], batch_first=True, padding_value=0)
    text_mask = torch.ones(text.shape, dtype=torch.float64)
    text_mask[text == 0] = 0
    vis = torch.as_tensor(_stack(ve))
    vis_mask = torch.ones(vis.shape, dtype=torch.float64)
    vis_mask[vis == 0] = 0
    sp = torch.as_tensor(_stack(se))
    sp_mask = torch.ones(sp.shape, dtype=torch.int64)
    sp_mask[sp == 0] = 0
    twv_t, tws_t = torch.as_tensor(_stack(twv)), torch.as_tensor(_stack(tws))
    twv_mask = torch.ones(twv_t.shape, dtype=torch.float64)                 # quirk: never zeroed
    tws_mask = torch.ones(tws_t.shape, dtype=torch.int64)
    lab = lambda x: torch.tensor([int(v) for v in x], dtype=torch.int64)
    text_batch = (text, lab(tl), pad_sequence(list(tti), batch_first=True, padding_value=0).long(), text_mask, mk_sent(ts))
    visual_batch = (twv_t, vis, lab(vl), pad_sequence(list(vti), batch_first=True, padding_value=0), vis_mask, mk_sent(vs))
    speech_batch = (tws_t, sp, lab(sl), pad_sequence(list(sti), batch_first=True, padding_value=0), sp_mask, mk_sent(ss))
    return text_batch, visual_batch, speech_batch, (twv_mask, tws_mask), list(seg), list(raw)


def _stack(seq):
    import numpy as np
    return np.stack([np.asarray(x) for x in seq])


def pack_step_inputs(batch, args, device, generator=None):
    """REF:trainer.py:42-64: MLM masking (re-drawn independently for text / twv / tws), label
    duplication for the pair positions, tuple packing -- returns the model's keyword arguments."""
    text_batch, visual_batch, speech_batch, attention_batch = batch[:4]
    dev = torch.device(device)
    t0, v0, s0 = text_batch[0].to(dev), visual_batch[0].to(dev), speech_batch[0].to(dev)
    if args.mlm:
        text_inputs, text_labels = mask_tokens(t0.clone(), args, generator)
        twv_ids, v_labels = mask_tokens(v0.clone(), args, generator)
        tws_ids, s_labels = mask_tokens(s0.clone(), args, generator)
    else:
        text_inputs, text_labels, twv_ids, v_labels, tws_ids, s_labels = t0, t0, v0, v0, s0, s0
    v_labels = torch.cat((v_labels, v_labels), dim=-1)                       # REF:trainer.py:50 (needs P == T)
    s_labels = torch.cat((s_labels, s_labels), dim=-1)                       # REF:trainer.py:53
    return dict(
        input_ids=(text_inputs, visual_batch[1].to(dev), speech_batch[1].to(dev), twv_ids, tws_ids),
        token_type_ids=(text_batch[2].to(dev), visual_batch[3].to(dev), speech_batch[3].to(dev)),
        attention_mask=(text_batch[3].to(dev), (attention_batch[0].to(dev), visual_batch[4].to(dev)),
                        (attention_batch[1].to(dev), speech_batch[4].to(dev))),
        masked_labels=(text_labels, v_labels, s_labels),
        ap_label=(visual_batch[2].to(dev), speech_batch[2].to(dev)),
        sentiment=text_batch[-1].to(dev),
    )


def should_step(step: int, gas: int, quirk: bool = True) -> bool:
    return (((step + 1) & gas) == 0) if quirk else (((step + 1) % gas) == 0)


def train_epoch(args, model, traindata, optimizer, scheduler, tokenizer=None, *, device="cuda", dp=None, quirk_step=True,
                generator=None, shuffle=True, batches=None):
    """One epoch.  ``traindata``: a Dataset of the reference's 16-tuples (collated here), or pass
    ``batches`` = an iterable of ready model-kwargs dicts (the synthetic generator).
    Returns the reference's 6-tuple (REF:trainer.py:101): (train_loss, text_loss, visual_loss,
    speech_loss, ap_loss_of_the_LAST_step, label_loss) each divided by the number of steps."""
    if batches is None:
        from torch.utils.data import DataLoader, RandomSampler, SequentialSampler
        sampler = RandomSampler(traindata) if shuffle else SequentialSampler(traindata)
        loader = DataLoader(traindata, sampler=sampler, batch_size=args.train_batch_size, collate_fn=collate)
        batches = (pack_step_inputs(b, args, device, generator) for b in loader)
    gas = args.gradient_accumulation_step
    model.train()
    train_loss = torch.zeros((), device=device)
    label_loss = torch.zeros((), device=device)
    ap_loss = torch.zeros((), device=device)
    n = 0
    for step, kwargs in enumerate(batches):
        stepping = should_step(step, gas, quirk_step)
        ctx = dp.no_sync() if (dp is not None and not stepping) else _null()
        with ctx:
            outputs, _ = model(**kwargs)
            loss = outputs[0]
            loss.mean().backward()                                           # REF:trainer.py:83
        train_loss += loss.detach().mean()
        label_loss += outputs[5].detach().mean()
        ap_loss = outputs[4].detach()
        n += 1
        if stepping:                                                         # REF:trainer.py:96-99
            if dp is not None:
                dp.finish_backward()
            optimizer.step()
            scheduler.step()
            optimizer.zero_grad()
    if n == 0:
        return (0.0,) * 6
    return (float(train_loss) / n, 0.0, 0.0, 0.0, float(ap_loss) / n, float(label_loss) / n)


def on_input_stream(model, batches):
    """Wraps an iterable of model-kwargs dicts whose tensors are BUILT ON THE GPU (dataset.DeviceBatchBuilder, the MLM masking
    kernel): every batch is produced on ``model.input_stream`` instead of the compute stream and TAGGED for the model (its first
    tensor is remembered until the next batch is asked for), so that forward() runs the step prologue on the input stream too --
    batch building and the prologue then run ahead of the previous step's backward / optimizer tail, and the host enqueues a step
    ahead of the GPU (a slow host no longer shows up as GPU idle time).  Only the tagged batch takes that path: an eval_epoch or a
    plain train_epoch afterwards, whose batches are built on the compute stream, is synchronised as usual (the model's own
    ``async_prologue`` switch is left alone).  The source tensors the builder reads must be complete (a resident dataset).
    ``train_epoch(..., batches=on_input_stream(model, builder_batches))``."""
    side = model.input_stream

    def mark(x, main):
        if torch.is_tensor(x):
            if x.is_cuda:
                x.record_stream(main)              # allocated on the input stream's pool, read by the compute stream
        elif isinstance(x, (tuple, list)):
            for y in x:
                mark(y, main)
        elif isinstance(x, dict):
            for y in x.values():
                mark(y, main)

    def first_tensor(b):
        ids = b.get("input_ids")
        return ids[0] if isinstance(ids, (tuple, list)) else ids

    it = iter(batches)
    try:
        while True:
            main = torch.cuda.current_stream()
            with torch.cuda.stream(side):
                try:
                    b = next(it)
                except StopIteration:
                    return
            mark(b, main)
            model.__dict__["_input_stream_batch"] = first_tensor(b)
            yield b
            model.__dict__["_input_stream_batch"] = None
    finally:                                       # exhausted, closed or abandoned: nothing stays tagged
        model.__dict__["_input_stream_batch"] = None


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def default_args(**kw):
    """The reference's argparse defaults that the loop reads (REF:train.py:24-42)."""
    d = dict(train_batch_size=32, mlm=True, mlm_probability=0.15, gradient_accumulation_step=1, learning_rate=5e-4,
             warmup_proportion=1.0, n_epochs=200, max_seq_length=40)
    d.update(kw)
    return types.SimpleNamespace(**d)


def build_optimizer(model, args, num_train_optimization_steps, mode="hf"):
    """REF:train.py:76-97: two parameter groups by NAME substring, AdamW, linear warm-up with
    warmup = N and total = warmup_proportion * N."""
    from .optim import AdamW, get_linear_schedule_with_warmup
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=args.learning_rate, mode=mode)
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=num_train_optimization_steps,
                                            num_training_steps=args.warmup_proportion * num_train_optimization_steps)
    return opt, sched


# ================================================================================================
# evaluation, metrics, epoch loop (REF:trainer.py:103-290)
# ================================================================================================
def eval_epoch(args, model, valdata, tokenizer=None, *, device="cuda", generator=None, batches=None):
    """REF:trainer.py:103-199.  The model in eval mode under ``no_grad``; MLM masking is applied at evaluation too
    when ``args.mlm`` (the reference does); shuffled order (RandomSampler) like the reference.  Returns its 8-tuple
    (dev_loss, text_loss, visual_loss, speech_loss, ap_loss, label_loss, preds, labels), every loss divided by the number
    of steps -- ``text/visual/speech_loss`` are 0 because the model returns ``None`` for them, and ``ap_loss`` is the LAST
    step's value divided by the number of steps (REF:trainer.py:199 divides the loop variable, not a sum)."""
    import numpy as np
    if batches is None:
        from torch.utils.data import DataLoader, RandomSampler
        loader = DataLoader(valdata, sampler=RandomSampler(valdata), batch_size=args.val_batch_size, collate_fn=collate)
        batches = (pack_step_inputs(b, args, device, generator) for b in loader)
    model.eval()
    dev_loss = torch.zeros((), device=device)
    label_loss = torch.zeros((), device=device)
    ap_loss = torch.zeros((), device=device)
    preds, labels = [], []
    n = 0
    with torch.no_grad():
        for kwargs in batches:
            outputs, logits = model(**kwargs)
            dev_loss += outputs[0].mean()
            label_loss += outputs[5].mean()
            ap_loss = outputs[4]
            preds.append(logits.detach().float())
            labels.append(kwargs["sentiment"].detach())
            n += 1
    if n == 0:
        return (0.0,) * 6 + (np.zeros((0,)), np.zeros((0,)))
    preds = torch.cat(preds).cpu().numpy()                          # one device->host transfer per epoch, not per step
    labels = torch.cat(labels).cpu().numpy()
    return (float(dev_loss) / n, 0.0, 0.0, 0.0, float(ap_loss) / n, float(label_loss) / n, preds, labels)


def _weighted_f1(y_true, y_pred):
    """F1 per class weighted by the class's support in ``y_true`` (sklearn ``f1_score(average="weighted")``)."""
    import numpy as np
    y_true, y_pred = np.asarray(y_true).reshape(-1), np.asarray(y_pred).reshape(-1)
    total, score = y_true.size, 0.0
    for c in np.unique(np.concatenate((y_true, y_pred))):
        tp = float(np.sum((y_pred == c) & (y_true == c)))
        fp = float(np.sum((y_pred == c) & (y_true != c)))
        fn = float(np.sum((y_pred != c) & (y_true == c)))
        f1 = 2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 0.0
        score += f1 * float(np.sum(y_true == c)) / max(total, 1)
    return score


def test_CE_score_model(preds, y_test):
    """REF:trainer.py:201-215 (classification heads): (accuracy, mean absolute error, weighted F1) of class predictions."""
    import numpy as np
    preds, y_test = np.asarray(preds), np.asarray(y_test)
    mae = float(np.mean(np.absolute(preds - y_test)))
    return float(np.mean(preds.reshape(-1) == y_test.reshape(-1))), mae, _weighted_f1(y_test, preds)


def test_MSE_score_model(preds, y_test, use_zero=False):
    """REF:trainer.py:217-228 (regression heads, num_labels 1 / 7): MAE of the raw scores, then accuracy and weighted F1 of
    the sign (>= 0) -- note the broadcasting of the reference: ``preds`` is [N,1], ``y_test`` [N], so its MAE is the mean over
    the [N,N] difference table; kept (callers compare epochs with it), documented here."""
    import numpy as np
    preds, y_test = np.asarray(preds), np.asarray(y_test)
    mae = float(np.mean(np.absolute(preds - y_test)))
    p, y = (preds >= 0), (y_test >= 0)
    if p.ndim == 2 and y.ndim == 1:                                  # sklearn flattens a [N,1] prediction column
        p = p.reshape(-1)
    return float(np.mean(p == y)), mae, _weighted_f1(y, p)

test_CE_score_model.__test__ = False                                # names start with "test_": not pytest cases
test_MSE_score_model.__test__ = False


def _dated_dir(path):
    """``path/<YYYYMMDD>-NN`` with the first unused NN, created: where train() below saves (the reference's directory naming, REF:utils.py:35-50;
    the reference's logging / path utilities themselves are out of scope)."""
    import datetime
    import os
    os.makedirs(path, exist_ok=True)
    stamp = datetime.datetime.now().strftime("%Y%m%d")
    i = 0
    while os.path.exists(os.path.join(path, f"{stamp}-{i:02d}")):
        i += 1
    out = os.path.join(path, f"{stamp}-{i:02d}")
    os.mkdir(out)
    return out


def train(args, model, train_dataset, val_dataset, test_dataset, optimizer, scheduler, tokenizer=None, logger=None, *, device="cuda",
          dp=None, save_root="./model_save", numpy_root="./numpy_save", patience_limit=25, epoch_batches=None):
    """REF:trainer.py:230-290: per epoch train -> validate -> test; the state dict is saved (``model_<epoch>.pt``, the
    reference's checkpoint format = its state-dict keys) whenever the TEST accuracy improves; after ``patience_limit`` epochs
    without improvement the best predictions / targets go to ``predict.npy`` / ``target.npy`` and the loop stops.
    ``epoch_batches``: optional callable ``(split, epoch) -> iterable of model kwargs`` replacing the datasets (synthetic data).
    Returns a dict with the best epoch's numbers and the per-epoch history."""
    import os
    import numpy as np
    log = logger.info if logger is not None else (lambda *a, **k: None)
    save_dir = _dated_dir(save_root)
    log("Model save path: {}".format(save_dir))
    score = test_MSE_score_model if getattr(args, "num_labels", 7) in (1, 7) else test_CE_score_model
    best = dict(epoch=-1, acc=0.0, loss=float("inf"), mae=None, f_score=None, preds=None, labels=None, path=None)
    history = []
    patience = 0
    rank0 = dp is None or torch.distributed.get_rank() == 0
    for epoch in range(int(args.n_epochs)):
        patience += 1
        tb = epoch_batches("train", epoch) if epoch_batches else None
        tr = train_epoch(args, model, train_dataset, optimizer, scheduler, tokenizer, device=device, dp=dp, batches=tb)
        log("[Train Epoch {}] Joint Loss : {} AP Loss : {} Label Loss : {}".format(epoch + 1, tr[0], tr[4], tr[5]))
        va = eval_epoch(args, model, val_dataset, tokenizer, device=device, batches=epoch_batches("val", epoch) if epoch_batches else None)
        log("[Val Epoch {}] Joint Loss : {} AP Loss : {} Label Loss : {}".format(epoch + 1, va[0], va[4], va[5]))
        te = eval_epoch(args, model, test_dataset, tokenizer, device=device, batches=epoch_batches("test", epoch) if epoch_batches else None)
        acc, mae, f_score = score(te[6], te[7])
        if dp is not None and torch.distributed.get_world_size() > 1:
            # ranks evaluate with their own sampler order and MLM draws, so their metrics can differ in the last digits: rank 0's
            # numbers decide the save rule and the patience stop on EVERY rank (a rank that stopped alone would leave the
            # others waiting in the next epoch's gradient all-reduce)
            t = torch.tensor([acc, mae, f_score], dtype=torch.float64, device=device)
            torch.distributed.broadcast(t, src=0)
            acc, mae, f_score = (float(x) for x in t.tolist())
        log("[Epoch {}] Test_ACC : {}, Test_MAE : {}, Test_F_Score: {}".format(epoch + 1, acc, mae, f_score))
        history.append(dict(epoch=epoch + 1, train_loss=tr[0], valid_loss=va[0], test_acc=acc, test_mae=mae, test_f_score=f_score))
        if acc > best["acc"]:
            path = os.path.join(save_dir, "model_" + str(epoch + 1) + ".pt")
            if rank0:
                torch.save(model.state_dict(), path)
            best.update(epoch=epoch, acc=acc, loss=va[0], mae=mae, f_score=f_score, preds=te[6], labels=te[7], path=path if rank0 else None)
            patience = 0
        if patience == patience_limit:
            if rank0 and best["preds"] is not None:
                out = _dated_dir(numpy_root)
                np.save(os.path.join(out, "predict.npy"), best["preds"])
                np.save(os.path.join(out, "target.npy"), best["labels"])
            break
    log("[Best Epoch {}] Best_ACC : {}, Best_MAE : {}, Best_F_Score: {}".format(best["epoch"] + 1, best["acc"], best["mae"], best["f_score"]))
    best["history"] = history
    best["save_dir"] = save_dir
    return best
