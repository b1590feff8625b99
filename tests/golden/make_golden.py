#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REAL reference (kimkyeonghun/MSA at
/root/reference, read-only) on CPU in the build container.

The reference has no tests and no golden vectors, so these files are the pin for
``oracle/mmbert_oracle.py``.  Run from the repo root:  ``python tests/golden/make_golden.py``.
It needs /root/reference and ``transformers`` (5.15.0 here); neither exists on the GPU box, which
only ever reads the committed ``.npz`` files.  Nothing of the reference's source text is stored:
the fixtures hold inputs' seeds/shapes and the reference's numeric outputs.

Shim: transformers>=5 needs post_init() before init_weights(); TEXTDIM and
the CPC x_size are hard-coded to 1024 in the reference and are set to H; DEVICE -> cpu.
Weights are ``oracle.mmbert_oracle.seeded_params`` loaded with ``load_state_dict`` so that the
tests can rebuild them from the seed instead of committing 16 MB of embeddings.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from transformers import BertConfig, PreTrainedModel

_orig_iw = PreTrainedModel.init_weights


def _iw(self):
    if "all_tied_weights_keys" not in self.__dict__:
        return self.post_init()
    return _orig_iw(self)


PreTrainedModel.init_weights = _iw

import config as ref_config                         # noqa: E402  (reference module)
ref_config.DEVICE = torch.device("cpu")
import MMBertEmbedding                              # noqa: E402
import MMBertForPretraining as M                    # noqa: E402

from oracle import mmbert_oracle as O               # noqa: E402
from msa_amd.data import synthetic_batch            # noqa: E402
from tests.golden.dataset_features import synthetic_features   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def build_reference(cfg, seed=0, dropout=None):
    H = cfg["hidden"]
    MMBertEmbedding.TEXTDIM = H
    kw = {}
    if dropout is not None:
        kw = dict(hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    bc = BertConfig(vocab_size=cfg["vocab"], hidden_size=H, num_hidden_layers=cfg["layers"],
                    num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"],
                    max_position_embeddings=512, **kw)
    bc._attn_implementation = "eager"
    model = M.MMBertForPretraining(bc)
    model.bert.set_joint_embeddings(cfg["dataset"])
    for n in ("cpc_zt", "cpc_zv", "cpc_za"):
        getattr(model, n).net = torch.nn.Linear(H, H)
    if dropout is not None:
        model.bert.jointEmbeddings.dropout.p = dropout
    model.set_alpha_beta(cfg.get("alpha", 1.0), cfg.get("beta", 1.0))
    sd = O.seeded_params(cfg, seed)
    sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    sd["cls.predictions.decoder.bias"] = sd["cls.predictions.bias"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    assert model.cls.predictions.decoder.weight.data_ptr() == model.bert.embeddings.word_embeddings.weight.data_ptr()
    return model


def np_(x):
    return x.detach().cpu().numpy()


CFG1 = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)


def gen_full(name, cfg, B, T, Pv, Pa, seed):
    """G4 + G5: full forward (eval), per-layer hidden states of each pass, losses, grads."""
    model = build_reference(cfg).eval()
    batch = synthetic_batch(B, T, Pv, Pa, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=seed)
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(np_(o[0] if isinstance(o, tuple) else o)))
             for l in model.bert.encoder.layer]
    emb = []
    hooks.append(model.bert.embeddings.register_forward_hook(lambda m, i, o: emb.append(np_(o))))
    jemb = []
    hooks.append(model.bert.jointEmbeddings.register_forward_hook(lambda m, i, o: jemb.append(np_(o))))
    pooled = []
    hooks.append(model.bert.pooler.register_forward_hook(lambda m, i, o: pooled.append(np_(o))))
    outputs, logits = model(**batch)
    for h in hooks:
        h.remove()
    outputs[0].mean().backward()                               # REF:trainer.py:83
    d = {}
    d["meta"] = np.array([B, T, Pv, Pa, seed])
    d["joint_loss"], d["ap_loss"], d["label_loss"], d["nce"] = (np_(outputs[i]) for i in (0, 4, 5, 6))
    assert outputs[1] is None and outputs[2] is None and outputs[3] is None
    d["logits"] = np_(logits)
    L = cfg["layers"]
    for pi, tag in enumerate("tvs"):
        sc, rel = outputs[7 + 2 * pi], outputs[8 + 2 * pi]
        d[f"{tag}_scores_shape"] = np.array(sc.shape)
        d[f"{tag}_scores_head"] = np_(sc[:, :, :48])
        d[f"{tag}_scores_stride"] = np_(sc[:, :, 5::611])
        d[f"{tag}_scores_lse"] = np_(torch.logsumexp(sc, -1))
        d[f"{tag}_rel"] = np_(rel)
        for l in range(L):
            d[f"{tag}_hidden{l}"] = hidden[pi * L + l]
        d[f"{tag}_emb"] = emb[pi]
        d[f"{tag}_pooled"] = pooled[pi]
    d["v_jemb"], d["s_jemb"] = jemb
    nograd = []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
            continue
        g = p.grad
        d["gnorm/" + n] = np.array(g.norm().item())
        d["gsum/" + n] = np.array(g.double().sum().item())
        flat = g.reshape(-1)
        d["ghead/" + n] = np_(flat[:16])
        if n == "bert.embeddings.word_embeddings.weight":
            # rows that received gradient: keep a few complete ones (tied decoder + lookup)
            rows = torch.tensor([0, 101, 102, 103, 1000, 2000])
            d["grows/" + n] = np_(g[rows])
    d["nograd"] = np.array(sorted(nograd))
    d["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "joint_loss", float(outputs[0]), "ap", float(outputs[4]), "label", float(outputs[5]),
          "nce", float(outputs[6]), "nograd", nograd)


def gen_units(cfg):
    """G1 JointEmbeddings, G2 extended mask, G3 CPC -- module-level calls on seeded inputs."""
    model = build_reference(cfg).eval()
    H = cfg["hidden"]
    rng = np.random.Generator(np.random.PCG64(77))
    d = {}
    text_emb = torch.from_numpy(rng.standard_normal((2, 50, H)).astype(np.float32))
    for tag, D in (("v", 35), ("s", 74)):
        pair = rng.standard_normal((2, 64, D))
        pair[1, 40:] = 0.0
        d[f"g1_{tag}_pair"] = pair
        d[f"g1_{tag}_out"] = np_(model.bert.jointEmbeddings(text_emb, torch.from_numpy(pair)))
    d["g1_text_emb"] = np_(text_emb)
    try:
        model.bert.jointEmbeddings(text_emb, torch.zeros(2, 4, 33))
        d["g1_wrongdim_raises"] = np.array(0)
    except Exception as e:                                    # REF:MMBertEmbedding.py:66
        d["g1_wrongdim_raises"] = np.array(1)
        d["g1_wrongdim_msg"] = np.array(str(e))
    # G2
    m2 = (rng.random((3, 11)) > 0.3).astype(np.float64)
    m3 = rng.standard_normal((3, 9, 5))
    m3[0, 6:] = 0.0
    m3[1, 2, 0] = 0.0                                          # a live frame whose feature 0 is exactly 0
    m3m = (m3 != 0).astype(np.float64)
    d["g2_m2"], d["g2_m3"] = m2, m3m
    for joint in (False, True):
        d[f"g2_out2_j{int(joint)}"] = np_(model.bert.get_extended_attention_mask(torch.from_numpy(m2), (3,), "cpu", joint))
        d[f"g2_out3_j{int(joint)}"] = np_(model.bert.get_extended_attention_mask(torch.from_numpy(m3m), (3,), "cpu", joint))
    d["g2_out3i_j1"] = np_(model.bert.get_extended_attention_mask(torch.from_numpy(m3m.astype(np.int64)), (3,), "cpu", True))
    # G3
    x = torch.from_numpy(rng.standard_normal((4, H)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((4, H)).astype(np.float32))
    d["g3_x"], d["g3_y"] = np_(x), np_(y)
    for n in ("cpc_zt", "cpc_zv", "cpc_za"):
        d["g3_" + n] = np_(getattr(model, n)(x, y))
    np.savez_compressed(os.path.join(OUT, "units.npz"), **d)
    print("units ok")


def gen_collate():
    """G6: the collate() output contract (dtypes, shapes, mask quirks) on seeded examples."""
    import model_utils
    rng = np.random.Generator(np.random.PCG64(5))
    B, T = 3, 8
    ex = []
    for b in range(B):
        n = 3 + b
        te = [101] + list(rng.integers(1000, 2000, n)) + [102] + [0] * (T - n - 2)
        ve = rng.standard_normal((T, 35)); ve[n + 2:] = 0
        se = rng.standard_normal((T, 74)); se[n + 2:] = 0
        tti = torch.zeros(T)
        vti = torch.cat((torch.zeros(T), torch.ones(T)))
        ex.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(0.5 * b),
                   te, ve, torch.tensor(b % 2), vti, torch.tensor(0.5 * b),
                   te, se, torch.tensor(1), vti, torch.tensor(0.5 * b), "seg%d" % b, "raw"))
    text_b, vis_b, sp_b, att_b, seg, raw = model_utils.collate(ex)
    d = {}
    for gname, grp in (("text", text_b), ("visual", vis_b), ("speech", sp_b), ("attention", att_b)):
        for i, t in enumerate(grp):
            d[f"{gname}{i}"] = np_(t)
            d[f"{gname}{i}_dtype"] = np.array(str(t.dtype))
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **d)
    print("collate ok")


def gen_dataset():
    """MMBertDataset.__getitem__ (REF:MMBertDataset.py:194-202) and collate of three of its items under a seeded
    `random`, for every (dataset, task, num_labels) rule of sentiment_selection."""
    import random
    import MMBertDataset as refds
    import model_utils
    refds.cudas = torch.device("cpu")
    d = {}
    cases = [("mosei", "sentiment", 1), ("mosei", "sentiment", 7), ("mosei", "sentiment", 2), ("mosei", "happy", 2),
             ("mosei", "sad", 6), ("mosi", "sentiment", 1), ("mosi", "sentiment", 2), ("ur_funny", "sentiment", 2)]
    for ci, (ds_name, task, nl) in enumerate(cases):
        feats = synthetic_features(dataset=ds_name, seed=11 + ci)
        ds = refds.MMBertDataset(None, feats, ds_name, task, nl)
        random.seed(100 + ci)
        tag = f"c{ci}"
        d[tag + "/meta"] = np.array([ds_name, task, str(nl)])
        for i in range(len(ds)):
            item = ds[i]
            for f, v in enumerate(item[:14]):
                a = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v)
                d[f"{tag}/item{i}/f{f}"] = a[:, :4] if f in (5, 10) else a          # pair features: 4 columns identify the source item
                if torch.is_tensor(v):
                    d[f"{tag}/item{i}/f{f}_dtype"] = np.array(str(v.dtype))
        random.seed(200 + ci)
        text_b, vis_b, sp_b, att_b, seg, raw = model_utils.collate([ds[i] for i in (0, 3, 4, 2)])
        for gname, grp in (("text", text_b), ("visual", vis_b), ("speech", sp_b), ("attention", att_b)):
            for i, t in enumerate(grp):
                a = np_(t)
                d[f"{tag}/batch/{gname}{i}"] = a[..., :4] if a.ndim == 3 else a       # feature tensors / their masks: 4 columns
                d[f"{tag}/batch/{gname}{i}_dtype"] = np.array(str(t.dtype))
        d[f"{tag}/batch/seg"] = np.array(seg)
    np.savez_compressed(os.path.join(OUT, "dataset.npz"), **d)
    print("dataset ok", len(d), "arrays")


def gen_train(cfg):
    """G8: four micro-batches through REF trainer.train_epoch (dropout 0, mlm off so that no torch
    RNG enters the arithmetic), torch.optim.AdamW(eps=1e-6) + linear warm-up.  Pins the
    every-2nd-step rule (REF:trainer.py:96), the loss bookkeeping and the parameter update."""
    import model_utils
    import trainer
    from transformers.optimization import get_linear_schedule_with_warmup
    trainer.DEVICE = model_utils.DEVICE = torch.device("cpu")
    trainer.tqdm = lambda x, **k: x
    model = build_reference(cfg, dropout=0.0)
    rng = np.random.Generator(np.random.PCG64(9))
    N, T = 8, 12
    items = []
    for b in range(N):
        n = 4 + (b % 5)
        te = [101] + list(map(int, rng.integers(1000, cfg["vocab"], n))) + [102] + [0] * (T - n - 2)
        ve = rng.standard_normal((T, 35)); ve[n + 2:] = 0
        se = rng.standard_normal((T, 74)); se[n + 2:] = 0
        tti = torch.zeros(T)
        vti = torch.cat((torch.zeros(T), torch.ones(T)))
        sent = float(rng.uniform(-3, 3))
        items.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(sent),
                      te, ve, torch.tensor(int(rng.integers(0, 2))), vti, torch.tensor(sent),
                      te, se, torch.tensor(int(rng.integers(0, 2))), vti, torch.tensor(sent), "s", "r"))
    order = []

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return N

        def __getitem__(self, i):
            order.append(int(i))
            return items[i]
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]                 # REF:train.py:78
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    lr = 5e-4
    opt = torch.optim.AdamW(groups, lr=lr, eps=1e-6)
    n_opt_steps = 4
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=n_opt_steps, num_training_steps=1.0 * n_opt_steps)
    args = types.SimpleNamespace(train_batch_size=2, mlm=False, mlm_probability=0.15, gradient_accumulation_step=1)
    losses = []
    orig_fwd = model.forward

    def rec_fwd(*a, **k):
        out = orig_fwd(*a, **k)
        losses.append([float(out[0][0]), float(out[0][4]), float(out[0][5]), float(out[0][6])])
        return out
    model.forward = rec_fwd
    before = {n: p.detach().clone() for n, p in named}
    torch.manual_seed(3)
    ret = trainer.train_epoch(args, model, DS(), opt, sched, None)
    d = {"order": np.array(order), "losses": np.array(losses), "lr": np.array(lr),
         "ret": np.array([float(r) for r in ret]), "n_opt_steps": np.array(n_opt_steps)}
    for i, it in enumerate(items):
        d[f"item{i}_text"] = np.array(it[4])
        d[f"item{i}_visual"] = np.array(it[5])
        d[f"item{i}_speech"] = np.array(it[10])
        d[f"item{i}_ap"] = np.array([int(it[6]), int(it[11])])
        d[f"item{i}_sent"] = np.array(float(it[3]))
    for n, p in named:
        delta = (p.detach() - before[n])
        d["dnorm/" + n] = np.array(delta.norm().item())
        d["dhead/" + n] = np_(delta.reshape(-1)[:16])
        d["pnorm/" + n] = np.array(p.detach().norm().item())
    np.savez_compressed(os.path.join(OUT, "train4.npz"), **d)
    print("train4 losses", losses, "ret", d["ret"])


def _make_items(rng, N, T, vocab):
    """N examples in the 16-tuple layout MMBertDataset.__getitem__ yields (REF:MMBertDataset.py:194-202), padded to T."""
    items = []
    for b in range(N):
        n = 3 + (b % (T - 5))
        te = [101] + list(map(int, rng.integers(1000, vocab, n))) + [102] + [0] * (T - n - 2)
        ve = rng.standard_normal((T, 35)); ve[n + 2:] = 0
        se = rng.standard_normal((T, 74)); se[n + 2:] = 0
        tti = torch.zeros(T)
        vti = torch.cat((torch.zeros(T), torch.ones(T)))
        sent = float(rng.uniform(-3, 3))
        items.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(sent),
                      te, ve, torch.tensor(int(rng.integers(0, 2))), vti, torch.tensor(sent),
                      te, se, torch.tensor(int(rng.integers(0, 2))), vti, torch.tensor(sent), "s", "r"))
    return items


def gen_eval(cfg):
    """G9: REF trainer.eval_epoch (REF:trainer.py:103-194) on 6 examples with val_batch_size 4 (a full and a short batch), mlm off,
    model.eval(): its 8-tuple -- dev loss, the three always-zero modality losses, ap_loss of the LAST batch / steps, label loss,
    predictions [N,1], labels [N] -- plus the per-batch losses, the sampler's order, and what test_MSE_score_model
    (REF:trainer.py:217-228, sklearn metrics) makes of the predictions."""
    import model_utils
    import trainer
    trainer.DEVICE = model_utils.DEVICE = torch.device("cpu")
    trainer.tqdm = lambda x, **k: x
    model = build_reference(cfg, dropout=0.0)
    rng = np.random.Generator(np.random.PCG64(19))
    N, T = 6, 12
    items = _make_items(rng, N, T, cfg["vocab"])
    order = []

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return N

        def __getitem__(self, i):
            order.append(int(i))
            return items[i]
    args = types.SimpleNamespace(val_batch_size=4, mlm=False, mlm_probability=0.15)
    losses = []
    orig_fwd = model.forward

    def rec_fwd(*a, **k):
        out = orig_fwd(*a, **k)
        losses.append([float(out[0][0]), float(out[0][4]), float(out[0][5]), float(out[0][6])])
        return out
    model.forward = rec_fwd
    torch.manual_seed(7)
    ret = trainer.eval_epoch(args, model, DS(), None)
    d = {"order": np.array(order), "losses": np.array(losses), "ret6": np.array([float(r) for r in ret[:6]]),
         "preds": np.asarray(ret[6]), "labels": np.asarray(ret[7]), "val_batch_size": np.array(4)}
    d["mse_scores"] = np.array([float(x) for x in trainer.test_MSE_score_model(ret[6], ret[7])])
    for i, it in enumerate(items):
        d[f"item{i}_text"] = np.array(it[4])
        d[f"item{i}_visual"] = np.array(it[5])
        d[f"item{i}_speech"] = np.array(it[10])
        d[f"item{i}_ap"] = np.array([int(it[6]), int(it[11])])
        d[f"item{i}_sent"] = np.array(float(it[3]))
    np.savez_compressed(os.path.join(OUT, "eval6.npz"), **d)
    print("eval6 losses", losses, "ret", d["ret6"], "scores", d["mse_scores"])


def gen_mask_tokens():
    """G10: REF model_utils.mask_tokens (REF:model_utils.py:6-39) under a seeded global torch RNG on CPU, with a stand-in for the
    tokenizer of the pinned transformers 2.8: ``get_special_tokens_mask(already_has_special_tokens=True)`` flags [CLS] and [SEP]
    only.  ``_pad_token`` is None here so that the reference's PAD branch (:24-26) is skipped: it computes a non-in-place
    ``masked_fill`` whose result is dropped (and calls ``.cuda()``, which this CPU container lacks) -- skipping it changes nothing."""
    import model_utils
    model_utils.DEVICE = torch.device("cpu")

    class Tok28:
        mask_token = "[MASK]"
        _pad_token = None
        pad_token_id = 0

        def get_special_tokens_mask(self, ids, already_has_special_tokens=False):
            assert already_has_special_tokens
            return [1 if x in (101, 102) else 0 for x in ids]

        def convert_tokens_to_ids(self, tok):
            assert tok == "[MASK]"
            return 103
    rng = np.random.Generator(np.random.PCG64(23))
    B, T = 48, 20
    ids = np.zeros((B, T), np.int64)
    for b in range(B):
        n = int(rng.integers(3, T - 2))
        ids[b, 0] = 101
        ids[b, 1:1 + n] = rng.integers(1000, 30000, n)
        ids[b, 1 + n] = 102
    args = types.SimpleNamespace(mlm_probability=0.15)
    d = {"inputs": ids}
    for seed in (1, 2):
        torch.manual_seed(seed)
        out, labels = model_utils.mask_tokens(torch.from_numpy(ids.copy()), Tok28(), args)
        d[f"out_seed{seed}"], d[f"labels_seed{seed}"] = np_(out), np_(labels)
    sel = d["labels_seed1"] != -100
    assert sel[ids == 0].any(), "the reference selects [PAD] positions"
    assert not sel[(ids == 101) | (ids == 102)].any()
    np.savez_compressed(os.path.join(OUT, "mask_tokens.npz"), **d)
    print("mask_tokens ok: selected", int(sel.sum()), "of", sel.size, "| on PAD", int(sel[ids == 0].sum()))


if __name__ == "__main__":
    if sys.argv[1:] == ["dataset"]:                 # only the MMBertDataset fixture (the others are unchanged)
        gen_dataset()
        sys.exit(0)
    if sys.argv[1:] == ["eval"]:
        gen_eval(dict(CFG1, vocab=4096))
        gen_mask_tokens()
        sys.exit(0)
    gen_units(CFG1)
    gen_collate()
    gen_full("cfg1_T50_P64", CFG1, 2, 50, 64, 64, seed=1)
    gen_full("cfg1_T50_P50", CFG1, 2, 50, 50, 50, seed=2)
    gen_full("h64_L1_T16_P24x40", dict(CFG1, hidden=64, layers=1, heads=4, intermediate=128, vocab=2048,
                                       alpha=0.7, beta=0.3), 3, 16, 24, 40, seed=3)
    gen_train(dict(CFG1, vocab=4096))
    gen_eval(dict(CFG1, vocab=4096))
    gen_mask_tokens()
    gen_dataset()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
