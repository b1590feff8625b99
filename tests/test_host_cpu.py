"""CPU-only tests of the host logic: the collate / masking / stepping counterparts of the reference's
trainer against the golden vectors, the schedule, and the world_size-2 (gloo) gradient bucketer."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from msa_amd import trainer as T
from msa_amd.parallel import GradBucketer


def _collate_examples():
    """The same seeded examples tests/golden/make_golden.py:gen_collate feeds the REAL collate()."""
    rng = np.random.Generator(np.random.PCG64(5))
    B, Tn = 3, 8
    ex = []
    for b in range(B):
        n = 3 + b
        te = [101] + list(rng.integers(1000, 2000, n)) + [102] + [0] * (Tn - n - 2)
        ve = rng.standard_normal((Tn, 35)); ve[n + 2:] = 0
        se = rng.standard_normal((Tn, 74)); se[n + 2:] = 0
        tti = torch.zeros(Tn)
        vti = torch.cat((torch.zeros(Tn), torch.ones(Tn)))
        ex.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(0.5 * b),
                   te, ve, torch.tensor(b % 2), vti, torch.tensor(0.5 * b),
                   te, se, torch.tensor(1), vti, torch.tensor(0.5 * b), "seg%d" % b, "raw"))
    return ex


def test_collate_reproduces_reference_output(golden_dir):
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    text_b, vis_b, sp_b, att_b, seg, raw = T.collate(_collate_examples())
    for gname, grp in (("text", text_b), ("visual", vis_b), ("speech", sp_b), ("attention", att_b)):
        for i, t in enumerate(grp):
            assert str(t.dtype) == str(g[f"{gname}{i}_dtype"]), (gname, i, t.dtype)
            np.testing.assert_array_equal(t.numpy(), g[f"{gname}{i}"], err_msg=f"{gname}{i}")
    assert seg == ["seg0", "seg1", "seg2"]


def test_collate_rejects_ragged_modalities():
    ex = _collate_examples()
    bad = list(ex[0]); bad[5] = bad[5][:-1]
    with pytest.raises(AssertionError):
        T.collate([tuple(bad)] + ex[1:])


def test_mask_tokens_rule_and_pack():
    args = T.default_args(mlm_probability=0.5)
    g = torch.Generator().manual_seed(0)
    ids = torch.tensor([[101, 7, 8, 9, 10, 11, 102, 0, 0]] * 64)
    out, labels = T.mask_tokens(ids.clone(), args, g)
    sel = labels != -100
    assert not sel[:, 0].any() and not sel[:, 6].any()                  # [CLS] / [SEP] never selected
    assert 0.3 < sel[:, 7:].float().mean() < 0.7                        # [PAD] IS selectable in the reference (dropped masked_fill)
    assert (labels[:, 7:][sel[:, 7:]] == 0).all()                       # ... with label 0
    assert 0.4 < sel[:, 1:6].float().mean() < 0.6
    assert ((out == 103) <= sel).all() and (labels[sel] == ids[sel]).all()
    assert 0.7 < (out[sel] == 103).float().mean() < 0.9                 # 80 % -> [MASK]
    assert (out[~sel] == ids[~sel]).all()
    # the opt-in deviation: PAD never selected
    _, lab2 = T.mask_tokens(ids.clone(), args, torch.Generator().manual_seed(0), special_ids=(T.PAD, T.CLS, T.SEP))
    assert not (lab2[:, 6:] != -100).any()
    batch = T.collate(_collate_examples())
    kw = T.pack_step_inputs(batch, T.default_args(mlm=False), "cpu")
    assert kw["masked_labels"][1].shape == (3, 16) and (kw["masked_labels"][1][:, :8] == kw["masked_labels"][1][:, 8:]).all()
    assert kw["input_ids"][1].dtype == torch.float64 and kw["attention_mask"][2][1].dtype == torch.int64


def test_mask_tokens_reproduces_the_reference_draws(golden_dir):
    """G10: REF model_utils.mask_tokens run for real (transformers-2.8-style tokenizer stand-in, seeded global RNG, CPU): with
    ``generator=None`` ours makes the same two ``torch.bernoulli`` draws in the same order -- inputs and labels bit for bit,
    [PAD] positions selected like the reference selects them."""
    g = np.load(os.path.join(golden_dir, "mask_tokens.npz"))
    ids = torch.from_numpy(g["inputs"])
    for seed in (1, 2):
        torch.manual_seed(seed)
        out, labels = T.mask_tokens(ids.clone(), T.default_args(mlm_probability=0.15))
        np.testing.assert_array_equal(out.numpy(), g[f"out_seed{seed}"])
        np.testing.assert_array_equal(labels.numpy(), g[f"labels_seed{seed}"])
    assert (g["labels_seed1"][g["inputs"] == 0] != -100).any()
    # the synthetic generator follows the same rule: some PAD position of the text carries label 0
    from msa_amd.data import synthetic_batch
    b = synthetic_batch(16, 50, 8, 8, seed=1)
    lab, msk = b["masked_labels"][0], b["attention_mask"][0]
    assert ((lab == 0) & (msk == 0)).any() and not (lab[:, 0] != -100).any()
    b2 = synthetic_batch(16, 50, 8, 8, seed=1, pad_selectable=False)
    assert not ((b2["masked_labels"][0] != -100) & (msk == 0)).any()


def test_eval_scores_reproduce_the_reference(golden_dir):
    """G9 (host part): test_MSE_score_model on the reference's own eval predictions gives the reference's (acc, MAE, F1),
    the [N,1] - [N] broadcast of its MAE included."""
    g = np.load(os.path.join(golden_dir, "eval6.npz"))
    acc, mae, f1 = T.test_MSE_score_model(g["preds"], g["labels"])
    np.testing.assert_allclose([acc, mae, f1], g["mse_scores"], rtol=1e-6)
    assert g["preds"].shape == (6, 1) and g["labels"].shape == (6,)


def test_step_rule_and_schedule():
    from msa_amd.optim import AdamW, get_linear_schedule_with_warmup
    assert [T.should_step(s, 1) for s in range(4)] == [False, True, False, True]       # REF:trainer.py:96 quirk
    assert [T.should_step(s, 2) for s in range(4)] == [True, False, False, True]
    assert [T.should_step(s, 2, quirk=False) for s in range(4)] == [False, True, False, True]
    p = torch.nn.Parameter(torch.zeros(4))
    opt = AdamW([{"params": [p], "weight_decay": 0.01}], lr=1.0)
    s = get_linear_schedule_with_warmup(opt, 4, 4.0)
    lrs = [opt.lr]
    for _ in range(5):
        s.step(); lrs.append(opt.lr)
    assert lrs == [0.0, 0.25, 0.5, 0.75, 0.0, 0.0]                     # warmup == total: ramps, then 0
    with pytest.raises(RuntimeError, match="flat storage"):
        opt.step()                                                       # no silent CPU/torch fallback


# ---------------------------------------------------------------------------------- DP, world 2
def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    bounds = [0, 100, 350, 600, 900, n]
    base = torch.arange(n, dtype=torch.float32)
    flat = base * (rank + 1)
    bk = GradBucketer(flat, bounds, bucket_mb=1000 * 4 / (1 << 20) * 0.3)      # 300-element buckets
    res = {}
    for k in range(len(bounds) - 2):           # "backward": slices 0..3 complete in order, slice 4 is the tail
        bk.ready(k)
    calls_before_finish = bk.calls
    bk.finish()
    res["sum_ok"] = bool(torch.equal(flat, base * 3))                     # rank0*1 + rank1*2
    res["calls"] = calls_before_finish
    # accumulation micro-step: nothing is exchanged
    flat2 = base * (rank + 1)
    bk2 = GradBucketer(flat2, bounds, bucket_mb=1e-9)
    bk2.enabled = False
    for k in range(4):
        bk2.ready(k)
    bk2.reset()
    res["nosync_ok"] = bool(torch.equal(flat2, base * (rank + 1))) and bk2.calls == 0
    # frozen ranges are cut out of the buckets: never exchanged, whatever bucket they fall into (one straddles a bucket edge)
    flat3 = base * (rank + 1)
    bk3 = GradBucketer(flat3, bounds, bucket_mb=1000 * 4 / (1 << 20) * 0.3, skip=[(120, 200), (340, 360)])
    for k in range(len(bounds) - 2):
        bk3.ready(k)
    bk3.finish()
    keep = torch.zeros(n, dtype=torch.bool)
    keep[120:200] = True
    keep[340:360] = True
    res["skip_ok"] = bool(torch.equal(flat3[~keep], (base * 3)[~keep])) and bool(torch.equal(flat3[keep], (base * (rank + 1))[keep]))
    # out-of-order readiness is a bug
    try:
        bk2.ready(2)
        res["order_checked"] = False
    except AssertionError:
        res["order_checked"] = True
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucketer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["sum_ok"] and res["nosync_ok"] and res["order_checked"] and res["skip_ok"]
    assert res["calls"] == 2, res          # slices merged into >=300-element buckets: [0,350) and [350,900); tail in finish()


def _exchange_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from msa_amd.parallel import exchange_rows, _scatter_rows
    res = {}
    # ---- compact exchange of embedding-lookup rows == dense all-reduce of the scattered table
    V, H, n = 97, 16, 40
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(0, V + 6, (n,), generator=g)                       # duplicates, the padding row 0 and ids past the vocabulary
    ids[:3] = torch.tensor([0, 5, 5])
    rows = torch.randn(n, H, generator=g)
    dense = torch.zeros(V, H)
    _scatter_rows(dense, ids, rows)
    dist.all_reduce(dense)
    union, block = exchange_rows(ids, rows, V)
    table = torch.zeros(V, H)
    table.index_add_(0, union, block)
    res["rows_ok"] = bool(torch.allclose(table, dense, rtol=1e-6, atol=1e-6)) and bool((union[1:] > union[:-1]).all()) and int(union.min()) > 0 and int(union.max()) < V
    res["rows_touch"] = int(union.numel())
    # ---- the wire-dtype exchange: same sums as an fp32 all-reduce up to ONE rounding of each contribution and one of the sum;
    # bit-identical on every rank; ranges out of layout order (reduce_range) and skip ranges honoured
    nel = 1003
    base = torch.linspace(-3.0, 3.0, nel)
    flat = (base * (rank + 1) + 0.01 * rank).clone()
    want = sum(base * (r + 1) + 0.01 * r for r in range(world))
    bounds = [0, 100, 350, 600, 900, nel]
    bk = GradBucketer(flat, bounds, bucket_mb=300 * 4 / (1 << 20), skip=[(120, 200)], wire_dtype=torch.bfloat16)
    bk.reduce_range(900, nel)                                              # the "table" first, as DataParallel does
    for k in range(4):
        bk.ready(k)
    bk.finish(upto=900)
    keep = torch.ones(nel, dtype=torch.bool)
    keep[120:200] = False
    err = ((flat - want).abs() / (want.abs() + 1e-3))[keep].max()
    res["wire_err"] = float(err)
    res["wire_skip_ok"] = bool(torch.equal(flat[~keep], (base * (rank + 1) + 0.01 * rank)[~keep]))
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    res["wire_identical"] = all(bool(torch.equal(o[keep], other[0][keep])) for o in other)      # (the skipped range is never exchanged)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_compact_row_exchange_and_wire_dtype_world2_gloo():
    """parallel.exchange_rows (the embedding lookup's rows in compact form: all-gather of ids, sorted union, ONE all-reduce of the
    [union, H] block) equals the dense all-reduce of the scattered table; GradBucketer(wire_dtype=bfloat16) gives every rank the
    same bits, within bf16 rounding of the fp32 sums, honours skip ranges and out-of-order ranges (DESIGN S6)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["rows_ok"] and res["rows_touch"] > 20, res
    assert res["wire_err"] < 2.0 ** -7 and res["wire_skip_ok"] and res["wire_identical"], res


def test_score_functions_match_their_definitions():
    """REF:trainer.py:201-228 restated: accuracy / MAE / support-weighted F1 (checked against scikit-learn when present)."""
    rng = np.random.default_rng(0)
    p, y = rng.integers(0, 5, 200), rng.integers(0, 5, 200)
    acc, mae, f1 = T.test_CE_score_model(p, y)
    assert abs(acc - float(np.mean(p == y))) < 1e-12 and abs(mae - float(np.mean(np.abs(p - y)))) < 1e-12
    pr, yr = rng.normal(size=(50, 1)), rng.normal(size=50)
    acc2, mae2, f12 = T.test_MSE_score_model(pr, yr)
    assert abs(acc2 - float(np.mean((pr.reshape(-1) >= 0) == (yr >= 0)))) < 1e-12
    assert abs(mae2 - float(np.mean(np.abs(pr - yr)))) < 1e-12          # the reference's [N,1] - [N] broadcast, kept
    try:
        from sklearn.metrics import f1_score
    except Exception:
        return
    assert abs(f1 - f1_score(y, p, average="weighted")) < 1e-12
    assert abs(f12 - f1_score(yr >= 0, pr.reshape(-1) >= 0, average="weighted")) < 1e-12
    # hand-checked tiny case: classes {0,1}; y = 0,0,1,1  p = 0,1,1,1 -> F1(0) = 2/3 (support 2), F1(1) = 4/5 (support 2)
    assert abs(T._weighted_f1([0, 0, 1, 1], [0, 1, 1, 1]) - (2 / 3 + 4 / 5) / 2) < 1e-12


# ------------------------------------------------------------------------------- MMBertDataset / device batch builder
def _np(v):
    return np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v)


def check_dataset_fixture(golden_dir, device):
    """msa_amd.dataset.MMBertDataset against the REAL reference class (tests/golden/make_golden.py:gen_dataset) on the same
    synthetic items under the same seeded `random`: every field of every item (values, dtypes, the label-1-for-the-true-pair
    quirk, the self-paired last item), every sentiment rule, then collate() of four items -- and the device batch builder's
    batch on ``device`` for the same indices from the same `random` state (fp32 features, otherwise identical: integer and
    mask tensors BIT-exact).  Runs on "cpu" here and on "cuda" from tests/test_train_gpu.py."""
    import random
    from tests.golden.dataset_features import synthetic_features
    from msa_amd.dataset import MMBertDataset, DeviceBatchBuilder
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    ncases = len({k.split("/")[0] for k in g.files})
    assert ncases == 8
    for ci in range(ncases):
        tag = f"c{ci}"
        ds_name, task, nl = (str(x) for x in g[tag + "/meta"])
        feats = synthetic_features(dataset=ds_name, seed=11 + ci)
        ds = MMBertDataset(None, feats, ds_name, task, int(nl))
        random.seed(100 + ci)
        paired = 0
        for i in range(len(ds)):
            item = ds[i]
            assert len(item) == 16 and item[14] == "seg%d" % i
            for f, v in enumerate(item[:14]):
                a = _np(v)
                a = a[:, :4] if f in (5, 10) else a
                ref = g[f"{tag}/item{i}/f{f}"]
                assert a.shape == ref.shape and np.array_equal(a, ref), (tag, i, f)
                if torch.is_tensor(v):
                    assert str(v.dtype) == str(g[f"{tag}/item{i}/f{f}_dtype"]), (tag, i, f, v.dtype)
            paired += int(item[6]) + int(item[11])
        assert int(ds[len(ds) - 1][6]) == 1                        # last item: always its own pair, label 1
        # collate of the build's items == collate of the reference's items
        random.seed(200 + ci)
        idx = (0, 3, 4, 2)
        batch = T.collate([ds[i] for i in idx])
        random.seed(200 + ci)
        dbatch = DeviceBatchBuilder(ds, device).batch(idx)
        for which, bt in (("collate", batch), ("device builder", dbatch)):
            for gname, grp in (("text", bt[0]), ("visual", bt[1]), ("speech", bt[2]), ("attention", bt[3])):
                for i, t in enumerate(grp):
                    ref = g[f"{tag}/batch/{gname}{i}"]
                    if which == "device builder":
                        assert t.device.type == torch.device(device).type, (tag, gname, i, t.device)
                    a = _np(t)
                    a = a[..., :4] if a.ndim == 3 else a
                    is_feature = gname in ("visual", "speech") and i == 1
                    if which == "device builder" and is_feature:
                        assert t.dtype == torch.float32 and np.array_equal(a, ref.astype(np.float32)), (tag, gname, i)
                    else:
                        assert np.array_equal(a, ref), (which, tag, gname, i)
                        assert str(t.dtype) == str(g[f"{tag}/batch/{gname}{i}_dtype"]), (which, tag, gname, i, t.dtype)
            assert list(bt[4]) == [str(x) for x in g[f"{tag}/batch/seg"]]


def test_dataset_items_and_batches_match_the_reference(golden_dir):
    check_dataset_fixture(golden_dir, "cpu")


def test_device_batch_feeds_pack_step_inputs():
    """The builder's batch goes through the trainer's packing unchanged (labels duplicated for the pair positions when
    P == T, REF:trainer.py:50,53) and yields the model's six keyword arguments."""
    from tests.golden.dataset_features import synthetic_features
    from msa_amd.dataset import MMBertDataset, DeviceBatchBuilder
    ds = MMBertDataset(None, synthetic_features(n_items=6, L=10, seed=3), "mosei", "sentiment", 1)
    b = DeviceBatchBuilder(ds, "cpu").batch([1, 4, 5])
    args = T.default_args(mlm=True, mlm_probability=0.3)
    kw = T.pack_step_inputs(b, args, "cpu", generator=torch.Generator().manual_seed(0))
    assert set(kw) == {"input_ids", "token_type_ids", "attention_mask", "masked_labels", "ap_label", "sentiment"}
    assert kw["input_ids"][1].shape == (3, 10, 35) and kw["input_ids"][2].shape == (3, 10, 74)
    assert kw["masked_labels"][1].shape == (3, 20) and kw["sentiment"].dtype == torch.float32
    # an epoch: every item exactly once over the rank shards, same permutation on every rank, short last batch kept
    bld = DeviceBatchBuilder(ds, "cpu")
    seen = []
    for rank in range(2):
        orig, got = bld.batch, []
        bld.batch = lambda idx, _o=orig, _g=got: (_g.extend(idx), _o(idx))[1]
        sizes = [kw["input_ids"][0].shape[0] for kw in bld.epoch(args, batch_size=2, generator=torch.Generator().manual_seed(9), rank=rank, world=2)]
        bld.batch = orig
        assert sizes == [2, 1]
        seen += got
    assert sorted(seen) == list(range(6))
    # n % world != 0: the permutation wraps around to a multiple of world (DistributedSampler's rule) -- every rank yields the SAME
    # batch sizes (a rank with one batch more would wait in an all-reduce nobody joins), every item is still visited
    ds7 = MMBertDataset(None, synthetic_features(n_items=7, L=10, seed=4), "mosei", "sentiment", 1)
    bld = DeviceBatchBuilder(ds7, "cpu")
    per_rank, seen = [], []
    for rank in range(2):
        orig, got = bld.batch, []
        bld.batch = lambda idx, _o=orig, _g=got: (_g.extend(idx), _o(idx))[1]
        per_rank.append([kw["input_ids"][0].shape[0] for kw in bld.epoch(args, batch_size=3, generator=torch.Generator().manual_seed(5), rank=rank, world=2)])
        bld.batch = orig
        seen += got
    assert per_rank[0] == per_rank[1] == [3, 1] and sorted(set(seen)) == list(range(7)) and len(seen) == 8
    with pytest.raises(ValueError, match="seeded identically"):
        next(bld.epoch(args, batch_size=3, rank=0, world=2))


def _epoch_worker(rank, world, port, q):
    """Two ranks walk an epoch of 7 items and all-reduce once per batch, the way DataParallel does per stepping micro-batch."""
    from tests.golden.dataset_features import synthetic_features
    from msa_amd.dataset import MMBertDataset, DeviceBatchBuilder
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = MMBertDataset(None, synthetic_features(n_items=7, L=10, seed=4), "mosei", "sentiment", 1)
    bld = DeviceBatchBuilder(ds, "cpu")
    args = T.default_args(mlm=True, mlm_probability=0.3)
    n, total = 0, torch.zeros(1)
    for kw in bld.epoch(args, batch_size=2, generator=torch.Generator().manual_seed(11), rank=rank, world=world):
        t = torch.tensor([float(kw["input_ids"][0].shape[0])])
        dist.all_reduce(t)                      # hangs (-> test timeout) if the ranks disagree on the number of batches
        total += t
        n += 1
    if rank == 0:
        q.put((n, float(total)))
    dist.barrier()
    dist.destroy_process_group()


def test_epoch_shards_have_equal_batch_counts_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 200
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, total = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n == 2 and total == 8.0              # 7 items -> 8 after wrapping, 4 per rank, batches of 2


# ------------------------------------------------------------------------------- from_pretrained: the checkpoints the reference loads
def _hf_checkpoint(cfg, seed=3):
    from oracle import mmbert_oracle as O
    sd = O.seeded_params(cfg, seed)
    hf = {k: v for k, v in sd.items() if k.startswith(("bert.embeddings", "bert.encoder", "bert.pooler", "cls.predictions", "cls.seq_relationship"))}
    hf["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    conf = dict(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                intermediate_size=cfg["intermediate"], max_position_embeddings=512, type_vocab_size=2, model_type="bert")
    return hf, conf


def _write_ckpt(d, conf, sd):
    import json
    d.mkdir()
    (d / "config.json").write_text(json.dumps(conf))
    torch.save(sd, d / "pytorch_model.bin")
    return str(d)


def test_from_pretrained_reads_legacy_and_prefixless_checkpoints_and_is_loud(tmp_path):
    """REF:train.py:70 loads ``bert-base/large-uncased``: those files carry the TF-era ``LayerNorm.gamma`` / ``LayerNorm.beta`` names
    (HF renames them on load: conversion_mapping.py:1274-1283, "legacy"), and a ``BertModel``-only checkpoint has no ``bert.`` prefix.
    Both must load to the SAME parameters as the current-named file; a missing encoder tensor or a key that matches nothing raises
    (round 2 loaded such files with every LayerNorm silently left at (1, 0))."""
    import warnings
    from msa_amd.model import MMBertForPretraining
    cfg = dict(hidden=64, layers=2, heads=2, intermediate=128, vocab=300, dataset="mosei")
    hf, conf = _hf_checkpoint(cfg)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                       # a complete BertForPreTraining file: nothing to warn about
        ref = MMBertForPretraining.from_pretrained(_write_ckpt(tmp_path / "current", conf, hf))
    assert ref.load_report == dict(missing=[], unexpected=[])
    want = {k: v for k, v in ref.state_dict().items() if k.startswith(("bert.", "cls.")) and not k.startswith("cls.align")}   # (align: fresh)
    for k, v in hf.items():
        assert torch.equal(want[k], v), k
    ln = [k for k in hf if "LayerNorm" in k]
    assert len(ln) == 2 * (1 + 2 * cfg["layers"] + 1) and any(float((hf[k] - (1.0 if k.endswith("weight") else 0.0)).abs().max()) > 0.05 for k in ln)

    # (1) legacy names
    legacy = {(k[:-6] + "gamma" if k.endswith("LayerNorm.weight") else k[:-4] + "beta" if k.endswith("LayerNorm.bias") else k): v for k, v in hf.items()}
    assert sum(k.endswith((".gamma", ".beta")) for k in legacy) == len(ln)
    m = MMBertForPretraining.from_pretrained(_write_ckpt(tmp_path / "legacy", conf, legacy))
    got = m.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    assert m.load_report == dict(missing=[], unexpected=[])

    # (2) BertModel-only file: no prefix, no heads -> same encoder, heads fresh WITH a warning that names them
    bare = {k[len("bert."):]: v for k, v in legacy.items() if k.startswith("bert.")}
    with pytest.warns(UserWarning, match="newly initialised.*cls.predictions.transform.dense.weight"):
        m = MMBertForPretraining.from_pretrained(_write_ckpt(tmp_path / "bare", conf, bare))
    got = m.state_dict()
    for k, v in want.items():
        if k.startswith("bert."):
            assert torch.equal(got[k], v), k
    assert torch.equal(got["cls.predictions.decoder.weight"], want["bert.embeddings.word_embeddings.weight"])      # still tied
    assert all(k.startswith("cls.") for k in m.load_report["missing"]) and m.load_report["unexpected"] == []

    # (3) one encoder tensor removed -> raises and names it
    broken = dict(hf)
    del broken["bert.encoder.layer.1.output.LayerNorm.weight"]
    with pytest.raises(ValueError, match="bert.encoder.layer.1.output.LayerNorm.weight"):
        MMBertForPretraining.from_pretrained(_write_ckpt(tmp_path / "broken", conf, broken))

    # (4) a key that matches nothing: warned about and recorded, as HF's from_pretrained (what REF:train.py:70 calls) does;
    # ignore_unexpected=False makes it an error
    extra = dict(hf, **{"bert.encoder.layer.0.attention.self.distance_embedding.weight": torch.zeros(3, 4)})
    path = _write_ckpt(tmp_path / "extra", conf, extra)
    with pytest.raises(ValueError, match="distance_embedding"):
        MMBertForPretraining.from_pretrained(path, ignore_unexpected=False)
    with pytest.warns(UserWarning, match="distance_embedding"):
        m = MMBertForPretraining.from_pretrained(path)
    assert m.load_report["unexpected"] == ["bert.encoder.layer.0.attention.self.distance_embedding.weight"]
    assert torch.equal(m.state_dict()["bert.pooler.dense.weight"], hf["bert.pooler.dense.weight"])

    # (5) decoder.bias as the only name of the tied prediction bias (BertForMaskedLM-style files) is not "missing"
    alias = {k: v for k, v in hf.items() if k != "cls.predictions.bias"}
    alias["cls.predictions.decoder.bias"] = hf["cls.predictions.bias"]
    m = MMBertForPretraining.from_pretrained(_write_ckpt(tmp_path / "alias", conf, alias))
    assert m.load_report["missing"] == [] and torch.equal(m.state_dict()["cls.predictions.bias"], hf["cls.predictions.bias"])


def test_inputs_embeds_alone_raises_like_the_reference_does():
    """REF:MMBertForPretraining.py:264 calls ``input_ids.long()`` whatever was passed: with ``inputs_embeds`` alone the reference raises
    (AttributeError) before any arithmetic; so does this model (NotImplementedError naming that line), and both refuse ids + embeds together
    with the reference's ValueError (REF :231)."""
    import pytest
    from msa_amd.model import MMBertConfig, MMBertModel
    m = MMBertModel(MMBertConfig(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128))
    m.set_joint_embeddings("mosei")
    x = torch.zeros(2, 5, 64)
    with pytest.raises(NotImplementedError, match="input_ids.long"):
        m(inputs_embeds=x)
    with pytest.raises(ValueError, match="both input_ids and inputs_embeds"):
        m(input_ids=torch.zeros(2, 5, dtype=torch.long), inputs_embeds=x)
    with pytest.raises(ValueError, match="either input_ids or inputs_embeds"):
        m()
