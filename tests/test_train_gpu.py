"""GPU tests of the training loop: the golden four-micro-batch trajectory of the REAL reference's
trainer.train_epoch (G8), the fused AdamW through the model, and 2-rank data parallelism."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import mmbert_oracle as O
from msa_amd.data import synthetic_batch, batch_to

DEV = "cuda"
CFG = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)


def build(cfg=CFG, dropout=0.0):
    from msa_amd.model import MMBertConfig, MMBertForPretraining
    c = MMBertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                     intermediate_size=cfg["intermediate"], hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    m = MMBertForPretraining(c)
    m.bert.set_joint_embeddings(cfg["dataset"])
    m.bert.jointEmbeddings.dropout_prob = dropout if dropout == 0.0 else 0.5
    m.load_state_dict(O.seeded_params(cfg), strict=False)
    return m.to(DEV)


def test_train_epoch_matches_reference_trajectory(golden_dir):
    """G8: same items, same order, mlm off, dropout 0, torch-AdamW semantics + warm-up (warmup == total):
    per-micro-batch losses and the parameter update after the 2nd optimizer step."""
    from msa_amd import trainer as T
    g = np.load(os.path.join(golden_dir, "train4.npz"))
    order = [int(i) for i in g["order"]]
    Tn = len(g["item0_text"])
    items = []
    for i in range(8):
        te = [int(x) for x in g[f"item{i}_text"]]
        tti, vti = torch.zeros(Tn), torch.cat((torch.zeros(Tn), torch.ones(Tn)))
        sent = float(g[f"item{i}_sent"])
        items.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(sent), te, g[f"item{i}_visual"], torch.tensor(int(g[f"item{i}_ap"][0])), vti,
                      torch.tensor(sent), te, g[f"item{i}_speech"], torch.tensor(int(g[f"item{i}_ap"][1])), vti, torch.tensor(sent), "s", "r"))
    data = [items[i] for i in order]                        # the reference's RandomSampler order, replayed sequentially
    m = build()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    args = T.default_args(train_batch_size=2, mlm=False, learning_rate=float(g["lr"]), warmup_proportion=1.0)
    opt, sched = T.build_optimizer(m, args, int(g["n_opt_steps"]), mode="torch")
    losses = []
    orig = m.forward

    def rec(*a, **k):
        out = orig(*a, **k)
        losses.append([float(out[0][i].detach()) for i in (0, 4, 5, 6)])
        return out
    m.forward = rec
    ret = T.train_epoch(args, m, data, opt, sched, device=DEV, shuffle=False)
    ref = g["losses"]
    # step 0/1 run on identical weights (first optimizer step has lr 0); steps 2/3 after one real update
    for s in range(4):
        for j in range(4):
            assert abs(losses[s][j] - ref[s][j]) < 4e-3 * max(1.0, abs(ref[s][j])), (s, j, losses[s], ref[s])
    assert abs(ret[0] - g["ret"][0]) < 4e-3 * g["ret"][0] and abs(ret[5] - g["ret"][5]) < 1e-2 * g["ret"][5]
    assert abs(ret[4] - g["ret"][4]) < 1e-2 * abs(g["ret"][4]) + 1e-4          # LAST step's ap_loss / steps (REF:trainer.py:101)
    assert opt._steps == 2                                                       # the `&` quirk: steps after micro-batch 2 and 4
    big = ("bert.encoder", "bert.embeddings", "cls.predictions", "bert.jointEmbeddings.W")
    for n, p in m.named_parameters():
        dn = float(g["dnorm/" + n])
        delta = (p.detach() - before[n]).float().cpu()
        if dn == 0.0:
            assert float(delta.abs().max()) == 0.0, n                           # never-differentiated parameters stay put
            continue
        if n.startswith(big) and "key.bias" not in n:
            assert abs(float(delta.norm()) - dn) < 0.05 * dn, (n, float(delta.norm()), dn)
    m.forward = orig


def test_fused_adamw_hf_mode_matches_oracle_on_model_grads():
    from msa_amd import trainer as T
    m = build()
    batch = batch_to(synthetic_batch(2, 16, 16, 16, vocab=CFG["vocab"], seed=3), DEV)
    m.eval()
    out, _ = m(**batch)
    out[0].mean().backward()
    flat = m._flat
    p0, g0 = flat.params.clone().cpu(), flat.grads.clone().cpu()
    opt, sched = T.build_optimizer(m, T.default_args(learning_rate=1e-3), 1)
    for grp in opt.param_groups:
        grp["lr"] = 1e-3
    opt.step()
    torch.cuda.synchronize()
    # fused zero_grad: everything zero except the LAZY blocks (the dense weights and the tied table: 3/4 of the parameters), whose old
    # gradient is dropped -- flagged stale, to be overwritten by the next backward -- and zero-filled only on demand
    assert not flat.grads_dirty and flat.stale == set(flat.lazy) and len(flat.lazy) == 4 * CFG["layers"] + 1
    keep = torch.ones(flat.total, dtype=torch.bool)
    for gv, _ in flat.lazy.values():
        o = (gv.data_ptr() - flat.grads.data_ptr()) // 4
        keep[o:o + gv.numel()] = False
    assert float(flat.grads[keep.to(DEV)].abs().max()) == 0.0 and float(flat.grads.abs().max()) > 0.0
    flat.settle()
    assert float(flat.grads.abs().max()) == 0.0 and not flat.stale
    for n, p in m.named_parameters():
        o, k = flat.offset[n], flat.numel[n]
        pr, gr = p0[o:o + k].clone(), g0[o:o + k]
        if any(n.startswith(f) for f in ("bert.jointEmbeddings.W_c", "cls.seq_relationship")):
            assert torch.equal(p.detach().cpu().reshape(-1), pr), n               # frozen: skipped like grad=None params
            continue
        O.adamw_step(pr, gr, torch.zeros(k), torch.zeros(k), 1, 1e-3, 0.01 if O.decays(n) else 0.0, mode="hf")
        assert torch.allclose(p.detach().cpu().reshape(-1), pr, rtol=1e-5, atol=1e-7), n
        assert torch.allclose(flat.half[o:o + k].float().cpu(), pr, rtol=1e-2, atol=1e-4), n   # bf16 working copy refreshed
    # the transposed copies follow too: a second forward must see the new weights everywhere
    out2, _ = m(**batch)
    assert abs(float(out2[0]) - float(out[0])) > 1e-4


def test_lazy_zero_of_the_dense_weight_gradients_is_invisible():
    """Round 4: AdamW's fused zero_grad leaves the dense weights' gradients (and the tied table's) alone; the next backward's
    weight-gradient launches OVERWRITE them (torch's zero_grad(set_to_none=True): the old gradient is dropped, never read).  With the
    learning rate at 0 the parameters never move, so every gradient below has a reference computed from a zero-filled buffer:
    (1) the first backward after a step overwrites; (2) a second micro-batch accumulates; (3) between zero_grad() and the next backward
    the dropped gradients read None, after it they are the flat views again; (4) a batch without any MLM label leaves the tied table's
    gradient = the embedding rows alone (zero-filled on demand); (5) after a backward that never reaches the encoder (heads only) the
    optimizer sees zeros there: its first moments just decay."""
    from msa_amd import trainer as T
    from msa_amd import model as MM
    batches = [batch_to(synthetic_batch(2, 16, 40, 24, vocab=CFG["vocab"], seed=30 + i), DEV) for i in range(3)]
    nolabel = dict(batches[2]); nolabel["masked_labels"] = tuple(torch.full_like(x, -100) for x in nolabel["masked_labels"])
    m = build()
    m.eval()
    opt, sched = T.build_optimizer(m, T.default_args(learning_rate=1e-3), 10)
    for grp in opt.param_groups:
        grp["lr"] = 0.0
    def fb(b):
        out, _ = m(**b); out[0].mean().backward()
    m._ensure_ready(torch.device(DEV, 0))
    flat = m._flat
    ref = {}
    for k, b in (("b0", batches[0]), ("b1", batches[1]), ("nolabel", nolabel)):       # references: zero-filled start, nothing stale
        flat.grads.zero_(); fb(b); ref[k] = flat.grads.clone()
    close = lambda a, b_: float((a - b_).norm() / b_.norm()) < 1e-5                   # (biases / LayerNorm sums go through fp32 atomics)
    p0 = flat.params.clone()
    opt.step(); opt.zero_grad()
    assert torch.equal(flat.params, p0)
    wq = m.bert.encoder.layer[0].attention.self.query.weight
    assert flat.stale == set(flat.lazy) and wq.grad is None and m.bert.embeddings.word_embeddings.weight.grad is None
    assert m.bert.encoder.layer[0].attention.self.query.bias.grad is not None
    assert float(flat.grads.abs().max()) > 0.0                                       # the dropped gradients are still in the buffer
    fb(batches[0])                                                                    # (1) overwrites them
    assert not flat.stale and close(flat.grads, ref["b0"])
    assert wq.grad is not None and wq.grad.data_ptr() == flat.grads.data_ptr() + 4 * flat.offset["bert.encoder.layer.0.attention.self.query.weight"]
    fb(batches[1])                                                                    # (2) accumulates
    assert close(flat.grads, ref["b0"] + ref["b1"])
    opt.step(); opt.zero_grad()
    fb(nolabel)                                                                       # (4)
    assert not flat.stale and close(flat.grads, ref["nolabel"])
    o, k = flat.offset["bert.embeddings.word_embeddings.weight"], flat.numel["bert.embeddings.word_embeddings.weight"]
    assert float(flat.grads[o:o + k].abs().max()) > 0.0 and int((flat.grads[o:o + k].view(-1, CFG["hidden"]).abs().sum(1) > 0).sum()) < 200
    opt.step(); opt.zero_grad()
    first = torch.randn(6, CFG["hidden"], device=DEV, generator=torch.Generator(DEV).manual_seed(7)).requires_grad_(True)
    hl, *_ = MM._HeadsFn.apply(first, m, torch.tensor([0, 1, 1, 0], device=DEV), torch.tensor([0.5, -1.0], device=DEV))
    hl.backward()                                                                     # (5) the trunk's backward never runs
    lo = flat.offset["bert.encoder.layer.1.output.dense.weight"]
    m_before = opt._m[lo:lo + 4096].clone()
    assert float(m_before.abs().max()) > 0.0 and flat.stale
    opt.step()
    assert torch.equal(opt._m[lo:lo + 4096], m_before * opt.betas[0])
    # ... and the zero-filling optimizer (lazy_zero = False before the first step) leaves nothing stale
    m2 = build(); m2.eval()
    opt2, _ = T.build_optimizer(m2, T.default_args(learning_rate=1e-3), 10)
    opt2.lazy_zero = False
    out, _ = m2(**batches[0]); out[0].mean().backward(); opt2.step()
    assert not m2._flat.stale and float(m2._flat.grads.abs().max()) == 0.0


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)         # 1-GPU box: both ranks share cuda:0, gloo moves the bytes
    from msa_amd import parallel
    from msa_amd import trainer as T
    torch.cuda.set_device(0)
    m = build()
    if rank == 1:                                                         # ranks start different: broadcast must fix it
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    opt, sched = T.build_optimizer(m, T.default_args(learning_rate=1e-3), 4)
    dp = parallel.DataParallel(m, opt, bucket_mb=0.25)
    m.eval()
    batch = batch_to(synthetic_batch(2, 16, 40, 24, vocab=CFG["vocab"], seed=10 + rank), DEV)
    out, _ = m(**batch)
    out[0].mean().backward()
    n_calls = dp.bucketer.calls
    dp.finish_backward()
    torch.cuda.synchronize()
    if rank == 0:
        q.put(dict(grads=m._flat.grads.cpu(), loss=float(out[0]), scale=opt.grad_scale, calls=n_calls))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_equals_mean_of_single_rank_grads():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:                                           # a free port, not a guess: reruns on one box must not collide
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res["scale"] == 0.5 and res["calls"] >= 1                      # averaging folded into AdamW; overlap buckets were issued
    # single-process reference: same two shards, gradients summed
    m = build()
    m.eval()
    m._ensure_ready(torch.device(DEV, 0))
    total = None
    for r in range(2):
        m._flat.grads.zero_()
        out, _ = m(**batch_to(synthetic_batch(2, 16, 40, 24, vocab=CFG["vocab"], seed=10 + r), DEV))
        out[0].mean().backward()
        total = m._flat.grads.clone() if total is None else total + m._flat.grads
    got, ref = res["grads"], total.cpu()
    err = float((got - ref).norm() / ref.norm())
    assert err < 2e-3, err               # same kernels, same data: only fp32 atomic-order noise


def test_eval_epoch_and_train_loop_save_rule(tmp_path):
    """eval_epoch (no_grad, MLM masking at eval, the reference's 8-tuple), the epoch loop's
    best-on-test-accuracy checkpoint (a state dict with the reference's keys that loads back) and the early stop."""
    from msa_amd import trainer as T
    from msa_amd.trainer import build_optimizer, default_args
    m = build(dropout=0.1)
    m.manual_seed(3)
    args = default_args(train_batch_size=4, val_batch_size=4, learning_rate=1e-3, n_epochs=4, num_labels=7)
    opt, sched = build_optimizer(m, args, num_train_optimization_steps=16)

    def batches(split, epoch):
        base = {"train": 100, "val": 200, "test": 300}[split]
        return [batch_to(synthetic_batch(4, 50, 50, 50, vocab=CFG["vocab"], seed=base + i), DEV) for i in range(2)]

    ev = T.eval_epoch(args, m, None, batches=batches("val", 0))
    assert len(ev) == 8 and np.isfinite(ev[0]) and ev[1] == ev[2] == ev[3] == 0.0 and ev[6].shape == (8, 1) and ev[7].shape == (8,)
    assert not m.training
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    best = T.train(args, m, None, None, None, opt, sched, epoch_batches=batches, save_root=str(tmp_path / "model_save"),
                   numpy_root=str(tmp_path / "numpy_save"), patience_limit=2)
    assert len(best["history"]) >= 1 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["valid_loss"]) for h in best["history"])
    assert any(float((m.state_dict()[k].float() - before[k].float()).abs().max()) > 0 for k in before)        # it trained
    if best["path"] is not None:
        sd = torch.load(best["path"], map_location="cpu")
        assert set(sd.keys()) == set(m.state_dict().keys())
        m2 = build()
        m2.load_state_dict(sd)
    if len(best["history"]) < int(args.n_epochs):                    # stopped early: the prediction dump exists
        dumps = list((tmp_path / "numpy_save").glob("*/predict.npy"))
        assert len(dumps) == 1


def test_device_batch_builder_drives_train_epoch():
    """items resident in HBM, batches gathered on the device, fed to train_epoch -- the loss falls over a
    few epochs on a 24-item synthetic set and nothing but the pair draws happens on the host."""
    from tests.golden.dataset_features import synthetic_features
    from msa_amd.dataset import MMBertDataset, DeviceBatchBuilder
    from msa_amd import trainer as T
    ds = MMBertDataset(None, synthetic_features(n_items=24, L=16, seed=4), "mosei", "sentiment", 1)
    bld = DeviceBatchBuilder(ds, DEV)
    assert bld.visual.is_cuda and bld.visual.dtype == torch.float32 and bld.ids.shape == (24, 16)
    m = build()
    m.train()
    args = T.default_args(train_batch_size=8, learning_rate=2e-3, mlm=True)
    opt, sched = T.build_optimizer(m, args, 40)
    first = last = None
    for ep in range(6):
        ret = T.train_epoch(args, m, None, opt, sched, device=DEV, quirk_step=False,
                            batches=bld.epoch(args, generator=torch.Generator().manual_seed(ep)))
        first = ret[0] if first is None else first
        last = ret[0]
    assert np.isfinite(last) and last < first, (first, last)


def test_async_prologue_with_changing_batch_shapes():
    """The prologue's two persistent output sets are sized on demand: batches of different pair lengths and batch sizes in turn
    (each first use of a larger shape re-allocates a set behind the compute stream) give the losses of the synchronous prologue."""
    from msa_amd import trainer as T
    cfg = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=2048, dataset="mosei", alpha=1.0, beta=1.0)
    shapes = [(3, 20, 60, 40), (5, 24, 200, 170), (2, 12, 30, 30), (5, 24, 200, 170), (6, 30, 260, 90), (3, 20, 60, 40)]
    pool = [batch_to(synthetic_batch(b, t, pv, pa, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i, (b, t, pv, pa) in enumerate(shapes)]
    got = {}
    for async_ in (False, True):
        m = build(cfg)
        m.eval()                                                           # no dropout, no update: every step is a pure function of its batch
        m.async_prologue = async_
        losses = []
        for b in pool:
            out, _ = m(**b)
            out[0].mean().backward()
            losses.append(out[0].detach())
        torch.cuda.synchronize()
        got[async_] = torch.stack(losses).double().cpu()
    assert bool(torch.isfinite(got[True]).all())
    assert torch.allclose(got[True], got[False], rtol=3e-6, atol=1e-7), (got[True], got[False])


def test_batches_built_on_the_input_stream_train_the_same():
    """trainer.on_input_stream: DeviceBatchBuilder's gathers and the MLM masking kernel run on model.input_stream, the step
    prologue follows them there (model.async_prologue), the compute stream picks the batch up through an event -- the same epochs
    as with everything on the compute stream (same seeds: the first epoch's losses agree to rounding noise, the loss falls)."""
    from tests.golden.dataset_features import synthetic_features
    from msa_amd.dataset import MMBertDataset, DeviceBatchBuilder
    from msa_amd import trainer as T
    ds = MMBertDataset(None, synthetic_features(n_items=24, L=16, seed=4), "mosei", "sentiment", 1)
    got = {}
    for side in (False, True):
        bld = DeviceBatchBuilder(ds, DEV)
        m = build()
        m.train()
        m.manual_seed(3)
        torch.manual_seed(100)                                            # governs the masking kernel too (CUDA generator seed + offset)
        import random
        random.seed(9)                                                    # the pair draws follow the reference: Python's ``random``
        args = T.default_args(train_batch_size=8, learning_rate=2e-3, mlm=True)
        opt, sched = T.build_optimizer(m, args, 40)
        rets = []
        for ep in range(4):
            batches = bld.epoch(args, generator=torch.Generator().manual_seed(ep))
            rets.append(T.train_epoch(args, m, None, opt, sched, device=DEV, quirk_step=False,
                                      batches=T.on_input_stream(m, batches) if side else batches))
        # what follows an input-stream epoch in the same process is built on the COMPUTE stream: an evaluation pass
        # whose batches (gathers + masking kernel) are still being written when forward() is called must wait for them -- the tag
        # of on_input_stream is per batch and gone, the model's own switch was never touched
        assert not getattr(m, "async_prologue", False) and m.__dict__.get("_input_stream_batch") is None
        args.val_batch_size = 8
        rs, cs = random.getstate(), torch.cuda.get_rng_state()
        ev = T.eval_epoch(args, m, None, device=DEV, batches=bld.epoch(args, generator=torch.Generator().manual_seed(77)))
        torch.cuda.synchronize()
        random.setstate(rs)                                               # ... and once more from a drained GPU: the same numbers
        torch.cuda.set_rng_state(cs)                                      # (the masking kernel's seed is the CUDA generator's state)
        ev2 = T.eval_epoch(args, m, None, device=DEV, batches=bld.epoch(args, generator=torch.Generator().manual_seed(77)))
        assert all(abs(a - b) <= 1e-5 * abs(b) + 1e-7 for a, b in zip(ev[:6], ev2[:6])), (ev[:6], ev2[:6])
        assert np.array_equal(ev[7], ev2[7]) and np.allclose(ev[6], ev2[6], rtol=1e-4, atol=1e-5)
        got[side] = rets + [ev]
    assert all(np.isfinite(r[0]) for r in got[True]) and got[True][-2][0] < got[True][0][0]
    assert abs(got[True][0][0] - got[False][0][0]) <= 2e-3 * abs(got[False][0][0]), (got[True][0], got[False][0])
    # the evaluation pass after the input-stream epochs: finite, and the same numbers as after the compute-stream epochs up to what
    # the two training runs differ by (same seeds, same items)
    ea, eb = got[True][-1], got[False][-1]
    assert np.isfinite(ea[0]) and abs(ea[0] - eb[0]) <= 0.1 * abs(eb[0]), (ea[:6], eb[:6])
    assert np.array_equal(ea[7], eb[7])                                   # same items in the same order (labels)


# ================================================================================================ round 2
def test_device_batch_builder_is_bit_exact_on_cuda(golden_dir):
    """SURVEY S8(f) row 2 on the device: DeviceBatchBuilder("cuda").batch() against the REAL reference's collate() output for the
    same items and the same `random` state (tests/golden/dataset.npz) -- ids, labels, token types, masks and sentiments BIT-exact
    with the reference's dtypes; features equal to the reference's float64 values rounded to fp32 (the one declared difference)."""
    from tests.test_host_cpu import check_dataset_fixture
    check_dataset_fixture(golden_dir, DEV)


def _eval_items(g, n):
    Tn = len(g["item0_text"])
    items = []
    for i in range(n):
        te = [int(x) for x in g[f"item{i}_text"]]
        tti, vti = torch.zeros(Tn), torch.cat((torch.zeros(Tn), torch.ones(Tn)))
        sent = float(g[f"item{i}_sent"])
        items.append((torch.tensor(te), torch.tensor(0), tti, torch.tensor(sent), te, g[f"item{i}_visual"], torch.tensor(int(g[f"item{i}_ap"][0])), vti,
                      torch.tensor(sent), te, g[f"item{i}_speech"], torch.tensor(int(g[f"item{i}_ap"][1])), vti, torch.tensor(sent), "s", "r"))
    return items


def test_eval_epoch_matches_reference_golden(golden_dir):
    """G9: the REAL reference's trainer.eval_epoch (REF:trainer.py:103-194) on six items, val_batch_size 4 (one full and one short
    batch), mlm off, eval mode -- replayed in the order its RandomSampler drew: the 8-tuple (dev loss, three zero modality losses,
    LAST batch's ap_loss / steps, label loss, predictions [N,1], labels [N]), the per-batch losses, and what test_MSE_score_model
    makes of the predictions.  Stated: losses 4e-3 relative (bf16 path), predictions 2e-2 abs, labels exact, accuracy / F1 exact
    when no prediction lies within 2e-2 of the sign boundary (asserted on the fixture)."""
    from msa_amd import trainer as T
    g = np.load(os.path.join(golden_dir, "eval6.npz"))
    order = [int(i) for i in g["order"]]
    items = _eval_items(g, 6)
    data = [items[i] for i in order]
    m = build()
    args = T.default_args(val_batch_size=int(g["val_batch_size"]), mlm=False)
    losses = []
    orig = m.forward

    def rec(*a, **k):
        out = orig(*a, **k)
        losses.append([float(out[0][i]) for i in (0, 4, 5, 6)])
        return out
    m.forward = rec

    class Seq(torch.utils.data.Dataset):                       # replay: eval_epoch's own RandomSampler must see the recorded order
        def __len__(self):
            return len(data)

        def __getitem__(self, i):
            return data[i]
    from torch.utils.data import DataLoader, SequentialSampler
    loader = DataLoader(Seq(), sampler=SequentialSampler(Seq()), batch_size=args.val_batch_size, collate_fn=T.collate)
    ret = T.eval_epoch(args, m, None, device=DEV, batches=(T.pack_step_inputs(b, args, DEV) for b in loader))
    m.forward = orig
    assert not m.training and len(ret) == 8
    ref = g["losses"]
    assert len(losses) == len(ref) == 2
    for s_ in range(2):
        for j in range(4):
            assert abs(losses[s_][j] - ref[s_][j]) < 4e-3 * max(1.0, abs(ref[s_][j])), (s_, j, losses[s_], ref[s_])
    r6 = g["ret6"]
    assert abs(ret[0] - r6[0]) < 4e-3 * r6[0] and ret[1] == ret[2] == ret[3] == 0.0 == r6[1] == r6[2] == r6[3]
    assert abs(ret[4] - r6[4]) < 4e-3 * abs(r6[4]) + 1e-4              # LAST batch's ap_loss / number of steps (REF:trainer.py:194)
    assert abs(ret[5] - r6[5]) < 4e-3 * r6[5]
    assert ret[6].shape == g["preds"].shape == (6, 1) and ret[7].shape == g["labels"].shape == (6,)
    assert np.abs(ret[6] - g["preds"]).max() < 2e-2
    np.testing.assert_allclose(ret[7], g["labels"], rtol=0, atol=0)
    assert np.abs(g["preds"]).min() > 2e-2                               # no prediction near the sign boundary: the metrics must agree exactly
    acc, mae, f1 = T.test_MSE_score_model(ret[6], ret[7])
    assert acc == g["mse_scores"][0] and abs(f1 - g["mse_scores"][2]) < 1e-12 and abs(mae - g["mse_scores"][1]) < 2e-2


def test_mosei_shape_trainer_loop_50_steps():
    """BASELINE configs[4]: the full pretraining loop through msa_amd.trainer.train_epoch (the reference's trainer.py counterpart)
    at the MOSEI shape -- 12-layer d=768, T=50, speech 74-dim x 500, visual 35-dim x 500, batch 16, train mode (all dropouts),
    AdamW (HF mode) + the reference's warm-up schedule, the `&` stepping quirk: 52 micro-batches = 26 optimizer steps.  Asserts
    the size-independent properties: every loss finite, the joint loss falls, the step-count rule, parameters the reference never
    differentiates stay bit-identical while every other parameter moves, and the bf16 working copy follows the fp32 masters."""
    from msa_amd import trainer as T
    cfg = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    m = build(cfg, dropout=0.1)
    m.train()
    m.manual_seed(7)
    n_micro = 52
    args = T.default_args(train_batch_size=16, learning_rate=1e-4, mlm=True)
    opt, sched = T.build_optimizer(m, args, n_micro // 2, mode="hf")
    pool = [batch_to(synthetic_batch(16, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=300 + i), DEV) for i in range(4)]
    assert pool[0]["input_ids"][1].shape == (16, 500, 35) and pool[0]["input_ids"][2].shape == (16, 500, 74)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    losses = []
    orig = m.forward

    def rec(*a, **k):
        out = orig(*a, **k)
        losses.append(out[0][0].detach())
        return out
    m.forward = rec
    ret = T.train_epoch(args, m, None, opt, sched, device=DEV, batches=(pool[i % 4] for i in range(n_micro)))
    m.forward = orig
    torch.cuda.synchronize()
    ls = torch.stack(losses).float().cpu().numpy()
    assert len(ls) == n_micro and np.isfinite(ls).all() and all(np.isfinite(r) for r in ret)
    assert opt._steps == n_micro // 2 and sched.last_step == n_micro // 2           # (step + 1) & 1 == 0: every second micro-batch
    assert ls[-8:].mean() < ls[:8].mean() - 0.5, (ls[:8].mean(), ls[-8:].mean())    # MLM loss leaves ln(30522) = 10.3 quickly
    frozen = ("bert.jointEmbeddings.W_cv.", "bert.jointEmbeddings.W_cs.", "cls.seq_relationship.")
    moved = 0
    for n, p in m.named_parameters():
        assert bool(torch.isfinite(p).all()), n
        if n.startswith(frozen):
            assert torch.equal(p.detach(), before[n]), n                              # never differentiated in the reference (App. B-9)
        else:
            moved += int(not torch.equal(p.detach(), before[n]))
    assert moved >= len(before) - 6 - 12                                              # (key biases have a zero true gradient: may or may not move)
    f = m._flat
    assert torch.equal(f.half.float(), f.params.to(torch.bfloat16).float())           # AdamW refreshed the bf16 copy it computes with


def test_trained_state_gradients_match_oracle_without_calibrator():
    from msa_amd import ops as _ops
    was = _ops.deterministic()
    try:
        _trained_state_gradients()
    finally:
        _ops.set_deterministic(was)                              # (process-wide switch of the library)


def _trained_state_gradients():
    """Round 4: every other oracle comparison runs at INITIALISATION weights (seeded N(0, 0.02)), where nce = 3 ln B exactly and the CPC
    gradients are residues of cancelling O(1) terms, softmax is flat and LayerNorm is (1, 0).  Here the comparison runs at a TRAINED
    state: 52 micro-batches = 26 optimizer steps of msa_amd.trainer.train_epoch (train mode, all dropouts, HF-AdamW, the reference's
    warm-up schedule and `&` stepping quirk: REF:trainer.py:40,66,83,96) at two layers of BASELINE configs[1] (d = 768, T = 50, A = V =
    500, batch 8), learning rate 5e-4 so that 26 steps move every weight by up to ~1/3 of its initial spread; then the fp32 MASTER
    weights are copied into the CPU oracle and ONE eval-mode forward + backward on a batch the training never saw is compared:
    the four losses (3e-3), regression logits, and EVERY parameter gradient with no calibrator -- encoder / embedding / MLM-head
    gradients within 6 % relative L2 error and cosine >= 0.995 (measured 1.4 % median, 1.7 % max), pooler / gate / classifier gradients
    within 8 % (measured <= 2.3 %), the alignment head within 12 %, the three CPC projections (REF:MMBertEmbedding.py:21-32) by
    their conditioning (see the assertion).
    Deviations -> gpurun_out/parity_trained.json (-> profiles/r4_parity_trained.json)."""
    import json
    from msa_amd import trainer as T
    cfg = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    B = 8
    m = build(cfg, dropout=0.1)
    m.train()
    m.manual_seed(3)
    # Deterministic mode (round 5): with fp32 atomics the 26-step trajectory is chaotic (see the alignment head's note below) and this
    # test met a different trained state in every run -- one run in ~40 landed outside a bound.  Ordered sums make it ONE trajectory.
    m.deterministic = True
    n_micro = 52
    args = T.default_args(train_batch_size=B, learning_rate=5e-4, mlm=True)
    opt, sched = T.build_optimizer(m, args, n_micro // 2, mode="hf")
    pool = [batch_to(synthetic_batch(B, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i in range(4)]
    init = {n: p.detach().float().cpu().clone() for n, p in m.named_parameters()}
    ret = T.train_epoch(args, m, None, opt, sched, device=DEV, batches=(pool[i % 4] for i in range(n_micro)))
    torch.cuda.synchronize()
    assert opt._steps == n_micro // 2 and all(np.isfinite(r) for r in ret)
    # the state really left initialisation: LayerNorm scales off 1, dense weights moved by a visible fraction of their 0.02 spread
    sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    moved = float((sd["bert.encoder.layer.0.intermediate.dense.weight"] - init["bert.encoder.layer.0.intermediate.dense.weight"]).abs().mean())
    ln_off = float((sd["bert.encoder.layer.1.output.LayerNorm.weight"] - 1.0).abs().mean())
    assert moved > 1e-3 and ln_off > 1e-3, (moved, ln_off)
    # eval-mode comparison on an unseen batch
    m.eval()
    opt.zero_grad()
    batch = synthetic_batch(B, 50, 500, 500, dataset="mosei", vocab=cfg["vocab"], seed=977)
    out, logits = m(**batch_to(batch, DEV))
    out[0].mean().backward()
    torch.cuda.synchronize()
    ours = {n: q.grad.float().cpu() for n, q in m.named_parameters()}
    p = {k: sd[k].clone().requires_grad_(True) for k in O.seeded_params(cfg)}
    ocfg = dict(cfg, hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    oout, ologits = O.pretraining_forward(p, ocfg, **batch)
    oout[0].mean().backward()
    rep = {"moved_mean_abs": moved, "layernorm_off_one": ln_off, "losses": {}, "grads": {}}
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-6)
    for i, name in ((0, "joint"), (4, "ap"), (5, "label"), (6, "nce")):
        rep["losses"][name] = dict(ours=float(out[i].detach()), oracle=float(oout[i].detach()), rel=rel(out[i].detach(), oout[i].detach()))
    rep["nce_minus_3lnB"] = float(oout[6]) - 3.0 * float(np.log(B))
    rep["logits_max_abs"] = float((logits.float().cpu() - ologits.detach()).abs().max())
    for n, g in ours.items():
        og = p[n].grad
        if og is None or float(og.abs().sum()) == 0.0:
            assert float(g.abs().sum()) == 0.0, n
            continue
        if "attention.self.key.bias" in n:                       # true gradient 0 (softmax is shift invariant)
            continue
        rep["grads"][n] = dict(rel_err=float((g - og).norm() / og.norm()), norm=float(og.norm()),
                               cosine=float(torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1))))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_trained.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    for name, r in rep["losses"].items():
        assert r["rel"] < 3e-3, (name, r)
    assert rep["logits_max_abs"] < 3e-2
    pool_norm = rep["grads"]["bert.pooler.dense.weight"]["norm"]
    for n, r in rep["grads"].items():
        if n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")):
            assert r["rel_err"] < 0.06 and r["cosine"] > 0.995, (n, r)          # measured (round 4): median 1.4 %, max 1.7 %, cosine >= 0.9998
        elif n.startswith("cpc_"):
            # Measured at this trained state: nce - 3 ln B = 1e-5 -- 26 optimizer steps do NOT move CPC off its symmetric point (the [CLS]
            # rows of a batch stay nearly collinear, so the InfoNCE softmax stays uniform) -- and |d nce / d W| = 1e-4 .. 1e-5 against
            # 1 .. 30 for every other head: a difference of nearly equal unit vectors, on which one bf16 rounding of the pooled rows is
            # an absolute error of the gradient's own size on both sides (21-31 % here, 3.6e-5 absolute).  A CPC gradient within two
            # orders of the pooler's is held to the heads' 8 %; an ill-conditioned one to 50 % and 1e-4 absolute.  The kernel's own
            # arithmetic is pinned where it is well conditioned: test_fused_heads_match_oracle_in_fp32 (2e-3, random [CLS] rows).
            if r["norm"] >= 1e-2 * pool_norm:
                assert r["rel_err"] < 0.08, (n, r)
            else:
                assert r["rel_err"] < 0.5 and r["rel_err"] * r["norm"] < 1e-4, (n, r)
        else:
            # pooler, gate (attn, vt / vv / vs), classifier1_1 / 1_2 measured <= 2.3 %; the alignment head's weight 7.9 % (its
            # gradient sums +-0.5 / B residuals over nearly identical [CLS] rows: norm 0.23 against 12-34 for the others).  Late round 4:
            # 26 steps at lr 5e-4 do not reach ONE state -- the trajectory is chaotic in the fp32 atomics' order: seven runs of this test
            # ended with |d align| = 0.085 ... 3.1 and a joint loss of 5.6 ... 11.7 on the unseen batch (tools/dbg_trained.py) -- while
            # the ABSOLUTE error of the alignment gradient stayed at 0.014-0.018 in all of them (0.6 % ... 19.6 % relative): one bf16
            # rounding of the [CLS] rows, as for CPC above.  So: 12 % relative or 0.03 absolute (1e-3 of the pooler's gradient norm).
            if n.startswith("cls.align"):
                assert r["rel_err"] < 0.12 or r["rel_err"] * r["norm"] < 0.03, (n, r)
            else:
                assert r["rel_err"] < 0.08, (n, r)


def _nccl_world1_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from msa_amd import ops, parallel
    from msa_amd import trainer as T
    try:
        rank, local, world = parallel.init_from_env(force=True)                       # RCCL with world_size 1, before any GPU call
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        cfg = dict(hidden=256, layers=3, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
        batch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=31), "cuda")
        grads = {}
        for use_dp in (False, True):
            m = build(cfg, dropout=0.1)
            m.train()
            m.manual_seed(5)
            args = T.default_args(train_batch_size=4, learning_rate=1e-3)
            opt, sched = T.build_optimizer(m, args, 4)
            dp = parallel.DataParallel(m, opt, bucket_mb=0.5, force_dynamic_queue=True) if use_dp else None
            out, _ = m(**batch)
            out[0].mean().backward()
            if dp is not None:
                calls = dp.bucketer.calls
                dp.finish_backward()
                assert calls >= 2, calls                                              # layer buckets were reduced DURING backward
                assert ops.dynamic_tile_queue                                         # the persistent GEMM ran on its device-side queue
            torch.cuda.synchronize()
            grads[use_dp] = m._flat.grads.clone()
            ops.dynamic_tile_queue = False
        a, b = grads[True], grads[False]
        err = float((a - b).abs().max() / b.abs().max())
        # round 4: the bf16 WIRE path (all_to_all_single + fp32 sum in rank order + all_gather_into_tensor) executed over RCCL -- at
        # world 1 GradBucketer short-circuits it to a plain all-reduce, so the test-only switch force_wire_path sends it through
        m = build(cfg, dropout=0.1)
        m.train()
        m.manual_seed(5)
        opt, sched = T.build_optimizer(m, T.default_args(train_batch_size=4, learning_rate=1e-3), 4)
        dp = parallel.DataParallel(m, opt, bucket_mb=0.5, force_dynamic_queue=True, wire_dtype=torch.bfloat16)
        dp.bucketer.force_wire_path = True
        out, _ = m(**batch)
        out[0].mean().backward()
        staged = len(dp.bucketer.stage1) + len(dp.bucketer.stage2)
        dp.finish_backward()
        torch.cuda.synchronize()
        w = m._flat.grads.clone()
        ops.dynamic_tile_queue = False
        assert staged >= 1, staged                                                    # buckets really went through the two-stage exchange
        werr = float((w - b).abs().max() / b.abs().max())
        # round 5: differentiated forwards whose backward never comes (a validation pass with grad enabled) must not pile up
        # unions -- the list is bounded, a cut list makes finish_backward rebuild the union, and the next real step is still right
        m = build(cfg, dropout=0.1)
        m.train()
        m.manual_seed(5)
        opt, sched = T.build_optimizer(m, T.default_args(train_batch_size=4, learning_rate=1e-3), 4)
        dp = parallel.DataParallel(m, opt, bucket_mb=0.5, force_dynamic_queue=True)
        for _ in range(12):
            m(**batch)
        assert len(dp._unions) <= 4 and dp._unions_cut, (len(dp._unions), dp._unions_cut)
        m.manual_seed(5)
        m._calls = 0                                                                   # (the same dropout seeds as the reference runs above)
        out, _ = m(**batch)
        out[0].mean().backward()
        dp.finish_backward()
        torch.cuda.synchronize()
        cerr = float((m._flat.grads - b).abs().max() / b.abs().max())
        assert cerr < 2e-3, cerr
        assert not dp._unions and not dp._unions_cut
        # ... and in deterministic mode the whole data-parallel step (row block through the ordered id-run sums) is bit-reproducible
        # -- also through the two paths that scatter the lookup's rows LOCALLY (round 6, ADVICE r5: they called index_add_ with duplicate ids,
        # fp32 atomics in arrival order): a no_sync() accumulation micro-step in front of the exchanged one (DataParallel._fold_rows_locally),
        # and finish_backward() without an early word exchange (early_word_embedding=False)
        ops.set_deterministic(True)
        batch_b = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=32), "cuda")
        for variant in ("plain", "no_sync_then_step", "no_early_word"):
            det = []
            for _ in range(2):
                m = build(cfg, dropout=0.1)
                m.train()
                m.manual_seed(5)
                opt, sched = T.build_optimizer(m, T.default_args(train_batch_size=4, learning_rate=1e-3), 4)
                dp = parallel.DataParallel(m, opt, bucket_mb=0.5, force_dynamic_queue=True, early_word_embedding=variant != "no_early_word")
                if variant == "no_sync_then_step":
                    with dp.no_sync():
                        o0, _ = m(**batch_b)
                        o0[0].mean().backward()
                out, _ = m(**batch)
                out[0].mean().backward()
                dp.finish_backward()
                torch.cuda.synchronize()
                det.append(m._flat.grads.clone())
            assert torch.equal(det[0], det[1]), variant
        ops.set_deterministic(False)
        ops.dynamic_tile_queue = False
        q.put(("ok", err, float(b.abs().max()), werr))
        dist.destroy_process_group()
    except Exception as e:                                                            # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc(), 0.0, 0.0))


def test_async_prologue_gives_the_same_training_run():
    """model.async_prologue: the step prologue (key bias, unmasked lengths, labelled rows) and its device -> host copy run on the
    model's input stream, ahead of the previous step's tail, with two alternating output buffer sets -- the host enqueues a step
    ahead of the GPU.  Ten seeded train steps (dropout on, AdamW, four different ragged batches, no host synchronisation in
    between, so the host really runs ahead) against the same run with the prologue on the compute stream.  Training amplifies the
    fp32-atomics noise of the heads and the embedding scatter, so the yardstick is a SECOND run on the compute stream: the first
    step (before any update) agrees to rounding, the later ones no worse than two identical runs do."""
    from msa_amd import trainer as T
    cfg = dict(hidden=256, layers=3, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    pool = [batch_to(synthetic_batch(6, 30, 200, 170, dataset="mosei", vocab=cfg["vocab"], seed=500 + i), DEV) for i in range(4)]
    runs = {}
    for tag, async_ in (("a", False), ("b", False), ("async", True)):
        m = build(cfg, dropout=0.1)
        m.train()
        m.manual_seed(11)
        m.async_prologue = async_
        opt, sched = T.build_optimizer(m, T.default_args(train_batch_size=6, learning_rate=3e-4), 10, mode="hf")
        losses = []
        for i in range(10):
            out, _ = m(**pool[i % 4])
            out[0].mean().backward()
            opt.step(); sched.step(); opt.zero_grad()
            losses.append(out[0].detach())
        torch.cuda.synchronize()
        runs[tag] = (torch.stack(losses).double().cpu(), {n: q.detach().double().clone() for n, q in m.named_parameters()})
    la, lb, lc = runs["a"][0], runs["b"][0], runs["async"][0]
    assert bool(torch.isfinite(lc).all())
    assert abs(float(lc[0] - la[0])) <= 2e-6 * abs(float(la[0]))                     # the same forward
    noise = float((la - lb).abs().max())
    assert float((lc - la).abs().max()) <= 3.0 * noise + 1e-3, (lc, la, noise)       # (a wrong key bias or row list moves the loss by > 0.1)
    pn = max(float((runs["a"][1][n] - runs["b"][1][n]).abs().max()) for n in runs["a"][1])
    pa = max(float((runs["async"][1][n] - runs["a"][1][n]).abs().max()) for n in runs["a"][1])
    # (round 6: the heads no longer use fp32 atomics, two identical runs now agree to ~1e-7 and are no yardstick any more for runs whose FIRST
    # step already differs in rounding -- the packing is built before the embedding kernels in one and behind them in the other (2e-6 on the
    # loss above); ten Adam steps at lr 3e-4 move a parameter by up to 3e-3 and turn such a difference into ~2e-4: bounded at 5e-4)
    assert pa <= 3.0 * pn + 5e-4, (pa, pn)


def test_side_streams_and_late_weight_gradients_give_bit_identical_training_runs():
    """Round 6: (a) the weight gradients of the few loss-carrying rows ride in the deferred multi-layer call (model.late_wgrads), (b) that
    call goes out on a side stream beside the embedding stage's backward (model.wgrad_side_stream), (c) the optimizer's transposed-copy
    launch runs on a side stream beside the next step's prologue (ops.SIDE_TRANSPOSES), (d) the heads' forward levels and backward run on
    a side stream beside the MLM head (model.heads_side_stream), (e) the pair projections on side streams beside the packing and embedding
    launches (model.pairs_side_stream).  None of them changes what is computed: in
    deterministic mode (ordered sums everywhere) six seeded train steps -- dropout on, AdamW, four ragged batches, no host synchronisation
    in between, and a forced deferred call (five layers) -- give the SAME BITS with every one of them switched off: losses and every
    parameter.  A missing wait (a backward reading transposed weights of the step before, a gradient added before the call overwrote
    it) shows up here as a difference."""
    from msa_amd import trainer as T
    from msa_amd import ops as _ops
    cfg = dict(hidden=256, layers=5, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    pool = [batch_to(synthetic_batch(6, 30, 200, 170, dataset="mosei", vocab=cfg["vocab"], seed=700 + i), DEV) for i in range(4)]
    runs = {}
    was, was_side = _ops.deterministic(), _ops.SIDE_TRANSPOSES
    try:
        _ops.set_deterministic(True)
        for tag, late, side_w, side_t, side_h, side_p in (("all", True, True, True, True, True), ("no_side_wgrads", True, False, True, True, True),
                                                          ("no_side_transposes", True, True, False, True, True), ("no_side_heads", True, True, True, False, True),
                                                          ("no_side_pairs", True, True, True, True, False), ("none", False, False, False, False, False)):
            _ops.SIDE_TRANSPOSES = side_t
            m = build(cfg, dropout=0.1)
            m.train()
            m.manual_seed(13)
            m.defer_wgrads, m.late_wgrads, m.wgrad_side_stream, m.heads_side_stream, m.pairs_side_stream = True, late, side_w, side_h, side_p
            opt, sched = T.build_optimizer(m, T.default_args(train_batch_size=6, learning_rate=3e-4), 10, mode="hf")
            losses = []
            for i in range(6):
                out, _ = m(**pool[i % 4])
                out[0].mean().backward()
                opt.step(); sched.step(); opt.zero_grad()
                losses.append(out[0].detach())
            torch.cuda.synchronize()
            runs[tag] = (torch.stack(losses).cpu(), {n: q.detach().clone().cpu() for n, q in m.named_parameters()})
    finally:
        _ops.set_deterministic(was)
        _ops.SIDE_TRANSPOSES = was_side
    ref_l, ref_p = runs["all"]
    assert bool(torch.isfinite(ref_l).all())
    for tag in ("no_side_wgrads", "no_side_transposes", "no_side_heads", "no_side_pairs"):
        assert torch.equal(runs[tag][0], ref_l), (tag, runs[tag][0], ref_l)
        for n in ref_p:
            assert torch.equal(runs[tag][1][n], ref_p[n]), (tag, n)
    # with the few-row gradients launched where they arise, the top layer's QKV gradient splits its token axis (another fp32 order): close, not equal
    assert float((runs["none"][0] - ref_l).abs().max()) <= 1e-3, (runs["none"][0], ref_l)


def test_data_parallel_over_rccl_world1_equals_plain_step():
    """The DP code path on hardware with ONE GPU: torch.distributed "nccl" (= RCCL) initialised with world_size 1, the model
    wrapped in parallel.DataParallel (weight broadcast, bucketed all-reduce hooks fired from backward as layers finish, the
    persistent GEMM on its dynamic tile queue), against the same seeded train-mode step without the wrapper: gradients equal up
    to fp32 atomic-order noise.  Runs in a child process (the process group must exist before the first GPU call)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    p = ctx.Process(target=_nccl_world1_worker, args=(port, q))
    p.start()
    status, err, scale, werr = q.get(timeout=600)
    p.join(120)
    assert status == "ok", err
    assert p.exitcode == 0
    assert err < 2e-3, (err, scale)
    # the bf16 wire exchange (all_to_all_single + all_gather_into_tensor over RCCL): one rounding in, one out -> within 2^-7 of the fp32 path
    assert werr < 2.0 ** -7, (werr, scale)


def test_layer_gradient_slices_are_final_when_their_hook_fires():
    """Round 4: data parallelism hands slice k of the flat gradient buffer to an all-reduce the moment
    ``model.grad_hook(i)`` fires for encoder layer i (parallel.DataParallel._on_layer_done -> GradBucketer.ready).  With TWO layers'
    weight gradients per launch (model.pair_wgrads) the upper layer of a pair is complete only after the pair's launch, and the
    LayerNorm gamma / beta sums of a layer arrive through a deferred batched reduce: the slice must hold its FINAL value when the hook
    fires, or an overlapped all-reduce would ship a partial gradient.  A recording hook clones every layer's slice on the compute
    stream at hook time (exactly what a collective enqueued there would read) -- five layers (pairs (3, 2), (1, 0) behind the sparse
    top layer; pairs (4, 3), (2, 1) and a single 0 with the short cuts off), train mode with all dropouts -- and every snapshot must equal
    the buffer after backward bit for bit; same for the tied word-embedding table at ``head_grad_hook`` time with the lookup's rows
    deferred (DataParallel's early reduce).  No process group involved: this is the ordering contract itself."""
    cfg = dict(hidden=256, layers=5, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    batch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=41), DEV)
    for shortcuts in (True, False):
        m = build(cfg, dropout=0.1)
        m.train()
        m.manual_seed(8)
        if not shortcuts:
            m.skip_padded_backward = False
            m.sparse_top_layer_backward = False
        m._ensure_ready(next(m.parameters()).device)
        flat, L = m._flat, cfg["layers"]
        bounds = [0]
        for i in reversed(range(L)):                                   # the slices DataParallel builds (parallel.py)
            last = f"bert.encoder.layer.{i}.attention.self.value.bias"
            bounds.append(flat.offset[last] + flat.numel[last])
        snaps, order = {}, []

        def hook(i, _f=flat, _b=bounds, _s=snaps, _o=order, _L=L):
            k = _L - 1 - i
            _o.append(i)
            _s[i] = (_b[k], _b[k + 1], _f.grads[_b[k]:_b[k + 1]].clone())
        wname = "bert.embeddings.word_embeddings.weight"
        wlo = flat.offset[wname]
        whi = wlo + flat.vpad * cfg["hidden"]
        head_snap = []
        m.grad_hook = hook
        m.head_grad_hook = lambda _f=flat, _h=head_snap: _h.append(_f.grads[wlo:whi].clone())
        m.defer_embed_rows = True
        out, _ = m(**batch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        assert order == list(reversed(range(L))), order
        for i, (lo, hi, snap) in snaps.items():
            assert hi > lo and float(snap.abs().sum()) > 0.0, i
            assert torch.equal(snap, flat.grads[lo:hi]), (shortcuts, i, float((snap - flat.grads[lo:hi]).abs().max()))
        assert len(head_snap) == 1 and torch.equal(head_snap[0], flat.grads[wlo:whi]) and float(head_snap[0].abs().sum()) > 0.0
        assert len(m.__dict__.get("_deferred_embed_rows", [])) == 1


@pytest.mark.parametrize("size", ["small5", "headline"])
def test_deterministic_mode_gives_bit_identical_steps(size):
    """Round 5, model.deterministic = True (the library's mmbert_set_deterministic): ordered sums instead of fp32 atomics -- the CE loss sums,
    the heads' skinny products, the weight-gradient kernel's bias sums, the LayerNorm partial-sum fold, the embedding scatter through
    sorted keys + segment sums.  The same seeded TRAIN-mode step (dropout on) of two freshly built models gives BIT-IDENTICAL losses,
    regression logits, and flat gradient buffers; two optimizer steps later the parameters are bit-identical too.  And two launch paths
    of the same function (all layers' weight gradients in one call / per layer pair; at five layers also the exact-zero short cuts on /
    off) agree within the ORIGINAL same-function bound again -- 2e-3 in L2 (test_model_gpu.same_grads had to go to 5e-3 in round 4 for the
    atomics' order).
    small5: five layers (a pair launch and a single-layer launch in the paired form); headline: the benchmarked model at batch 4."""
    from msa_amd import ops as _ops
    from msa_amd import trainer as T_
    if size == "small5":
        cfg, shape = dict(hidden=256, layers=5, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0), (4, 24, 200, 130)
    else:
        cfg, shape = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0), (4, 50, 500, 500)
    batch = batch_to(synthetic_batch(*shape, dataset="mosei", vocab=cfg["vocab"], seed=91), DEV)
    batch2 = batch_to(synthetic_batch(*shape, dataset="mosei", vocab=cfg["vocab"], seed=92), DEV)

    def run(flags=None, steps=0, accumulate=False):
        m = build(cfg, dropout=0.1)
        m.train()
        m.manual_seed(31)
        for k, v in (flags or {}).items():
            if k == "training":
                m.train(v)
            else:
                setattr(m, k, v)
        if accumulate:                                            # a micro-batch whose gradients the timed one is accumulated onto
            o0, _l0 = m(**batch2)
            o0[0].mean().backward()
        out, logits = m(**batch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        res = dict(losses=[out[i].detach().clone() for i in (0, 4, 5, 6)], logits=logits.detach().clone(), grads=m._flat.grads.clone(),
                   named={n: q.grad.detach().float().clone() for n, q in m.named_parameters() if q.grad is not None})
        if steps:
            opt, sched = T_.build_optimizer(m, T_.default_args(train_batch_size=shape[0], learning_rate=1e-3), 10, mode="hf")
            sched.step()
            opt.step(); opt.zero_grad()
            for _ in range(steps - 1):
                o, _l = m(**batch)
                o[0].mean().backward()
                opt.step(); sched.step(); opt.zero_grad()
            torch.cuda.synchronize()
            res["params"] = m._flat.params.clone()
        return res
    was = _ops.deterministic()
    try:
        _ops.set_deterministic(True)
        a, b = run(steps=2), run(steps=2)
        for x, y in zip(a["losses"], b["losses"]):
            assert torch.equal(x, y), (float(x), float(y))
        assert torch.equal(a["logits"], b["logits"])
        diff = [n for n in a["named"] if not torch.equal(a["named"][n], b["named"][n])]
        assert not diff, diff[:8]
        assert torch.equal(a["grads"], b["grads"]) and torch.equal(a["params"], b["params"])
        # Gradient accumulation (the reference steps on every second micro-batch): adding onto NON-ZERO gradients is where several adders
        # per address first show -- a + b == b + a, but (g + a) + b != (g + b) + a.  Round 5: the joint embedding's LayerNorm (one pass per
        # pair modality, two items of one batched fold) differed in the last bit from the second micro-batch on.
        c, d = run(accumulate=True), run(accumulate=True)
        diff = [n for n in c["named"] if not torch.equal(c["named"][n], d["named"][n])]
        assert not diff and torch.equal(c["grads"], d["grads"]), diff[:8]
        # same function, other launch paths, at the ORIGINAL tolerance: the weight gradients per layer pair instead of in one call (train
        # mode, the same masks), and -- in EVAL mode, where the packed row order does not select other dropout masks -- the exact-zero
        # short cuts off
        ulps = 2.0 ** -7 if size == "small5" else 2.0 ** -6       # (one flipped bf16 rounding of an addend; two at twelve layers)

        def close(o, ref, tag, l2=2e-3):
            for i in range(4):
                assert abs(float(o["losses"][i]) - float(ref["losses"][i])) <= 1e-6 * abs(float(ref["losses"][i])), (tag, i)
            for n, g in ref["named"].items():
                if "attention.self.key.bias" in n:
                    continue
                d = o["named"][n] - g
                assert float(d.abs().max()) <= ulps * float(g.abs().max()) + 2e-7, (tag, n, float(d.abs().max()), float(g.abs().max()))
                assert float(d.norm()) <= l2 * float(g.norm()) + 2e-7 * float(g.numel()) ** 0.5, (tag, n, float(d.norm()), float(g.norm()))
        close(run(dict(defer_wgrads=False)), a, "paired weight gradients")
        if size == "small5":
            # (the short cuts change WHERE bf16 roundings of activation gradients happen -- the sparse top layer's split-K products, the
            # packed rows' tile shapes --, not only the order of fp32 sums: at five layers the original bound holds; at twelve the flipped
            # roundings compound to ~1 % on the embedding tables -- test_model_gpu's timed_depth case bounds that comparison)
            ev = dict(training=False)
            close(run(dict(ev, skip_padded_backward=False, sparse_top_layer_backward=False)), run(ev), "short cuts off (eval)")
    finally:
        _ops.set_deterministic(was)
