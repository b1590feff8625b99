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
    assert float(flat.grads.abs().max()) == 0.0 and not flat.grads_dirty
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
