"""The CPU oracle against every golden vector generated from the real reference
(tests/golden/make_golden.py).  fp32 vs fp32: tolerances are rounding-order only."""
import os

import numpy as np
import pytest
import torch

from oracle import mmbert_oracle as O
from msa_amd.data import synthetic_batch

CFG1 = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
CASES = {
    "cfg1_T50_P64": CFG1,
    "cfg1_T50_P50": CFG1,
    "h64_L1_T16_P24x40": dict(CFG1, hidden=64, layers=1, heads=4, intermediate=128, vocab=2048, alpha=0.7, beta=0.3),
}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def close(a, b, rtol=2e-5, atol=2e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_param_count_matches_reference(golden_dir):
    g = load(golden_dir, "cfg1_T50_P64")
    assert O.count_params(CFG1) == int(g["n_params"]) == 4627394


def test_units_joint_embeddings_mask_cpc(golden_dir):
    g = load(golden_dir, "units")
    p = O.seeded_params(CFG1)
    te = torch.from_numpy(g["g1_text_emb"])
    for tag in "vs":
        out = O.joint_embeddings(p, te, torch.from_numpy(g[f"g1_{tag}_pair"]), (35, 74))
        close(out, g[f"g1_{tag}_out"])
    assert int(g["g1_wrongdim_raises"]) == 1
    with pytest.raises(Exception, match=str(g["g1_wrongdim_msg"])):
        O.joint_embeddings(p, te, torch.zeros(2, 4, 33), (35, 74))
    m2, m3 = torch.from_numpy(g["g2_m2"]), torch.from_numpy(g["g2_m3"])
    for j in (0, 1):
        close(O.extended_attention_mask(m2, bool(j)), g[f"g2_out2_j{j}"])
        close(O.extended_attention_mask(m3, bool(j)), g[f"g2_out3_j{j}"])
    close(O.extended_attention_mask(m3.long(), True), g["g2_out3i_j1"])
    # quirk B-2: the live frame with feature-0 == 0 is masked out in joint mode
    assert g["g2_out3_j1"][1, 0, 0, 2] == -10000.0
    x, y = torch.from_numpy(g["g3_x"]), torch.from_numpy(g["g3_y"])
    for n in ("cpc_zt", "cpc_zv", "cpc_za"):
        close(O.cpc(p, n, x, y), g["g3_" + n])


@pytest.mark.parametrize("name", list(CASES))
def test_full_forward_backward(golden_dir, name):
    cfg = CASES[name]
    g = load(golden_dir, name)
    B, T, Pv, Pa, seed = (int(x) for x in g["meta"])
    batch = synthetic_batch(B, T, Pv, Pa, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=seed)
    p = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
    inter = {}
    outputs, logits = O.pretraining_forward(p, cfg, **batch, collect=inter)
    hidden = inter["t.hidden"]
    for tag in "tvs":                                           # every pass's embeddings and per-layer hidden states
        close(inter[tag + ".emb"].detach(), g[f"{tag}_emb"], rtol=1e-4, atol=1e-5)
        for l in range(cfg["layers"]):
            close(inter[tag + ".hidden"][l].detach(), g[f"{tag}_hidden{l}"], rtol=1e-4, atol=2e-5)
    close(inter["v.jemb"].detach(), g["v_jemb"], rtol=1e-4, atol=1e-5)
    close(inter["s.jemb"].detach(), g["s_jemb"], rtol=1e-4, atol=1e-5)
    close(outputs[0].detach(), g["joint_loss"], rtol=1e-5)
    close(outputs[4].detach(), g["ap_loss"], rtol=1e-5)
    close(outputs[5].detach(), g["label_loss"], rtol=1e-5)
    close(outputs[6].detach(), g["nce"], rtol=1e-5)
    assert outputs[1] is None and outputs[2] is None and outputs[3] is None
    close(logits.detach(), g["logits"], rtol=1e-4, atol=1e-5)
    for l in range(cfg["layers"]):
        close(hidden[l].detach(), g[f"t_hidden{l}"], rtol=1e-4, atol=2e-5)
    for pi, tag in enumerate("tvs"):
        sc = outputs[7 + 2 * pi].detach()
        assert tuple(sc.shape) == tuple(g[f"{tag}_scores_shape"])
        close(sc[:, :, :48], g[f"{tag}_scores_head"], rtol=1e-4, atol=2e-5)
        close(sc[:, :, 5::611], g[f"{tag}_scores_stride"], rtol=1e-4, atol=2e-5)
        close(torch.logsumexp(sc, -1), g[f"{tag}_scores_lse"], rtol=1e-5, atol=1e-5)
        close(outputs[8 + 2 * pi].detach(), g[f"{tag}_rel"], rtol=1e-4, atol=1e-5)
    outputs[0].mean().backward()
    nograd = sorted(k for k, v in p.items() if v.grad is None or not bool(v.grad.abs().sum() > 0))
    assert nograd == list(g["nograd"])                          # quirk B-9
    for k, v in p.items():
        if v.grad is None or k in nograd:
            continue
        gn = float(g["gnorm/" + k])
        assert abs(float(v.grad.norm()) - gn) <= 2e-4 * gn + 1e-7, k
        close(v.grad.reshape(-1)[:16], g["ghead/" + k], rtol=2e-3, atol=2e-6 + 1e-4 * gn)
    rows = torch.tensor([0, 101, 102, 103, 1000, 2000])
    k = "bert.embeddings.word_embeddings.weight"
    close(p[k].grad[rows], g["grows/" + k], rtol=2e-3, atol=1e-6)


def test_intermediates_joint_pass(golden_dir):
    cfg = CASES["cfg1_T50_P64"]
    g = load(golden_dir, "cfg1_T50_P64")
    B, T, Pv, Pa, seed = (int(x) for x in g["meta"])
    b = synthetic_batch(B, T, Pv, Pa, seed=seed)
    p = O.seeded_params(cfg)
    ids, am = b["input_ids"], b["attention_mask"]
    for tag, pair, twx, mask in (("v", ids[1], ids[3], am[1]), ("s", ids[2], ids[4], am[2])):
        hid = []
        seq, pooled = O.mmbert_model(p, cfg, (twx, pair), mask, None, True, collect=hid)
        for l in range(cfg["layers"]):
            close(hid[l], g[f"{tag}_hidden{l}"], rtol=1e-4, atol=2e-5)
        close(pooled, g[f"{tag}_pooled"], rtol=1e-4, atol=1e-5)
        emb = O.bert_embeddings(p, twx, torch.zeros_like(twx))
        close(emb, g[f"{tag}_emb"], rtol=1e-4, atol=1e-5)
        close(O.joint_embeddings(p, emb, pair, (35, 74)), g[f"{tag}_jemb"], rtol=1e-4, atol=1e-5)


def test_collate_contract(golden_dir):
    """G6: dtypes/shapes and the mask quirks the synthetic generator must reproduce."""
    g = load(golden_dir, "collate")
    assert str(g["text3_dtype"]) == "torch.float64" and str(g["text0_dtype"]) == "torch.int64"
    assert str(g["visual1_dtype"]) == "torch.float64" and str(g["visual4_dtype"]) == "torch.float64"
    assert str(g["speech4_dtype"]) == "torch.int64" and str(g["attention1_dtype"]) == "torch.int64"
    assert str(g["attention0_dtype"]) == "torch.float64" and str(g["text4_dtype"]) == "torch.float32"
    # quirk B-1: text-with-pair masks stay all ones although the text has PAD
    assert (g["text0"] == 0).any() and g["attention0"].all() and g["attention1"].all()
    assert ((g["text3"] == 0) == (g["text0"] == 0)).all()
    assert ((g["visual4"] != 0) == (g["visual1"] != 0)).all()
    b = synthetic_batch(3, 8, 8, 8, seed=4)
    assert b["input_ids"][0].dtype == torch.int64 and b["input_ids"][1].dtype == torch.float64
    assert b["attention_mask"][0].dtype == torch.float64
    assert b["attention_mask"][1][0].dtype == torch.float64 and b["attention_mask"][1][1].dtype == torch.float64
    assert b["attention_mask"][2][0].dtype == torch.int64 and b["attention_mask"][2][1].dtype == torch.int64
    assert b["attention_mask"][1][0].all() and b["attention_mask"][2][0].all()
    assert b["sentiment"].dtype == torch.float32
    assert tuple(b["attention_mask"][1][1].shape) == tuple(b["input_ids"][1].shape)
    # P == T: pair labels are a copy of the text labels (quirk B-5)
    lv = b["masked_labels"][1]
    assert (lv[:, :8] == lv[:, 8:]).all()


def _batch_from_items(g, idx):
    T = len(g["item0_text"])
    text = torch.tensor(np.stack([g[f"item{i}_text"] for i in idx]))
    vis = torch.tensor(np.stack([g[f"item{i}_visual"] for i in idx]))
    sp = torch.tensor(np.stack([g[f"item{i}_speech"] for i in idx]))
    B = len(idx)
    lab2 = torch.cat((text, text), dim=-1)                       # mlm off: labels = inputs (REF:trainer.py:45-53)
    return dict(
        input_ids=(text, vis, sp, text, text),
        token_type_ids=(torch.zeros(B, T, dtype=torch.long), None, None),
        attention_mask=((text != 0).double(), (torch.ones(B, T).double(), (vis != 0).double()),
                        (torch.ones(B, T).long(), (sp != 0).long())),
        masked_labels=(text, lab2, lab2),
        ap_label=(torch.tensor([int(g[f"item{i}_ap"][0]) for i in idx]), torch.tensor([int(g[f"item{i}_ap"][1]) for i in idx])),
        sentiment=torch.tensor([float(g[f"item{i}_sent"]) for i in idx], dtype=torch.float32),
    )


def test_train_epoch_trajectory(golden_dir):
    """G8: four micro-batches of REF trainer.train_epoch; optimizer steps after micro-batch 2 and 4
    (the ``&`` quirk), torch.optim.AdamW(eps=1e-6) + linear warm-up (warmup == total)."""
    g = load(golden_dir, "train4")
    cfg = dict(CFG1, vocab=4096, hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    p = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
    p0 = {k: v.detach().clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    lr0, nopt = float(g["lr"]), int(g["n_opt_steps"])
    order = list(g["order"])
    opt_step = 0
    for step in range(4):
        batch = _batch_from_items(g, order[2 * step:2 * step + 2])
        out, _ = O.pretraining_forward(p, cfg, **batch, train=True)
        for j, i in enumerate((0, 4, 5, 6)):
            close(out[i].detach(), g["losses"][step][j], rtol=2e-5)
        out[0].mean().backward()
        if O.should_step(step, 1):
            assert step in (1, 3)
            lr = lr0 * O.linear_schedule_lambda(opt_step, nopt, 1.0 * nopt)
            opt_step += 1
            with torch.no_grad():
                for k, t in p.items():
                    if t.grad is None:
                        continue
                    O.adamw_step(t, t.grad, m[k], v2[k], opt_step, lr, 0.01 if O.decays(k) else 0.0, mode="torch")
                    t.grad = None
    for k, t in p.items():
        dn = float(g["dnorm/" + k])
        delta = (t.detach() - p0[k])
        assert abs(float(delta.norm()) - dn) <= 2e-3 * dn + 1e-9, k
        close(delta.reshape(-1)[:16], g["dhead/" + k], rtol=5e-3, atol=1e-7 + 2e-3 * dn / max(1, t.numel()) ** 0.5)


def test_hf_adamw_differs_from_torch_only_in_documented_ways():
    """The 'hf' mode is restated from the published transformers-2.8.0 algorithm (parity unpinned);
    sanity: with wd=0 and eps->0 both modes agree."""
    torch.manual_seed(0)
    p1 = torch.randn(64); p2 = p1.clone(); g = torch.randn(64)
    m1 = torch.zeros(64); v1 = torch.zeros(64); m2 = torch.zeros(64); v2 = torch.zeros(64)
    for t in range(1, 4):
        O.adamw_step(p1, g, m1, v1, t, 1e-3, 0.0, eps=1e-12, mode="hf")
        O.adamw_step(p2, g, m2, v2, t, 1e-3, 0.0, eps=1e-12, mode="torch")
    close(p1, p2, rtol=1e-5, atol=1e-6)


def test_hf_adamw_matches_the_hand_computed_vector():
    """G11 (round 5): the oracle's mode "hf" against a known-answer vector computed by hand from the published transformers-2.8.0
    update rule (tests/golden/hf_adamw_hand.py: 40-digit decimal arithmetic, no optimizer implementation) -- three steps of three
    scalar trajectories in float64: parameters, first and second moments.  With this the optimizer bench.py and the 52-step loop
    run (trainer mode "hf") is pinned by vectors, not by reading."""
    from tests.golden import hf_adamw_hand as G
    for name, (p0, wd, grads, ps, ms, vs) in G.TRAJECTORIES.items():
        p = torch.tensor([p0], dtype=torch.float64)
        m, v = torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
        for t, g in enumerate(grads, 1):
            O.adamw_step(p, torch.tensor([g], dtype=torch.float64), m, v, t, G.LR, wd, beta1=G.BETA1, beta2=G.BETA2, eps=G.EPS, mode="hf")
            assert abs(float(p) - ps[t - 1]) <= 1e-13 * abs(ps[t - 1]), (name, t, float(p), ps[t - 1])
            assert abs(float(m) - ms[t - 1]) <= 1e-13 * abs(ms[t - 1]) and abs(float(v) - vs[t - 1]) <= 1e-13 * vs[t - 1], (name, t)
        # ... and the torch rule does NOT give these numbers (eps inside the bias correction, decay first): the vector discriminates
        q = torch.tensor([p0], dtype=torch.float64)
        m, v = torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
        for t, g in enumerate(grads, 1):
            O.adamw_step(q, torch.tensor([g], dtype=torch.float64), m, v, t, G.LR, wd, beta1=G.BETA1, beta2=G.BETA2, eps=G.EPS, mode="torch")
        assert abs(float(q) - ps[-1]) > 1e-7 * abs(ps[-1]), name


def test_schedule_and_step_rule():
    assert [O.should_step(s, 1) for s in range(4)] == [False, True, False, True]
    assert [O.should_step(s, 1, quirk=False) for s in range(3)] == [True, True, True]
    assert O.linear_schedule_lambda(0, 10, 10) == 0.0 and O.linear_schedule_lambda(5, 10, 10) == 0.5
    assert O.linear_schedule_lambda(10, 10, 10) == 0.0
    assert O.decays("bert.encoder.layer.0.output.dense.weight")
    assert not O.decays("bert.encoder.layer.0.output.LayerNorm.weight") and not O.decays("cls.predictions.bias")


def test_mask_tokens_rule_invariants(golden_dir):
    b = synthetic_batch(4, 32, 8, 8, seed=11)
    ids, lab = b["input_ids"][0], b["masked_labels"][0]
    sel = lab != -100
    assert not sel[:, 0].any() and not sel[ids == 102].any()             # [CLS] / [SEP] never selected
    assert ((ids == 103) <= sel).all()
    inp = torch.tensor([[101, 5, 6, 102, 0]])
    out, labels = O.mask_tokens_rule(inp, torch.ones_like(inp), torch.tensor([[1, 1, 0, 1, 1]]))
    assert out.tolist() == [[101, 103, 6, 102, 103]] and labels.tolist() == [[-100, 5, 6, -100, 0]]     # [PAD] is selectable
    # G10: the rule against the REAL mask_tokens -- its outputs are reproduced by feeding the rule the draws it made
    g = load(golden_dir, "mask_tokens")
    ids = torch.from_numpy(g["inputs"])
    for seed in (1, 2):
        torch.manual_seed(seed)                                          # the reference's two draws, in its order (:27,30)
        prob = torch.full(ids.shape, 0.15)
        prob[(ids == 101) | (ids == 102)] = 0.0
        select = torch.bernoulli(prob)
        replace = torch.bernoulli(torch.full(ids.shape, 0.8))
        out, labels = O.mask_tokens_rule(ids, select, replace)
        np.testing.assert_array_equal(out.numpy(), g[f"out_seed{seed}"])
        np.testing.assert_array_equal(labels.numpy(), g[f"labels_seed{seed}"])
