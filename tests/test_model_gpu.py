"""End-to-end parity of the HIP path (through the reference's module API) against the CPU oracle
and against the golden vectors generated from the real reference.

Stated tolerances (bf16 operands, fp32 accumulation): losses 3e-3 relative,
regression logits 2e-2 absolute, MLM scores 3e-2 absolute, parameter-gradient cosine >= 0.995 and
every parameter gradient within 4.5 % relative error (== cosine 0.999) of the oracle's, except
ill-conditioned head gradients, which are bounded by 3x the deviation that bf16 STORAGE ALONE
causes in the oracle itself (oracle.bf16_storage_emulation, a calibrator -- not a parity pin)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mmbert_oracle as O
from msa_amd.data import synthetic_batch, batch_to, to_fused

DEV = "cuda"


def same_grads(a, b, tag):
    """Two evaluations of the SAME function by different launch paths (sparse / dense rows, packed / unpacked, with / without returned
    scores): the paths differ in fp32 summation order (atomics, split-K), which flips single bf16 roundings of activation gradients
    downstream.  One flip moves an entry by one bf16 ulp of an addend -- at most 2^-7 of the largest entry --, so: every entry within
    2^-7 of the largest one AND the whole tensor within 2e-3 in L2 (a wrong row, mask or scale moves it by tens of per cent); absolute
    floors per ELEMENT (2e-7: gradients that are ~0 by cancellation -- CPC at init -- carry that much atomic-order noise).  Bounds of
    2e-3 ... 4e-3 of the largest entry, as first written, sat inside the noise: tools/stress_repeat.py found 1 failure in 30-50
    repetitions for three of these tests.  Round 4: the L2 part is 5e-3 -- test_deferred_weight_gradients_equal_the_per_layer_launches
    failed 3 of 30 repetitions at the round's first commit and 11 of 30 at its last, always with the SAME numbers (the speech
    projection's weight gradient 3.4e-3 off in L2: ONE bf16 rounding of an activation gradient upstream has two outcomes, and that
    gradient is the outer product of few rows)."""
    a, b = a.float(), b.float()
    scale = float(b.abs().max())
    d = a - b
    assert float(d.abs().max()) <= 2.0 ** -7 * scale + 2e-7, (tag, float(d.abs().max()), scale)
    assert float(d.norm()) <= 5e-3 * float(b.norm()) + 2e-7 * math.sqrt(a.numel()), (tag, float(d.norm()), float(b.norm()))

CFG1 = dict(hidden=128, layers=2, heads=2, intermediate=512, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)


def build(cfg, train=False, seed=0):
    from msa_amd.model import MMBertConfig, MMBertForPretraining
    c = MMBertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                     num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"])
    m = MMBertForPretraining(c)
    m.bert.set_joint_embeddings(cfg["dataset"])
    m.set_alpha_beta(cfg.get("alpha", 1.0), cfg.get("beta", 1.0))
    sd = O.seeded_params(cfg, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and sorted(missing) == ["cls.predictions.decoder.bias", "cls.predictions.decoder.weight"], (missing, unexpected)
    m = m.to(DEV)
    m.train(train)
    return m


def oracle_run(cfg, batch, train=False, emulate_bf16=False, masks=None, probs=None):
    """``masks`` (+ ``probs`` = the exact drop probabilities the HIP path quantises to): the oracle in TRAIN mode replaying the
    keep masks the HIP kernels drew (oracle._dropout's replay hook) -- every dropout site must be in ``masks``."""
    p = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
    ocfg = dict(cfg)
    if not train:
        ocfg.update(hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    elif probs is not None:
        ocfg.update(probs)
    kw = dict(train=True, masks=masks) if train else {}
    if emulate_bf16:
        with O.bf16_storage_emulation():
            out, logits = O.pretraining_forward(p, ocfg, **batch, **kw)
            out[0].mean().backward()
    else:
        out, logits = O.pretraining_forward(p, ocfg, **batch, **kw)
        out[0].mean().backward()
    return p, out, logits


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-6)


def hip_dropout_masks(m, B, lens, split):
    """The keep mask of EVERY dropout site of the forward call ``m`` has just run, rebuilt from the model's own (seed, site, element
    index) mapping through the library's mask exports (mmbert_dropout_mask / mmbert_attn_dropout_mask) and keyed the way the
    oracle's replay hook reads them (oracle._dropout): ``{tag}emb`` [B,T,H], ``{tag}joint`` [B,S,H], ``{tag}l{i}.attn`` [B,h,S,S],
    ``{tag}l{i}.h1`` / ``.h2`` [B,S,H] for the passes tag = t. / v. / s.  Sites (model._encode / _EncoderFn): 1000 = embeddings over the
    [3BT, H] text rows of all passes, 1001 + pass = JointEmbeddings over that pass's [B*S, H]; layer i: 8i = attention probabilities
    (element index = the sequence's elem_base + (head*S + query)*Spad + key), 8i + 1 / 8i + 2 = the two hidden dropouts, indexed by the
    row of the matrix the encoder RUNS on -- the valid-first packing's row order when ``split`` is in use (split.inv[original row]).
    Also returns the exact drop probabilities (thr16 / 65536) for the oracle's 1 / (1 - p) scale."""
    from msa_amd import ops
    cfg = m.config
    H, L, heads = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
    T = lens[0]
    dev = next(m.parameters()).device
    plan = m._plan(lens, B, dev)
    lay, bounds = plan["layout"], plan["bounds"]
    seed = m._seed * 1000003 + m._calls                            # _next_seed() of the call that just ran
    ph, pa, pj = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob, m.bert.jointEmbeddings.dropout_prob
    tags = ("t.", "v.", "s.")
    masks = {}
    flat = lambda n, d: ops.dropout_mask(n, d, dev)
    emb = flat(3 * B * T * H, ops.make_drop(ph, seed, 1000)).view(3, B, T, H).cpu()
    for k, tag in enumerate(tags):
        masks[tag + "emb"] = emb[k]
        if k:
            masks[tag + "joint"] = flat(B * lens[k] * H, ops.make_drop(pj, seed, 1001 + k)).view(B, lens[k], H).cpu()
    rows = split.rows_packed if split is not None else lay.tokens
    for i in range(L):
        for name, site in (("h1", 8 * i + 1), ("h2", 8 * i + 2)):
            mk = flat(rows * H, ops.make_drop(ph, seed, site)).view(rows, H)
            if split is not None:
                mk = mk.index_select(0, split.inv)                  # original packed row r ran as row inv[r]
            for k, tag in enumerate(tags):
                masks[f"{tag}l{i}.{name}"] = mk[bounds[k]:bounds[k + 1]].view(B, lens[k], H).cpu()
        d = ops.make_drop(pa, seed, 8 * i)
        for k, tag in enumerate(tags):
            S = lens[k]
            masks[f"{tag}l{i}.attn"] = torch.stack([torch.stack([ops.attn_dropout_mask(S, lay.elem_base_host[k * B + b], h, d, dev) for h in range(heads)])
                                                    for b in range(B)]).cpu()
    q = lambda p_: ops.make_drop(p_, seed, 0)[1] / 65536.0
    return masks, dict(hidden_dropout=q(ph), attn_dropout=q(pa), joint_dropout=q(pj))


def check_against_oracle(cfg, B, T, Pv, Pa, seed, loss_tol=3e-3, train=False, flags=None, model_seed=7, score_tol=3e-2, logit_tol=2e-2,
                         grad_tol=0.045, score_mean_tol=None, head_grad_tol=None, report=None, calibrate=True):
    """``train``: the whole step in TRAIN mode (all three dropouts on, REF:trainer.py:40,66,83) -- the HIP model runs first, its masks
    are rebuilt (hip_dropout_masks) and the oracle replays them.  ``flags``: model switches set before the call.  The default
    tolerances are the L = 2 ones of the file header; deep models pass the bounds re-derived at depth (test_bert_base_12_layers_match_oracle)."""
    batch = synthetic_batch(B, T, Pv, Pa, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=seed)
    m = build(cfg, train=train)
    for k, v in (flags or {}).items():
        assert hasattr(m, k) or k in ("skip_padded_backward", "sparse_top_layer_backward", "skip_masked_keys", "sparse_mlm_backward"), k
        setattr(m, k, v)
    masks = probs = None
    if train:
        m.manual_seed(model_seed)
        seen, orig = [], m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
    out, logits = m(**batch_to(batch, DEV))
    if train:
        m._split_layout = orig
        torch.cuda.synchronize()
        masks, probs = hip_dropout_masks(m, B, [T, T + Pv, T + Pa], seen[0])
        keep = float(masks["t.l0.h1"].float().mean())
        assert 0.85 < keep < 0.95 and 0.4 < float(masks["v.joint"].float().mean()) < 0.6, keep       # dropout really was on
        m.last_split = seen[0]
    p, oout, ologits = oracle_run(cfg, batch, train=train, masks=masks, probs=probs)
    # calibrator: what bf16 storage alone does to each loss / gradient.  ``calibrate=False`` (the batch-16 run at the timed depth: a second
    # fp32 oracle pass there is another ~70 s and ~45 GB): fixed bounds only -- the ones the eval-mode batch-16 test states
    pe, eout = (None, None)
    if calibrate:
        pe, eout, _ = oracle_run(cfg, batch, train=train, masks=masks, probs=probs, emulate_bf16=True)
    for i, name in ((0, "joint"), (4, "ap"), (5, "label"), (6, "nce")):
        # 3e-3 relative, or 3x the deviation bf16 STORAGE alone causes in the oracle where that is larger (the 2-way alignment
        # CE of a 2-sample batch at H = 1024 moves by 1.9e-3 under storage rounding alone)
        tol = max(loss_tol, 3.0 * rel(eout[i].detach(), oout[i].detach())) if calibrate else loss_tol
        assert rel(out[i].detach(), oout[i].detach()) < tol, (name, float(out[i].detach()), float(oout[i].detach()), tol)
    assert out[1] is None and out[2] is None and out[3] is None
    assert float((logits.detach().float().cpu() - ologits.detach()).abs().max()) < logit_tol
    V = cfg["vocab"]
    for k in (7, 9, 11):
        assert tuple(out[k].shape) == tuple(oout[k].shape)
        d = (out[k].float().cpu() - oout[k].detach()).abs()
        assert float(d.max()) < score_tol, (k, float(d.max()))
        assert score_mean_tol is None or float(d.mean()) < score_mean_tol, (k, float(d.mean()))
    for k in (8, 10, 12):
        assert float((out[k].float().cpu() - oout[k].detach()).abs().max()) < logit_tol
    out[0].mean().backward()
    torch.cuda.synchronize()
    devs = {} if report else None
    worst = compare_gradients(m, p, pe, grad_tol, head_grad_tol, devs)
    if report:                                                    # every parameter's measured deviation -> gpurun_out/<report>.json (_report)
        _report(report, dict(losses={name: dict(hip=float(out[i].detach()), oracle=float(oout[i].detach())) for i, name in ((0, "joint"), (4, "ap"), (5, "label"), (6, "nce"))},
                             grads=devs, worst_cosine=worst))
    return m, out, worst


def compare_gradients(m, p, pe, grad_tol=0.045, head_grad_tol=None, devs=None):
    """Every parameter gradient of the HIP model ``m`` against the oracle's (``p``; ``pe`` = the oracle under bf16 storage emulation,
    the calibrator for the ill-conditioned head gradients).  Returns (worst cosine, its name)."""
    worst = (1.0, None)
    loose = []
    for n, q in m.named_parameters():
        og = p[n].grad
        g = q.grad.float().cpu()
        if og is None or float(og.abs().sum()) == 0.0:
            assert float(g.abs().sum()) == 0.0, f"{n}: reference has no gradient here"
            continue
        if "attention.self.key.bias" in n:
            # softmax is invariant to a per-query constant: the true gradient is 0, the reference holds
            # fp32 rounding noise (~1e-9); ours must be bf16-noise small
            assert float(og.norm()) < 1e-6 and float(g.norm()) < 2e-3, (n, float(g.norm()))
            continue
        dev = float((g - og).norm() / og.norm())
        if pe is None:
            # no calibrator (check_against_oracle(calibrate=False)): the fixed bounds of test_bert_base_12_layers_batch8_gradients_without_calibrator --
            # encoder side grad_tol, [B, H] heads head_grad_tol; a CPC projection whose gradient is orders below the pooler's (differences of nearly
            # equal unit vectors: see there) 50 % AND 1e-3 absolute, otherwise the heads' bound (train mode: dropout decorrelates the [CLS] rows)
            if devs is not None:
                devs[n] = dict(rel_err=dev, norm=float(og.norm()))
            enc = n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions"))
            pool_n = float(p["bert.pooler.dense.weight"].grad.norm())
            if n.startswith("cpc_") and float(og.norm()) < 1e-2 * pool_n:
                # (the eval-at-initialisation case: a difference of nearly equal unit vectors, orders below the other head gradients)
                assert dev < 0.5 and dev * float(og.norm()) < 1e-3, (n, dev, float(og.norm()))
            else:
                assert dev < (grad_tol if enc or head_grad_tol is None else head_grad_tol), (n, dev, float(og.norm()))
            cos = float(torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1)))
            if cos < worst[0]:
                worst = (cos, n)
            continue
        dev_emul = float((pe[n].grad - og).norm() / og.norm())
        if devs is not None:
            devs[n] = dict(rel_err=dev, emulated_oracle_rel_err=dev_emul, norm=float(og.norm()))
        # 4.5 % relative error == cosine 0.999; ill-conditioned head gradients (CPC, pooler, gates at
        # init) are allowed 3x what bf16 storage alone does to the ORACLE's gradient
        # ... and a gradient whose norm is three orders below the other head gradients (the CPC biases at initialisation: ~5e-5
        # against 0.4-1.3, see DESIGN numerics) is noise at bf16 precision on BOTH sides (the emulated oracle itself moves it by
        # 30-55 %): bounded absolutely, 2e-4
        # (head_grad_tol: the [B, H]-sized head gradients -- pooler, gates, classifier, align -- at depth: sums over the batch of per-sample
        # terms that partly cancel; L = 12, batch 8: 12 %, measured <= 9.9 % in eval mode, test_bert_base_12_layers_batch8_gradients_without_calibrator)
        enc_side = n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions"))
        tol_n = grad_tol if enc_side or head_grad_tol is None else head_grad_tol
        assert dev < max(tol_n, 3.0 * dev_emul) or float((g - og).norm()) < 2e-4, (n, dev, dev_emul, float(og.norm()))
        if dev > grad_tol:
            loose.append((n, round(dev, 3), round(dev_emul, 3)))
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1)))
        if cos < worst[0]:
            worst = (cos, n)
    print("gradients checked against the bf16-emulation calibrator instead of the 4.5% bound:", loose)
    return worst


def test_cfg1_matches_oracle_forward_backward():
    m, out, worst = check_against_oracle(CFG1, 2, 50, 64, 64, seed=1)
    print("worst gradient cosine", worst)


def test_cfg1_matches_reference_golden(golden_dir):
    """Direct comparison with the numbers the REAL reference produced (tests/golden/make_golden.py)."""
    for name in ("cfg1_T50_P64", "cfg1_T50_P50"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        B, T, Pv, Pa, seed = (int(x) for x in g["meta"])
        batch = synthetic_batch(B, T, Pv, Pa, seed=seed)
        m = build(CFG1)
        out, logits = m(**batch_to(batch, DEV))
        assert rel(out[0], g["joint_loss"]) < 3e-3 and rel(out[4], g["ap_loss"]) < 3e-3
        assert rel(out[5], g["label_loss"]) < 3e-3 and rel(out[6], g["nce"]) < 3e-3
        assert np.abs(logits.detach().float().cpu().numpy() - g["logits"]).max() < 2e-2
        for pi, tag in enumerate("tvs"):
            sc = out[7 + 2 * pi].detach().float().cpu()
            assert np.abs(sc[:, :, :48].numpy() - g[f"{tag}_scores_head"]).max() < 3e-2
            assert np.abs(torch.logsumexp(sc, -1).numpy() - g[f"{tag}_scores_lse"]).max() < 2e-2
        out[0].mean().backward()
        nograd = sorted(n for n, q in m.named_parameters() if float(q.grad.abs().sum()) == 0.0)
        assert nograd == list(g["nograd"])
        for n, q in m.named_parameters():
            if n in nograd or "attention.self.key.bias" in n:       # key.bias: true gradient is 0 (noise only)
                continue
            if not n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")):
                continue        # ill-conditioned [B,H] head gradients: bounded by the calibrator in check_against_oracle
            gn = float(g["gnorm/" + n])
            assert abs(float(q.grad.norm()) - gn) < 0.03 * gn + 1e-7, n


def test_bert_base_shapes_two_layers_match_oracle():
    cfg = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    check_against_oracle(cfg, 2, 50, 500, 500, seed=5)


def test_bert_large_width_matches_oracle():
    """The reference's own default width (TEXTDIM = 1024, CPC x_size 1024: REF:config.py:12, REF:MMBertForPretraining.py:327-344
    -- bert-large: 16 heads, I = 4096), two layers."""
    cfg = dict(hidden=1024, layers=2, heads=16, intermediate=4096, vocab=8192, dataset="mosei", alpha=1.0, beta=1.0)
    check_against_oracle(cfg, 2, 50, 96, 80, seed=8)


def test_mosi_dims_and_unequal_pair_lengths():
    cfg = dict(CFG1, dataset="mosi", vocab=4096, alpha=0.5, beta=0.25)
    check_against_oracle(cfg, 3, 24, 70, 33, seed=6)


def test_long_fusion_stress_and_ur_funny_dims_match_oracle():
    """BASELINE configs[3] shape class (A = V = 1375: S = 1425 per joint pass, attention tiles far past one LDS tile) at a
    width the CPU oracle finishes in seconds, with the UR-FUNNY feature widths (371 / 81: odd, not multiples of 8)."""
    cfg = dict(hidden=256, layers=1, heads=4, intermediate=1024, vocab=2048, dataset="ur_funny", alpha=1.0, beta=1.0)
    check_against_oracle(cfg, 2, 50, 1375, 1375, seed=9)


def test_full_size_long_fusion_step_is_finite_and_seeded():
    """BASELINE configs[3] at full size (12-layer d=768, T=50, A=V=1375, batch 4, train mode): too large for the CPU oracle,
    so the size-independent properties: finite losses, finite gradients on every parameter the reference differentiates,
    and the same seed -> the same loss (up to the fp32 atomics of the loss sums)."""
    cfg = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    batch = batch_to(synthetic_batch(4, 50, 1375, 1375, seed=21), DEV)
    m = build(cfg, train=True)
    m.manual_seed(5)
    out, logits = m(**batch)
    a = float(out[0])
    out[0].mean().backward()
    torch.cuda.synchronize()
    assert all(np.isfinite(float(out[i])) for i in (0, 4, 5, 6)) and bool(torch.isfinite(logits).all())
    assert tuple(out[9].shape) == (4, 1425, cfg["vocab"])
    nz = 0
    for n, q in m.named_parameters():
        assert bool(torch.isfinite(q.grad).all()), n
        nz += int(float(q.grad.abs().sum()) > 0.0)
    assert nz >= len(list(m.named_parameters())) - 6            # the six parameters the reference never differentiates
    m.zero_grad()
    m.manual_seed(5)
    b = float(m(**batch)[0][0])
    assert abs(b - a) <= 1e-6 * abs(a)                        # same seed -> same masks; the loss sums themselves use fp32 atomics


def test_full_size_fused_long_sequence_step_is_finite_and_seeded():
    """BASELINE configs[3] in its fused form at full size (12-layer d=768, ONE sequence text | visual | speech of S = 50 + 1375 + 1375 =
    2800 per sample, batch 4, train mode): far too large for the CPU oracle, so the size-independent properties -- finite losses and
    gradients on every parameter the reference differentiates, the same seed -> the same loss, and the row-set packing really
    skipping the padding of BOTH pair blocks (pair lengths are drawn from half to full: ~25 % of the rows)."""
    cfg = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    fb = batch_to(to_fused(synthetic_batch(4, 50, 1375, 1375, seed=23)), DEV)
    m = build(cfg, train=True)
    m.manual_seed(6)
    seen, orig = [], m._split_layout
    m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
    out, logits = m.forward_fused(**fb)
    a = float(out[0])
    out[0].mean().backward()
    torch.cuda.synchronize()
    m._split_layout = orig
    assert all(np.isfinite(float(out[i])) for i in (0, 4, 5, 6)) and bool(torch.isfinite(logits).all())
    assert tuple(out[7].shape) == (4, 2800, cfg["vocab"])
    lay = seen[0]
    assert lay is not None and lay.rows_a < 0.9 * lay.tokens and lay.tokens == 4 * 2800
    # both blocks contribute: more rows are skipped than the trailing (speech) padding alone accounts for
    sp_mask = fb["attention_mask"][2][:, :, 0]
    tail_pad = int((sp_mask == 0).sum())
    assert lay.tokens - lay.rows_a > tail_pad
    nz = 0
    for n, q in m.named_parameters():
        assert bool(torch.isfinite(q.grad).all()), n
        nz += int(float(q.grad.abs().sum()) > 0.0)
    assert nz >= len(list(m.named_parameters())) - 6
    m.zero_grad()
    m.manual_seed(6)
    b = float(m.forward_fused(**fb)[0][0])
    assert abs(b - a) <= 1e-6 * abs(a)


def test_fused_sequence_extension_matches_its_oracle():
    """forward_fused (text | visual | speech in ONE sequence: a declared extension, BASELINE's "fused seq_len~1050" shape class)
    against oracle.fused_forward, the same extension of the CPU restatement: losses, regression logits, MLM scores, and the
    gradients of every embedding / encoder / MLM-head parameter; unequal block lengths (the two pair blocks sit at different
    offsets of every sequence)."""
    cfg = dict(CFG1, vocab=4096)
    batch = to_fused(synthetic_batch(3, 24, 70, 33, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=12))
    p = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
    ocfg = dict(cfg, hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    oout, ologits = O.fused_forward(p, ocfg, **batch)
    oout[0].mean().backward()
    m = build(cfg)
    out, logits = m.forward_fused(**batch_to(batch, DEV))
    for i, name in ((0, "joint"), (4, "ap"), (5, "label"), (6, "nce")):
        assert rel(out[i].detach(), oout[i].detach()) < 3e-3, (name, float(out[i]), float(oout[i]))
    assert float((logits.float().cpu() - ologits.detach()).abs().max()) < 2e-2
    assert tuple(out[7].shape) == tuple(oout[7].shape) == (3, 24 + 70 + 33, cfg["vocab"])
    assert float((out[7].float().cpu() - oout[7].detach()).abs().max()) < 3e-2
    assert float((out[8].float().cpu() - oout[8].detach()).abs().max()) < 2e-2
    out[0].mean().backward()
    torch.cuda.synchronize()
    checked = 0
    for n, q in m.named_parameters():
        og = p[n].grad
        if og is None or float(og.abs().sum()) == 0.0 or "attention.self.key.bias" in n:
            continue
        if not n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")):
            continue            # [B,H]-sized head gradients: ill-conditioned at init, bounded in check_against_oracle's calibrated form
        dev_ = float((q.grad.float().cpu() - og).norm() / og.norm())
        assert dev_ < 0.045, (n, dev_)
        checked += 1
    assert checked > 30
    # both projection matrices received a gradient (each pair block found its rows)
    assert float(m.bert.jointEmbeddings.Wv.weight.grad.abs().sum()) > 0 and float(m.bert.jointEmbeddings.Ws.weight.grad.abs().sum()) > 0
    # the same step without the returned scores (sparse MLM head forward, masked-out tail rows left out): same losses, same gradients
    g_full = {n: q.grad.detach().float().clone() for n, q in m.named_parameters()}
    m2 = build(cfg)
    m2.return_scores = False
    out2, logits2 = m2.forward_fused(**batch_to(batch, DEV))
    out2[0].mean().backward()
    assert out2[7] is None and abs(float(out2[0]) - float(out[0])) <= 2e-6 * abs(float(out[0]))
    for n, q in m2.named_parameters():
        if "attention.self.key.bias" in n:
            continue
        same_grads(q.grad, g_full[n], n)


def test_fused_sequence_packs_over_a_row_set():
    """forward_fused with padding in BOTH pair blocks (the visual block's padded rows sit in the middle of the sequence) and labels on
    padded rows: the valid-first packing over the row SET (default) against the prefix packing (``fused_rowset_packing = False``,
    which can only skip the trailing speech padding) and against no packing at all.  A sequence run in its own valid-first order
    computes the same function up to rounding (attention sums its keys in another order -> some context values round to the other
    bf16 neighbour, 2^-9 relative, and two layers carry that on), so: losses to 2e-4 (measured 3e-5), prediction scores to two
    bf16 ulps of the largest logit, every parameter gradient to 1.5 % of its L2 norm (the prefix form, same key order as
    the dense one: 4e-3 of its largest entry) -- and backward visits fewer rows than the prefix form."""
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    b3 = synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=43)
    fb = to_fused(b3)
    lab = fb["masked_labels"].clone()
    lab[0, 24 + 199] = 77                                   # a label on the last (padded) visual row of sample 0: mid-sequence
    lab[1, -1] = 5                                          # ... and on the last (padded) speech row of sample 1
    fb = batch_to(dict(fb, masked_labels=lab), DEV)
    res = {}
    for mode in ("rowset", "prefix", "dense"):
        m = build(cfg)
        m.fused_rowset_packing = mode == "rowset"
        m.skip_padded_backward = mode != "dense"
        seen, orig = [], m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen, **k: (_s.append(_o(*a, **k)), _s[-1])[1]
        out, logits = m.forward_fused(**fb)
        out[0].mean().backward()
        torch.cuda.synchronize()
        res[mode] = (out, logits, {n: q.grad.detach().float().clone() for n, q in m.named_parameters()}, seen[0])
    lr, lp = res["rowset"][3], res["prefix"][3]
    assert lr is not None and lp is not None and res["dense"][3] is None
    assert lr.rows_a < 0.9 * lp.rows_a and lr.rows_a < 0.85 * lr.tokens, (lr.rows_a, lp.rows_a, lr.tokens)      # the mid-sequence padding is skipped too
    ob, gb = res["dense"][0], res["dense"][2]
    for mode in ("rowset", "prefix"):
        oa, ga = res[mode][0], res[mode][2]
        for i in (0, 4, 5, 6):
            assert abs(float(oa[i]) - float(ob[i])) <= (2e-4 if mode == "rowset" else 1e-5) * abs(float(ob[i])), (mode, i, float(oa[i]), float(ob[i]))
        d = float((oa[7].float() - ob[7].float()).abs().max())
        assert d <= 2.0 ** -6 * float(ob[7].float().abs().max()), (mode, d)                # bf16 ulps of the largest logit
        worst = (0.0, None)
        for n in ga:
            if "attention.self.key.bias" in n:
                continue
            scale = float(gb[n].abs().max())
            e = float((ga[n] - gb[n]).abs().max()) / (scale + 1e-12)
            worst = max(worst, (e, n))
            if mode == "prefix":
                same_grads(ga[n], gb[n], (mode, n))
            else:                                           # another key order: single entries flip; bounded in the L2 sense, 1.5 % of the norm
                l2 = float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-12))
                enc = n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions"))
                # ([B,H]-sized head gradients at batch 4: a flipped bf16 digit of a [CLS] hidden state moves them by per cent, see
                # test_bert_base_12_layers_batch8_gradients_without_calibrator)
                assert l2 <= (1.5e-2 if enc else 0.1) or float((ga[n] - gb[n]).norm()) < 1e-4, (mode, n, l2, e)      # (1e-4: the CPC gradients at init, norm ~1e-5)
                worst_l2 = max(locals().get("worst_l2", (0.0, None)), (l2, n)) if enc else locals().get("worst_l2", (0.0, None))
        if mode == "rowset":
            print("rowset worst encoder-side L2 deviation", worst_l2)
        print(mode, "worst gradient deviation relative to the largest entry", worst)


@pytest.mark.parametrize("size", ["small", "timed_depth"])
def test_backward_on_unmasked_rows_only_equals_full_backward(size):
    """The valid-first packing (ops.SplitLayout): rows behind a sequence's last unmasked key and without a label have zero
    gradients in every layer, so backward runs on the other rows only.  Same model, same batch (heavy padding: pair lengths
    from half to full), dropout off: losses and scores identical, every parameter gradient equal to the full backward up to
    fp32 summation order; a label placed on a padded row switches the short cut off (and the gradients still agree).
    ``timed_depth`` (round 5): the same proof at the BENCHMARKED model -- 12 layers, d = 768, 12 heads, T = 50, A = V = 500 (batch 4:
    4 600 packed rows) -- where the induction "zero in every layer" runs over twelve layers and the skipped rows meet the 192- / 224-row
    GEMM tiles and the one-call weight gradients of the timed step."""
    if size == "small":
        cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
        T, Pv, Pa = 24, 200, 130
    else:
        cfg = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
        T, Pv, Pa = 50, 500, 500
    batch = synthetic_batch(4, T, Pv, Pa, dataset="mosei", vocab=cfg["vocab"], seed=31)
    dbatch = batch_to(batch, DEV)
    res = {}
    for skip in (True, False):
        m = build(cfg)
        m.skip_padded_backward = skip
        seen = []
        orig = m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
        out, logits = m(**dbatch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        assert (seen[0] is not None) == skip
        if skip:
            assert seen[0].rows_a < 0.9 * seen[0].tokens                 # a real saving on this batch
        res[skip] = (out, logits, {n: q.grad.detach().float().clone() for n, q in m.named_parameters()})
    (oa, la, ga), (ob, lb, gb) = res[True], res[False]
    for i in (0, 4, 5, 6):
        assert abs(float(oa[i]) - float(ob[i])) <= 1e-6 * abs(float(ob[i]))
    for k in (7, 9, 11):
        assert torch.equal(oa[k], ob[k])                                    # forward is the same arithmetic row by row
    for n in ga:
        if "attention.self.key.bias" in n:                  # true gradient 0: rounding noise on both sides
            continue
        same_grads(ga[n], gb[n], n)
    # labels on rows behind the last unmasked key (what the reference's pipeline produces: mask_tokens selects [PAD] positions,
    # trainer.py copies text labels onto pair positions): such a row is a query WITH a gradient, so its sequence keeps every row up
    # to its last labelled one in the leading region -- per sequence, the others keep their saving -- and the gradients still
    # equal the full backward
    lab_v = batch["masked_labels"][1].clone()
    lab_v[0, -1] = 1234                                                     # sample 0: label on its very last (padded) pair row
    lab_t = batch["masked_labels"][0].clone()
    lab_t[1, -1] = 0                                                        # sample 1: a labelled [PAD] row of the text pass
    b2 = dict(batch, masked_labels=(lab_t, lab_v, batch["masked_labels"][2]))
    res2 = {}
    for skip in (True, False):
        m = build(cfg)
        m.skip_padded_backward = skip
        seen = []
        orig = m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
        out, _ = m(**batch_to(b2, DEV))
        out[0].mean().backward()
        torch.cuda.synchronize()
        if skip:
            lay = seen[0]
            assert lay is not None and lay.rows_a < 0.95 * lay.tokens            # still a saving: only two sequences grew
            B_ = 4
            assert lay.valid_host[B_ + 0] == T + Pv and lay.valid_host[1] == T     # the two labelled sequences keep all their rows
        res2[skip] = (out, {n: q.grad.detach().float().clone() for n, q in m.named_parameters()})
    for i in (0, 4, 5, 6):
        assert abs(float(res2[True][0][i]) - float(res2[False][0][i])) <= 1e-6 * abs(float(res2[False][0][i]))
    for n in res2[True][1]:
        if "attention.self.key.bias" in n:
            continue
        same_grads(res2[True][1][n], res2[False][1][n], n)


@pytest.mark.parametrize("train", [False, True])
def test_sparse_backward_of_the_top_layer_equals_dense(train):
    """Only the MLM-labelled rows and the [CLS] rows of the top encoder layer's output have a gradient, so its output sublayer,
    LayerNorms and output projection run their backward on those rows only.  Against the dense backward of the same model on
    the same batch -- also in TRAIN mode with the same seed (the compact rows must regenerate the dropout masks of their
    original rows): every parameter gradient equal up to fp32 summation order."""
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    dbatch = batch_to(synthetic_batch(4, 24, 120, 90, dataset="mosei", vocab=cfg["vocab"], seed=33), DEV)
    res = {}
    for sparse in (True, False):
        m = build(cfg, train=train)
        m.sparse_top_layer_backward = sparse
        m.manual_seed(17)
        from msa_amd import model as MM
        calls, orig = [], MM._EncoderFn._last_layer_sparse
        MM._EncoderFn._last_layer_sparse = staticmethod(lambda *a, _o=orig, _c=calls: (_c.append(1), _o(*a))[1])
        try:
            out, _ = m(**dbatch)
            out[0].mean().backward()
        finally:
            MM._EncoderFn._last_layer_sparse = staticmethod(orig)
        torch.cuda.synchronize()
        assert len(calls) == (1 if sparse else 0)                        # the short cut really ran (once: the top layer)
        res[sparse] = (float(out[0]), {n: q.grad.detach().float().clone() for n, q in m.named_parameters()})
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])        # same forward (fp32 atomics in the loss sums)
    for n in res[True][1]:
        if "attention.self.key.bias" in n:                  # true gradient 0: rounding noise on both sides
            continue
        same_grads(res[True][1][n], res[False][1][n], n)      # (the compact path rounds its few-row products to bf16 at other points: split-K + residual)


def test_inference_dedupes_masked_rows_exactly():
    """Without dropout and autograd the masked-out rows of a sequence are identical in every layer (same input, same keys), so
    inference keeps one of them per sequence: every output -- losses, regression logits, all prediction scores incl. those of
    the padded positions -- is bit-identical to the full computation."""
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    dbatch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=35), DEV)
    m = build(cfg)
    outs = {}
    for dd in (True, False):
        m.dedupe_masked_rows = dd
        seen = []
        orig = m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
        with torch.no_grad():
            outs[dd] = m(**dbatch)
        m._split_layout = orig
        assert (seen[0] is not None) == dd
        if dd:
            assert seen[0].rows_packed < 0.9 * seen[0].tokens
    (oa, la), (ob, lb) = outs[True], outs[False]
    assert torch.allclose(la, lb, rtol=1e-5, atol=1e-7)           # (the heads' dense layers sum their split products with fp32 atomics)
    for i in (0, 4, 5, 6):
        assert abs(float(oa[i]) - float(ob[i])) <= 1e-6 * abs(float(ob[i]))      # (the loss sums use fp32 atomics: not bit-stable run to run)
    for k in (7, 9, 11):
        assert torch.equal(oa[k], ob[k]), k                                      # every prediction score, bit for bit
    for k in (8, 10, 12):
        assert torch.allclose(oa[k], ob[k], rtol=1e-5, atol=1e-7), k            # relationship scores: summed with fp32 atomics


@pytest.mark.parametrize("train", [False, True])
def test_training_without_returned_scores_equals_the_faithful_step(train):
    """model.return_scores = False (what trainer.py needs: it never reads outputs[7..12]'s score tensors): the MLM head runs on the
    labelled rows only -- forward too -- and the encoder leaves out the rows that only the returned scores would read.  Losses
    and every parameter gradient equal the faithful step's (also in TRAIN mode with the same seed: the rows that remain keep
    their packed positions, hence their dropout masks); the score slots are None."""
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    dbatch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=37), DEV)
    res = {}
    for scores in (True, False):
        m = build(cfg, train=train)
        m.return_scores = scores
        m.manual_seed(23)
        seen, orig = [], m._split_layout
        m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
        out, logits = m(**dbatch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        assert (out[7] is None and out[9] is None and out[11] is None) == (not scores)
        assert seen[0] is not None and seen[0].dropped == (not scores)
        if not scores:
            assert seen[0].rows_packed == seen[0].rows_a < 0.9 * seen[0].tokens
        res[scores] = ([float(out[i]) for i in (0, 4, 5, 6)], logits.detach().float().clone(),
                       {n: q.grad.detach().float().clone() for n, q in m.named_parameters()})
    for a, b in zip(res[False][0], res[True][0]):
        assert abs(a - b) <= 2e-6 * abs(b) + 1e-7, (res[False][0], res[True][0])
    assert float((res[False][1] - res[True][1]).abs().max()) <= 1e-5
    for n in res[True][2]:
        if "attention.self.key.bias" in n:
            continue
        same_grads(res[False][2][n], res[True][2][n], n)


def test_dropout_train_mode_is_seeded_and_unbiased():
    batch = batch_to(synthetic_batch(2, 50, 64, 64, seed=1), DEV)
    m = build(CFG1, train=True)
    m.manual_seed(11)
    a = float(m(**batch)[0][0])
    b = float(m(**batch)[0][0])
    m.manual_seed(11)
    c = float(m(**batch)[0][0])
    # same seed -> same masks (the CE loss sum uses fp32 atomics, so allow last-bit differences)
    assert abs(a - c) < 1e-5 * abs(a) and abs(a - b) > 1e-4 * abs(a)
    m.eval()
    e = float(m(**batch)[0][0])
    vals = []
    m.train()
    for _ in range(8):
        vals.append(float(m(**batch)[0][0]))
    assert all(np.isfinite(vals)) and abs(np.mean(vals) - e) < 0.15 * abs(e)
    out, _ = m(**batch)
    out[0].mean().backward()
    assert all(torch.isfinite(q.grad).all() for q in m.parameters())


# (long2 with the short cuts OFF went with the GPU-suite budget, round 6: 32 s of fp32 oracle for the dense backward at S = 1425, which
# test_full_size_long_fusion_step_is_finite_and_seeded and the dense cases of the three other shapes cover)
@pytest.mark.parametrize("case,shortcuts", [(c_, s_) for c_ in ("cfg1", "base2", "deep4", "long2") for s_ in (True, False) if not (c_ == "long2" and not s_)])
def test_train_mode_step_matches_oracle_with_replayed_masks(case, shortcuts):
    """The BENCHMARKED configuration -- model.train(), dropout 0.1 / 0.1 / 0.5 on, as REF:trainer.py:40,66,83 runs it -- end to end
    against the oracle: the HIP step's keep masks of every site (embeddings, JointEmbeddings, and per layer and pass the attention
    probabilities [B,h,S,S] and both hidden dropouts) are rebuilt from the model's (seed, site) mapping and replayed in the oracle
    (oracle._dropout), so the two sides compute the same function: the 4 losses at 3e-3, regression logits, prediction scores and
    EVERY parameter gradient at the eval-mode tolerances.  A wrong site or seed handed to a backward launch, a swapped h1 / h2 drop
    tuple or a mask indexed by the wrong row order gives gradients that fail here (and nowhere in eval mode).
    cfg1 = BASELINE configs[0]'s model (B=2, T=50, P=64); base2 = two layers of configs[1] (d=768, T=50, A=V=500); long2 = two layers at
    configs[3]'s lengths (A=V=1375).  All with the
    default exact-zero short cuts (valid-first packing: hidden-dropout masks follow the packed row order; sparse top-layer
    backward: masks of the ORIGINAL rows regenerated on gathered rows) and with them off (dense backward on every row).
    deep4 (round 4) = FOUR layers (d=256, B=2, T=24, unequal pair lengths 70 / 33): the only depth at which the dropout sites 8i + k
    exist for i >= 2 and at which backward pairs two layers' weight gradients into one 8-problem launch (model.pair_wgrads: layers
    (2, 1) with the sparse top layer, (3, 2) and (1, 0) with the short cuts off) WITH dropout on -- a swapped pair index or a site
    handed to the wrong layer's launch fails here and nowhere at L = 2."""
    if case == "cfg1":
        cfg, shape = CFG1, (2, 50, 64, 64)
    elif case == "long2":
        # round 5: BASELINE configs[3]'s sequence lengths (A = V = 1375: S = 1425 per joint pass, 23 key tiles, a 17-row last query tile)
        # at two layers of the headline width and configs[3]'s batch 4, TRAIN mode -- the long-sequence ends of the dropout index arithmetic (12 x 1425 x 1428
        # elements per sequence), of the attention tile lists and of the packed row maps, with every mask replayed in the oracle
        cfg, shape = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0), (4, 50, 1375, 1375)
    elif case == "deep4":
        cfg, shape = dict(hidden=256, layers=4, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0), (2, 24, 70, 33)
    else:
        cfg, shape = dict(hidden=768, layers=2, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0), (2, 50, 500, 500)
    flags = {} if shortcuts else dict(skip_padded_backward=False, sparse_top_layer_backward=False)
    from msa_amd import model as MM
    calls, orig = [], MM._EncoderFn._last_layer_sparse
    MM._EncoderFn._last_layer_sparse = staticmethod(lambda *a, _o=orig, _c=calls: (_c.append(1), _o(*a))[1])
    try:
        from msa_amd import ops as _ops
        tn_calls, tn_orig = [], _ops.gemm_tn_grouped
        _ops.gemm_tn_grouped = lambda probs, *a, _o=tn_orig, _c=tn_calls, **k: (_c.append(len(probs)), _o(probs, *a, **k))[1]
        m, out, worst = check_against_oracle(cfg, *shape, seed={"cfg1": 1, "base2": 5, "deep4": 6, "long2": 9}[case], train=True, flags=flags)
    finally:
        MM._EncoderFn._last_layer_sparse = staticmethod(orig)
        _ops.gemm_tn_grouped = tn_orig
    if case == "deep4":                                            # the paired (8-problem) weight-gradient launches really ran
        assert tn_calls.count(8) == (1 if shortcuts else 2), tn_calls
    # the short cuts really ran when asked for (and only then)
    assert (m.last_split is not None) == shortcuts and len(calls) == (1 if shortcuts else 0), (m.last_split, calls)
    print("train-mode worst gradient cosine", worst)


@pytest.mark.parametrize("shortcuts", [True, False])
def test_train_mode_step_at_the_timed_depth_matches_oracle_with_replayed_masks(shortcuts):
    """Rounds 5-6: the replay test above at the configuration bench.py TIMES -- 12 layers, d = 768, 12 heads, I = 3072, vocabulary 30 522,
    T = 50, A = V = 500, model.train() with dropout 0.1 / 0.1 / 0.5 (REF:trainer.py:40,66,83).  With the short cuts ON (the timed step) at the
    TIMED BATCH, 16 (round 6: 18 400 packed rows, 2.75 attention rounds, the 44-problem weight-gradient call at full height; one fp32 oracle
    pass with fixed bounds; batch 8 on hosts below 70 GiB); with them OFF at batch 8 (9 200 packed rows: the smallest batch at which
    nt_choose's cost rule sends FFN-up + GELU and the GELU' input gradient to the MULTI-TILE 8-phase form on 224-row tiles, the form that
    carries the timed step; at batch 2 every launch is single-round, at batch 6 the rule takes 192-row tiles) with the bf16-storage
    calibrator.  Every dropout site 8i + k for i < 12 is replayed in the oracle; asserted besides the numbers: the weight gradients of all
    dense layers went out in ONE call (model._auto_defer_wgrads: 11 layers x 4 problems behind the sparse top layer, 12 x 4 with the short cuts
    off; the few-row weight gradients behind them: 50 problems) and mmbert_gemm_nt dispatched a forward and a backward shape to the multi-tile 8-phase form on 224-row tiles."""
    cfg = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)
    flags = {} if shortcuts else dict(skip_padded_backward=False, sparse_top_layer_backward=False)
    from msa_amd import model as MM
    from msa_amd import ops as _ops
    calls, orig = [], MM._EncoderFn._last_layer_sparse
    MM._EncoderFn._last_layer_sparse = staticmethod(lambda *a, _o=orig, _c=calls: (_c.append(1), _o(*a))[1])
    tn_calls, tn_orig = [], _ops.gemm_tn_grouped
    nt_shapes, nt_orig = set(), _ops.gemm_nt

    def nt_spy(A, B, **k):
        epi = (1 if k.get("bias") is not None else 0) | (2 if k.get("gelu") else 0) | (4 if k.get("resid") is not None else 0) | \
              (8 if k.get("gelu_bwd_u") is not None else 0) | (16 if k.get("out_f32") else 0)
        nt_shapes.add((A.shape[0], B.shape[0], A.shape[1], epi))
        return nt_orig(A, B, **k)

    try:
        _ops.gemm_tn_grouped = lambda probs, *a, _o=tn_orig, _c=tn_calls, **k: (_c.append(len(probs)), _o(probs, *a, **k))[1]
        _ops.gemm_nt = nt_spy
        # tolerances at L = 12 (re-derived at depth, test_bert_base_12_layers_match_oracle): losses 4e-3, scores 8e-2 max / 8e-3 mean (a bf16
        # score of magnitude >= 8 carries 3e-2 of rounding alone), regression logits 3e-2, encoder-side gradients max(6 %, 3 x the bf16-storage
        # calibrator), the [B, H]-sized head gradients max(12 %, 3 x calibrator) as in the eval-mode L = 12 batch-8 test
        # round 6: with the short cuts ON -- the step bench.py times -- at the TIMED BATCH, 16 (18 400 packed rows: 2.75
        # attention rounds, the 44-problem weight-gradient call at full height), one fp32 oracle pass, the fixed bounds of the eval-mode
        # batch-16 test (encoder side 8 %, heads 12 %, CPC 50 % and 1e-3 absolute); with them OFF at batch 8 with the bf16-storage calibrator
        if shortcuts and _host_mem_gib() >= 70:
            m, out, worst = check_against_oracle(cfg, 16, 50, 500, 500, seed=8, train=True, flags=flags, loss_tol=4e-3, score_tol=8e-2,
                                                 score_mean_tol=8e-3, logit_tol=3e-2, grad_tol=0.08, head_grad_tol=0.12,
                                                 report="parity_train_L12_B16", calibrate=False)
        else:
            m, out, worst = check_against_oracle(cfg, 8, 50, 500, 500, seed=8, train=True, flags=flags, loss_tol=4e-3, score_tol=8e-2,
                                                 score_mean_tol=8e-3, logit_tol=3e-2, grad_tol=0.06, head_grad_tol=0.12,
                                                 report="parity_train_L12_B8" + ("" if shortcuts else "_shortcuts_off"))
    finally:
        MM._EncoderFn._last_layer_sparse = staticmethod(orig)
        _ops.gemm_tn_grouped = tn_orig
        _ops.gemm_nt = nt_orig
    assert (m.last_split is not None) == shortcuts and len(calls) == (1 if shortcuts else 0), (m.last_split, calls)
    # one call for all dense layers, no paired launches; round 6: the few-row weight gradients ride in it (the tied decoder's and the MLM
    # transform's; with the sparse top layer its three few-row ones and its QKV gradient as a 45th long problem): 50 problems either way
    assert tn_calls == [50], tn_calls
    multi = {}
    for (M, N, K, epi) in nt_shapes:
        d = _ops.gemm_nt_describe(M, N, K, epi)
        if d["kernel"] == "8phase" and d["tiles"] > d["workgroups"] and N < 30000:
            multi[(N, K, epi)] = (M, d["tile"])
    print("multi-tile 8-phase launches (N, K, epilogue) -> (rows, tile):", multi)
    assert multi.get((3072, 768, 3), (0, ""))[1] == "224x256", multi                       # FFN-up + bias + GELU (+ pre-activation store), forward
    # the GELU' input gradient, backward: on all 9 200 rows with the short cuts off (224-row tiles), on the ~ 7 000 unmasked ones with them
    # on (whatever height the rule takes there)
    assert (3072, 768, 8) in multi and (shortcuts or multi[(3072, 768, 8)][1] == "224x256"), multi
    assert (2304, 768, 1) in multi, multi                                                  # QKV + bias
    print("train-mode worst gradient cosine at L = 12", worst)


@pytest.mark.parametrize("train", [False, True])
def test_composite_layer_calls_are_bit_identical(train):
    """Round 5: mmbert_layer_fwd / mmbert_layer_bwd issue a layer's forward (7 launches) and the dense part of its backward (7) from ONE C
    call each -- the same kernels with the same arguments.  Against the per-launch path (model.composite_layers = False) in deterministic
    mode (so that the loss sums do not depend on atomics' arrival order): losses, regression logits, prediction scores and the whole flat
    gradient buffer BIT-identical, eval and train mode (dropout seeds handed through the structures), short cuts on (split layout, sparse
    top layer: L - 1 composite backward calls) and off (L calls); and the composite path really ran."""
    from msa_amd import ops as _ops
    cfg = dict(hidden=256, layers=3, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    dbatch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=61), DEV)
    was = _ops.deterministic()
    try:
        _ops.set_deterministic(True)
        for shortcuts in (True, False):
            res = {}
            for comp in (True, False):
                m = build(cfg, train=train)
                m.manual_seed(29)
                m.composite_layers = comp
                if not shortcuts:
                    m.skip_padded_backward = m.sparse_top_layer_backward = False
                calls = {"fwd": 0, "bwd": 0}
                of, ob = _ops.layer_fwd, _ops.layer_bwd
                _ops.layer_fwd = lambda *a, _o=of: (calls.__setitem__("fwd", calls["fwd"] + 1), _o(*a))[1]
                _ops.layer_bwd = lambda *a, _o=ob: (calls.__setitem__("bwd", calls["bwd"] + 1), _o(*a))[1]
                try:
                    out, logits = m(**dbatch)
                    out[0].mean().backward()
                    torch.cuda.synchronize()
                finally:
                    _ops.layer_fwd, _ops.layer_bwd = of, ob
                assert calls == ({"fwd": 3, "bwd": 2 if shortcuts else 3} if comp else {"fwd": 0, "bwd": 0}), (comp, shortcuts, calls)
                res[comp] = ([out[i].detach().clone() for i in (0, 4, 5, 6)], logits.detach().clone(), [out[k].detach().clone() for k in (7, 9, 11)],
                             m._flat.grads.clone())
            for x, y in zip(res[True][0], res[False][0]):
                assert torch.equal(x, y), (shortcuts, float(x), float(y))
            assert torch.equal(res[True][1], res[False][1])
            for x, y in zip(res[True][2], res[False][2]):
                assert torch.equal(x, y)
            assert torch.equal(res[True][3], res[False][3]), float((res[True][3] - res[False][3]).abs().max())
    finally:
        _ops.set_deterministic(was)


def test_deferred_weight_gradients_equal_the_per_layer_launches():
    """Round 4: without a gradient hook (one GPU) every dense layer's weight gradients go out in ONE call at the end of backward
    (model.defer_wgrads, whole rounds of tiles) instead of per layer pair: the same products on the same operands -- every parameter
    gradient equal to the paired form's up to the fp32 accumulation order of the token split that the deferred form no longer has
    (layer 0 alone splits the token axis in the paired form).  Train mode (dropout on), five layers."""
    cfg = dict(hidden=256, layers=5, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    batch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=52), DEV)
    from msa_amd import ops as _ops
    grads, calls = {}, {}
    for defer in (True, False):
        m = build(cfg, train=True)
        m.manual_seed(12)
        m.defer_wgrads = defer
        seen, orig = [], _ops.gemm_tn_grouped
        _ops.gemm_tn_grouped = lambda probs, *a, _o=orig, _c=seen, **k: (_c.append(len(probs)), _o(probs, *a, **k))[1]
        try:
            out, _ = m(**batch)
            out[0].mean().backward()
            torch.cuda.synchronize()
        finally:
            _ops.gemm_tn_grouped = orig
        grads[defer], calls[defer] = {n: q.grad.float().clone() for n, q in m.named_parameters()}, seen
    # 4 dense layers behind the sparse top layer: one call of 16 + (round 6) the top layer's QKV gradient, its three few-row ones and the MLM head's two
    assert calls[True] == [22] and 8 in calls[False] and 22 not in calls[False], calls
    for n in grads[True]:
        if "attention.self.key.bias" in n:
            continue
        same_grads(grads[True][n], grads[False][n], n)


def test_two_graphs_in_flight_one_backward_equals_two_backward_passes():
    """Round 6: two forward passes, then ONE backward of the sum of their losses.  autograd then runs both heads' and both MLM heads' backward
    stages before either trunk's: the heads' side stream forks twice before the first join, and the few-row weight gradients of BOTH graphs
    wait for the deferred call -- two writers of the tied decoder's, the MLM transform's and the top layer's gradients, which must never
    share a launch (model._late_wgrad flushes the first set).  Against the same two forward passes differentiated one after the other (the
    second accumulates): the same gradients up to fp32 summation order.  Train mode, forced deferred call, five layers."""
    cfg = dict(hidden=256, layers=5, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    b1 = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=61), DEV)
    b2 = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=62), DEV)
    grads = {}
    for mode in ("sum", "apart"):
        m = build(cfg, train=True)
        m.manual_seed(21)
        m.defer_wgrads = True
        o1, _ = m(**b1)
        if mode == "apart":
            o1[0].mean().backward()
        o2, _ = m(**b2)
        if mode == "apart":
            o2[0].mean().backward()
        else:
            (o1[0] + o2[0]).backward()
        torch.cuda.synchronize()
        assert not m.__dict__.get("_late_wgrads") and not m.__dict__.get("_heads_join") and m.__dict__.get("_wgrad_join") is None
        grads[mode] = {n: q.grad.float().clone() for n, q in m.named_parameters() if q.grad is not None}
    assert grads["sum"].keys() == grads["apart"].keys()
    for n in grads["sum"]:
        if "attention.self.key.bias" in n:
            continue
        same_grads(grads["sum"][n], grads["apart"][n], n)


def test_one_layernorm_reduce_per_backward_and_no_leftovers_after_a_failed_one():
    """Round 4: every LayerNorm' of a backward pass (MLM head, sparse top layer, dense layers, embedding stage) shares ONE collector:
    ONE mmbert_ln_bwd_reduce_rows launch per backward (the trunk's flush at its end) instead of one per call, the
    same gamma / beta gradients as with a reduce per call; sums a half-finished backward left behind are dropped by the next forward,
    not folded into its gradients."""
    cfg = dict(hidden=256, layers=3, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    batch = batch_to(synthetic_batch(4, 24, 200, 130, dataset="mosei", vocab=cfg["vocab"], seed=53), DEV)
    from msa_amd import ops as _ops
    ln_names = lambda m: [n for n, _ in m.named_parameters() if "LayerNorm" in n]
    grads, flushes = {}, {}
    for shared in (True, False):
        m = build(cfg, train=False)
        calls, orig_flush, orig_slot = [], _ops.LnDeferred.flush, _ops.LnDeferred.slot
        _ops.LnDeferred.flush = lambda self, _o=orig_flush, _c=calls: (_c.append(len(self.items)), _o(self))[1]
        if not shared:                                             # a reduce per call: flush right behind every slot
            orig_ln = _ops.ln_bwd
            _ops.ln_bwd = lambda *a, _o=orig_ln, **k: (lambda r, d: (d.flush() if d is not None else None, r)[1])(_o(*a, **k), k.get("deferred"))
        try:
            if shared:
                out, _ = m(**batch)                                # (a first pass grows the collector's workspace, which may flush early)
                out[0].mean().backward()
                m._flat.grads.zero_()
                calls.clear()
                lnd = m._shared_lnd()                              # leftovers of a backward that "raised": a stale call in the collector
                H = cfg["hidden"]
                x = torch.randn(50, H, device=DEV).bfloat16()
                _, mean, rstd = _ops.ln_fwd(x, torch.ones(H, device=DEV), torch.zeros(H, device=DEV), 1e-12)
                stale_g, stale_b = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
                _ops.ln_bwd(x, x, mean, rstd, torch.ones(H, device=DEV), stale_g, stale_b, deferred=lnd)
                assert len(lnd.items) == 1
            out, _ = m(**batch)
            if shared:
                assert len(m._shared_lnd().items) == 0             # dropped at the forward
            out[0].mean().backward()
            torch.cuda.synchronize()
        finally:
            _ops.LnDeferred.flush = orig_flush
            if not shared:
                _ops.ln_bwd = orig_ln
        if shared:
            assert float(stale_g.abs().max()) == 0.0 and float(stale_b.abs().max()) == 0.0
        grads[shared] = {n: dict(m.named_parameters())[n].grad.float().clone() for n in ln_names(m)}
        flushes[shared] = [c for c in calls if c]
    assert len(flushes[True]) == 1 and sum(flushes[True]) == sum(flushes[False]) and len(flushes[False]) == sum(flushes[False]), flushes
    for n in grads[True]:
        assert float(grads[True][n].abs().max()) > 0
        same_grads(grads[True][n], grads[False][n], n)


def test_sparse_mlm_backward_equals_dense_backward():
    """The MLM head's backward over the labelled rows only (default) against the dense backward over all rows: the
    CE gradient of an unlabelled row is exactly zero, so every parameter gradient must agree up to fp32 summation order."""
    batch = batch_to(synthetic_batch(2, 50, 64, 64, seed=3), DEV)
    grads = []
    for sparse in (True, False):
        m = build(CFG1)
        m.sparse_mlm_backward = sparse
        out, _ = m(**batch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        grads.append({n: q.grad.float().clone() for n, q in m.named_parameters()})
    for n in grads[0]:
        if "attention.self.key.bias" in n:                  # true gradient 0 (softmax is shift invariant): rounding noise on both sides
            continue
        same_grads(grads[0][n], grads[1][n], n)
    lab = batch["masked_labels"]
    n_act = sum(int((x != -100).sum()) for x in lab)
    assert 0 < n_act < sum(x.numel() for x in lab) // 2          # the sparse path was really taken


@pytest.mark.parametrize("num_labels", [7, 1])
def test_fused_heads_equal_eager_heads(num_labels):
    """_HeadsFn (hand-written backward, csrc/heads.hip) against the eager autograd form of the same arithmetic (_heads):
    losses, returned scores and EVERY parameter gradient (3e-4 relative to the largest entry: the two forms differ by fp32
    summation order in the heads, and downstream of them one bf16 rounding flip of an activation gradient is 2^-9 relative
    on that element -- observed: a single 2^-14 difference in one word-embedding row at scale 0.47)."""
    batch = batch_to(synthetic_batch(4, 50, 64, 64, seed=9), DEV)
    res = []
    for fused in (True, False):
        m = build(CFG1)
        m.num_labels = num_labels
        m.fused_heads = fused
        m.set_alpha_beta(0.7, 1.3)
        out, logits_out = m(**batch)
        out[0].mean().backward()
        torch.cuda.synchronize()
        res.append((out, logits_out, {n: q.grad.float().clone() for n, q in m.named_parameters()}))
    (o1, l1, g1), (o2, l2, g2) = res
    for i in (0, 4, 5, 6):
        assert rel(o1[i], o2[i]) < 1e-5, (i, float(o1[i]), float(o2[i]))
    for i in (8, 10, 12):
        assert float((o1[i] - o2[i]).abs().max()) < 1e-5
    assert float((l1 - l2).abs().max()) < 1e-5
    for n in g1:
        if "attention.self.key.bias" in n:                  # true gradient 0: both sides hold rounding noise
            continue
        scale = float(g2[n].abs().max()) + 1e-12
        err = float((g1[n] - g2[n]).abs().max())
        # 3e-7 absolute: the CPC / classifier gradients at initialisation are differences of nearly equal 1e-3-sized terms
        # (norm ~1e-6, see DESIGN numerics): their last digits depend on the summation order on BOTH sides
        assert err <= 3e-4 * scale + 3e-7, f"{n}: err {err:.3e} scale {scale:.3e}"


def test_loss_mean_backward_is_the_reference_call_without_aten_launches():
    """Round 6.  The reference differentiates ``outputs[0].mean()`` (REF:trainer.py:83).  outputs[0] is a 0-dim
    ``_ScalarLoss``: ``mean()`` of it is the tensor itself and ``backward()`` seeds a cached device-side 1.0 -- the reduction, the fill and
    the division torch launches for that line are gone, the graph and the values are the same: gradients BIT-identical (deterministic mode)
    to the plain-tensor path (``torch.Tensor.mean`` on the same scalar, implicit ones gradient), ``.item()`` / arithmetic / ``mean(dim)``
    behave as on any tensor, and the pair features are read as float64 (the reference's dtype, rounded on load) -- modifying them in place
    between forward and backward raises like a saved tensor would."""
    from msa_amd import ops as _ops
    from msa_amd.model import _ScalarLoss
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    batch = batch_to(synthetic_batch(4, 24, 90, 70, dataset="mosei", vocab=cfg["vocab"], seed=77), DEV)
    assert batch["input_ids"][1].dtype == torch.float64 and batch["input_ids"][2].dtype == torch.float64      # collate's contract
    was = _ops.deterministic()
    grads = []
    try:
        _ops.set_deterministic(True)
        for plain in (False, True):
            m = build(cfg, train=True)
            m.manual_seed(3)
            out, _ = m(**batch)
            loss = out[0]
            assert isinstance(loss, _ScalarLoss) and loss.dim() == 0 and loss.mean() is loss
            assert isinstance(float(loss), float) and float(loss + 1.0) == pytest.approx(float(loss) + 1.0)
            assert loss.mean(dim=None).shape == () and float(loss.detach().mean()) == float(loss)
            if plain:
                torch.Tensor.mean(loss.as_subclass(torch.Tensor)).backward()        # what torch does for REF:trainer.py:83 on a plain tensor
            else:
                loss.mean().backward()
            torch.cuda.synchronize()
            grads.append(m._flat.grads.clone())
    finally:
        _ops.set_deterministic(was)
    assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().sum()) > 0.0
    # fp32 features (DeviceBatchBuilder's dtype) give the same step as float64 ones whose values are fp32-representable
    m = build(cfg)
    o64, _ = m(**batch)
    b32 = dict(batch, input_ids=tuple(t.float() if t.dtype == torch.float64 else t for t in batch["input_ids"]))
    o32, _ = m(**b32)
    assert abs(float(o64[0]) - float(o32[0])) <= 1e-6 * abs(float(o64[0]))
    # the caller's feature tensor is read again in backward: an in-place change in between is an error, not a silently wrong gradient
    out, _ = m(**batch)
    batch["input_ids"][1].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out[0].mean().backward()


def test_submodule_api_matches_oracle():
    """MMBertModel.forward(joint) / JointEmbeddings.forward / heads via the reference's call sites."""
    cfg = CFG1
    batch = synthetic_batch(2, 50, 64, 64, seed=1)
    p = O.seeded_params(cfg)
    m = build(cfg)
    ids, am = batch["input_ids"], batch["attention_mask"]
    ocfg = dict(cfg, hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    with torch.no_grad():
        seq, pooled = m.bert((ids[3].to(DEV), ids[1].to(DEV)), attention_mask=(am[1][0].to(DEV), am[1][1].to(DEV)), joint=True)
        oseq, opooled = O.mmbert_model(p, ocfg, (ids[3], ids[1]), am[1], None, True)
        assert float((seq.cpu() - oseq).abs().max()) < 6e-2 and float((pooled.cpu() - opooled).abs().max()) < 2e-2
        seq, pooled = m.bert(ids[0].to(DEV), attention_mask=am[0].to(DEV), token_type_ids=batch["token_type_ids"][0].to(DEV))
        oseq, opooled = O.mmbert_model(p, ocfg, ids[0], am[0], batch["token_type_ids"][0], False)
        assert float((seq.cpu() - oseq).abs().max()) < 6e-2
        te = torch.randn(2, 50, cfg["hidden"], generator=torch.Generator().manual_seed(3))
        je = m.bert.jointEmbeddings(te.to(DEV), ids[2].to(DEV))
        oje = O.joint_embeddings(p, te, ids[2], (35, 74))
        assert float((je.cpu() - oje).abs().max()) < 5e-2
        with pytest.raises(Exception, match="Wrong Dimension"):
            m.bert.jointEmbeddings(te.to(DEV), torch.zeros(2, 4, 33, device=DEV))
        with pytest.raises(ValueError):
            m.bert(ids[0].to(DEV), attention_mask=torch.ones(2, 50, 1, 1, device=DEV))


def test_mmbert_model_standalone_matches_oracle():
    """``MMBertModel(config)`` constructed on its own (REF:MMBertForPretraining.py:13-22 allows it; round 3 raised): it adopts a private
    owner for the flat parameter storage on first use, keeps its own parameters and prefix-less state-dict keys, matches the oracle's
    ``mmbert_model`` in both modes, differentiates into its own ``.grad`` views, and says so when handed ``inputs_embeds``."""
    from msa_amd.model import MMBertConfig, MMBertModel
    cfg = CFG1
    m = MMBertModel(MMBertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                                 num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"]))
    m.set_joint_embeddings(cfg["dataset"])
    p = O.seeded_params(cfg)
    own = {k[len("bert."):]: v for k, v in p.items() if k.startswith("bert.")}
    missing, unexpected = m.load_state_dict(own, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    assert all(not k.startswith("bert.") for k in m.state_dict())
    m = m.to(DEV).eval()
    batch = synthetic_batch(2, 50, 64, 64, seed=1)
    ids, am = batch["input_ids"], batch["attention_mask"]
    ocfg = dict(cfg, hidden_dropout=0.0, attn_dropout=0.0, joint_dropout=0.0)
    seq, pooled = m((ids[3].to(DEV), ids[1].to(DEV)), attention_mask=(am[1][0].to(DEV), am[1][1].to(DEV)), joint=True)
    oseq, opooled = O.mmbert_model(p, ocfg, (ids[3], ids[1]), am[1], None, True)
    assert float((seq.detach().cpu() - oseq).abs().max()) < 6e-2 and float((pooled.detach().cpu() - opooled).abs().max()) < 2e-2
    (seq.float().square().mean() + pooled.square().mean()).backward()
    torch.cuda.synchronize()
    g = m.encoder.layer[0].intermediate.dense.weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0.0
    assert float(m.pooler.dense.weight.grad.abs().sum()) > 0.0
    with torch.no_grad():
        seq2, _ = m(ids[0].to(DEV), attention_mask=am[0].to(DEV), token_type_ids=batch["token_type_ids"][0].to(DEV))
        oseq2, _ = O.mmbert_model(p, ocfg, ids[0], am[0], batch["token_type_ids"][0], False)
        assert float((seq2.cpu() - oseq2).abs().max()) < 6e-2
    with pytest.raises(NotImplementedError, match="inputs_embeds"):
        m(inputs_embeds=torch.zeros(2, 50, cfg["hidden"], device=DEV))
    with pytest.raises(ValueError):
        m(ids[0].to(DEV), inputs_embeds=torch.zeros(2, 50, cfg["hidden"], device=DEV))


def test_cpu_tensors_are_rejected_loudly():
    from msa_amd.model import MMBertConfig, MMBertForPretraining
    m = MMBertForPretraining(MMBertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256))
    m.bert.set_joint_embeddings("mosei")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(**synthetic_batch(2, 8, 8, 8, vocab=512, seed=1))


# ================================================================================================ round 2: headline depth, intermediates
BASE12 = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)


def _host_mem_gib():
    """MemAvailable of the host in GiB (the fp32 CPU oracle at the timed batch needs ~45 GB); a large number when it cannot be read."""
    try:
        with open("/proc/meminfo") as fh:
            return {l.split(":")[0]: int(l.split()[1]) for l in fh}.get("MemAvailable", 0) // (1 << 20)
    except OSError:
        return 1 << 20


def _report(name, payload):
    """Measured deviations go to gpurun_out/ (when it exists) so that the stated tolerances can be read against them."""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, name + ".json"), "w") as fh:
            json.dump(payload, fh, indent=1)


def _deep_check_against_oracle(cfg, B, T, Pv, Pa, seed, report, min_cos=0.995, loss_tol=4e-3, grad_tol=0.06):
    """Eval-mode forward + backward of a DEEP model against the fp32 oracle with tolerances re-derived at depth from the oracle itself
    run with bf16 storage (see test_bert_base_12_layers_match_oracle for the stated bounds); every deviation is MEASURED and written to
    gpurun_out/<report>.json first, the assertions follow."""
    batch = synthetic_batch(B, T, Pv, Pa, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=seed)
    p, oout, ologits = oracle_run(cfg, batch)
    pe, eout, elogits = oracle_run(cfg, batch, emulate_bf16=True)
    m = build(cfg)
    out, logits = m(**batch_to(batch, DEV))
    rep = {"losses": {}, "grads": {}}
    for i, name in ((0, "joint"), (4, "ap"), (5, "label"), (6, "nce")):
        ours, emul = rel(out[i].detach(), oout[i].detach()), rel(eout[i].detach(), oout[i].detach())
        rep["losses"][name] = dict(ours=ours, emulated_oracle=emul, value=float(oout[i]), hip=float(out[i]))
    dl = float((logits.float().cpu() - ologits.detach()).abs().max())
    rep["logits_max_abs"] = dict(ours=dl, emulated_oracle=float((elogits.detach() - ologits.detach()).abs().max()))
    for k in (7, 9, 11):
        assert tuple(out[k].shape) == tuple(oout[k].shape)
        d = (out[k].float().cpu() - oout[k].detach()).abs()
        de = (eout[k].detach() - oout[k].detach()).abs()
        rep[f"scores{k}"] = dict(max_abs=float(d.max()), mean_abs=float(d.mean()), emulated_max_abs=float(de.max()), emulated_mean_abs=float(de.mean()))
    for k in (8, 10, 12):
        rep[f"rel{k}_max_abs"] = float((out[k].float().cpu() - oout[k].detach()).abs().max())
    out[0].mean().backward()
    torch.cuda.synchronize()
    worst_cos = (1.0, None)
    special = {}
    for n, q in m.named_parameters():
        og = p[n].grad
        g = q.grad.float().cpu()
        if og is None or float(og.abs().sum()) == 0.0:
            special[n] = ("nograd", float(g.abs().sum()))
            continue
        if "attention.self.key.bias" in n:                 # true gradient 0 (softmax is shift invariant): noise on both sides
            special[n] = ("keybias", float(og.norm()), float(g.norm()))
            continue
        dev = float((g - og).norm() / og.norm())
        dev_emul = float((pe[n].grad - og).norm() / og.norm())
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1)))
        rep["grads"][n] = dict(rel_err=dev, emulated_oracle_rel_err=dev_emul, cosine=cos, norm=float(og.norm()), abs_err=float((g - og).norm()))
        if n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")) and cos < worst_cos[0]:
            worst_cos = (cos, n)
    rep["worst_encoder_cosine"] = worst_cos
    enc = sorted(r["rel_err"] for n, r in rep["grads"].items() if n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")))
    rep["encoder_rel_err_median_max"] = (enc[len(enc) // 2], enc[-1])
    _report(report, rep)
    print(report, "worst encoder-side gradient cosine", worst_cos, "encoder gradient rel err median / max", rep["encoder_rel_err_median_max"])
    # ---- assertions ----
    for name, r in rep["losses"].items():
        assert r["ours"] < max(loss_tol, 3.0 * r["emulated_oracle"]), (name, r)
    assert dl < 3e-2, dl
    for k in (7, 9, 11):
        r = rep[f"scores{k}"]
        assert r["max_abs"] < 8e-2 and r["mean_abs"] < 8e-3, (k, r)
    for k in (8, 10, 12):
        assert rep[f"rel{k}_max_abs"] < 3e-2
    for n, sp in special.items():
        if sp[0] == "nograd":
            assert sp[1] == 0.0, f"{n}: reference has no gradient here"
        else:
            assert sp[1] < 1e-5 and sp[2] < 5e-3, (n, sp)
    # A gradient whose norm is below 2 % of the median encoder weight-gradient norm is a residue of cancellation (measured at L = 24: the
    # top layers' query / key weights, norm 0.013-0.019 against 2-3 -- softmax is nearly uniform at initialisation, so dS = P (dP - delta)
    # sums to zero over the keys and dQ, dK are what is left): the bf16 rounding of the dS operand of the MFMA (2^-9 per element, not
    # cancelling) is 14-24 % of it, where bf16 STORAGE alone (the calibrator) moves it by 6.5-7 %.  Such gradients are held to 30 % and
    # cosine 0.95; every other one to max(grad_tol, 2 x calibrator) and min_cos.
    enc_w = sorted(r["norm"] for n, r in rep["grads"].items() if n.startswith("bert.encoder") and n.endswith("weight") and "LayerNorm" not in n)
    small = 0.02 * enc_w[len(enc_w) // 2]
    for n, r in rep["grads"].items():
        tiny = r["norm"] < small
        assert r["rel_err"] < max(grad_tol, 2.0 * r["emulated_oracle_rel_err"]) or r["abs_err"] < 2e-4 or (tiny and r["rel_err"] < 0.30), (n, r, small)
        if n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")):
            assert r["cosine"] > (0.95 if tiny else min_cos), (n, r)


def test_bert_base_12_layers_match_oracle():
    """BASELINE configs[1] at its full depth and width (12 layers, d = 768, 12 heads, T = 50, A = V = 500), batch 2 so that the
    fp32 CPU oracle finishes in seconds, eval mode: losses, regression logits, prediction scores, and the gradient of EVERY
    parameter.  bf16 rounding compounds over 12 residual blocks, so the tolerances are re-derived at this depth from
    the oracle itself run with bf16 storage (oracle.bf16_storage_emulation: fp32 arithmetic, activations / GEMM weights rounded
    where the HIP path stores bf16) -- stated, L = 12:
      losses 4e-3 relative; regression logits 3e-2 abs; prediction scores 8e-2 abs (max over 1.1e8 values) and 8e-3 mean abs;
      per-parameter gradient: relative L2 error <= max(6 %, 2 x the emulated oracle's own deviation) (or 2e-4 absolute for
      gradients of norm < 1e-3: the CPC biases), encoder-side cosine >= 0.995.
    Measured (round 2, gpurun_out/parity_L12.json -> profiles/r2_parity_L12.json): losses 5e-5 / 5e-5 / 5e-6 / 1e-7 (emulated oracle:
    4e-5 / 2e-3 / 5e-4 / 1e-5); scores 0.034-0.044 max, 0.0052-0.0057 mean (emulated: 0.036-0.044 / 0.0054-0.0059); encoder-side
    gradients median 4.5 % / max 8.7 % (emulated: 3.9 % / 7.5 %), cosine >= 0.9962; [B,H] head gradients at batch 2: 17-26 %
    (emulated: 18-55 %) -- the HIP path deviates from the fp32 oracle by what bf16 storage alone does to the oracle."""
    _deep_check_against_oracle(BASE12, 2, 50, 500, 500, seed=5, report="parity_L12")


BERT_LARGE = dict(hidden=1024, layers=24, heads=16, intermediate=4096, vocab=30522, dataset="mosei", alpha=1.0, beta=1.0)


def test_reference_default_bert_large_24_layers_match_oracle():
    """Round 4: the reference's ACTUAL default model at full depth -- `bert-large-uncased` (REF:train.py:28,70: L = 24, H = 1024, 16
    heads, I = 4096; REF:config.py:12 TEXTDIM = 1024, REF:MMBertForPretraining.py:327-344 CPC x_size = 1024) with max_seq_length 40
    (REF:train.py:32) and pair length == text length (collate asserts equal lengths, REF:model_utils.py:92, so the MLM labels of the
    pair positions are a COPY of the text labels: REF:trainer.py:50,53 -- labelled rows behind the last unmasked key), batch 4 so
    that the fp32 CPU oracle finishes in seconds.  Stated at L = 24 (twice the depth of the L = 12 bounds): losses 8e-3 relative or
    3 x emulated (measured at batch 2: 3e-4 / 6.2e-3 / 5e-4 / 1e-5 -- the 2-way alignment CE of a 2-sample batch moves by 4e-3
    absolute, a logit difference of 1e-2), scores 8e-2 max / 8e-3 mean (measured 0.048 / 0.0072 = the emulated oracle's own 0.047 /
    0.0072), every gradient within max(6 %, 2 x the bf16-storage-emulated oracle's own deviation) (measured at batch 2: median 6.4 %
    against the emulated oracle's 10 %), encoder-side cosine >= 0.97; deviations -> gpurun_out/parity_L24_bert_large.json (-> profiles/r4_parity_L24_bert_large.json)."""
    _deep_check_against_oracle(BERT_LARGE, 4, 40, 40, 40, seed=14, report="parity_L24_bert_large", min_cos=0.97, loss_tol=8e-3)


def test_reference_default_bert_large_full_size_train_step_properties():
    """The reference's default configuration at FULL size -- bert-large, T = P = 40, train_batch_size 32 (REF:train.py:38), train mode
    with all dropouts, AdamW -- is beyond the CPU oracle at this batch, so the size-independent properties: finite losses and
    gradients on every parameter the reference differentiates, the same seed -> the same loss, the valid-first packing keeps the
    labelled padded pair rows (P == T: every pair position carries the text label), four optimizer steps keep everything finite,
    move every differentiated parameter and leave the never-differentiated ones bit-identical."""
    from msa_amd import trainer as T_
    cfg = BERT_LARGE
    B = 32
    batch = batch_to(synthetic_batch(B, 40, 40, 40, dataset="mosei", vocab=cfg["vocab"], seed=33), DEV)
    m = build(cfg, train=True)
    m.manual_seed(9)
    seen, orig = [], m._split_layout
    m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
    out, logits = m(**batch)
    a = float(out[0])
    out[0].mean().backward()
    torch.cuda.synchronize()
    m._split_layout = orig
    assert all(np.isfinite(float(out[i])) for i in (0, 4, 5, 6)) and bool(torch.isfinite(logits).all())
    assert tuple(out[7].shape) == (B, 40, cfg["vocab"]) and tuple(out[9].shape) == (B, 80, cfg["vocab"]) and tuple(logits.shape) == (B, 1)
    lab_v = batch["masked_labels"][1]
    assert bool((lab_v[:, :40] == lab_v[:, 40:]).all()) and int((lab_v[:, 40:] != -100).sum()) > 0     # REF:trainer.py:50,53
    nz = 0
    for n, q in m.named_parameters():
        assert bool(torch.isfinite(q.grad).all()), n
        nz += int(float(q.grad.abs().sum()) > 0.0)
    assert nz >= len(list(m.named_parameters())) - 6
    m.zero_grad()
    m.manual_seed(9)
    b = float(m(**batch)[0][0])
    assert abs(b - a) <= 1e-6 * abs(a)
    frozen = ("bert.jointEmbeddings.W_cv.", "bert.jointEmbeddings.W_cs.", "cls.seq_relationship.")
    before = {n: p.detach().clone() for n, p in m.named_parameters() if n.startswith(frozen)}
    allp = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt, sched = T_.build_optimizer(m, T_.default_args(train_batch_size=B, learning_rate=5e-5), 8, mode="hf")
    losses = []
    for i in range(4):
        o, _ = m(**batch)
        o[0].mean().backward()
        opt.step(); sched.step(); opt.zero_grad()
        losses.append(float(o[0]))
    # (joint = alpha mlm + ap + label - beta nce starts at ~10.4 + 0.7 + 3 - 3 ln 32 = 4.2 and is not monotone over a handful of
    # warm-up steps with dropout 0.5 on the joint embeddings: finite, and every differentiated parameter moved)
    assert all(np.isfinite(losses)) and max(losses) < 30.0, losses
    moved = 0
    for n, p in m.named_parameters():
        assert bool(torch.isfinite(p).all()), n
        if n.startswith(frozen):
            assert torch.equal(p.detach(), before[n]), n
        else:
            moved += int(not torch.equal(p.detach(), allp[n]))
    assert moved >= len(allp) - 6 - 24, moved                        # (key biases: zero true gradient, may or may not move)


@pytest.mark.parametrize("B", [16])
def test_bert_base_12_layers_batch8_gradients_without_calibrator(B):
    """(B = 16, round 4: the bench's own batch -- BASELINE configs[1] exactly as benchmarked, in eval mode; ~40 GB and about three
    minutes of fp32 CPU oracle, falling back to batch 8 below 70 GiB of available host memory; deviations -> gpurun_out/parity_L12_B16.json.
    Round 6: the batch-8 case of this test went with the GPU-suite budget -- it is implied by this one plus the batch-8 TRAIN-mode replay.)
    The headline depth again at batch 8 (half the bench's batch; the fp32 CPU oracle needs about a minute and ~20 GB for it), with NO
    calibrator: the [B,H]-sized head gradients are sums over the batch of per-sample terms that partly cancel -- 17-26 % off at
    batch 2, where only the bf16-emulation calibrator bounds them -- and are better conditioned here.  Stated at L = 12, B = 8:
      losses 4e-3 relative (measured 2e-4); every encoder / embedding / MLM-head gradient within 8 % relative L2 error, cosine
      >= 0.995 (measured: median 5.1 %, max 6.1 %, cosine >= 0.9982); pooler, gate (``attn``, vt / vv / vs), classifier and align
      gradients within 12 % (measured <= 9.9 %); the three CPC projections within 30 % AND 1e-3 absolute (measured 19-26 %, 1.6e-4 ..
      7.3e-4 absolute): at initialisation nce = 3 ln B exactly and their gradient is what is left of O(1) per-sample terms that
      cancel to a norm of 2e-3 .. 4e-3 -- three orders below the other head gradients (0.1 .. 2.4) -- so one bf16 rounding of the
      pooled vectors (2^-9 relative) is an absolute error of that size whatever the kernel.  Deviations go to
      gpurun_out/parity_L12_B8.json (-> profiles/r3_parity_L12_B8.json)."""
    if B == 16 and _host_mem_gib() < 70:
        B = 8                                                            # (a small host: the batch-8 form of the same check, never a skip)
    cfg = BASE12
    batch = synthetic_batch(B, 50, 500, 500, dataset=cfg["dataset"], vocab=cfg["vocab"], seed=5)
    m = build(cfg)
    out, logits = m(**batch_to(batch, DEV))
    out[0].mean().backward()
    torch.cuda.synchronize()
    ours = {n: q.grad.float().cpu() for n, q in m.named_parameters()}
    vals = [float(out[i]) for i in (0, 4, 5, 6)]
    lg = logits.float().cpu()
    del m, out, logits
    torch.cuda.empty_cache()
    p, oout, ologits = oracle_run(cfg, batch)
    rep = {"losses": {}, "grads": {}}
    for v, i, name in zip(vals, (0, 4, 5, 6), ("joint", "ap", "label", "nce")):
        rep["losses"][name] = dict(ours=rel(v, oout[i].detach()), value=float(oout[i]))
        assert rel(v, oout[i].detach()) < 4e-3, (name, v, float(oout[i]))
    assert float((lg - ologits.detach()).abs().max()) < 3e-2
    for n, g in ours.items():
        og = p[n].grad
        if og is None or float(og.abs().sum()) == 0.0:
            assert float(g.abs().sum()) == 0.0, n
            continue
        if "attention.self.key.bias" in n:
            continue
        dev = float((g - og).norm() / og.norm())
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1)))
        rep["grads"][n] = dict(rel_err=dev, cosine=cos, norm=float(og.norm()))
    _report(f"parity_L12_B{B}", rep)
    for n, r in rep["grads"].items():
        if n.startswith(("bert.embeddings", "bert.encoder", "bert.jointEmbeddings", "cls.predictions")):
            assert r["rel_err"] < 0.08 and r["cosine"] > 0.995, (n, r)
        elif n.startswith("cpc_"):
            # conditioning decides the bound (round 4, justified by profiles/r4_parity_trained.json): a CPC gradient whose norm is
            # within two orders of the pooler's is held to the heads' 8 % -- never the case so far: the [CLS] rows of a batch are nearly
            # collinear at initialisation AND after training (nce stays 3 ln B to 1e-5), so d nce / d W is a difference of nearly equal
            # unit vectors, 3-5 orders below the other head gradients, and one bf16 rounding of the pooled rows (2^-9) is an absolute
            # error of its size on BOTH sides: bounded absolutely (1e-3 of an O(1) input scale; measured 1.3e-4 .. 7.3e-4) and at 50 %
            if r["norm"] >= 1e-2 * rep["grads"]["bert.pooler.dense.weight"]["norm"]:
                assert r["rel_err"] < 0.08, (n, r)
            else:
                assert r["rel_err"] < 0.5 and r["rel_err"] * r["norm"] < 1e-3, (n, r)
        else:
            assert r["rel_err"] < 0.12, (n, r)


def test_hidden_states_match_reference_golden(golden_dir):
    """Per-layer hidden states, the text embeddings and the JointEmbeddings outputs of all three passes against what forward hooks
    recorded on the REAL reference (tests/golden/make_golden.py:88-102): stated 8e-2 max abs over ~3e4 values per tensor / 6e-3 mean
    abs (bf16 storage of post-LayerNorm values of magnitude up to ~8: half an ulp there is 1.6e-2, two layers; measured 6.6e-2 /
    3.4e-3).  Both packings: the plain one and the valid-first one (rows un-permuted)."""
    for name in ("cfg1_T50_P64", "cfg1_T50_P50"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        B, T, Pv, Pa, seed = (int(x) for x in g["meta"])
        batch = batch_to(synthetic_batch(B, T, Pv, Pa, seed=seed), DEV)
        for split in (False, True):
            m = build(CFG1)
            m.skip_padded_backward = split
            m.debug_hidden = {}
            seen, orig = [], m._split_layout
            m._split_layout = lambda *a, _o=orig, _s=seen: (_s.append(_o(*a)), _s[-1])[1]
            out, _ = m(**batch)
            # the valid-first packing was really in use when asked for (the P64 batch has padded pair rows; the P50 one too: pair
            # lengths are drawn from half to full) and only then
            assert (seen[0] is not None) == split and (not split or seen[0].rows_a < seen[0].tokens), (name, split)
            dbg = m.debug_hidden
            lens = [T, T + Pv, T + Pa]
            starts = np.cumsum([0] + [B * n for n in lens])

            def cmp(got, ref, what):
                d = np.abs(got.float().cpu().numpy() - ref)
                assert d.max() < 8e-2 and d.mean() < 6e-3, (name, split, what, float(d.max()), float(d.mean()))
            for pi, tag in enumerate("tvs"):
                cmp(dbg["emb"][pi * B * T:(pi + 1) * B * T].view(B, T, -1), g[f"{tag}_emb"], tag + "_emb")
                rows = slice(int(starts[pi]), int(starts[pi + 1]))
                if tag != "t":
                    cmp(dbg["x"][rows].view(B, lens[pi], -1), g[f"{tag}_jemb"], tag + "_jemb")
                assert len(dbg["layers"]) == CFG1["layers"]
                for l in range(CFG1["layers"]):
                    cmp(dbg["layers"][l][rows].view(B, lens[pi], -1), g[f"{tag}_hidden{l}"], f"{tag}_hidden{l}")
                    # pooled = tanh(pooler(hidden[L-1][:, 0])) is what the heads consume


@pytest.mark.parametrize("B,H,num_labels,path", [(4, 128, 7, "step"), (16, 768, 7, "step"), (2, 1024, 1, "step"), (5, 192, 7, "step"), (32, 1024, 7, "step"),
                                                 (21, 768, 1, "step"), (17, 128, 7, "step"), (48, 768, 7, "step"), (128, 256, 7, "step"), (3, 64, 7, "step"),
                                                 (16, 768, 7, "launches"), (21, 768, 1, "launches"), (32, 1024, 7, "launches")])
def test_fused_heads_match_oracle_in_fp32(B, H, num_labels, path):
    """csrc/heads.hip (+ the dense products around it: model._HeadsFn, hand-derived backward) against the ORACLE's restatement of
    the same objective (oracle.heads_from_cls -> fusion_objective, the functions the pinned pretraining_forward runs) on the
    same fp32 [CLS] rows: no bf16 anywhere on this path, so the comparison is tight -- losses 2e-5 relative, the gradient wrt the
    [CLS] rows and wrt EVERY head parameter (pooler, align, attn, vt/vv/vs, classifier1_1/2, the three CPC projections) within
    2e-3 of its own norm (fp32 summation order; hipBLASLt vs MKL), without any calibrator.
    ``path``: "step" = one launch per dependency level (round 6: csrc/heads_coop.hip, model._HeadsStepFn: 7 + 6 launches -- also beyond the older
    form's 32-sample limit: batch 48 at the headline width, 128 at d = 256, and three samples), "launches" = the 19-launch form (model._HeadsFn)."""
    cfg = dict(hidden=H, layers=1, heads=max(1, H // 64), intermediate=4 * H, vocab=512, dataset="mosei", alpha=1.0, beta=0.7, num_labels=num_labels)
    m = build(cfg)
    m.num_labels = num_labels
    m._ensure_ready(torch.device(DEV, 0))
    gen = torch.Generator().manual_seed(B * 1000 + H)
    first = torch.randn(3 * B, H, generator=gen)
    ap_v, ap_s = torch.randint(0, 2, (B,), generator=gen), torch.randint(0, 2, (B,), generator=gen)
    sent = torch.rand(B, generator=gen) * 6 - 3
    p = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(cfg).items()}
    # move the head weights off their N(0, 0.02) initialisation scale so that no term is negligible
    with torch.no_grad():
        for k in p:
            if not k.startswith(("bert.embeddings", "bert.encoder", "cls.predictions", "bert.jointEmbeddings")):
                p[k].mul_(8.0 if num_labels == 7 else 1.5)      # (the tanh of the 1-label head saturates to an exactly-zero gradient at 3x)
    sd = {k: v.detach().clone() for k, v in p.items()}
    m.load_state_dict(sd, strict=False)
    f_o = first.clone().requires_grad_(True)
    loss_o, ap_o, label_o, nce_o, logits_o = O.heads_from_cls(p, cfg, f_o, ap_v, ap_s, sent)
    loss_o.backward()
    from msa_amd import model as MM
    f_g = first.to(DEV).requires_grad_(True)
    m._flat.grads.zero_()
    fn = MM._HeadsStepFn if path == "step" else MM._HeadsFn
    heads_loss, aux, logits_out, t_rel, relv = fn.apply(f_g, m, torch.cat((ap_v, ap_s)).to(DEV).long(), sent.to(DEV).float())
    heads_loss.backward()
    torch.cuda.synchronize()
    for got, ref, what in ((heads_loss, loss_o, "loss"), (aux[0], ap_o, "ap"), (aux[1], label_o, "label"), (aux[2], nce_o, "nce")):
        # (B = 1: nce is log(1) - 0 = 0 exactly here and 1.7e-8 of rounding in the oracle: an absolute floor beside the relative bound)
        assert rel(got.detach(), ref.detach()) < 2e-5 or abs(float(got) - float(ref)) < 1e-6, (what, float(got), float(ref))
    assert float((logits_out.cpu() - logits_o.detach()).abs().max()) < 1e-4 * max(1.0, float(logits_o.abs().max()))
    dn = float(f_o.grad.norm())
    assert float((f_g.grad.cpu() - f_o.grad).norm()) < 2e-3 * dn, ("dfirst", float((f_g.grad.cpu() - f_o.grad).norm()), dn)
    checked = 0
    for n, q in m.named_parameters():
        if n.startswith(("bert.embeddings", "bert.encoder", "cls.predictions", "bert.jointEmbeddings", "cls.seq_relationship")):
            continue
        og = p[n].grad
        assert og is not None and float(og.norm()) > 0, n
        err = float((q.grad.float().cpu() - og).norm())
        assert err < 2e-3 * float(og.norm()) + 1e-9, (n, err, float(og.norm()))
        checked += 1
    assert checked == 2 + 2 + 2 + 6 + 4 + 6          # pooler, align, attn, vt/vv/vs, classifier1_1/2, cpc_z{t,v,a}.net: weights + biases


def test_level_launch_heads_equal_the_multi_launch_heads_and_read_their_rows_from_the_encoder_output():
    """Round 6: model._HeadsStepFn (one launch per dependency level: 7 + 6) against model._HeadsFn (19 launches) on the same fp32 [CLS] rows and
    parameters: the four losses, the returned scores, the gradient of the rows, of the per-pass MLM losses and of every head parameter agree
    to fp32 summation order (1e-5 relative on losses / 2e-4 of a gradient's norm); the gradients ACCUMULATE (a second backward doubles them);
    two runs give the same bits (no atomics on data); and the form the model uses -- rows gathered by the kernel from the bf16 encoder output
    through a row list, label tensors passed separately -- gives bit for bit what the fp32 form gives on those rows converted to fp32."""
    from msa_amd import model as MM
    B, H = 16, 768
    cfg = dict(hidden=H, layers=1, heads=H // 64, intermediate=4 * H, vocab=512, dataset="mosei", alpha=0.6, beta=0.7)
    m = build(cfg)
    m.num_labels = 7
    m._ensure_ready(torch.device(DEV, 0))
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, q in m.named_parameters():
            if not n.startswith(("bert.embeddings", "bert.encoder", "cls.predictions", "bert.jointEmbeddings")):
                q.mul_(6.0)
    m._flat.maybe_refresh()
    y = (torch.randn(40 + 3 * B * 7, H, generator=gen)).to(torch.bfloat16).to(DEV)
    rows = (torch.arange(3 * B) * 7 + 3).to(DEV)
    first = y.index_select(0, rows).float()
    ap_v, ap_s = torch.randint(0, 2, (B,), generator=gen).to(DEV), torch.randint(0, 2, (B,), generator=gen).to(DEV)
    sent = (torch.rand(B, generator=gen) * 6 - 3).to(DEV)
    mlm = torch.tensor([7.1, 6.9, 7.3], device=DEV, requires_grad=True)
    names = [n for n, _ in m.named_parameters() if not n.startswith(("bert.embeddings", "bert.encoder", "cls.predictions", "bert.jointEmbeddings", "cls.seq_relationship"))]
    params = dict(m.named_parameters())

    def run(fn, first_in, ap, src=None, twice=False):
        m._flat.grads.zero_()
        f = first_in.clone().requires_grad_(True)
        ml = mlm.detach().clone().requires_grad_(True)
        args = (f, m, ap, sent, ml) + ((src,) if fn is MM._HeadsStepFn else ())
        loss, aux, logits, t_rel, rel = fn.apply(*args)
        loss.backward(retain_graph=twice)
        if twice:
            loss.backward()
        torch.cuda.synchronize()
        return dict(loss=loss.detach().clone(), aux=aux.clone(), logits=logits.clone(), t_rel=t_rel.clone(), rel=rel.clone(), dfirst=f.grad.clone(), dmlm=ml.grad.clone(),
                    grads={n: params[n].grad.detach().float().clone() for n in names})
    ap_cat = torch.cat((ap_v, ap_s))
    old = run(MM._HeadsFn, first, ap_cat)
    new = run(MM._HeadsStepFn, first, ap_cat)
    assert rel(new["loss"], old["loss"]) < 1e-5 and float((new["aux"] - old["aux"]).abs().max()) < 1e-5 * float(old["aux"].abs().max())
    for k in ("logits", "t_rel", "rel"):
        assert float((new[k] - old[k]).abs().max()) < 1e-4 * max(1.0, float(old[k].abs().max())), k
    assert float((new["dfirst"] - old["dfirst"]).norm()) < 2e-4 * float(old["dfirst"].norm())
    assert torch.allclose(new["dmlm"], old["dmlm"], rtol=1e-6, atol=0)
    for n in names:
        assert float(old["grads"][n].norm()) > 0, n
        assert float((new["grads"][n] - old["grads"][n]).norm()) < 2e-4 * float(old["grads"][n].norm()) + 1e-9, n
    again = run(MM._HeadsStepFn, first, ap_cat)
    assert torch.equal(again["loss"], new["loss"]) and torch.equal(again["dfirst"], new["dfirst"]) and all(torch.equal(again["grads"][n], new["grads"][n]) for n in names)
    dbl = run(MM._HeadsStepFn, first, ap_cat, twice=True)
    for n in names:
        assert float((dbl["grads"][n] - 2.0 * new["grads"][n]).norm()) <= 1e-6 * float(new["grads"][n].norm()), n
    # the model's form: rows read from the bf16 matrix through the list, the two label tensors separately
    placeholder = torch.empty_like(first)
    viay = run(MM._HeadsStepFn, placeholder, (ap_v, ap_s), src=(y, rows))
    assert torch.equal(viay["loss"], new["loss"]) and torch.equal(viay["rel"], new["rel"]) and torch.equal(viay["dfirst"], new["dfirst"])
    assert all(torch.equal(viay["grads"][n], new["grads"][n]) for n in names)


def test_scores_dtype_float32_for_numpy_consumers():
    """outputs[7/9/11] are bf16 views by default (documented deviation); ``model.scores_dtype = torch.float32`` hands out what the
    reference does -- fp32 tensors, straight from the vocabulary GEMM's accumulators -- that ``.cpu().numpy()`` accepts
    (REF:sampling.py-style consumers).  Losses are identical (the CE kernels round fp32 logits to bf16 as they load them) and the
    bf16 scores are the rounded fp32 ones."""
    batch = batch_to(synthetic_batch(2, 50, 64, 64, seed=1), DEV)
    m = build(CFG1)
    assert m.scores_dtype == torch.bfloat16
    with torch.no_grad():
        o16, _ = m(**batch)
        m.scores_dtype = torch.float32
        o32, _ = m(**batch)
    for k, S in ((7, 50), (9, 114), (11, 114)):
        assert o16[k].dtype == torch.bfloat16 and o32[k].dtype == torch.float32
        a = o32[k].cpu().numpy()
        assert a.shape == (2, S, CFG1["vocab"]) and np.isfinite(a).all()
        assert torch.equal(o32[k].to(torch.bfloat16), o16[k])
        with pytest.raises(TypeError):
            o16[k].cpu().numpy()
    assert o32[8].dtype == torch.float32 and o32[8].cpu().numpy().shape == (2, 2)
    for i in (0, 4, 5, 6):
        assert abs(float(o32[i]) - float(o16[i])) <= 1e-6 * abs(float(o16[i]))
    # ... and in training: same loss, same gradients, whatever the score dtype
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        m2 = build(CFG1)
        m2.scores_dtype = dt
        out, _ = m2(**batch)
        out[0].mean().backward()
        grads[dt] = (float(out[0]), {n: q.grad.detach().float().clone() for n, q in m2.named_parameters()})
    assert abs(grads[torch.float32][0] - grads[torch.bfloat16][0]) <= 1e-6 * abs(grads[torch.bfloat16][0])
    for n in grads[torch.float32][1]:
        if "attention.self.key.bias" in n:
            continue
        a, b = grads[torch.float32][1][n], grads[torch.bfloat16][1][n]
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-7, n   # (1e-7: the CPC gradients at init are ~1e-7, summed with atomics)


def test_label_on_a_cls_row_and_bad_labels():
    """A caller-supplied label on a [CLS] row: the sparse top-layer backward would gather that row twice (labelled rows + [CLS]
    rows), so it steps aside (dense top layer) and the gradients equal the dense path's.  A label outside the vocabulary that is
    not -100 raises like torch's CrossEntropyLoss does in the reference."""
    cfg = dict(hidden=256, layers=2, heads=4, intermediate=1024, vocab=4096, dataset="mosei", alpha=1.0, beta=1.0)
    batch = synthetic_batch(4, 24, 120, 90, dataset="mosei", vocab=cfg["vocab"], seed=41)
    lab_t = batch["masked_labels"][0].clone()
    lab_t[2, 0] = 77                                            # [CLS] row of sample 2, text pass
    b2 = batch_to(dict(batch, masked_labels=(lab_t,) + tuple(batch["masked_labels"][1:])), DEV)
    from msa_amd import model as MM
    res = {}
    for sparse in (True, False):
        m = build(cfg)
        m.sparse_top_layer_backward = sparse
        calls, orig = [], MM._EncoderFn._last_layer_sparse
        MM._EncoderFn._last_layer_sparse = staticmethod(lambda *a, _o=orig, _c=calls: (_c.append(1), _o(*a))[1])
        try:
            out, _ = m(**b2)
            out[0].mean().backward()
        finally:
            MM._EncoderFn._last_layer_sparse = staticmethod(orig)
        torch.cuda.synchronize()
        assert calls == []                                       # never taken: not asked for, or a [CLS] row is labelled
        res[sparse] = {n: q.grad.detach().float().clone() for n, q in m.named_parameters()}
    for n in res[True]:
        if "attention.self.key.bias" in n:
            continue
        same_grads(res[True][n], res[False][n], n)
    lab_bad = batch["masked_labels"][0].clone()
    lab_bad[0, 3] = cfg["vocab"] + 5
    m = build(cfg)
    with pytest.raises(IndexError, match="out of bounds"):
        out, _ = m(**batch_to(dict(batch, masked_labels=(lab_bad,) + tuple(batch["masked_labels"][1:])), DEV))
        out[0].mean().backward()


def test_from_pretrained_local_directory(tmp_path):
    """SURVEY S8(f) row 4 / REF:train.py:70: ``MMBertForPretraining.from_pretrained(dir)`` on a local HuggingFace-style checkpoint
    directory -- ``config.json`` + ``pytorch_model.bin`` (or ``model.safetensors``) with BertForPreTraining's key names, incl. the
    transformers-4.x ``bert.embeddings.position_ids`` buffer and the tied ``cls.predictions.decoder.*`` aliases: the loaded model
    computes exactly what a model built from the same tensors with load_state_dict computes; keys the checkpoint does not have
    (jointEmbeddings, fusion head, CPC: they are the reference's additions) keep their fresh initialisation."""
    import json
    cfg = dict(CFG1, vocab=2048)
    sd = O.seeded_params(cfg, seed=3)
    hf_keys = {k: v for k, v in sd.items() if k.startswith(("bert.embeddings", "bert.encoder", "bert.pooler", "cls.predictions", "cls.seq_relationship"))}
    hf_keys["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    hf_keys["cls.predictions.decoder.bias"] = sd["cls.predictions.bias"]
    hf_keys["bert.embeddings.position_ids"] = torch.arange(512).unsqueeze(0)
    conf = dict(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                intermediate_size=cfg["intermediate"], max_position_embeddings=512, type_vocab_size=2, hidden_act="gelu",
                hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, initializer_range=0.02,
                pad_token_id=0, model_type="bert", architectures=["BertForPreTraining"])
    from msa_amd.model import MMBertForPretraining
    batch = batch_to(synthetic_batch(2, 50, 64, 64, vocab=cfg["vocab"], seed=2), DEV)
    outs = []
    for fmt in ("bin", "safetensors"):
        d = tmp_path / fmt
        d.mkdir()
        (d / "config.json").write_text(json.dumps(conf))
        if fmt == "bin":
            torch.save(hf_keys, d / "pytorch_model.bin")
        else:
            from safetensors.torch import save_file
            save_file({k: v.clone().contiguous() for k, v in hf_keys.items()}, str(d / "model.safetensors"))
        torch.manual_seed(123)                                   # the fresh (non-checkpoint) parameters: same draw for both models
        m = MMBertForPretraining.from_pretrained(str(d))
        assert m.config.hidden_size == cfg["hidden"] and m.config.num_hidden_layers == cfg["layers"]
        m.bert.set_joint_embeddings("mosei")
        torch.manual_seed(123)
        ref = MMBertForPretraining(m.config)
        ref.bert.set_joint_embeddings("mosei")
        # ref: the same fresh draw, then the checkpoint's tensors through load_state_dict
        own = {k: v for k, v in m.state_dict().items()}
        missing, unexpected = ref.load_state_dict({k: v for k, v in hf_keys.items()}, strict=False)
        assert not unexpected and all(not k.startswith(("bert.embeddings", "bert.encoder", "bert.pooler", "cls.predictions")) for k in missing)
        for k, v in hf_keys.items():
            if k.endswith("position_ids"):
                continue
            assert torch.equal(own[k].cpu(), v), k               # every checkpoint tensor arrived, aliases included
        assert m.cls.predictions.decoder.weight.data_ptr() == m.bert.embeddings.word_embeddings.weight.data_ptr()
        # non-checkpoint parameters of m and ref were drawn from the same seed but at different points of the RNG stream
        # (jointEmbeddings is created after from_pretrained): copy them so that the two models are the same function
        ref.load_state_dict(m.state_dict())
        m, ref = m.to(DEV).eval(), ref.to(DEV).eval()
        with torch.no_grad():
            a, la = m(**batch)
            b, lb = ref(**batch)
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-7) and torch.equal(a[7], b[7]) and abs(float(a[0]) - float(b[0])) <= 1e-6 * abs(float(b[0]))
        outs.append(float(a[0]))
    assert abs(outs[0] - outs[1]) <= 1e-6 * abs(outs[0])           # .bin and .safetensors load the same model
    with pytest.raises(OSError, match="local checkpoint directory"):
        MMBertForPretraining.from_pretrained("bert-base-uncased")
