"""``torch.ops.mmbert.*`` (msa_amd/torch_ops.py): the C-ABI kernel families as PyTorch custom operators with autograd, each
against the plain fp32 PyTorch computation of the same op on the same bf16-rounded inputs (forward and gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, rtol, atol, what):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g}"


@pytest.fixture(scope="module", autouse=True)
def _register():
    import msa_amd.torch_ops  # noqa: F401


@pytest.mark.parametrize("act", ["none", "gelu"])
def test_linear_operator_forward_and_gradients(act):
    M, N, K = 640, 512, 256
    x = rnd(M, K, seed=1).bfloat16()
    w = (rnd(N, K, seed=2) * 0.05).bfloat16()
    b = rnd(N, seed=3)
    dy = rnd(M, N, seed=4).bfloat16()
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = torch.ops.mmbert.linear(xd, wd, bd, act)
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    y.backward(dy.to(DEV))
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.clone().requires_grad_(True)
    pre = xr @ wr.t() + br
    ref = torch.nn.functional.gelu(pre) if act == "gelu" else pre
    ref.backward(dy.float())
    close(y, ref, 1e-2, 2e-2, "y")
    close(xd.grad, xr.grad, 3e-2, 3e-2, "dx")
    close(wd.grad, wr.grad, 2e-2, 1.5e-1, "dW")            # sums of 640 bf16 products
    close(bd.grad, br.grad, 2e-2, 1.5e-1, "db")


def test_layer_norm_operator_forward_and_gradients():
    M, H = 300, 768
    x = rnd(M, H, seed=5).bfloat16()
    g, b = 1.0 + 0.1 * rnd(H, seed=6), 0.1 * rnd(H, seed=7)
    dy = rnd(M, H, seed=8).bfloat16()
    xd, gd, bd = x.to(DEV).requires_grad_(True), g.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = torch.ops.mmbert.layer_norm(xd, gd, bd, 1e-12)
    y.backward(dy.to(DEV))
    xr, gr, br = x.float().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    ref.backward(dy.float())
    close(y, ref, 1e-2, 2e-2, "y")
    close(xd.grad, xr.grad, 2e-2, 2e-2, "dx")
    close(gd.grad, gr.grad, 2e-2, 1e-1, "dgamma")
    close(bd.grad, br.grad, 2e-2, 1e-1, "dbeta")


def test_attention_operator_forward_and_gradients():
    lens, heads, H = [50, 114, 64], 2, 128
    M = sum(lens)
    qkv = rnd(M, 3 * H, seed=9).bfloat16()
    dctx = rnd(M, H, seed=10).bfloat16()
    bias = torch.zeros(M)
    bias[torch.rand(M, generator=torch.Generator().manual_seed(11)) < 0.2] = -10000.0
    qd = qkv.to(DEV).requires_grad_(True)
    ctx = torch.ops.mmbert.attention(qd, bias.to(DEV), lens, heads)
    ctx.backward(dctx.to(DEV))
    s = 0
    for n in lens:
        x = qkv[s:s + n].float().requires_grad_(True)
        q, k, v = (x[:, j * H:(j + 1) * H].view(n, heads, 64).transpose(0, 1) for j in range(3))
        sc = (q @ k.transpose(-1, -2)) * 0.125 + bias[s:s + n][None, None, :]
        ref = (torch.softmax(sc, -1) @ v).transpose(0, 1).reshape(n, H)
        ref.backward(dctx[s:s + n].float())
        close(ctx[s:s + n], ref, 2e-2, 2e-2, "ctx")
        close(qd.grad[s:s + n], x.grad, 3e-2, 3e-2, "dqkv")
        s += n
    # dropout: seeded -> reproducible, and different from the deterministic output
    a = torch.ops.mmbert.attention(qd.detach(), bias.to(DEV), lens, heads, 0.1, 5)
    b = torch.ops.mmbert.attention(qd.detach(), bias.to(DEV), lens, heads, 0.1, 5)
    assert torch.equal(a, b) and not torch.equal(a, ctx)


def test_operators_have_no_cpu_fallback():
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mmbert.layer_norm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5)


# ---------------------------------------------------------------------------------- round 2: the remaining kernel families
def test_embed_ln_operator_forward_and_gradients():
    """torch.ops.mmbert.embed_ln = BertEmbeddings (HF:53-108): word + type + position -> LayerNorm(1e-12) (-> dropout), against
    the plain fp32 PyTorch form; gradients of the three tables (padding row 0 of the word table receives none, HF:58) and of the
    LayerNorm parameters."""
    B, T, H, V = 3, 20, 128, 500
    g_ = torch.Generator().manual_seed(21)
    ids = torch.randint(1, V, (B, T), generator=g_)
    ids[0, -3:] = 0
    tt = torch.randint(0, 2, (B, T), generator=g_)
    word, typ, pos = rnd(V, H, seed=1) * 0.1, rnd(2, H, seed=2) * 0.1, rnd(64, H, seed=3) * 0.1
    gam, bet = 1.0 + 0.1 * rnd(H, seed=4), 0.1 * rnd(H, seed=5)
    dy = rnd(B * T, H, seed=6).bfloat16()
    dv = [t.to(DEV).requires_grad_(True) for t in (word, typ, pos, gam, bet)]
    y = torch.ops.mmbert.embed_ln(ids.to(DEV), tt.to(DEV), *dv, 1e-12)
    assert y.shape == (B * T, H) and y.dtype == torch.bfloat16
    y.backward(dy.to(DEV))
    rv = [t.clone().requires_grad_(True) for t in (word, typ, pos, gam, bet)]
    e = torch.nn.functional.embedding(ids, rv[0], padding_idx=0) + rv[1][tt] + rv[2][:T][None]
    ref = torch.nn.functional.layer_norm(e, (H,), rv[3], rv[4], 1e-12).reshape(B * T, H)
    ref.backward(dy.float())
    close(y, ref, 1e-2, 2e-2, "y")
    for got, want, what in zip(dv, rv, ("dword", "dtype", "dpos", "dgamma", "dbeta")):
        # table gradients are sums over up to B*T = 60 rows of a bf16-stored gradient (entries up to ~10: half an ulp 2e-2 each)
        close(got.grad, want.grad, 3e-2, 3e-1 if what == "dtype" else 1e-1, what)
    assert float(dv[0].grad[0].abs().max()) == 0.0                       # padding_idx row
    # dropout: seeded and unbiased
    a = torch.ops.mmbert.embed_ln(ids.to(DEV), None, *[t.detach() for t in dv], 1e-12, 0.25, 7)
    b = torch.ops.mmbert.embed_ln(ids.to(DEV), None, *[t.detach() for t in dv], 1e-12, 0.25, 7)
    assert torch.equal(a, b) and 0.15 < float((a == 0).float().mean()) < 0.35


@pytest.mark.parametrize("D", [35, 74])
def test_joint_embed_operator_forward_and_gradients(D):
    """torch.ops.mmbert.joint_embed = JointEmbeddings (REF:MMBertEmbedding.py:57-72): cat(text_emb, relu(W pair + b)) -> LayerNorm(1e-5)."""
    B, T, P, H = 2, 10, 24, 128
    te, pair = rnd(B, T, H, seed=1).bfloat16(), rnd(B, P, D, seed=2)
    W, b = rnd(H, D, seed=3) * 0.2, rnd(H, seed=4) * 0.1
    gam, bet = 1.0 + 0.1 * rnd(H, seed=5), 0.1 * rnd(H, seed=6)
    dy = rnd(B, T + P, H, seed=7).bfloat16()
    ted = te.to(DEV).requires_grad_(True)
    dv = [t.to(DEV).requires_grad_(True) for t in (W, b, gam, bet)]
    y = torch.ops.mmbert.joint_embed(ted, pair.to(DEV), *dv, 1e-5)
    assert y.shape == (B, T + P, H)
    y.backward(dy.to(DEV))
    ter = te.float().requires_grad_(True)
    rv = [t.clone().requires_grad_(True) for t in (W, b, gam, bet)]
    cat = torch.cat((ter, torch.relu(torch.nn.functional.linear(pair, rv[0], rv[1]))), dim=1)
    ref = torch.nn.functional.layer_norm(cat, (H,), rv[2], rv[3], 1e-5)
    ref.backward(dy.float())
    close(y, ref, 1e-2, 2e-2, "y")
    close(ted.grad, ter.grad, 3e-2, 3e-2, "dtext")
    for got, want, what in zip(dv, rv, ("dW", "db", "dgamma", "dbeta")):
        close(got.grad, want.grad, 3e-2, 8e-2, what)


def test_mlm_head_ce_operator_forward_and_gradient():
    """torch.ops.mmbert.mlm_head_ce = CrossEntropyLoss(ignore_index=-100, mean) over vocabulary rows with a padded leading dimension."""
    M, V, ld = 64, 1000, 1024
    logits = torch.zeros(M, ld)
    logits[:, :V] = rnd(M, V, seed=1) * 2
    logits = logits.bfloat16()
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2))
    labels[::3] = -100
    ld_ = logits.to(DEV).requires_grad_(True)
    loss = torch.ops.mmbert.mlm_head_ce(ld_, labels.to(DEV), V)
    (loss * 1.7).backward()
    lr = logits.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr[:, :V], labels, ignore_index=-100)
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 2e-3 * abs(float(ref))
    close(ld_.grad[:, :V], lr.grad[:, :V], 3e-2, 2e-4, "dlogits")
    assert float(ld_.grad[:, V:].abs().max()) == 0.0 and float(ld_.grad[::3].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["hf", "torch"])
def test_adamw_multi_tensor_operator(mode):
    n = 256 * 40
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2) * 0.1
    flags = torch.tensor([0, 1, 2, 1] * 10, dtype=torch.uint8)
    p, g = p0.to(DEV), g0.to(DEV)
    m, v, pb = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    torch.ops.mmbert.adamw_multi_tensor(p, g, m, v, pb, flags.to(DEV), 1e-2, 0.9, 0.999, 1e-6, 0.01, 1, 1.0, mode, True)
    pr = p0.clone()
    f = flags.repeat_interleave(256)
    mm, vv = 0.1 * g0, 0.001 * g0 * g0
    if mode == "hf":
        upd = pr - (1e-2 * (1 - 0.999) ** 0.5 / (1 - 0.9)) * mm / (vv.sqrt() + 1e-6)
        upd = torch.where(f == 1, upd - 1e-2 * 0.01 * upd, upd)
    else:
        base = torch.where(f == 1, pr * (1 - 1e-2 * 0.01), pr)
        upd = base - (1e-2 / (1 - 0.9)) * mm / (vv.sqrt() / (1 - 0.999) ** 0.5 + 1e-6)
    want = torch.where(f == 2, pr, upd)
    close(p, want, 1e-5, 1e-6, "p")
    assert float(g.abs().max()) == 0.0 and torch.equal(pb.float().cpu(), p.cpu().bfloat16().float())


def test_mlm_mask_rng_operator():
    """torch.ops.mmbert.mlm_mask_rng (mmbert_mlm_mask): the reference's masking rule (REF:model_utils.py:6-39) from the library's
    counter RNG -- selection rate, the special ids never selected ([PAD] is NOT special in the reference), 80 % of the selected ->
    [MASK], labels = original ids where selected and -100 elsewhere, untouched inputs elsewhere; seeded."""
    ids0 = torch.randint(1000, 30000, (64, 50), generator=torch.Generator().manual_seed(3))
    ids0[:, 0], ids0[:, 30], ids0[:, 31:] = 101, 102, 0
    a = ids0.clone().to(DEV)
    lab = torch.ops.mmbert.mlm_mask_rng(a, 0.15, 9, [101, 102], 103)
    sel = (lab != -100).cpu()
    assert not sel[:, 0].any() and not sel[:, 30].any() and sel[:, 31:].any()
    assert 0.12 < float(sel[:, 1:30].float().mean()) < 0.18 and 0.10 < float(sel[:, 31:].float().mean()) < 0.20
    assert torch.equal(lab.cpu()[sel], ids0[sel]) and torch.equal(a.cpu()[~sel], ids0[~sel])
    rep = (a.cpu() == 103) & sel
    assert 0.7 < float(rep.sum()) / float(sel.sum()) < 0.9 and torch.equal(a.cpu()[sel & ~rep], ids0[sel & ~rep])
    b = ids0.clone().to(DEV)
    lab2 = torch.ops.mmbert.mlm_mask_rng(b, 0.15, 9, [101, 102], 103)
    assert torch.equal(lab, lab2) and torch.equal(a, b)
    c = ids0.clone().to(DEV)
    assert not torch.equal(torch.ops.mmbert.mlm_mask_rng(c, 0.15, 10, [101, 102], 103), lab)
    # trainer.mask_tokens on CUDA tensors goes through the same kernel
    from msa_amd import trainer as T
    torch.manual_seed(5)
    out, labels = T.mask_tokens(ids0.clone().to(DEV), T.default_args(mlm_probability=0.15))
    s2 = labels != -100
    assert 0.10 < float(s2.float().mean()) < 0.20 and not bool(s2[:, 0].any()) and bool(((out == 103) <= s2).all())
    # ... seeded by torch's CUDA generator: re-seeding in the middle of a process reproduces the draws, consecutive calls differ,
    # and the CUDA RNG state captures / restores them (the round-2 seed was a process-wide call counter)
    out2, labels2 = T.mask_tokens(ids0.clone().to(DEV), T.default_args(mlm_probability=0.15))
    assert not torch.equal(labels2, labels)
    state = torch.cuda.get_rng_state()
    out3, labels3 = T.mask_tokens(ids0.clone().to(DEV), T.default_args(mlm_probability=0.15))
    torch.cuda.set_rng_state(state)
    out4, labels4 = T.mask_tokens(ids0.clone().to(DEV), T.default_args(mlm_probability=0.15))
    assert torch.equal(labels3, labels4) and torch.equal(out3, out4)
    torch.manual_seed(5)
    out5, labels5 = T.mask_tokens(ids0.clone().to(DEV), T.default_args(mlm_probability=0.15))
    assert torch.equal(labels5, labels) and torch.equal(out5, out)
