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
