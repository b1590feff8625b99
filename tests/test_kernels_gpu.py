"""Per-kernel parity: every C-ABI entry point against a plain fp32 CPU computation of the same op on
the same (bf16-rounded) inputs.  Tolerances: outputs are bf16 (rel 2^-8) of fp32-accumulated sums."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def bf(x):
    return x.to(torch.bfloat16)


def assert_close(got, ref, rtol, atol, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} at ref {float(ref.flatten()[err.argmax()]):.4g}"


@pytest.fixture(scope="module")
def ops():
    from msa_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------- GEMM
def test_gemm_nt_exact_integers_asymmetric(ops):
    """A = [I | 0] picks rows of an ASYMMETRIC integer B: catches swapped row/col fragment maps."""
    M, N, K = 128, 256, 128
    A = torch.zeros(M, K)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    B = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 7) % 61 - 30.0
    out = ops.gemm_nt(bf(A).to(DEV), bf(B).to(DEV))
    assert torch.equal(out.float().cpu(), (A @ B.t()))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (800, 768, 768), (1150, 2304, 768), (333, 512, 3072), (64, 30592, 128)])
def test_gemm_nt_plain_and_bias(ops, M, N, K):
    A, B, bias = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.05)), rnd(N, seed=3)
    ref = A.float() @ B.float().t()
    assert_close(ops.gemm_nt(A.to(DEV), B.to(DEV)), ref, 1e-2, 2e-2, "plain")
    assert_close(ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV)), ref + bias, 1e-2, 2e-2, "bias")
    assert_close(ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), out_f32=True, alpha=0.5), 0.5 * ref + bias, 1e-4, 1e-3, "f32 out")


class forced_nt:
    """mmbert_gemm_nt_force(mode) for a block: 0 by shape | 1 the 128 x 128 kernel | 8 the 8-phase kernel wherever eligible | 128 / 192 /
    224 / 256 the 8-phase kernel on that tile height (single- or multi-tile form by the tile count)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from msa_amd import _lib
        _lib.load().mmbert_gemm_nt_force(self.mode)

    def __exit__(self, *exc):
        from msa_amd import _lib
        _lib.load().mmbert_gemm_nt_force(0)


EPI_CODE = {"plain": 0, "bias": 1, "gelu": 3, "bias_resid_drop": 5, "resid_drop": 5, "resid": 4, "gelu_bwd": 8, "bias_f32": 17}


def epilogue_cases(ops, A, B, N, seeds, drop_site):
    """Every epilogue instantiation of mmbert_gemm_nt on (A, B) with its fp32 torch reference (dropout through the exported mask)."""
    M = A.shape[0]
    bias, R = rnd(N, seed=seeds[0]).to(DEV), bf(rnd(M, N, seed=seeds[1])).to(DEV)
    ref = A.float() @ B.float().t()
    drop = ops.make_drop(0.1, *drop_site)
    keep = ops.dropout_mask(M * N, drop, DEV).view(M, N).float()
    u = R.float().requires_grad_(True)
    torch.nn.functional.gelu(u).sum().backward()
    return {"plain": ({}, ref), "bias": (dict(bias=bias), ref + bias), "resid": (dict(resid=R), ref + R.float()),
            "bias_resid_drop": (dict(bias=bias, resid=R, drop=drop), (ref + bias) * keep * drop[2] + R.float()),
            "gelu": (dict(bias=bias, gelu=True), torch.nn.functional.gelu(ref + bias)), "gelu_bwd": (dict(gelu_bwd_u=R), ref * u.grad),
            "bias_f32": (dict(bias=bias, out_f32=True), ref + bias)}


# (the K = 128 shapes are not eligible for the 8-phase kernel: forcing a tile height on them re-runs the 128 x 128 kernel -- one forced mode
# besides the plain one is kept for the small shape, the vocabulary-sized one runs once: round 6, GPU-suite budget)
_NT_FORM_CASES = ([(mode, M, N, K) for mode in (1, 8, 128, 192, 224, 256) for (M, N, K) in ((700, 768, 768), (1150, 2304, 768), (520, 768, 3072))]
                  + [(1, 256, 256, 128), (224, 256, 256, 128), (1, 2048, 30592, 128)])


@pytest.mark.parametrize("mode,M,N,K", _NT_FORM_CASES)
def test_gemm_nt_every_kernel_form_all_epilogues(ops, mode, M, N, K):
    """The 128 x 128 kernel and the 8-phase kernel at each of its tile heights, forced, on ragged shapes (M, N not tile multiples; the
    K = 128 shapes are not eligible for the 8-phase kernel and stay on the 128 x 128 one whatever is forced)."""
    with forced_nt(mode):
        d = ops.gemm_nt_describe(M, N, K)
        assert d["kernel"] == ("8phase" if mode != 1 and K >= 256 else "128x128"), d
        assert mode in (1, 8) or K < 256 or d["tile"] == f"{mode}x256", d
        A, B, bias, R, U = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.05)), rnd(N, seed=3), bf(rnd(M, N, seed=4)), bf(rnd(M, N, seed=5))
        ref = A.float() @ B.float().t()
        Ad, Bd = A.to(DEV), B.to(DEV)
        assert_close(ops.gemm_nt(Ad, Bd), ref, 1e-2, 2e-2, "plain")
        assert_close(ops.gemm_nt(Ad, Bd, bias=bias.to(DEV)), ref + bias, 1e-2, 2e-2, "bias")
        assert_close(ops.gemm_nt(Ad, Bd, bias=bias.to(DEV), gelu=True), torch.nn.functional.gelu(ref + bias), 1e-2, 2e-2, "gelu")
        assert_close(ops.gemm_nt(Ad, Bd, bias=bias.to(DEV), resid=R.to(DEV)), ref + bias + R.float(), 1e-2, 2e-2, "resid")
        u = U.float().requires_grad_(True)
        torch.nn.functional.gelu(u).sum().backward()
        assert_close(ops.gemm_nt(Ad, Bd, gelu_bwd_u=U.to(DEV)), ref * u.grad, 1e-2, 2e-2, "gelu bwd")


@pytest.mark.parametrize("mode,M", [(0, 18400), (1, 384), (8, 450), (8, 18400), (128, 450), (128, 18400), (192, 450), (192, 18400), (224, 384),
                                    (224, 18400), (256, 384), (256, 18400), (0, 5000), (0, 18400 // 8)])
def test_gemm_nt_forced_forms_exact_integers(ops, mode, M):
    """Every tile of every form exactly once, on the right rows: small integers are exact in bf16 and in the fp32 accumulators."""
    with forced_nt(mode):
        N, K = 512, 256
        A = ((torch.arange(M)[:, None] * 5 + torch.arange(K)[None, :] * 3) % 7 - 3.0)
        B = ((torch.arange(N)[:, None] * 2 + torch.arange(K)[None, :] * 11) % 5 - 2.0)
        out = ops.gemm_nt(bf(A).to(DEV), bf(B).to(DEV), out_f32=True)
        assert torch.equal(out.cpu(), A @ B.t())


@pytest.mark.parametrize("rows,M,N,K", [(224, 18400 - 37, 768, 768), (224, 18400 - 37, 768, 3072), (192, 13850 - 37, 768, 768), (192, 13850 - 37, 768, 2304),
                                        (192, 13850 - 37, 768, 3072), (128, 6400 - 37, 1024, 1024), (128, 6400 - 37, 1024, 4096)])
def test_gemm_nt_single_round_tile_heights_every_epilogue(ops, rows, M, N, K):
    """The launches whose tiles fit the chip in ONE round take the smallest tile height that still fits: at the step's own sizes, ragged last
    row panel -- 224-row tiles (A half 0 = 128 rows: two LDS-DMA pieces per wave; A half 1 = 96: two for waves 0-3, one for waves 4-7) for
    the forward N = 768 shapes at 18 400 rows (249 tiles instead of 216), 192-row tiles (two compile-time wave classes with their own
    counted vmcnt) for the input gradients at ~13 850 packed rows (219 instead of 165), 128-row tiles for the reference's default model
    (bert-large: 6 400 rows x N = 1024, REF:train.py:28,32,38, where 256-row tiles would leave 60 % of the chip idle).  The default
    dispatch picks that height; every epilogue against fp32 torch; the SAME BITS as the 256-row form (the K order per output element
    does not depend on the tile height); exact on small integers."""
    d = ops.gemm_nt_describe(M, N, K)
    assert d["kernel"] == "8phase" and d["tile"] == f"{rows}x256" and d["tiles"] <= d["cus"], d
    A, B = bf(rnd(M, K, seed=21, scale=0.5)).to(DEV), bf(rnd(N, K, seed=22, scale=0.05)).to(DEV)
    for name, (kw, want) in epilogue_cases(ops, A, B, N, (23, 24), (11, 5)).items():
        got = ops.gemm_nt(A, B, **kw)
        assert_close(got, want, 1e-2, 4e-2 if rows == 128 else 3e-2, name)
        with forced_nt(256):
            assert ops.gemm_nt_describe(M, N, K)["tile"] == "256x256"
            other = ops.gemm_nt(A, B, **kw)
        assert torch.equal(got, other), (name, float((got.float() - other.float()).abs().max()))
    aux = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)                      # the GELU epilogue's second output
    bias = rnd(N, seed=23).to(DEV)
    ops.gemm_nt(A, B, bias=bias, gelu=True, aux=aux)
    assert_close(aux, A.float() @ B.float().t() + bias, 1e-2, 3e-2, "gelu aux")
    Ai = ((torch.arange(M)[:, None] * 5 + torch.arange(K)[None, :] * 3) % 3 - 1.0)
    Bi = ((torch.arange(N)[:, None] * 2 + torch.arange(K)[None, :] * 11) % 2).float()
    assert torch.equal(ops.gemm_nt(bf(Ai).to(DEV), bf(Bi).to(DEV), out_f32=True), Ai.to(DEV) @ Bi.to(DEV).t())


@pytest.mark.parametrize("mode", [0, 192, 256])
@pytest.mark.parametrize("M,N,K", [(18400, 2304, 256), (14000, 3072, 256), (5000, 3072, 256), (9000, 1792, 256), (18400 - 37, 2304, 768)])
def test_gemm_nt_grouped_tile_walk_exact(ops, M, N, K, mode):
    """More tiles than CUs: the multi-tile form (the half-tile stream crosses the tile seams) walks them in one group per XCD
    (ceil(row tiles / 8) row panels swept over the column panels, the last group short) -- every tile exactly once, checked on small
    integers (exact in bf16 and in the fp32 accumulators), at the default height and at the forced ones."""
    with forced_nt(mode):
        d = ops.gemm_nt_describe(M, N, K)
        assert d["kernel"] == "8phase", d
        if mode == 0 and (M, N) in ((18400, 2304), (14000, 3072), (18400 - 37, 2304)):
            assert d["tile"] == "224x256" and d["tiles"] > d["cus"] and d["workgroups"] == d["cus"] and d["group_m"] > 1, d
        A = ((torch.arange(M)[:, None] * 5 + torch.arange(K)[None, :] * 3) % 3 - 1.0)      # entries in {-1, 0, 1} x {0, 1}: |sums| <= K, exact in bf16 up to 256
        B = ((torch.arange(N)[:, None] * 2 + torch.arange(K)[None, :] * 11) % 2).float()
        out = ops.gemm_nt(bf(A).to(DEV), bf(B).to(DEV), out_f32=K > 256)
        assert torch.equal(out.float(), A.to(DEV) @ B.to(DEV).t())


@pytest.mark.parametrize("epi", ["bias", "gelu", "gelu_bwd", "resid_drop"])
def test_gemm_nt_multi_tile_form_same_bits_at_every_height(ops, epi):
    """The multi-tile form with the fused epilogues of the step's multi-round launches (QKV: bias; FFN-up: bias + GELU + pre-activation;
    GELU' input gradient; a dropout + residual epilogue) and a ragged last row panel: the default (224-row tiles) against fp32 torch,
    reproducible run to run, and the SAME BITS at 192- and 256-row tiles (the K order per element does not depend on the height; the
    224-row form reproduced the retired ring kernel's bits in round 4) and from the 128 x 128 kernel to one bf16 ulp of the largest
    entry (another summation order); the 192-row form (two wave classes) also on the device tile queue."""
    import msa_amd.ops as O
    M, N, K = 18400 - 37, 2304, 768
    A, B = bf(rnd(M, K, seed=31, scale=0.5)).to(DEV), bf(rnd(N, K, seed=32, scale=0.05)).to(DEV)
    bias, R = rnd(N, seed=33).to(DEV), bf(rnd(M, N, seed=34)).to(DEV)
    kw = {"bias": dict(bias=bias), "gelu": dict(bias=bias, gelu=True), "gelu_bwd": dict(gelu_bwd_u=R),
          "resid_drop": dict(bias=bias, resid=R, drop=ops.make_drop(0.1, 3, 4))}[epi]

    def run():
        aux = torch.empty((M, N), device=DEV, dtype=torch.bfloat16) if epi == "gelu" else None
        return ops.gemm_nt(A, B, aux=aux, **kw), aux
    d = ops.gemm_nt_describe(M, N, K, epi=EPI_CODE[epi])
    assert d["kernel"] == "8phase" and d["tile"] == "224x256" and d["tiles"] > d["cus"], d
    got, gaux = run()
    assert torch.equal(run()[0], got)
    if epi != "resid_drop":
        want = epilogue_cases(ops, A, B, N, (33, 34), (3, 4))[epi][1]
        assert_close(got, want, 1e-2, 3e-2, epi)
    with forced_nt(1):
        small, saux = run()
    scale = float(small.float().abs().max())
    assert float((got.float() - small.float()).abs().max()) <= 2.0 ** -7 * scale
    assert gaux is None or float((gaux.float() - saux.float()).abs().max()) <= 2.0 ** -7 * float(saux.float().abs().max())
    for mh in (192, 256):
        with forced_nt(mh):
            d = ops.gemm_nt_describe(M, N, K, epi=EPI_CODE[epi])
            assert d["kernel"] == "8phase" and d["tile"] == f"{mh}x256" and d["tiles"] > d["cus"], d
            out, aux = run()
            assert torch.equal(out, got), mh
            assert aux is None or torch.equal(aux, gaux), mh
            if mh == 192:
                was = O.dynamic_tile_queue
                try:
                    O.dynamic_tile_queue = True
                    for rep in range(2):
                        assert torch.equal(run()[0], got), (mh, "queue", rep)
                finally:
                    O.dynamic_tile_queue = was
    # the height rule itself (host-only): more valid rows in backward -> 256-row tiles save a round; bert-large's QKV -> 192-row tiles
    assert ops.gemm_nt_describe(14400, 3072, 768, epi=8)["tile"] == "256x256"
    assert ops.gemm_nt_describe(13850, 3072, 768, epi=8)["tile"] == "224x256"
    assert ops.gemm_nt_describe(6400, 3072, 1024, epi=1)["tile"] == "192x256"


def test_gemm_nt_vocabulary_projection_on_the_multi_tile_form(ops):
    """The vocabulary projection (N = 30 592: 120 column panels, tile walk in groups of 4 row panels) through the default dispatch -- the
    multi-tile form on 224-row tiles -- against fp32 torch on a row sample, bf16 and fp32 scores, ragged last row panel; the same bits
    on 256-row tiles."""
    M, N, K = 4600 - 37, 30592, 768
    A, B, bias = bf(rnd(M, K, seed=401, scale=0.5)).to(DEV), bf(rnd(N, K, seed=402, scale=0.05)).to(DEV), rnd(N, seed=403).to(DEV)
    d = ops.gemm_nt_describe(M, N, K, epi=1)
    assert d["kernel"] == "8phase" and d["tile"] == "224x256" and d["tiles"] > 4 * d["cus"] and d["group_m"] == 4, d
    got, got32 = ops.gemm_nt(A, B, bias=bias), ops.gemm_nt(A, B, bias=bias, out_f32=True)
    with forced_nt(256):
        ref, ref32 = ops.gemm_nt(A, B, bias=bias), ops.gemm_nt(A, B, bias=bias, out_f32=True)
    assert torch.equal(got, ref) and torch.equal(got32, ref32)
    assert torch.equal(got, got32.bfloat16())
    rows = torch.arange(0, M, 97, device=DEV)
    assert_close(got32[rows], A[rows].float() @ B.float().t() + bias, 1e-2, 3e-2, "vocabulary rows")


@pytest.mark.parametrize("mode", [0, 192, 256])
@pytest.mark.parametrize("epi", ["plain", "bias", "gelu", "resid_drop", "resid", "gelu_bwd", "bias_f32"])
def test_gemm_nt_tile_queue_bit_identical_every_instantiation(ops, epi, mode):
    """the queue's fetch is an inline-asm atomic whose result is read behind a hand-counted s_waitcnt (a compiler-inserted copy
    of that register before the wait would read stale data -> duplicate or missing tiles).  So: EVERY epilogue instantiation of the
    multi-tile form at each tile height (gemm_nt8_kernel<EPI, true, 7 / 3 / 4>: the draw sits between K tiles 0 and 1, its result is
    published behind K tile 1's counted wait) on a multi-round shape whose last row panel AND last column panel are ragged, the device
    tile queue (what DataParallel switches on for N > 1) against the static walk: the same bits, three launches in a row on one queue
    buffer (the last workgroup out hands it back zeroed)."""
    import msa_amd.ops as O
    M, N, K = 18400 - 37, 3072 - 40, 768
    A, B = bf(rnd(M, K, seed=311, scale=0.1)).to(DEV), bf(rnd(N, K, seed=312, scale=0.1)).to(DEV)
    bias, R = rnd(N, seed=313).to(DEV), bf(rnd(M, N, seed=314)).to(DEV)
    kw = {"plain": {}, "bias": dict(bias=bias), "gelu": dict(bias=bias, gelu=True), "resid_drop": dict(bias=bias, resid=R, drop=ops.make_drop(0.1, 5, 9)),
          "resid": dict(resid=R), "gelu_bwd": dict(gelu_bwd_u=R), "bias_f32": dict(bias=bias, out_f32=True)}[epi]
    was = O.dynamic_tile_queue
    try:
        with forced_nt(mode):
            d = ops.gemm_nt_describe(M, N, K, with_queue=True)
            assert d["kernel"] == "8phase" and d["tiles"] > d["cus"] and d["tile"] == ("224x256" if mode == 0 else f"{mode}x256"), d
            O.dynamic_tile_queue = False
            aux0 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16) if epi == "gelu" else None
            ref = ops.gemm_nt(A, B, aux=aux0, **kw)
            O.dynamic_tile_queue = True
            for rep in range(3):
                aux1 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16) if epi == "gelu" else None
                out = ops.gemm_nt(A, B, aux=aux1, **kw)
                assert torch.equal(out, ref), (epi, mode, rep)
                assert aux0 is None or torch.equal(aux1, aux0), (epi, mode, rep)
    finally:
        O.dynamic_tile_queue = was


def test_dropout_hash_pairwise_keep_correlations(ops):
    """mmb_pair_mix is one xorshift16-multiply-xorshift16 round on a Weyl sequence and each 16-bit half of a word decides one
    element, so rates alone do not show whether the two halves of a word, neighbouring pairs or neighbouring rows are independent.
    Pearson correlation of the keep flags at p = 0.1 and 0.5 over 2^23 elements laid out as rows of H = 768: element 2i vs 2i + 1 (the
    two halves of one word), pair i vs i + 1 (same half of neighbouring words), row r vs r + 1 (768 elements = 384 words apart), and
    two different sites (streams) at the same index.  Independent bits give |rho| ~ 1 / sqrt(n) = 3.5e-4; bound 2e-3."""
    H, rows = 768, 8192
    n = H * rows
    for p_drop in (0.1, 0.5):
        k = ops.dropout_mask(n, ops.make_drop(p_drop, 12345, 77), DEV).float().view(rows, H)
        k2 = ops.dropout_mask(n, ops.make_drop(p_drop, 12345, 78), DEV).float().view(rows, H)
        rate = float(k.mean())
        assert abs(rate - (1 - p_drop)) < 2e-3, (p_drop, rate)

        def rho(a, b):
            a, b = a.reshape(-1) - a.mean(), b.reshape(-1) - b.mean()
            return float((a * b).mean() / (a.std() * b.std() + 1e-12))
        checks = {"halves of one word": rho(k[:, 0::2], k[:, 1::2]), "low halves of neighbouring words": rho(k[:, 0:-2:2], k[:, 2::2]),
                  "high half vs next word's low half": rho(k[:, 1:-1:2], k[:, 2::2]), "rows r, r + 1": rho(k[:-1], k[1:]),
                  "rows r, r + 2": rho(k[:-2], k[2:]), "two sites": rho(k, k2)}
        for what, r in checks.items():
            assert abs(r) < 2e-3, (p_drop, what, r)


@pytest.mark.parametrize("M,N,K", [(300, 256, 4096), (77, 768, 30592), (130, 132, 1024)])
def test_gemm_nt_splitk(ops, M, N, K):
    A, B = bf(rnd(M, K, seed=15, scale=0.1)), bf(rnd(N, K, seed=16, scale=0.1))
    out = ops.gemm_nt_splitk(A.to(DEV), B.to(DEV))
    assert_close(out, A.float() @ B.float().t(), 1e-2, 2e-2, "split-K")
    again = ops.gemm_nt_splitk(A.to(DEV), B.to(DEV))
    assert torch.equal(out, again)                                   # slabs + ordered reduction: reproducible
    R = bf(rnd(M, N + 8, seed=17))[:, :N]                            # the residual form (row pitch != N)
    with_r = ops.gemm_nt_splitk(A.to(DEV), B.to(DEV), resid=R.to(DEV)[:, :N])
    assert_close(with_r, A.float() @ B.float().t() + R.float(), 1e-2, 2e-2, "split-K + residual")


def test_gemm_nt_strided_views_and_alpha_dev(ops):
    M, N, K = 300, 256, 192
    big = bf(rnd(M, 3 * K, seed=4)).to(DEV)
    A = big[:, K:2 * K]
    B = bf(rnd(N, K, seed=5, scale=0.1)).to(DEV)
    ad = torch.tensor([0.25], device=DEV)
    out = torch.zeros(M, 2 * N, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(A, B, out=out[:, N:], alpha=2.0, alpha_dev=ad)
    ref = 0.5 * (A.float().cpu() @ B.float().cpu().t())
    assert_close(out[:, N:], ref, 1e-2, 2e-2, "strided")
    assert float(out[:, :N].abs().max()) == 0.0


def test_gelu_and_its_derivative_on_a_dense_grid(ops):
    """The single-transcendental GELU / GELU' of csrc/common.h against torch's erf form on every bf16 value of [-14, 14]
    (plus +-100, +-1e4): the only error left is the bf16 rounding of the result (half an ulp = 2^-9 relative) plus the
    approximation floor (5.7e-7 / 4.1e-6 absolute).  GELU is reached through the persistent GEMM's epilogue (K = 256,
    A = [x | 0], B = [I | 0] -> the product is x exactly), GELU' through mmbert_gelu_bwd with dy = 1."""
    xs = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
    xs = xs[torch.isfinite(xs) & (xs.abs() <= 14.0)]
    xs = torch.cat([xs, torch.tensor([100.0, -100.0, 1e4, -1e4])])
    xs = torch.cat([xs, torch.zeros(-xs.numel() % 256)])                      # whole rows of 256 (kernels take multiples of 8)
    n = xs.numel()
    x64 = xs.double()
    gelu = (0.5 * x64 * (1.0 + torch.erf(x64 / math.sqrt(2.0))))
    grad = 0.5 * (1.0 + torch.erf(x64 / math.sqrt(2.0))) + x64 * torch.exp(-0.5 * x64 * x64) / math.sqrt(2.0 * math.pi)
    # derivative
    u = bf(xs).to(DEV).contiguous()
    du = ops.gelu_bwd(torch.ones_like(u), u)
    assert_close(du, grad, 2.0 ** -8, 5e-6, "gelu'")
    # forward, through the GEMM epilogue: one grid value per output element
    M = (n + 255) // 256
    X = torch.zeros(M * 256)
    X[:n] = xs
    ref = torch.zeros(M * 256, dtype=torch.float64)
    ref[:n] = gelu
    got = ops.gemm_nt(bf(X.view(M, 256)).to(DEV), bf(torch.eye(256)).to(DEV), gelu=True, bias=torch.zeros(256, device=DEV))
    assert_close(got.reshape(-1), ref, 2.0 ** -8, 1e-6, "gelu")


def test_gemm_nt_gelu_resid_gelubwd(ops):
    M, N, K = 257, 512, 128
    A, B, bias = bf(rnd(M, K, seed=6)), bf(rnd(N, K, seed=7, scale=0.1)), rnd(N, seed=8)
    pre = A.float() @ B.float().t() + bias
    aux = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), gelu=True, aux=aux)
    assert_close(aux, pre, 1e-2, 1e-2, "pre-activation")
    assert_close(out, torch.nn.functional.gelu(pre), 1e-2, 1e-2, "gelu")
    R = bf(rnd(M, N, seed=9))
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), resid=R.to(DEV))
    assert_close(out, pre + R.float(), 1e-2, 2e-2, "bias+resid")
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), resid=R.to(DEV))
    assert_close(out, pre - bias + R.float(), 1e-2, 2e-2, "resid")
    U = bf(rnd(M, N, seed=10))
    u = U.float().requires_grad_(True)
    torch.nn.functional.gelu(u).sum().backward()
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), gelu_bwd_u=U.to(DEV))
    assert_close(out, (pre - bias) * u.grad, 1e-2, 2e-2, "gelu bwd")


def test_gemm_nt_dropout_epilogue_matches_exported_mask(ops):
    M, N, K = 130, 256, 64
    A, B, bias, R = bf(rnd(M, K, seed=11)), bf(rnd(N, K, seed=12, scale=0.1)), rnd(N, seed=13), bf(rnd(M, N, seed=14))
    drop = ops.make_drop(0.1, seed=1234, site=7)
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), resid=R.to(DEV), drop=drop)
    mask = ops.dropout_mask(M * N, drop, DEV).view(M, N).float().cpu()
    assert 0.88 < float(mask.mean()) < 0.92
    ref = (A.float() @ B.float().t() + bias) * mask * drop[2] + R.float()
    assert_close(out, ref, 1e-2, 2e-2, "dropout epilogue")


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (300, 128, 256), (1150, 768, 768), (2000, 256, 3072), (999, 2304, 128), (1000, 2048, 64)])
def test_gemm_tn(ops, M, N, K):
    A, B = bf(rnd(M, N, seed=20, scale=0.1)), bf(rnd(M, K, seed=21))
    ref = A.float().t() @ B.float()
    W0 = rnd(N, K, seed=22)
    W = W0.clone().to(DEV)
    ops.gemm_tn(A.to(DEV), B.to(DEV), W, accumulate=True, alpha=0.5)
    assert_close(W, W0 + 0.5 * ref, 2e-3, 2e-3 * math.sqrt(M), "accumulate")
    ops.gemm_tn(A.to(DEV), B.to(DEV), W, accumulate=False)
    assert_close(W, ref, 2e-3, 2e-3 * math.sqrt(M), "overwrite")


@pytest.mark.parametrize("shapes", [[(3072, 768), (768, 3072), (2304, 768), (768, 768)],
                                    [(3072, 768), (768, 3072), (2304, 768), (768, 768), (1024, 256), (768, 768), (256, 1024), (520, 136)]])
def test_gemm_tn_fused_bias_and_grouped(ops, shapes):
    """Four problems (an encoder layer's dense layers) and eight (two layers: the paired launch of the encoder's backward) sharing M."""
    M = 1700
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        A, B = bf(rnd(M, N, seed=70 + i, scale=0.1)), bf(rnd(M, K, seed=80 + i))
        W0, b0 = rnd(N, K, seed=90 + i), rnd(N, seed=95 + i)
        bias = b0.clone().to(DEV) if i % 2 == 0 else None
        probs.append((A.to(DEV), B.to(DEV), W0.clone().to(DEV), bias))
        refs.append((W0 + A.float().t() @ B.float(), (b0 + A.float().sum(0)) if bias is not None else None))
    ops.gemm_tn_grouped(probs)
    for (A, B, W, bias), (rw, rb) in zip(probs, refs):
        assert_close(W, rw, 2e-3, 2e-3 * math.sqrt(M), "grouped W")
        if bias is not None:
            assert_close(bias, rb, 2e-3, 2e-2, "grouped bias")
    A, B = bf(rnd(999, 520, seed=60, scale=0.1)), bf(rnd(999, 136, seed=61))
    W, bias = torch.zeros(520, 136, device=DEV), torch.ones(520, device=DEV)
    ops.gemm_tn(A.to(DEV), B.to(DEV), W, accumulate=False, alpha=0.5, bias_out=bias)
    assert_close(W, 0.5 * (A.float().t() @ B.float()), 2e-3, 5e-2, "single W")
    assert_close(bias, 1.0 + 0.5 * A.float().sum(0), 2e-3, 2e-2, "single bias")


def test_gemm_tn_grouped_many_layers_in_whole_rounds(ops):
    """Round 4: up to 48 problems per call (the deferred weight gradients of up to 12 encoder layers: nothing needs them before the
    optimizer on one GPU).  More tiles than CUs -> no token split, whole rounds of CUs-many tiles, one launch per round.  Five "layers"
    of four problems each (540 tiles of 256 x 256 = 2.1 rounds) on small-integer operands: every weight gradient and every
    ones-operand bias gradient EXACT in fp32, accumulated onto a non-zero start; the planner reports no slabs."""
    import ctypes
    from msa_amd import _lib
    M = 1184
    shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)] * 5
    n = len(shapes)
    Ns, Ks = (ctypes.c_int * n)(*[a for a, _ in shapes]), (ctypes.c_int * n)(*[b for _, b in shapes])
    sp = ctypes.c_int(-1)
    assert _lib.load().mmbert_gemm_tn_grouped_workspace(n, Ns, Ks, M, ctypes.byref(sp)) == 0 and sp.value == 1
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        A = ((torch.arange(M)[:, None] * (3 + i) + torch.arange(N)[None, :] * 5) % 3 - 1.0)          # {-1, 0, 1}
        B = ((torch.arange(M)[:, None] * 2 + torch.arange(K)[None, :] * (7 + i)) % 2).float()        # {0, 1}: |sums| <= M, exact in fp32
        W0 = torch.full((N, K), float(i + 1))
        b0 = torch.full((N,), -2.0) if i % 2 == 0 else None
        probs.append((bf(A).to(DEV), bf(B).to(DEV), W0.clone().to(DEV), b0.clone().to(DEV) if b0 is not None else None))
        refs.append((W0 + A.t() @ B, (b0 + A.sum(0)) if b0 is not None else None))
    ops.gemm_tn_grouped(probs)
    for i, ((A, B, W, bias), (rw, rb)) in enumerate(zip(probs, refs)):
        assert torch.equal(W.cpu(), rw), i
        if bias is not None:
            assert torch.equal(bias.cpu(), rb), i


@pytest.mark.parametrize("layers", [1, 3])
def test_gemm_tn_grouped_few_row_problems_ride_behind_the_long_tiles(ops, layers):
    """Round 6 (mmbert_gemm_tn_grouped_rows): problems of FEWER token rows behind the long ones in one call -- the tied decoder's, the MLM
    transform's and the sparse top layer's weight gradients in the deferred multi-layer call.  One "layer" (108 long tiles: a single launch,
    the few-row tiles behind them) and three (324: a full round, then 68 long tiles + the few-row ones); small-integer operands, so every
    fp32 sum is EXACT whatever the order; per-problem accumulate flags (an overwritten gradient starts from garbage); problems handed over
    in mixed order (ops sorts by rows); ragged row counts."""
    M = 1184
    shapes = [(3072, 768, M), (768, 3072, M), (2304, 768, M), (768, 768, M)] * layers
    shapes += [(30592, 768, 368), (768, 768, 368), (3072, 768, 391), (768, 3072, 391), (768, 768, 391), (520, 136, 7)]
    probs, refs, flags = [], [], []
    for i, (N, K, m) in enumerate(shapes):
        A = ((torch.arange(m)[:, None] * (3 + i) + torch.arange(N)[None, :] * 5) % 3 - 1.0)          # {-1, 0, 1}
        B = ((torch.arange(m)[:, None] * 2 + torch.arange(K)[None, :] * (7 + i)) % 2).float()        # {0, 1}
        acc = (i % 3 != 1)
        W0 = torch.full((N, K), float(i + 1))
        b0 = torch.full((N,), -2.0) if i % 2 == 0 else None
        Wd = W0.clone().to(DEV) if acc else torch.full((N, K), float("nan"), device=DEV)
        probs.append((bf(A).to(DEV), bf(B).to(DEV), Wd, b0.clone().to(DEV) if b0 is not None else None))
        flags.append(acc)
        refs.append(((W0 if acc else 0) + A.t() @ B, (b0 + A.sum(0)) if b0 is not None else None))
    order = list(range(len(probs)))
    order = order[-3:] + order[:-3]                                # three few-row problems in front: the call takes them longest first
    ops.gemm_tn_grouped([probs[i] for i in order], accumulate=[flags[i] for i in order])
    for i, ((A, B, W, bias), (rw, rb)) in enumerate(zip(probs, refs)):
        assert torch.equal(W.cpu(), rw), (i, shapes[i])
        if bias is not None:
            assert torch.equal(bias.cpu(), rb), (i, shapes[i])


@pytest.mark.parametrize("M,splits", [(1700, 0), (1700, 3), (70, 0), (4129, 2), (33, 0)])
def test_gemm_tn_ragged_tokens_forced_splits_and_accumulate(ops, M, splits):
    """The weight-gradient kernel (gemm_tn8_kernel: 64-token K tiles, half-tile stream, transposed fragment reads) on ragged token counts
    (the buffer range check zero-fills past the split's end), forced token splits (fp32 slabs + deterministic reduce), tiles that hang
    over N and K, accumulate on and off, against fp32 torch; and twice in a row: the same bits (round 4 showed these weight gradients
    bit-identical to the retired 4-slot-ring kernel's: same 32-token summation blocks)."""
    from msa_amd import _lib
    lib = _lib.load()
    shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768), (520, 136), (256, 1024)]
    data = []
    for i, (N, K) in enumerate(shapes):
        data.append((bf(rnd(M, N, seed=170 + i, scale=0.1)).to(DEV), bf(rnd(M, K, seed=180 + i)).to(DEV), rnd(N, K, seed=190 + i), rnd(N, seed=195 + i)))
    outs = {}
    try:
        lib.mmbert_gemm_tn_force_splits(splits)
        for rep in (0, 1):
            for accumulate in (True, False):
                probs = [(A, B, W0.clone().to(DEV), (b0.clone().to(DEV) if i % 3 != 2 else None)) for i, (A, B, W0, b0) in enumerate(data)]
                ops.gemm_tn_grouped(probs, accumulate=accumulate, alpha=0.5)
                torch.cuda.synchronize()
                outs[(rep, accumulate)] = [(p[2].cpu(), p[3].cpu() if p[3] is not None else None) for p in probs]
    finally:
        lib.mmbert_gemm_tn_force_splits(0)
    for accumulate in (True, False):
        for i, ((w0, b0), (w1, b1)) in enumerate(zip(outs[(0, accumulate)], outs[(1, accumulate)])):
            assert torch.equal(w0, w1), (accumulate, i, float((w0 - w1).abs().max()))
        for i, (A, B, W0, b0) in enumerate(data):
            ref = (W0 if accumulate else 0) + 0.5 * (A.float().t() @ B.float()).cpu()
            assert_close(outs[(0, accumulate)][i][0], ref, 2e-3, 2e-3 * math.sqrt(M), f"W {i}")
            if outs[(0, accumulate)][i][1] is not None:
                assert_close(outs[(0, accumulate)][i][1], b0 + 0.5 * A.float().sum(0).cpu(), 2e-3, 2e-2, f"bias {i}")


def test_gemm_tn_exact_integers(ops):
    M, N, K = 192, 128, 128
    A = ((torch.arange(M)[:, None] * 5 + torch.arange(N)[None, :] * 3) % 7 - 3.0)
    B = ((torch.arange(M)[:, None] * 2 + torch.arange(K)[None, :] * 11) % 5 - 2.0)
    W = torch.zeros(N, K, device=DEV)
    ops.gemm_tn(bf(A).to(DEV), bf(B).to(DEV), W, accumulate=False)
    assert torch.equal(W.cpu(), A.t() @ B)


def test_gemm_tn_tied_vocab_rows(ops):
    """decoder weight gradient: N = V (not a multiple of 128) rows into a [V,H] fp32 table."""
    M, V, H = 500, 2008, 128
    A, B = bf(rnd(M, V, seed=23, scale=0.05)), bf(rnd(M, H, seed=24))
    W = torch.zeros(V, H, device=DEV)
    ops.gemm_tn(A.to(DEV), B.to(DEV), W, accumulate=True)
    assert_close(W, A.float().t() @ B.float(), 2e-3, 5e-2, "vocab rows")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rows_to_block_equals_the_torch_form(ops, dtype):
    """mmbert_rows_to_block (the local side of the data-parallel compact row exchange) against searchsorted + index_add: duplicates,
    the padding row 0, ids past the vocabulary and ids that are not in the union are all in the batch."""
    V, H, n = 500, 768, 2400
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V + 40, (n,), generator=g)
    ids[:5] = torch.tensor([0, 7, 7, V, V + 3])
    rows = torch.randn(n, H, generator=g).to(dtype)
    other = torch.randint(1, V, (300,), generator=g)
    keep = (ids > 0) & (ids < V)
    union = torch.unique(torch.cat((ids[keep][::2], other)))          # every second local row is NOT in the list
    block = torch.zeros(union.numel(), H, device=DEV)
    ops.rows_to_block(ids.to(DEV), rows.to(DEV), union.to(DEV), V, block)
    ref = torch.zeros(union.numel(), H)
    pos = torch.searchsorted(union, ids.clamp(0, V - 1)).clamp(max=union.numel() - 1)
    hit = keep & (union[pos] == ids)
    ref.index_add_(0, pos[hit], rows[hit].float())
    assert_close(block, ref, 1e-5, 1e-5, "rows_to_block")
    assert float(block.abs().sum()) > 0


def test_colsum(ops):
    X = bf(rnd(1234, 768, seed=25))
    out = torch.ones(768, device=DEV)
    ops.colsum(X.to(DEV), out, alpha=2.0)
    assert_close(out, 1.0 + 2.0 * X.float().sum(0), 1e-4, 1e-2, "colsum")


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,H,eps", [(7, 128, 1e-12), (1000, 768, 1e-12), (513, 1024, 1e-5), (50, 64, 1e-5), (1, 768, 1e-12), (5, 256, 1e-5),
                                     (18400, 768, 1e-12)])
def test_ln_fwd_bwd(ops, M, H, eps):
    x, gamma, beta, dy = bf(rnd(M, H, seed=30)), 1 + 0.1 * rnd(H, seed=31), 0.1 * rnd(H, seed=32), bf(rnd(M, H, seed=33))
    xr = x.float().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (H,), g_, b_, eps)
    y.backward(dy.float())
    out, mean, rstd = ops.ln_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), eps)
    assert_close(out, y, 1e-2, 1e-2, "ln fwd")
    assert_close(mean, x.float().mean(1), 1e-4, 1e-5, "mean")
    dgamma, dbeta = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx = ops.ln_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dgamma, dbeta)
    assert_close(dx, xr.grad, 1e-2, 1e-2, "ln dx")
    assert_close(dgamma, g_.grad, 1e-3, 1e-3 * math.sqrt(M), "dgamma")
    assert_close(dbeta, b_.grad, 1e-3, 1e-3 * math.sqrt(M), "dbeta")


def test_ln_bwd_deferred_reduce_over_calls_of_different_row_counts(ops):
    """One ops.LnDeferred over LayerNorm' calls of different M (the MLM head's few hundred rows, the encoder layers' thousands, the
    embedding stage) -- one mmbert_ln_bwd_reduce_rows launch -- gives every call the gamma / beta sums of its own immediate reduce; more
    calls than slots, and a workspace that has to grow mid-collection, flush on their own."""
    H = 768
    Ms = [360, 13850, 13850, 37, 8000, 800, 1]
    gamma = (1 + 0.1 * rnd(H, seed=70)).to(DEV)
    calls = []
    for q, M in enumerate(Ms):
        x, dy = bf(rnd(M, H, seed=71 + 2 * q)).to(DEV), bf(rnd(M, H, seed=72 + 2 * q)).to(DEV)
        _, mean, rstd = ops.ln_fwd(x, gamma, torch.zeros(H, device=DEV), 1e-12)
        calls.append((x, dy, mean, rstd))
    ref = []
    for x, dy, mean, rstd in calls:
        dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        dx = ops.ln_bwd(dy, x, mean, rstd, gamma, dg, db)
        ref.append((dx, dg, db))
    for slots in (32, 3):
        ops._lnd_cache.clear()                                     # (the persistent workspace starts small again: it has to grow)
        lnd = ops.LnDeferred(slots)
        got = []
        for x, dy, mean, rstd in calls:
            dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
            got.append((ops.ln_bwd(dy, x, mean, rstd, gamma, dg, db, deferred=lnd), dg, db))
        lnd.flush()
        for q, ((dx, dg, db), (rx, rg, rb)) in enumerate(zip(got, ref)):
            assert torch.equal(dx, rx), f"call {q}: dx differs"
            assert_close(dg, rg.cpu(), 1e-5, 1e-5 * math.sqrt(Ms[q]), f"call {q} dgamma (slots {slots})")
            assert_close(db, rb.cpu(), 1e-5, 1e-5 * math.sqrt(Ms[q]), f"call {q} dbeta (slots {slots})")


@pytest.mark.parametrize("M,H,n", [(40, 128, 24), (40, 768, 24), (3000, 256, 2500), (2100, 1024, 2050)])
def test_ln_row_maps_and_dropouts(ops, M, H, n):
    """Row maps on every operand, the post-LN dropout (embedding form) and the branch dropout + second output + bias gradient (encoder
    form) against torch, masks replayed from mmbert_dropout_mask.  H = 128: the generic kernels; H = 256 k: the lean ones (several
    trips per wave at the larger row counts: the request-ahead pipeline and the scalar row records)."""
    x, gamma, beta = bf(rnd(M, H, seed=34)), 1 + 0.1 * rnd(H, seed=35), 0.1 * rnd(H, seed=36)
    in_rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:n].int()
    out_rows = torch.randperm(M, generator=torch.Generator().manual_seed(2))[:n].int()
    post = ops.make_drop(0.5, 99, 1)
    out = torch.zeros(M, H, device=DEV, dtype=torch.bfloat16)
    _, mean, rstd = ops.ln_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, out=out, in_rows=in_rows.to(DEV), out_rows=out_rows.to(DEV), drop=post)
    mask = ops.dropout_mask(n * H, post, DEV).view(n, H).float().cpu()
    assert 0.45 < float(mask.mean()) < 0.55
    xs = x.float()[in_rows.long()].requires_grad_(True)
    y = torch.nn.functional.layer_norm(xs, (H,), gamma, beta, 1e-5) * mask * post[2]
    ref = torch.zeros(M, H)
    ref[out_rows.long()] = y.detach()
    assert_close(out, ref, 1e-2, 1e-2, "mapped ln + post dropout")
    dy = bf(rnd(M, H, seed=37))
    y.backward(dy.float()[out_rows.long()])
    dgamma, dbeta = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    pre = ops.make_drop(0.1, 99, 2)
    dx2 = torch.zeros(n, H, device=DEV, dtype=torch.bfloat16)
    dbias2 = torch.zeros(H, device=DEV)
    dx = ops.ln_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dgamma, dbeta, M=n, dy_rows=out_rows.to(DEV), x_rows=in_rows.to(DEV),
                    post_drop=post, dx2=dx2, pre_drop=pre, dbias2=dbias2)
    assert_close(dx, xs.grad, 1e-2, 1e-2, "mapped ln bwd")
    m2 = ops.dropout_mask(n * H, pre, DEV).view(n, H).float().cpu()
    assert_close(dx2, xs.grad * m2 * pre[2], 1e-2, 1e-2, "branch dropout grad")
    assert_close(dbias2, (xs.grad * m2 * pre[2]).sum(0), 3e-3, 1e-3 * math.sqrt(n), "fused bias gradient")     # (column sums of the bf16 values stored in dx2)
    g_ref = (dy.float()[out_rows.long()] * mask * post[2])
    xh = (xs.detach() - xs.detach().mean(1, keepdim=True)) * torch.rsqrt(xs.detach().var(1, unbiased=False, keepdim=True) + 1e-5)
    assert_close(dgamma, (g_ref * xh).sum(0), 2e-3, 2e-3 * math.sqrt(n), "dgamma")
    assert_close(dbeta, g_ref.sum(0), 2e-3, 2e-3 * math.sqrt(n), "dbeta")
    # the rows the dy map leaves out (dy_row_limit): zero gradient, nothing read
    lim = int(out_rows.max()) // 2
    dgamma.zero_(); dbeta.zero_()
    dx3 = ops.ln_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dgamma, dbeta, M=n, dy_rows=out_rows.to(DEV), x_rows=in_rows.to(DEV),
                     post_drop=post, dy_row_limit=lim)
    gone = (out_rows >= lim)
    assert bool(gone.any()) and float(dx3[gone.to(DEV)].abs().max()) == 0.0
    assert_close(dx3[(~gone).to(DEV)], xs.grad[~gone], 1e-2, 1e-2, "rows below the limit")


# ------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, bias, drop_mask=None, drop_scale=1.0):
    w = (q @ k.transpose(-1, -2)) * 0.125 + bias[None, None, :]
    w = torch.softmax(w, -1)
    if drop_mask is not None:
        w = w * drop_mask * drop_scale
    return w @ v


@pytest.mark.parametrize("lens,heads,p", [([50], 2, 0.0), ([64, 128], 2, 0.0), ([50, 114, 550], 2, 0.0), ([37, 200], 3, 0.1), ([1425], 1, 0.0)])
def test_attention_fwd_bwd(ops, lens, heads, p):
    H = heads * 64
    M = sum(lens)
    qkv = bf(rnd(M, 3 * H, seed=40))
    dctx = bf(rnd(M, H, seed=41))
    bias = torch.zeros(M)
    gsel = torch.Generator().manual_seed(5)
    bias[torch.rand(M, generator=gsel) < 0.2] = -10000.0
    layout = ops.SeqLayout(lens, heads, DEV)
    drop = ops.make_drop(p, 777, 3)
    ctx, lse = ops.attn_fwd(qkv.to(DEV), bias.to(DEV), layout, H, drop=drop)
    dqkv = ops.attn_bwd(qkv.to(DEV), ctx, dctx.to(DEV), lse, bias.to(DEV), layout, H, drop=drop)
    s = 0
    for i, n in enumerate(lens):
        x = qkv[s:s + n].float().requires_grad_(True)
        q, k, v = (x[:, j * H:(j + 1) * H].view(n, heads, 64).transpose(0, 1)[None] for j in range(3))
        mask = None
        if p > 0:
            mask = torch.stack([ops.attn_dropout_mask(n, layout.elem_base_host[i], h, drop, DEV).float().cpu() for h in range(heads)])[None]
            assert 0.85 < float(mask.mean()) < 0.95
        out = ref_attention(q, k, v, bias[s:s + n], mask, drop[2])          # [1, heads, n, 64]
        ref = out[0].transpose(0, 1).reshape(n, H)
        ref.backward(dctx[s:s + n].float())
        assert_close(ctx[s:s + n], ref, 2e-2, 2e-2, f"ctx seq{i}")
        sc = (q @ k.transpose(-1, -2)) * 0.125 + bias[s:s + n][None, None, :]
        assert_close(lse[s:s + n], torch.logsumexp(sc, -1)[0].t(), 1e-3, 2e-2, f"lse seq{i}")
        assert_close(dqkv[s:s + n], x.grad, 3e-2, 3e-2, f"dqkv seq{i}")
        s += n


@pytest.mark.parametrize("lens", [[64], [128, 256], [50, 114, 114, 50, 114], [128, 50]])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_is_bitwise_reproducible_and_finite(ops, lens, p):
    """Regression: an inline-asm VALU read of MFMA results without wait states gave run-to-run
    different outputs with sporadic NaN rows.  20 reruns must be bit-identical and finite."""
    heads, H = 2, 128
    M = sum(lens)
    qkv = bf(rnd(M, 3 * H, seed=43)).to(DEV)
    dctx = bf(rnd(M, H, seed=44)).to(DEV)
    bias = torch.zeros(M, device=DEV)
    bias[torch.rand(M, generator=torch.Generator().manual_seed(6)).to(DEV) < 0.2] = -10000.0
    layout = ops.SeqLayout(lens, heads, DEV)
    drop = ops.make_drop(p, 5, 3)
    first = None
    for _ in range(20):
        ctx, lse = ops.attn_fwd(qkv, bias, layout, H, drop=drop)
        dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, bias, layout, H, drop=drop)
        torch.cuda.synchronize()
        cur = (ctx.clone(), lse.clone(), dqkv.clone())
        assert all(bool(torch.isfinite(t.float()).all()) for t in cur)
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, cur))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_skips_trailing_masked_keys_exactly(ops, p):
    """Keys masked with the reference's -10000 at the tail of a sequence (padded pair rows) have probability exactly 0 in
    fp32, so the kernels may skip their tiles: context, LSE and all three gradients must be BIT-identical with and without
    ``kv_len`` -- also with masked keys in the middle, with a fully masked sequence (kv_len = S: nothing skipped, softmax over
    equally biased keys is uniform) and with a tail that ends inside a tile."""
    lens, heads, H = [550, 300, 50, 200, 130], 2, 128
    tails = [230, 100, 0, 200, 2]                      # masked keys at the end of each sequence (sequence 3: every key)
    M = sum(lens)
    qkv = bf(rnd(M, 3 * H, seed=51)).to(DEV)
    dctx = bf(rnd(M, H, seed=52)).to(DEV)
    bias = torch.zeros(M)
    bias[torch.rand(M, generator=torch.Generator().manual_seed(53)) < 0.1] = -10000.0
    s = 0
    for n, t in zip(lens, tails):
        if t:
            bias[s + n - t:s + n] = -10000.0
        bias[s] = 0.0 if t < n else -10000.0           # keep key 0 of the other sequences valid
        s += n
    layout = ops.SeqLayout(lens, heads, DEV)
    kb = ops.pad_key_bias(bias.to(DEV), layout)
    kv = ops.attn_kv_len(kb, layout)
    want = []
    s = 0
    for n, t in zip(lens, tails):
        valid = (bias[s:s + n] > -10000.0).nonzero()
        want.append(n if valid.numel() == 0 else int(valid.max()) + 1)
        s += n
    assert kv.cpu().tolist() == want and want[0] <= 320 and want[3] == 200
    drop = ops.make_drop(p, 99, 3)
    ctx0, lse0 = ops.attn_fwd(qkv, kb, layout, H, drop=drop)
    d0 = ops.attn_bwd(qkv, ctx0, dctx, lse0, kb, layout, H, drop=drop)
    ctx1, lse1 = ops.attn_fwd(qkv, kb, layout, H, drop=drop, kv_len=kv)
    d1 = ops.attn_bwd(qkv, ctx1, dctx, lse1, kb, layout, H, drop=drop, kv_len=kv)
    assert torch.equal(ctx0, ctx1) and torch.equal(lse0, lse1) and torch.equal(d0, d1)
    assert bool(torch.isfinite(d1.float()).all())
    # the gradients of fully masked trailing keys are exact zeros
    assert float(d1[320:550, H:].abs().max()) == 0.0


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_backward_query_limit_is_exact(ops, p):
    """Round 4: ``attn_bwd(q_limit=...)`` -- per sequence, the query rows at index >= q_limit[s] have an exactly-zero output gradient
    (the top encoder layer: only the MLM-labelled rows and [CLS] have one), so dQ is zero there and they add exact zeros to dK / dV;
    both backward kernels stop their query range at it.  dctx zero past the last "labelled" row of every sequence; limits from
    mmbert_attn_q_limit on the row list (a sequence with no such row gets 0: all-zero gradients); results BIT-identical to the run
    without the limit, with and without dropout, with masked-out trailing keys (kv_len) on top."""
    lens, heads, H = [550, 300, 50, 200, 130, 64], 2, 128
    M = sum(lens)
    starts = [sum(lens[:i]) for i in range(len(lens))]
    qkv = bf(rnd(M, 3 * H, seed=61)).to(DEV)
    dctx = bf(rnd(M, H, seed=62))
    # rows with a gradient: a few among the first 50 rows of each sequence (text positions), none in sequence 3
    g = torch.Generator().manual_seed(63)
    rows = []
    for i, (s0, n) in enumerate(zip(starts, lens)):
        if i == 3:
            continue
        k = torch.randperm(min(50, n), generator=g)[:7] + s0
        rows.append(torch.cat((k, torch.tensor([s0]))))
    rows = torch.cat(rows)
    keep = torch.zeros(M, dtype=torch.bool)
    keep[rows] = True
    dctx = (dctx.float() * keep[:, None]).to(torch.bfloat16).to(DEV)
    bias = torch.zeros(M)
    bias[starts[0] + 400:starts[0] + 550] = -10000.0
    layout = ops.SeqLayout(lens, heads, DEV)
    kb = ops.pad_key_bias(bias.to(DEV), layout)
    kv = ops.attn_kv_len(kb, layout)
    qlim = ops.attn_q_limit(rows.int().to(DEV), layout)
    want = [0] * len(lens)
    for r in rows.tolist():
        i = max(j for j, s0 in enumerate(starts) if s0 <= r)
        want[i] = max(want[i], r - starts[i] + 1)
    assert qlim.cpu().tolist() == want and want[3] == 0 and max(want) <= 50
    drop = ops.make_drop(p, 31, 3)
    ctx, lse = ops.attn_fwd(qkv, kb, layout, H, drop=drop, kv_len=kv)
    d0 = ops.attn_bwd(qkv, ctx, dctx, lse, kb, layout, H, drop=drop, kv_len=kv)
    d1 = ops.attn_bwd(qkv, ctx, dctx, lse, kb, layout, H, drop=drop, kv_len=kv, q_limit=qlim)
    assert torch.equal(d0, d1) and bool(torch.isfinite(d1.float()).all())
    assert float(d1[starts[3]:starts[3] + lens[3]].abs().max()) == 0.0            # the sequence without any gradient row
    assert float(d1[:, H:].abs().max()) > 0.0


@pytest.mark.parametrize("mode", ["split", "dedupe", "drop"])
def test_split_layout_row_maps_kernel_equals_host_form(ops, mode):
    """ops.SplitLayout builds its [tokens]-sized row maps with mmbert_split_rows on the GPU and with numpy on the CPU: same maps,
    same tile lists, and the maps are consistent (every kept row round-trips, every forward tile row is covered once)."""
    import numpy as np
    lens = [50] * 5 + [550, 300, 129, 128, 64]
    valid = [27, 50, 31, 44, 50, 377, 300, 1, 128, 63]
    base = ops.SeqLayout(lens, 4, DEV)
    kw = dict(dedupe=(mode == "dedupe"), drop=(mode == "drop"))
    g = ops.SplitLayout(base, valid, DEV, **kw)
    c = ops.SplitLayout(base, valid, "cpu", **kw)
    torch.cuda.synchronize()
    for name in ("perm", "inv", "ftile_seq", "ftile_r0", "ftile_qshift", "ftile_qend", "tile_seq", "tile_r0", "qtile_qshift", "qtile_qend",
                 "seq_start", "kv_len"):
        assert torch.equal(getattr(g, name).cpu(), getattr(c, name)), name
    assert g.rows_a == sum(valid) and g.rows_packed == c.rows_packed
    perm, inv = g.perm.cpu().numpy(), g.inv.cpu().numpy()
    kept = inv < g.rows_packed
    assert (inv[perm] == np.arange(g.rows_packed)).all()                    # packed -> original -> packed
    if mode == "split":
        assert kept.all() and sorted(perm.tolist()) == list(range(sum(lens)))
    if mode == "drop":
        assert kept.sum() == g.rows_a and (inv[~kept] == g.rows_a).all()
    cov = np.zeros(g.rows_packed, int)
    for r0, sh, en in zip(g.ftile_r0.tolist(), g.ftile_qshift.tolist(), g.ftile_qend.tolist()):
        cov[sh + r0: sh + min(r0 + 128, en)] += 1
    assert (cov == 1).all()


@pytest.mark.parametrize("heads", [12, 4, 3])
def test_device_split_layout_equals_host_form(ops, heads):
    """ops.DeviceSplitLayout (mmbert_split_layout + mmbert_split_rows: the packing built from DEVICE-side counts, no host round trip)
    against ops.SplitLayout's numpy form on the same counts: identical starts, row maps and -- entry for entry, in the same
    longest-work-first / XCD-grouped order -- tile lists; the worst-case-sized lists are padded with sequence -1; the lazily read
    host numbers (rows_a, ntiles, valid_host) agree.  Incl. a sequence with nothing valid past row 1, full ones, and ties."""
    import numpy as np
    g = torch.Generator().manual_seed(heads)
    lens = [50] * 6 + [550] * 9 + [300, 129, 128, 64, 1]
    valid = [int(torch.randint(1, n + 1, (1,), generator=g)) for n in lens]
    valid[0], valid[6], valid[7], valid[8], valid[-1] = 50, 550, 1, 377, 1
    valid[9] = valid[10] = 377                                              # ties in the longest-first order
    base = ops.SeqLayout(lens, heads, DEV)
    vd = torch.tensor(valid, dtype=torch.int32, device=DEV)
    host = torch.tensor(valid, dtype=torch.int32)
    ev = torch.cuda.Event(); ev.record()
    d = ops.DeviceSplitLayout(base, vd, DEV, words=(host, ev))
    h = ops.SplitLayout(base, valid, DEV)
    torch.cuda.synchronize()
    nf, nq, ra = (int(x) for x in d.counts[:3].cpu())
    assert (nf, nq, ra) == (h.nftiles, h.ntiles, h.rows_a) and d.rows_a == h.rows_a and d.ntiles == h.ntiles and d.valid_host == h.valid_host
    assert d.nftiles >= nf and d.rows_packed == h.rows_packed == sum(lens)
    for name in ("ftile_seq", "ftile_r0", "ftile_qshift", "ftile_qend"):
        assert torch.equal(getattr(d, name)[:nf].cpu(), getattr(h, name).cpu()), name
    for name in ("tile_seq", "tile_r0", "qtile_qshift", "qtile_qend"):
        assert torch.equal(getattr(d, name)[:nq].cpu(), getattr(h, name).cpu()), name
    assert bool((d.ftile_seq[nf:] == -1).all()) and bool((d.tile_seq[nq:] == -1).all())
    for name in ("seq_start", "kv_len", "perm", "inv", "perm32", "inv32"):
        assert torch.equal(getattr(d, name).cpu(), getattr(h, name).cpu()), name
    # attention over the padded list: same context as over the exact list
    H = heads * 64
    M = sum(lens)
    qkv = bf(torch.randn(M, 3 * H, generator=g)).to(DEV)
    bias = ops.pad_key_bias(torch.zeros(M, device=DEV), base)
    c1, l1 = ops.attn_fwd(qkv, bias, d, H)
    c2, l2 = ops.attn_fwd(qkv, bias, h, H)
    assert torch.equal(c1, c2) and torch.equal(l1, l2)


def test_attention_rescale_branch(ops):
    """Force the running max to jump at a later key tile (guide rule 26): spike one key."""
    n, heads, H = 200, 1, 64
    qkv = bf(rnd(n, 3 * H, seed=42) * 0.3)
    qkv[150, H:2 * H] = qkv[3, 0:H] * 40.0          # key 150 aligned with query 3 -> huge score in tile 2
    layout = ops.SeqLayout([n], heads, DEV)
    bias = torch.zeros(n)
    ctx, _ = ops.attn_fwd(qkv.to(DEV), bias.to(DEV), layout, H)
    x = qkv.float()
    ref = ref_attention(x[None, None, :, :H], x[None, None, :, H:2 * H], x[None, None, :, 2 * H:], bias)[0, 0]
    assert_close(ctx, ref, 2e-2, 2e-2, "rescale")


# ------------------------------------------------------------------------- embeddings, CE, AdamW
def test_embed_gather_scatter(ops):
    V, H, T, n = 500, 128, 10, 60
    word, typ, pos = rnd(V, H, seed=50), rnd(2, H, seed=51), rnd(64, H, seed=52)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (n,), generator=g)
    ids[::7] = 0
    tts = torch.randint(0, 2, (n,), generator=g)
    out = ops.embed_gather(ids.to(DEV), tts.to(DEV), word.to(DEV), typ.to(DEV), pos.to(DEV), T)
    ref = word[ids] + typ[tts] + pos[torch.arange(n) % T]
    assert_close(out, ref, 1e-2, 1e-2, "gather")
    d = bf(rnd(n, H, seed=53))
    gw, gt, gp = torch.zeros(V, H, device=DEV), torch.zeros(2, H, device=DEV), torch.zeros(64, H, device=DEV)
    ops.embed_scatter(ids.to(DEV), tts.to(DEV), d.to(DEV), T, gw, gt, gp)
    rw, rt, rp = torch.zeros(V, H), torch.zeros(2, H), torch.zeros(64, H)
    keep = ids != 0
    rw.index_add_(0, ids[keep], d.float()[keep])
    rt.index_add_(0, tts, d.float())
    rp.index_add_(0, torch.arange(n) % T, d.float())
    assert_close(gw, rw, 1e-4, 1e-4, "word grad")
    assert_close(gt, rt, 1e-4, 1e-3, "type grad")
    assert_close(gp, rp, 1e-4, 1e-4, "pos grad")


@pytest.mark.parametrize("D", [35, 74, 371])
def test_pair_proj(ops, D):
    B, P, T, H = 3, 21, 5, 128
    feat, W, b = rnd(B, P, D, seed=54), rnd(H, D, seed=55, scale=0.2), rnd(H, seed=56, scale=0.1)
    out = torch.zeros(B * (T + P), H, device=DEV, dtype=torch.bfloat16)
    ops.pair_proj_fwd(feat.to(DEV), W.to(DEV), b.to(DEV), out, T)
    Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = torch.relu(feat @ Wr.t() + br)
    o3 = out.view(B, T + P, H)
    assert float(o3[:, :T].abs().max()) == 0.0
    assert_close(o3[:, T:], y, 1e-2, 1e-2, "pair fwd")
    dJ = bf(rnd(B * (T + P), H, seed=57))
    (o3.float().cpu()[:, T:] > 0)
    y_b = o3[:, T:].float().cpu()
    gy = dJ.float().view(B, T + P, H)[:, T:] * (y_b > 0)
    ((feat @ Wr.t() + br) * gy).sum().backward()        # linear part only: relu gate taken from the bf16 output
    dW, db = torch.zeros(H, D, device=DEV), torch.zeros(H, device=DEV)
    ops.pair_proj_bwd(feat.to(DEV), out, dJ.to(DEV), T, dW, db)
    assert_close(dW, Wr.grad, 1e-3, 1e-3, "pair dW")
    assert_close(db, br.grad, 1e-3, 1e-3, "pair db")
    dW2, db2 = torch.zeros(H, D, device=DEV), torch.zeros(H, device=DEV)      # per-row-range slabs added in range order: reproducible
    ops.pair_proj_bwd(feat.to(DEV), out, dJ.to(DEV), T, dW2, db2)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    ops.pair_proj_bwd(feat.to(DEV), out, dJ.to(DEV), T, dW2, db2)             # and it ACCUMULATES (+=)
    assert_close(dW2, 2 * Wr.grad, 1e-3, 2e-3, "pair dW accumulated")
    # round 6: float64 features as the reference's collate hands them over (REF:model_utils.py:94-99), rounded on load like the reference's
    # .float() (REF:MMBertEmbedding.py:62,64): the SAME bits as the fp32 path on the rounded copy, forward and backward
    f64 = feat.double() + 1e-9 * rnd(B, P, D, seed=58).double()               # (values that really need the rounding)
    out64 = torch.zeros_like(out)
    ops.pair_proj_fwd(f64.to(DEV), W.to(DEV), b.to(DEV), out64, T)
    out32 = torch.zeros_like(out)
    ops.pair_proj_fwd(f64.float().to(DEV), W.to(DEV), b.to(DEV), out32, T)
    assert torch.equal(out64, out32)
    dW3, db3, dW4, db4 = (torch.zeros(H, D, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, D, device=DEV), torch.zeros(H, device=DEV))
    ops.pair_proj_bwd(f64.to(DEV), out64, dJ.to(DEV), T, dW3, db3)
    ops.pair_proj_bwd(f64.float().to(DEV), out32, dJ.to(DEV), T, dW4, db4)
    assert torch.equal(dW3, dW4) and torch.equal(db3, db4)


def test_cross_entropy_segments(ops):
    M, V, ldv = 90, 1000, 1024
    logits = torch.zeros(M, ldv)
    logits[:, :V] = rnd(M, V, seed=58, scale=3.0)
    logits = bf(logits)
    g = torch.Generator().manual_seed(4)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[torch.rand(M, generator=g) < 0.6] = -100
    bounds = torch.tensor([0, 20, 55, M], dtype=torch.int32)
    g = torch.tensor([0.5, 2.0, -1.0])
    loss, inv, lse = ops.ce_fwd(logits.to(DEV), V, labels.to(DEV), bounds.to(DEV), 3)
    dl = torch.full((M, ldv), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.ce_bwd(logits.to(DEV), V, labels.to(DEV), bounds.to(DEV), 3, inv, g.to(DEV), lse, dl)
    for s in range(3):
        a, b = int(bounds[s]), int(bounds[s + 1])
        lg = logits[a:b, :V].float().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(lg, labels[a:b])
        (ref * g[s]).backward()
        assert_close(loss[s], ref, 2e-3, 2e-3, f"ce loss seg{s}")
        assert_close(dl[a:b, :V], lg.grad, 2e-2, 1e-4, f"dlogits seg{s}")
        valid = labels[a:b] != -100
        assert_close(lse[a:b][valid.to(DEV)], torch.logsumexp(lg.detach(), -1)[valid], 1e-4, 1e-3, "row lse")
    assert float(dl[:, V:].abs().max()) == 0.0
    # in-place form (dlogits aliases logits)
    lg_dev = logits.to(DEV).clone()
    ops.ce_bwd(lg_dev, V, labels.to(DEV), bounds.to(DEV), 3, inv, g.to(DEV), lse, lg_dev)
    assert torch.equal(lg_dev, dl)


@pytest.mark.parametrize("mode", [0, 1])
def test_adamw_flat(ops, mode):
    import sys
    sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
    from oracle import mmbert_oracle as O
    n = 256 * 12
    p, g = rnd(n, seed=60), rnd(n, seed=61, scale=0.1)
    flags = torch.tensor([0, 1, 2] * 4, dtype=torch.uint8)
    pr, m, v = p.clone(), torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.clone().to(DEV), g.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        gd.copy_(g.to(DEV) * 4.0)
        ops.adamw(pd, gd, md, vd, pb, flags.to(DEV), lr=1e-2, wd=0.01, step=step, gscale=0.25, mode=mode, zero_grad=True)
        for blk in range(12):
            sl = slice(blk * 256, (blk + 1) * 256)
            if flags[blk] == 2:
                continue
            O.adamw_step(pr[sl], g[sl], m[sl], v[sl], step, 1e-2, 0.01 if flags[blk] == 1 else 0.0, mode="hf" if mode == 0 else "torch")
        assert float(gd.abs().max()) == 0.0
    assert_close(pd, pr, 1e-5, 1e-6, "adamw params")
    assert_close(pb, pr, 1e-2, 1e-3, "bf16 copy")
    assert torch.equal(pd[512:768].cpu(), p[512:768])                  # frozen block untouched


def test_adamw_hf_mode_matches_the_hand_computed_vector(ops):
    """G11 (round 5): the HIP AdamW kernel's mode 0 (transformers-2.8.0 AdamW, what REF:train.py:92 constructs) against the hand-computed
    known-answer vector of tests/golden/hf_adamw_hand.py -- through the C ABI (mmbert_adamw), one 256-element block per trajectory with
    the block's decay flag, three steps, fp32 state: parameters within 2e-6 relative, moments within 5e-7 (fp32 arithmetic against 40-digit
    decimals), the bf16 working copy equal to the rounded parameter, gradients zeroed.  (It caught the kernel forming 1 - beta2 and the
    bias corrections in fp32: 1 - 0.999f is 1.3e-5 off; mmbert_adamw takes doubles since.)"""
    import sys
    sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
    from tests.golden import hf_adamw_hand as G
    names = list(G.TRAJECTORIES)
    n = 256 * len(names)
    p = torch.empty(n)
    flags = torch.zeros(len(names), dtype=torch.uint8)
    for i, k in enumerate(names):
        p[i * 256:(i + 1) * 256] = G.TRAJECTORIES[k][0]
        flags[i] = 1 if G.TRAJECTORIES[k][1] > 0 else 0
    wd = max(t[1] for t in G.TRAJECTORIES.values())
    assert all(t[1] in (0.0, wd) for t in G.TRAJECTORIES.values())
    pd, md, vd = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    gd = torch.empty(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        for i, k in enumerate(names):
            gd[i * 256:(i + 1) * 256] = G.TRAJECTORIES[k][2][step - 1]
        ops.adamw(pd, gd, md, vd, pb, flags.to(DEV), lr=G.LR, beta1=G.BETA1, beta2=G.BETA2, eps=G.EPS, wd=wd, step=step, gscale=1.0, mode=0, zero_grad=True)
        torch.cuda.synchronize()
        assert float(gd.abs().max()) == 0.0
        for i, k in enumerate(names):
            want_p, want_m, want_v = (G.TRAJECTORIES[k][j][step - 1] for j in (3, 4, 5))
            sl = slice(i * 256, (i + 1) * 256)
            got = pd[sl].double().cpu()
            assert float((got - want_p).abs().max()) <= 2e-6 * abs(want_p), (k, step, float(got[0]), want_p)
            # (moments: fp32 roundings only, 2^-23 per operation -- the coefficients 1 - beta are formed in double, as the reference's are)
            assert float((md[sl].double().cpu() - want_m).abs().max()) <= 5e-7 * abs(want_m), (k, step, float(md[sl][0]), want_m)
            assert float((vd[sl].double().cpu() - want_v).abs().max()) <= 5e-7 * abs(want_v), (k, step, float(vd[sl][0]), want_v)
            assert torch.equal(pb[sl].cpu(), pd[sl].cpu().bfloat16()), (k, step)


def test_transpose_cast_and_casts(ops):
    src = rnd(5000, seed=62)
    descs, tile0 = [], 0
    mats = [(0, 0, 40, 50, 64), (2000, 4096, 30, 100, 32)]              # (src_off, dst_off, rows, cols, dst_ld)
    for so, do, r, c, ld in mats:
        descs.append([so, do, r | (c << 32), ld | (tile0 << 32)])
        tile0 += ((r + 63) // 64) * ((c + 63) // 64)
    import numpy as np
    raw = np.zeros((len(mats), 4), dtype=np.int64)
    for i, (so, do, r, c, ld) in enumerate(mats):
        raw[i, 0], raw[i, 1] = so, do
        raw[i, 2] = np.int64(r) | (np.int64(c) << 32)
        raw[i, 3] = np.int64(ld) | (np.int64(sum(((a[2] + 63) // 64) * ((a[3] + 63) // 64) for a in mats[:i])) << 32)
    dst = torch.zeros(10000, device=DEV, dtype=torch.bfloat16)
    ops.transpose_cast(src.to(DEV), dst, torch.from_numpy(raw).to(DEV), len(mats), tile0)
    for so, do, r, c, ld in mats:
        ref = src[so:so + r * c].view(r, c).t()
        got = dst[do:do + c * ld].view(c, ld).float().cpu()
        assert_close(got[:, :r], ref, 1e-2, 1e-2, "transpose")
    y = torch.empty(5000, device=DEV, dtype=torch.bfloat16)
    ops.cast_f32_bf16(src[:5000].to(DEV), y)
    assert torch.equal(y.cpu(), src.to(torch.bfloat16))
    dst2 = torch.zeros_like(dst)                                        # the bf16-source form: same bits
    ops.transpose_cast(y, dst2, torch.from_numpy(raw).to(DEV), len(mats), tile0)
    assert torch.equal(dst, dst2)


def test_compact_rows_and_scatter_rows_zero(ops):
    """mmbert_compact_rows(_inv) + mmbert_scatter_rows_zero: the row list map[cat(rows, extra)] in both integer widths, and its stamped
    inverse taking compact rows back to full height in one launch = zeros().index_copy_() per matrix; a second list on the same
    RowInverse must not see the first one's entries (nothing is cleared in between: the stamp ends their validity)."""
    g = torch.Generator().manual_seed(77)
    M, H = 700, 128
    inv_map = torch.randperm(M, generator=g).to(DEV)
    rinv = ops.RowInverse(M, DEV)
    for n, ne, width in ((40, 6, torch.int32), (3, 2, torch.int64), (0, 5, torch.int32)):
        pick = torch.randperm(M, generator=g)[:n + ne]
        rows, extra = pick[:n].to(width).to(DEV), pick[n:].long().to(DEV)
        ref = inv_map[torch.cat((rows.long(), extra))]
        r64, r32 = ops.compact_rows(rows, extra, inv_map)
        assert torch.equal(r64, ref) and torch.equal(r32, ref.int())
        q64, q32 = ops.compact_rows(rows, extra, inv_map, inverse=rinv)
        assert torch.equal(q64, ref) and torch.equal(q32, ref.int())
        a, b = bf(rnd(n + ne, H, seed=78)).to(DEV), rnd(n + ne, 36, seed=79).to(DEV)
        nrows = M - 50                                               # (the list may point past the rows asked for: those are left out)
        fa, fb = ops.scatter_rows_zero([a, b], rinv, nrows)
        keep = ref < nrows
        ra_, rb_ = torch.zeros(nrows, H, device=DEV, dtype=torch.bfloat16), torch.zeros(nrows, 36, device=DEV)
        ra_.index_copy_(0, ref[keep], a[keep]); rb_.index_copy_(0, ref[keep], b[keep])
        assert torch.equal(fa, ra_) and torch.equal(fb, rb_)


def test_gather_rows_batched(ops):
    """mmbert_gather_rows: one row list, several matrices (bf16 / fp32 rows of different widths, 1-D vectors, a strided view)."""
    g = torch.Generator().manual_seed(70)
    M = 500
    idx = torch.randint(0, M, (77,), generator=g).int().to(DEV)
    big = bf(rnd(M, 96, seed=71)).to(DEV)
    srcs = [bf(rnd(M, 768, seed=72)).to(DEV), rnd(M, seed=73).to(DEV), rnd(M, 36, seed=74).to(DEV), big[:, 32:64], bf(rnd(M, 3072, seed=75)).to(DEV)]
    outs = ops.gather_rows(srcs, idx)
    for t, o in zip(srcs, outs):
        assert torch.equal(o, t.index_select(0, idx.long()))
    assert ops.gather_rows(srcs[:2], idx[:0])[0].shape == (0, 768)


def test_step_prologue_matches_the_torch_formulation(ops):
    """mmbert_prologue (two launches) against the element-wise formulation it replaced: (1 - mask) * -10000 per key from the
    reference's mask dtypes (float64 text mask, float64 [B,P,D] visual mask and int64 speech mask read at feature 0 through
    their own strides, REF:MMBertForPretraining.py:76), padded per sequence; kv_len by mmbert_attn_kv_len's rule; valid =
    max(kv_len, last labelled position + 1); the ascending labelled-row list of mmbert_active_rows; the host words."""
    B, T, Pv, Pa, V = 5, 12, 40, 33, 1000
    g = torch.Generator().manual_seed(4)
    lens_t = torch.randint(3, T + 1, (B,), generator=g)
    tmask = (torch.arange(T)[None, :] < lens_t[:, None]).double()
    ones_f, ones_i = torch.ones(B, T, dtype=torch.float64), torch.ones(B, T, dtype=torch.int64)
    vis = torch.randn(B, Pv, 35, generator=g, dtype=torch.float64)
    sp = torch.randn(B, Pa, 74, generator=g, dtype=torch.float64)
    for b in range(B):
        vis[b, int(torch.randint(Pv // 2, Pv + 1, (1,), generator=g)):] = 0
        sp[b, int(torch.randint(1, Pa + 1, (1,), generator=g)):] = 0
    vis[1, 2, 0] = 0.0                                      # a live frame whose feature 0 is exactly 0 (quirk B-2): masked as a key
    vmask, smask = (vis != 0).double(), (sp != 0).long()
    lens = [T, T + Pv, T + Pa]
    tokens = B * sum(lens)
    labels = torch.full((tokens,), -100, dtype=torch.int64)
    sel = torch.rand(tokens, generator=g) < 0.05
    labels[sel] = torch.randint(0, V, (int(sel.sum()),), generator=g)
    labels[B * T + 2 * (T + Pv) + T + Pv - 1] = 7           # a label on the last (padded) pair row of sample 2, visual pass
    labels[0] = 3                                           # ... and on a [CLS] row
    labels[5] = V + 9                                       # ... and one outside the vocabulary
    dev_masks = [t.to(DEV) for t in (tmask, ones_f, vmask, ones_i, smask)]
    segs = [(dev_masks[0], 0, 0), (dev_masks[1], 1, 0), (dev_masks[2][:, :, 0], 1, T), (dev_masks[3], 2, 0), (dev_masks[4][:, :, 0], 2, T)]
    pro = ops.prologue(segs, lens, B, labels.to(DEV), V, DEV)
    torch.cuda.synchronize()
    # reference formulation
    layout = ops.SeqLayout([n for n in lens for _ in range(B)], 2, DEV)
    kb_rows = torch.cat([((1.0 - tmask.float()) * -10000.0).reshape(-1),
                         torch.cat(((1.0 - ones_f.float()) * -10000.0, (1.0 - vmask[:, :, 0].float()) * -10000.0), dim=1).reshape(-1),
                         torch.cat(((1.0 - ones_i.float()) * -10000.0, (1.0 - smask[:, :, 0].float()) * -10000.0), dim=1).reshape(-1)])
    kb_ref = ops.pad_key_bias(kb_rows.to(DEV), layout)
    assert torch.equal(pro.key_bias, kb_ref)
    kv_ref = ops.attn_kv_len(kb_ref, layout)
    assert torch.equal(pro.kv_len, kv_ref)
    row_seq = torch.from_numpy(layout._row_seq)
    row_pos = torch.from_numpy(layout._row_pos)
    lab_end = torch.zeros(3 * B, dtype=torch.int64).scatter_reduce_(0, row_seq, torch.where(labels != -100, row_pos + 1, 0), "amax")
    valid_ref = torch.maximum(kv_ref.cpu().long(), lab_end)
    assert torch.equal(pro.valid.cpu().long(), valid_ref)
    assert int(valid_ref[B + 2]) == T + Pv                  # the labelled padded row keeps its whole sequence
    ok = (labels >= 0) & (labels < V)
    idx_ref = ok.nonzero().reshape(-1)
    words = pro.words.cpu()
    n = int(words[3 * B])
    assert n == idx_ref.numel() and torch.equal(pro.idx[:n].cpu().long(), idx_ref)
    assert torch.equal(words[:3 * B].long(), valid_ref)
    first_rows = torch.tensor([int(s_) for s_ in layout.seq_start.cpu()])
    assert int(words[3 * B + 1]) == int(ok[first_rows].sum()) >= 1                           # labelled [CLS] (position-0) rows
    assert int(words[3 * B + 2]) == int(((labels != -100) & ~ok).sum()) == 1                 # the one out-of-vocabulary label
    # no labels: valid = kv_len, nothing labelled
    pro2 = ops.prologue(segs, lens, B, None, V, DEV)
    assert torch.equal(pro2.valid, kv_ref) and int(pro2.words[3 * B]) == 0 and torch.equal(pro2.key_bias, kb_ref)


def test_step_prologue_row_set_mode_and_its_packing(ops):
    """mmbert_prologue's row-set mode + mmbert_split_rows(rank): masked-out rows in the MIDDLE of a sequence (the fused text | visual
    | speech sequence: the visual block's padding precedes the speech block).  Active = unmasked key, or labelled, or position 0;
    valid[s] counts them; rank orders every sequence active-first with both groups in their original order; key_bias comes back in
    that order; and the SplitLayout built from it is a permutation whose leading rows_a rows are exactly the active rows."""
    B, T, Pv, Pa, V = 4, 10, 37, 29, 500
    g = torch.Generator().manual_seed(11)
    S = T + Pv + Pa
    tmask = torch.ones(B, T, dtype=torch.float64)
    tmask[1, 7:] = 0                                         # [PAD] text rows in the middle of the fused sequence too
    tmask[2, 0] = 0                                          # a masked [CLS] key: position 0 stays active (the heads read it)
    vis = torch.randn(B, Pv, 35, generator=g, dtype=torch.float64)
    sp = torch.randn(B, Pa, 74, generator=g, dtype=torch.float64)
    for b in range(B):
        vis[b, int(torch.randint(Pv // 3, Pv, (1,), generator=g)):] = 0
        sp[b, int(torch.randint(1, Pa + 1, (1,), generator=g)):] = 0
    vmask, smask = (vis != 0).double(), (sp != 0).long()
    labels = torch.full((B * S,), -100, dtype=torch.int64)
    sel = torch.rand(B * S, generator=g) < 0.06                # labels everywhere, masked rows included
    labels[sel] = torch.randint(0, V, (int(sel.sum()),), generator=g)
    dm = [t.to(DEV) for t in (tmask, vmask, smask)]
    segs = [(dm[0], 0, 0), (dm[1][:, :, 0], 0, T), (dm[2][:, :, 0], 0, T + Pv)]
    pro = ops.prologue(segs, [S], B, labels.to(DEV), V, DEV, rowset=True)
    plain = ops.prologue(segs, [S], B, labels.to(DEV), V, DEV)
    torch.cuda.synchronize()
    key_mask = torch.cat((tmask.float(), vmask[:, :, 0].float(), smask[:, :, 0].float()), dim=1)         # [B, S]
    bias = (1.0 - key_mask) * -10000.0
    active = (bias > -10000.0) | (labels.view(B, S) != -100)
    active[:, 0] = True
    assert torch.equal(pro.valid.cpu().long(), active.sum(1))
    assert torch.equal(pro.kv_len, plain.kv_len) and torch.equal(pro.idx[:int(pro.words[B])], plain.idx[:int(plain.words[B])])
    rank_ref = torch.empty(B, S, dtype=torch.int64)
    slots = (S + 127) // 128 * 128
    kb_ref = torch.full((B, slots), -1.0e30)
    for b in range(B):
        order = torch.cat((active[b].nonzero().reshape(-1), (~active[b]).nonzero().reshape(-1)))       # new position -> old position
        rank_ref[b, order] = torch.arange(S)
        kb_ref[b, :S] = bias[b, order]
    assert torch.equal(pro.rank.cpu().long(), rank_ref.reshape(-1))
    assert torch.equal(pro.key_bias.cpu(), kb_ref.reshape(-1))
    assert bool((~active).any()) and bool((active[:, T:T + Pv].sum(1) < Pv).all())                  # padding in the middle block indeed
    base = ops.SeqLayout([S] * B, 2, DEV)
    lay = ops.SplitLayout(base, pro.valid.cpu().numpy(), DEV, rank=pro.rank)
    perm, inv = lay.perm.cpu(), lay.inv.cpu()
    assert lay.rows_a == int(active.sum()) and perm.numel() == B * S
    assert torch.equal(perm[inv], torch.arange(B * S)) and torch.equal(torch.sort(perm).values, torch.arange(B * S))
    assert bool(active.reshape(-1)[perm[:lay.rows_a]].all()) and not bool(active.reshape(-1)[perm[lay.rows_a:]].any())
    # a sequence's active rows are contiguous in region A, in their original order
    sa = lay.seq_start.cpu()
    for b in range(B):
        rows = perm[int(sa[b]):int(sa[b]) + int(active[b].sum())]
        assert torch.equal(rows, b * S + active[b].nonzero().reshape(-1))


def test_skinny_products_of_the_heads(ops):
    """mmbert_skinny_mm / mmbert_skinny_wgrad (the heads' dense layers) against fp32 torch: y = x W^T + b with two sources and a row
    range, dx = dy W as the inner-major form accumulated onto an existing tensor, odd sizes (N = 2, inner = 1 / 2), and the weight
    / bias gradients of several layers in one launch."""
    g = torch.Generator().manual_seed(7)
    rnd_ = lambda *s: torch.randn(*s, generator=g)
    B, H = 16, 768
    X, W1, W2, b = rnd_(3 * B, H), rnd_(H, 2 * H) * 0.05, rnd_(2, H) * 0.05, rnd_(H)
    Xd, W1d, W2d, bd = (t.to(DEV) for t in (X, W1, W2, b))
    Y = torch.zeros(3 * B, H, device=DEV)
    rel = torch.zeros(2 * B, 2, device=DEV)
    b2 = rnd_(2).to(DEV)
    ops.skinny_mm([(Y, bd, 0, False, [(Xd, W1d[:, :H], 0, 0), (Xd, W1d[:, H:], 0, 0)]),
                   (rel, b2, 0, False, [(Xd[B:], W2d, 0, 0)])])
    ref = X @ (W1[:, :H] + W1[:, H:]).t() + b
    assert float((Y.cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    assert float((rel.cpu() - (X[B:] @ W2.t() + b2.cpu())).abs().max()) < 1e-4
    # input gradients: dX = dY W (+ a second source on the rows B..3B only), accumulated onto an existing tensor
    dY, drel = rnd_(3 * B, H), rnd_(2 * B, 2)
    base = rnd_(3 * B, H)
    dX = base.clone().to(DEV)
    Wp = rnd_(H, H) * 0.05
    ops.skinny_mm([(dX, None, 0, True, [(dY.to(DEV), Wp.to(DEV), 1, 0), (drel.to(DEV), W2d, 1, B)])])
    ref = base + dY @ Wp
    ref[B:] += drel @ W2
    assert float((dX.cpu() - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    dlo, Wc2 = rnd_(B, 1), rnd_(1, H)
    dT = torch.zeros(B, H, device=DEV)
    ops.skinny_mm([(dT, None, 0, False, [(dlo.to(DEV), Wc2.to(DEV), 1, 0)])])
    assert float((dT.cpu() - dlo @ Wc2).abs().max()) < 1e-5
    # weight / bias gradients, several layers in one launch (one of them into a column slice of a wider weight)
    T_ = rnd_(B, H)
    gW, gb = torch.zeros(H, H, device=DEV), torch.zeros(H, device=DEV)
    gW2, gb2 = torch.ones(2, H, device=DEV), torch.zeros(2, device=DEV)
    gWide = torch.zeros(H, 2 * H, device=DEV)
    dYb = dY[:B]
    ops.skinny_wgrad([(dYb.to(DEV), T_.to(DEV), gW, gb), (drel.to(DEV), Xd[B:], gW2, gb2), (dYb.to(DEV), T_.to(DEV), gWide[:, H:], None)])
    assert float((gW.cpu() - dYb.t() @ T_).abs().max()) < 2e-4 * float((dYb.t() @ T_).abs().max())
    assert float((gb.cpu() - dYb.sum(0)).abs().max()) < 1e-4
    assert float((gW2.cpu() - (1.0 + drel.t() @ X[B:])).abs().max()) < 2e-4 * float((drel.t() @ X[B:]).abs().max())
    assert float((gb2.cpu() - drel.sum(0)).abs().max()) < 1e-4
    assert float(gWide[:, :H].abs().max()) == 0.0 and torch.allclose(gWide[:, H:], gW, rtol=1e-5, atol=1e-6)


def test_attention_with_more_than_65535_tiles(ops):
    """Tile lists longer than gridDim.y's 16-bit range (a large no-grad evaluation batch: 3 passes x B x ~5 tiles) -- the launchers
    fall back to grid (tiles, heads) there; spot-checked sequences against fp32, forward and backward."""
    n, S, heads = 66000, 8, 1
    H = 64 * heads
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(n * S, 3 * H, generator=g))
    dctx = bf(torch.randn(n * S, H, generator=g))
    layout = ops.SeqLayout([S] * n, heads, DEV)
    assert layout.nftiles > 65535
    bias = torch.zeros(n * S)
    ctx, lse = ops.attn_fwd(qkv.to(DEV), bias.to(DEV), layout, H)
    dqkv = ops.attn_bwd(qkv.to(DEV), ctx, dctx.to(DEV), lse, bias.to(DEV), layout, H)
    torch.cuda.synchronize()
    for i in (0, 1, 32767, 65535, 65536, n - 1):
        x = qkv[i * S:(i + 1) * S].float().requires_grad_(True)
        q, k, v = (x[:, j * H:(j + 1) * H].view(S, heads, 64).transpose(0, 1)[None] for j in range(3))
        ref = ref_attention(q, k, v, bias[:S])[0].transpose(0, 1).reshape(S, H)
        ref.backward(dctx[i * S:(i + 1) * S].float())
        assert_close(ctx[i * S:(i + 1) * S], ref, 2e-2, 2e-2, f"ctx seq{i}")
        assert_close(dqkv[i * S:(i + 1) * S], x.grad, 3e-2, 3e-2, f"dqkv seq{i}")


def test_deterministic_forms_equal_the_atomic_forms(ops):
    """Round 5 (mmbert_set_deterministic): the ordered forms compute the same sums as the atomic ones -- the heads' skinny products through
    their slab + fold (mmbert_skinny_mm_ordered), the CE loss sums through the ordered one-workgroup sum, the embedding scatter and the
    data-parallel row block through sorted keys + mmbert_segment_sum_rows -- and give the same bits twice in a row."""
    was = ops.deterministic()
    try:
        H, B = 768, 48
        X, W, W2, bias = rnd(B, H, seed=501).to(DEV), rnd(2304, H, seed=502, scale=0.05).to(DEV), rnd(H, 512, seed=503, scale=0.05).to(DEV), rnd(2304, seed=504).to(DEV)
        Y0 = rnd(B, 2304, seed=505).to(DEV)

        def skinny():
            Y, Z = Y0.clone(), torch.zeros(B, 512, device=DEV)
            ops.skinny_mm([(Y, bias, 0, True, [(X, W, False, 0)]), (Z, None, 0, False, [(X[:16], W2, True, 8), (X[16:40], W2, True, 0)])])
            return Y, Z
        V, M = 1000, 1500
        logits = bf(rnd(M, 1024, seed=506)).to(DEV)
        labels = torch.full((M,), -100, dtype=torch.long)
        labels[::7] = torch.arange(0, M, 7) % V
        bounds = torch.tensor([0, 400, 900, M], dtype=torch.int32, device=DEV)

        def ce():
            return ops.ce_fwd(logits, V, labels.to(DEV), bounds, 3)[0].clone()
        n, T = 600, 50
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(0, 300, (n,), generator=g).to(DEV)
        tts = (torch.arange(n) % 3 == 0).long().to(DEV)
        d = bf(rnd(n, H, seed=507)).to(DEV)

        def scatter():
            gw, gt, gp = torch.zeros(300, H, device=DEV), torch.zeros(2, H, device=DEV), torch.zeros(512, H, device=DEV)
            ops.embed_scatter(ids, tts, d, T, gw, gt, gp)
            return gw, gt, gp
        union = torch.unique(ids[::2][ids[::2] > 0])
        rows32 = rnd(n, H, seed=508).to(DEV)

        def block():
            blk = torch.zeros(union.numel(), H, device=DEV)
            return ops.rows_to_block(ids, rows32, union, 300, blk)
        ops.set_deterministic(False)
        ref = [*skinny(), ce(), *scatter(), block()]
        ops.set_deterministic(True)
        got, again = [*skinny(), ce(), *scatter(), block()], [*skinny(), ce(), *scatter(), block()]
        names = ["skinny Y", "skinny Z", "ce loss sums", "word rows", "type rows", "position rows", "row block"]  # (ids 0 = padding: skipped by both forms)
        for nm, r, a, b in zip(names, ref, got, again):
            assert torch.equal(a, b), nm
            assert_close(a, r, 1e-5, 1e-4, nm)
        assert float(got[0].abs().sum()) > 0 and float(got[3].abs().sum()) > 0 and float(got[6].abs().sum()) > 0
        # more rows than one launch of the ordered kernel holds in LDS (8192): consecutive launches, still one fixed order
        big_ids = torch.randint(0, 300, (10000,), generator=g).to(DEV)
        big_rows = rnd(10000, H, seed=509).to(DEV)
        ops.set_deterministic(False)
        ref_blk = ops.rows_to_block(big_ids, big_rows, union, 300, torch.zeros(union.numel(), H, device=DEV))
        ops.set_deterministic(True)
        b1 = ops.rows_to_block(big_ids, big_rows, union, 300, torch.zeros(union.numel(), H, device=DEV))
        b2 = ops.rows_to_block(big_ids, big_rows, union, 300, torch.zeros(union.numel(), H, device=DEV))
        assert torch.equal(b1, b2)
        assert_close(b1, ref_blk, 1e-5, 2e-4, "row block, 10 000 rows")
    finally:
        ops.set_deterministic(was)
