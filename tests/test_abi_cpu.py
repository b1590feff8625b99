"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mmbert_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmbert_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmbert_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from msa_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        if not os.path.exists(build.HIPCC):
            pytest.skip("no prebuilt library and no hipcc on this machine")
        build.build(verbose=False)
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mmbert_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_host_side_helpers_run_without_gpu():
    from msa_amd import _lib
    lib = _lib.load()
    assert lib.mmbert_dropout_thr16(0.0) == 0
    assert lib.mmbert_dropout_thr16(0.1) == 6554
    assert lib.mmbert_dropout_thr16(0.5) == 32768
    a, b = lib.mmbert_rng_stream(1, 2), lib.mmbert_rng_stream(1, 3)
    assert a != b and a == lib.mmbert_rng_stream(1, 2)
    splits = ctypes.c_int(0)
    need = lib.mmbert_gemm_tn_workspace(18400, 768, 768, ctypes.byref(splits))
    # split 0 writes the gradient itself, the others slabs (+ one row of bias sums each: deterministic mode stores them there)
    assert splits.value > 1 and need == (splits.value - 1) * (768 * 768 + 768) * 4
    assert lib.mmbert_gemm_tn_workspace(100, 4096, 4096, ctypes.byref(splits)) == 0 and splits.value == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from msa_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_torch_operator_namespace_registers_without_a_gpu():
    """msa_amd.torch_ops defines torch.ops.mmbert.* (schemas + CUDA / AutogradCUDA kernels); on a CPU tensor an operator
    raises instead of falling back."""
    import pytest
    import torch
    import msa_amd.torch_ops  # noqa: F401
    for n in ("linear", "linear_pre", "linear_bwd", "layer_norm", "layer_norm_fwd", "layer_norm_bwd", "attention", "attention_fwd", "attention_bwd",
              "embed_ln", "embed_ln_fwd", "embed_ln_bwd", "joint_embed", "joint_embed_fwd", "joint_embed_bwd", "mlm_head_ce", "mlm_head_ce_fwd",
              "mlm_head_ce_bwd", "adamw_multi_tensor", "mlm_mask_rng"):
        assert hasattr(torch.ops.mmbert, n), n
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mmbert.layer_norm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5)
