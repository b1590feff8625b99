"""ISA regression guard (CPU: hipcc cross-compiles): the small kernels whose loads hipcc had serialized keep their loads in
flight together.  The measure is tools/scan_serialized_loads.py's: drains (s_waitcnt vmcnt(0)) that follow a SINGLE global load."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scan():
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.environ.get("HIPCC"):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("scan_serialized_loads", os.path.join(ROOT, "tools", "scan_serialized_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.scan_source


def _find(table, prefix):
    hits = [v for k, v in table.items() if k.startswith(prefix) or k.startswith("void " + prefix)]
    assert hits, prefix
    return hits


def test_rowwise_kernels_keep_their_loads_in_flight(scan):
    t = scan("rowwise.hip")
    # (only "no load waits alone" is pinned: the LOAD COUNT of a kernel is hipcc's business and moves with ROCm versions -- ADVICE r4)
    for prefix in ("pair_proj_fwd_kernel<6, float>", "pair_proj_fwd_kernel<7, float>", "pair_proj_fwd_kernel<6, double>", "pair_proj_fwd_kernel<7, double>",
                   "pair_wgrad_kernel<float>", "pair_wgrad_kernel<double>", "ce_count_kernel", "split_layout_kernel"):
        for loads, drains, serialized in _find(t, prefix):
            assert loads > 0 and serialized == 0, (prefix, loads, drains, serialized)
    for prefix in ("transpose_cast_kernel<float>", "transpose_cast_kernel<__bf16>", "_Z21transpose_cast_kernelIDF16b"):
        for loads, drains, serialized in [v for k, v in t.items() if prefix in k]:
            assert serialized <= 1, (prefix, loads, drains, serialized)    # (the edge-tile path reads one element at a time)


def test_heads_kernels_keep_their_loads_in_flight(scan):
    t = scan("heads.hip")
    (loads, drains, serialized), = _find(t, "skinny_wgrad_kernel")
    assert loads > 0 and serialized <= 6, (loads, drains, serialized)       # the 20 loads of a block together; the scalar edge paths remain
    for loads, drains, serialized in _find(t, "skinny_mm_kernel"):           # (the atomic and the ordered instantiation)
        assert serialized <= 1, (loads, drains, serialized)
