"""ctypes binding of libmmbert_hip.so (the C ABI declared in include/mmbert_hip.h).

There is NO fallback: if the shared library is absent or a symbol is missing, importing the product
path raises.  Build it with ``python -m msa_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (MMBERT_LIB_PATH: a diagnostic override -- another BUILD of the same library, for whole-process A/Bs of kernel changes: the library
# itself has no run-time switches; tools/ab_lib.sh; the stamp tools set LIB_PATH directly)
LIB_PATH = os.environ.get("MMBERT_LIB_PATH") or os.path.join(HERE, "libmmbert_hip.so")

P, I, F, D, U32, SZ, U64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_uint32, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); must list every symbol of include/mmbert_hip.h
SIGNATURES = {
    "mmbert_gemm_nt": (I, [P, P, I, P, I, P, I, I, I, I, I, P, P, I, P, I, P, I, F, P, U32, U32, F, P]),
    "mmbert_gemm_nt_force": (None, [I]),
    "mmbert_gemm_nt_describe": (I, [I, I, I, I, I, P]),
    "mmbert_gemm_nt_splitk": (I, [P, P, I, P, I, P, I, I, I, I, P, P, I]),
    "mmbert_gemm_nt_splitk_workspace": (SZ, [I, I, I]),
    "mmbert_gemm_tn_force_splits": (None, [I]),
    "mmbert_gemm_tn_force_one_launch": (None, [I]),
    "mmbert_gemm_tn_workspace": (SZ, [I, I, I, P]),
    "mmbert_gemm_tn": (I, [P, P, I, P, I, P, I, I, I, I, I, F, P, P, P]),
    "mmbert_gemm_tn_grouped_workspace": (SZ, [I, P, P, I, P]),
    "mmbert_gemm_tn_grouped": (I, [P, I, P, P, P, P, P, P, P, P, I, I, F, P, P]),
    "mmbert_gemm_tn_grouped_rows": (I, [P, I, P, P, P, P, P, P, P, P, P, I, P, F, P, P]),
    "mmbert_colsum": (I, [P, P, I, I, I, P, F, P]),
    "mmbert_rng_stream": (U32, [U64, U32]),
    "mmbert_dropout_thr16": (U32, [F]),
    "mmbert_dropout_mask": (I, [P, P, SZ, U32, U32]),
    "mmbert_mlm_mask": (I, [P, P, P, SZ, U32, U32, U32, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "mmbert_ln_fwd": (I, [P, P, I, P, P, I, P, I, I, P, P, F, P, P, U32, U32, F, I]),
    "mmbert_ln_bwd": (I, [P, P, I, P, P, I, P, P, P, P, I, I, P, I, P, P, I, P, P, P, U32, U32, F, U32, U32, F, P, P, I, I]),
    "mmbert_ln_bwd_reduce": (I, [P, I, P, P, P, P, I, I]),
    "mmbert_ln_bwd_reduce_rows": (I, [P, I, P, P, P, P, P, I]),
    "mmbert_ln_bwd_workspace": (SZ, [I, I]),
    "mmbert_embed_gather": (I, [P, P, P, P, P, P, I, I, I, I, P, I]),
    "mmbert_embed_scatter": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P]),
    "mmbert_pair_proj_fwd": (I, [P, P, I, I, I, I, P, P, I, P, I, I]),
    "mmbert_pair_proj_bwd": (I, [P, P, I, I, I, I, P, P, I, I, P, P, I, P]),
    "mmbert_pair_proj_bwd_workspace": (SZ, [I, I, I, I]),
    "mmbert_attn_tile_rows": (I, [I]),
    "mmbert_attn_kv_len": (I, [P, P, P, P, I, P]),
    "mmbert_attn_fwd": (I, [P, P, P, P, P, P, I, I, P, P, P, P, P, I, U32, U32, F, P, P, P]),
    "mmbert_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, P, P, P, P, P, I, P, P, I, U32, U32, F, P, P, P, I, P]),
    "mmbert_attn_q_limit": (I, [P, P, I, P, I, P]),
    "mmbert_attn_dropout_mask": (I, [P, P, I, C.c_uint, I, U32, U32]),
    "mmbert_ce_fwd": (I, [P, P, I, I, P, I, P, I, P, P, P, I, P]),
    "mmbert_ce_bwd": (I, [P, P, I, I, P, I, P, I, P, P, P, P, I, P, I, I]),
    "mmbert_active_rows": (I, [P, P, I, I, P, P]),
    "mmbert_prologue": (I, [P, I, P, P, P, P, P, P, P, I, P, I, P, I, P, P, P, P, P, P, P, P]),
    "mmbert_split_rows": (I, [P, P, P, P, P, P, I, I, I, P, P, P, P, P]),
    "mmbert_heads_gate_fwd": (I, [P, P, P, P, P, I, I, P, P]),
    "mmbert_heads_loss_fwd": (I, [P, P, P, P, P, P, P, I, I, F, I, P, P, P, P, P, P, P, I, F, P, P]),
    "mmbert_heads_scale": (I, [P, P, SZ, P]),
    "mmbert_heads_seed": (I, [P, P, SZ, P, P, P, SZ, P, I, F]),
    "mmbert_heads_gate_bwd": (I, [P, P, P, P, P, P, P, I, I, P, P, P, P]),
    "mmbert_heads_tanh": (I, [P, P, SZ]),
    "mmbert_heads_tanh_bwd": (I, [P, P, P, P, SZ]),
    "mmbert_heads_colsum": (I, [P, I, P, P, P, P, P]),
    "mmbert_skinny_mm": (I, [P, I, P]),
    "mmbert_skinny_mm_workspace": (SZ, [I, P]),
    "mmbert_skinny_mm_ordered": (I, [P, I, P, P]),
    "mmbert_heads_step_struct_size": (I, []),
    "mmbert_heads_step_workspace": (SZ, [I, I]),
    "mmbert_heads_step_fwd": (I, [P, P]),
    "mmbert_heads_step_bwd": (I, [P, P]),
    "mmbert_heads_step_fwd_levels": (I, [P, P, I, I]),
    "mmbert_heads_step_bwd_levels": (I, [P, P, I, I]),
    "mmbert_heads_step_dmlm": (I, [P, P]),
    "mmbert_layer_fwd": (I, [P, P, P]),
    "mmbert_layer_bwd": (I, [P, P, P]),
    "mmbert_layer_struct_sizes": (I, [P]),
    "mmbert_set_deterministic": (None, [I]),
    "mmbert_get_deterministic": (I, []),
    "mmbert_id_runs_sum_rows": (I, [P, P, I, I, P, I, I, I, P, I, P, I]),
    "mmbert_skinny_wgrad": (I, [P, I, P]),
    "mmbert_adamw": (I, [P, P, P, P, P, P, P, SZ, D, D, D, D, D, I, D, I, I]),
    "mmbert_gelu_bwd": (I, [P, P, P, P, SZ]),
    "mmbert_cast_f32_bf16": (I, [P, P, P, SZ]),
    "mmbert_cast_bf16_f32": (I, [P, P, P, SZ]),
    "mmbert_transpose_cast": (I, [P, P, P, P, I, I]),
    "mmbert_transpose_bf16": (I, [P, P, P, P, I, I]),
    "mmbert_gather_rows": (I, [P, I, P, P, P, P, P, P, I]),
    "mmbert_compact_rows": (I, [P, P, P, I, P, I, P, P, P]),
    "mmbert_compact_rows_inv": (I, [P, P, P, I, P, I, P, P, P, P, C.c_uint]),
    "mmbert_scatter_rows_zero": (I, [P, I, P, P, P, P, P, P, C.c_uint, I, I]),
    "mmbert_pack_i64": (I, [P, I, P, P, P, P, P]),
    "mmbert_rows_to_block": (I, [P, P, P, I, I, I, I, P, I, I, P]),
    "mmbert_split_layout": (I, [P, P, P, I, I, I, I, I, P]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the library and binds every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X kernels are not built and there is no CPU fallback. "
            "Run `python -m msa_amd.build` (needs hipcc) first.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code} "
                           f"({'invalid argument' if code < 0 else 'hipError_t'})")
