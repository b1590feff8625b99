#!/usr/bin/env python3
"""Builds and runs tools/ubench/valu_rate.hip: clk per wave-instruction per SIMD for the VALU operations the dropout hash / softmax use."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "valu_rate.so")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "valu_rate.hip")])
    if "--build" in sys.argv: sys.exit(0)
lib = ctypes.CDLL(so)
lib.valu_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(4, dtype=torch.int32, device="cuda")
names = ["v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32", "v_exp_f32", "v_xor_b32", "v_lshrrev_b32", "v_pk_sub_i16 clamp",
         "v_cvt_pk_bf16_f32", "v_fma_f32", "v_perm_b32", "v_alignbit_b32", "v_bfi_b32", "v_and_or_b32", "v_xad_u32"]
iters = 2000
for threads in (256, 512):
    print(f"{threads // 256} wave(s) per SIMD, 256 workgroups:")
    for k, nm in enumerate(names):
        for _ in range(2):
            lib.valu_probe(k, threads, 256, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.valu_probe(k, threads, 256, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n_inst = iters * 64 * (threads // 256)            # wave-instructions per SIMD
        print(f"  {nm:22s} {ms * 1e-3 * 2.4e9 / n_inst:6.2f} clk per wave-instruction per SIMD at 2.4 GHz nominal ({ms:.3f} ms)")
