"""Opcode mix of a kernel's hot loop, priced with the issue costs tools/ubench/ubench_valu.hip measured on MI355X (profiles/r5_ubench_valu.txt).

    python tools/isa_mix.py attention.hip attn_fwd_kernelILb1       # the loop with the most MFMAs of every kernel whose mangled name matches
    python tools/isa_mix.py gemm.hip 'gemm_nt8_kernelILi5ELb0ELi7' tail   # the code after the last MFMA (the epilogue)

Cost classes (clk of SIMD issue per wave64 instruction, >= 2 waves per SIMD contending): fast 2.4 (v_add/sub/mul/fma/fmac/fmamk/fmaak f32,
v_add/sub u32, v_and/or/xor, v_mov, v_lshrrev, v_ashrrev, v_bitop3; literals cost nothing extra), transcendental 8.2, everything else 4.3
(v_lshlrev (!), integer multiplies, SDWA and DPP forms, min/max/med3/max3, converts, packed maths, lshl_add/add3/xad/and_or/bfi/perm/bfe,
compares, v_cndmask, v_ldexp, v_add_co)."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msa_amd.build import HIPCC, FLAGS  # noqa: E402

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32",
        "v_fmaak_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32"}
TRANS = {"v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32"}


def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in TRANS:
        return 8.2
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return 4.3
    return 2.4 if base in FAST else 4.3


def assembly(name):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, name + ".s")
        flags = [f for f in FLAGS if f != "-fPIC"]
        subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", os.path.join(ROOT, "msa_amd", "csrc", name), "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def kernels(lines):
    cur, body = None, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            body[cur].append(l)
    return body


def loops(body):
    """(start, end) line ranges of backward branches."""
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    out = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            out.append((labels[m.group(1)], i))
    return out


def mix(seg):
    h = collections.Counter()
    for l in seg:
        t = l.strip().split()
        if t and re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_)", t[0]):
            h[t[0]] += 1
    return h


def report(title, seg):
    h = mix(seg)
    valu = {k: v for k, v in h.items() if k.startswith("v_") and "mfma" not in k and not k.startswith("v_accvgpr")}
    n, clk = sum(valu.values()), sum(cost(k) * v for k, v in valu.items())
    mf = sum(v for k, v in h.items() if "mfma" in k)
    print(f"== {title}: {len(seg)} lines, {n} VALU ({clk:.0f} clk priced), {mf} MFMA, {sum(v for k, v in h.items() if k.startswith('ds_'))} LDS, "
          f"{sum(v for k, v in h.items() if k.startswith('s_'))} scalar")
    for k, v in sorted(valu.items(), key=lambda kv: -cost(kv[0]) * kv[1]):
        print(f"   {k:28s} {v:5d}  x{cost(k):.1f} = {cost(k) * v:7.0f}")


if __name__ == "__main__":
    src, pat = sys.argv[1], sys.argv[2]
    tail = len(sys.argv) > 3 and sys.argv[3] == "tail"
    for name, body in kernels(assembly(src)).items():
        if pat not in name:
            continue
        lp = loops(body)
        if tail:
            last = max(i for i, l in enumerate(body) if "v_mfma" in l)
            report(name + " after the last MFMA", body[last + 1:])
        else:
            best = max(lp, key=lambda se: sum("mfma" in l for l in body[se[0]:se[1]]))
            report(name + " hot loop", body[best[0]:best[1] + 1])
