#!/usr/bin/env python3
"""Per-kernel register / scratch usage of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py msa_amd/csrc/gemm.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_regs.o"] + [a for a in sys.argv[3:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:60]:60s} VGPR {v.get('VGPRs','?'):>4s} AGPR {v.get('AGPRs','?'):>3s} spillV {v.get('VGPRs Spill','?'):>3s} spillS {v.get('SGPRs Spill','?'):>3s} scratch {v.get('ScratchSize [bytes/lane]','?'):>4s} occ {v.get('Occupancy [waves/SIMD]','?')}")
