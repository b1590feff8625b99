#!/bin/bash
# A variant library for the same-process A/B tools: ONE source replaced, the rest from the product objects (msa_amd/_obj).
#   tools/ab_build.sh /tmp/exp/attention.hip attention tools/_ab/lib_variant.so   ->  LIB_B=tools/_ab/lib_variant.so python tools/ab_attn.py
set -e
src=$1; which=$2; out=$3
cd "$(dirname "$0")/.."
python -m msa_amd.build >/dev/null
mkdir -p tools/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Imsa_amd/csrc -c "$src" -o /tmp/_ab_variant.o
objs=""
for f in gemm attention rowwise heads heads_coop layer; do
    if [ "$f" = "$which" ]; then objs="$objs /tmp/_ab_variant.o"; else objs="$objs msa_amd/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
echo "$out"
