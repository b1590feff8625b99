#!/bin/bash
# Counter passes over the three attention kernels at the headline shapes (tools/bench_attn.py); one rocprofv3 --pmc pass per group
# (SQ block: 8 slots per pass).  Output: gpurun_out/pmc_attn_<group>/ ; summarise with tools/pmc_attn.py
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc_counters_list.txt 2>&1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
G2="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
G3="GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INST_LEVEL_LDS"
i=1
for g in "$G1" "$G2" "$G3"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmc_attn_g$i -o p -- python3 tools/bench_attn.py > gpurun_out/pmc_attn_g$i.log 2>&1
  i=$((i+1))
done
ls gpurun_out/pmc_attn_g*/
