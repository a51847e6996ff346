#!/bin/bash
# Developer helper (GPU box): SQ / cache counters of the kernels of one training step, one rocprofv3 --pmc pass per counter
# group (counters are collected with --kernel-trace only).  usage: tools/pmc_sq.sh <tag> [f32|bf16]  ->  gpurun_out/sq_<tag>.txt
tag=$1; dt=${2:-f32}
out=gpurun_out/pmc_sq_$tag; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/p$i -o p -- python3 tools/pmc_steps.py 2 $dt > $out/p$i.log 2>&1 \
    || { echo "pass $i ($group) failed"; tail -3 $out/p$i.log; }
done <<'G'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
G
python3 tools/pmc_sq_table.py $out > gpurun_out/sq_$tag.txt
cat gpurun_out/sq_$tag.txt | cut -c1-240 | head -60
