#!/bin/bash
# PMC passes for one x6 GEMM shape (development aid).  usage: tools/pmc_x6.sh <outdir> M N K
out=$1; M=$2; N=$3; K=$4
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -- python tools/gemm_probe.py $M $N $K 3 1 > $out/set$i.log 2>&1
  echo "set $i done: $set"
done <<SETS
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM
SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
SETS
