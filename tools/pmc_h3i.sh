#!/bin/bash
# PMC passes for one fp16x3 GEMM shape (development aid).  usage: tools/pmc_h3i.sh <outdir> M N K [image|raw]
out=$1; M=$2; N=$3; K=$4; MODE=${5:-image}
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -- python3 tools/h3i_probe.py $M $N $K 3 $MODE > $out/set$i.log 2>&1 || exit 1
done <<SETS
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INST_CYCLES_VMEM
SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_IFETCH
TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
FETCH_SIZE
WRITE_SIZE
SETS
python3 tools/pmc_report.py $out gemm_h3i > $out/report.txt
find $out -name "*.csv" -delete; find $out -type d -empty -delete
