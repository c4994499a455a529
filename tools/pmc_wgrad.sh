#!/bin/bash
# PMC passes for the fp16x3 weight-gradient kernel on one shape.  usage: tools/pmc_wgrad.sh <outdir> M N K [kernel-name pattern]
out=$1; M=$2; N=$3; K=$4; pat=${5:-wgrad_}
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -- python3 tools/wgrad_probe.py $M $N $K h3 3 > $out/set$i.log 2>&1 || exit 1
done <<SETS
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
SETS
python3 tools/pmc_report.py $out $pat > $out/report.txt
find $out -name "*.csv" -delete; find $out -type d -empty -delete
