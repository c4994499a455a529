#!/bin/bash
# PMC passes for the head-image self-attention kernels (development aid).  usage: tools/pmc_attn.sh <outdir>
out=$1
mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -- python3 tools/aimg_probe.py > $out/set$i.log 2>&1 || exit 1
  echo "set $i done: $set"
done <<SETS
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM
SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
FETCH_SIZE
WRITE_SIZE
SETS
for k in attn_fwd_img attn_bwd_dq_img attn_bwd_dkv_img; do echo "== $k"; python3 tools/pmc_report.py $out $k; done > $out/report.txt
find $out -name "*.csv" -delete; find $out -type d -empty -delete
