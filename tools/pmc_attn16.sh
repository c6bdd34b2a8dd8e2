#!/bin/bash
# Counter passes over the perf-mode attention kernel (developer tool; run on the GPU box through gpurun): one pass per counter
# group, kernel trace only.  $1 = output tag; VH_ATTN16_X1=1 in the environment selects the one-sub-tile kernel.
R=$GRAFT_REPO_ROOT
TAG=${1:-x2}
OUT=$R/gpurun_out/pmc_attn16_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/p$i -o pmc --output-format csv -- python3 $R/tools/attn16_once.py $2 > $OUT/log$i.txt 2>&1 || echo "pass $i failed: $ctrs"
done
python3 $R/tools/summarize_prof.py $OUT $R/gpurun_out/pmc_attn16_$TAG.md "perf-mode attention ($TAG) PMC"
rm -rf $OUT/p*
