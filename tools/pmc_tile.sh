#!/bin/bash
# PMC passes over the large-M GEMM shapes (developer tool; run on the GPU box through gpurun).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_tile
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/pmc_tile/p$i -o pmc --output-format csv -- python3 $R/tools/tile_once.py > $R/gpurun_out/pmc_tile/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 $R/tools/summarize_prof.py $R/gpurun_out/pmc_tile $R/gpurun_out/pmc_tile_summary.md "tile GEMM (LDS-DMA staging) PMC"
