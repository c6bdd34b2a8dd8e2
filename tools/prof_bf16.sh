#!/bin/bash
# Perf-mode kernels under rocprofv3 (on the GPU box through gpurun): kernel statistics of tools/bench_bf16.py, then the MFMA-busy and
# LDS bank-conflict counters in their own passes (kernel trace only beside --pmc).  Summaries -> gpurun_out/prof_bf16/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bf16
mkdir -p $OUT
trap 'rm -rf $OUT/bf16 $OUT/bf16m $OUT/bf16l' EXIT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bf16 -o stats --output-format csv -- python3 $R/tools/bench_bf16.py --reps 3 > $OUT/bench_bf16_under_rocprof.log 2> $OUT/bf16.err
python3 $R/tools/summarize_prof.py $OUT/bf16 $OUT/bf16_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_bf16.py --reps 3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/bf16m -o pmc --output-format csv -- python3 $R/tools/bench_bf16.py --reps 1 --skip-model > $OUT/bf16m.log 2> $OUT/bf16m.err
python3 $R/tools/summarize_prof.py $OUT/bf16m $OUT/pmc_bf16_mfma_busy.md "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/bench_bf16.py --reps 1 --skip-model"
if [ "$1" = "lds" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/bf16l -o pmc --output-format csv -- python3 $R/tools/bench_bf16.py --reps 1 --skip-model > $OUT/bf16l.log 2> $OUT/bf16l.err
python3 $R/tools/summarize_prof.py $OUT/bf16l $OUT/pmc_bf16_lds.md "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/bench_bf16.py --reps 1 --skip-model"
fi
