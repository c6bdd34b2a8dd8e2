#!/bin/bash
# Developer tool (on the GPU box through gpurun): MFMA-busy counters of the training step, per kernel.
# usage: tools/pmc_train.sh [bench_train.py args, e.g. dropout=0.1]  ->  gpurun_out/pmc_train/pmc_train_mfma.md
set -euo pipefail
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
OUT="$R/gpurun_out/pmc_train"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# counters in their own run (kernel trace only beside --pmc); python3 directly after `--`
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/p" -o pmc --output-format csv -- python3 "$R/tools/bench_train.py" steps=2 "$@" > "$OUT/log.txt" 2>&1
python3 "$R/tools/summarize_prof.py" "$OUT/p" "$OUT/pmc_train_mfma.md" "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/bench_train.py steps=2 $*"
rm -rf "$OUT/p"
