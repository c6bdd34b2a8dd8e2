#!/bin/bash
# Developer tool (on the GPU box through gpurun): per-kernel durations of one command.
# usage: tools/prof_kernels.sh <name> <python script and args...>  ->  gpurun_out/<name>.md
set -euo pipefail
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
NAME=$1; shift
OUT="$R/gpurun_out/prof_$NAME"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the program itself directly after `--`: no env / shell hop between the profiler and python3)
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT" -o p --output-format csv -- python3 "$@" > "$R/gpurun_out/$NAME.log" 2> "$OUT.err"
python3 "$R/tools/summarize_prof.py" "$OUT" "$R/gpurun_out/$NAME.md" "rocprofv3 --kernel-trace --stats -- python3 $*"
rm -rf "$OUT"
