#!/bin/bash
# Developer tool (on the GPU box through gpurun): per-kernel durations of one command.
# usage: tools/prof_kernels.sh <name> <python script and args...>  ->  gpurun_out/<name>.md
R=$GRAFT_REPO_ROOT
NAME=$1; shift
OUT=$R/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 "$@" > $R/gpurun_out/$NAME.log 2> $OUT.err
python3 $R/tools/summarize_prof.py $OUT $R/gpurun_out/$NAME.md "rocprofv3 --kernel-trace --stats -- python3 $*"
rm -rf $OUT
