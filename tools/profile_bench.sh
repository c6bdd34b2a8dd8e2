#!/bin/bash
# Round-end evidence run (on the GPU box through gpurun): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the two PMC passes for the dominant kernel's HBM traffic
# (separate passes, no tracing domains beside --kernel-trace).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench
mkdir -p $OUT
cd $R
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-nar --no-roofline > $OUT/fetch.json 2> $OUT/fetch.err
python3 $R/tools/summarize_prof.py $OUT/fetch $OUT/pmc_fetch_size.md "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-nar --no-roofline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-nar --no-roofline > $OUT/write.json 2> $OUT/write.err
python3 $R/tools/summarize_prof.py $OUT/write $OUT/pmc_write_size.md "rocprofv3 --pmc WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-nar --no-roofline"
rm -rf $OUT/stats $OUT/fetch $OUT/write
