set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_beams
mkdir -p $OUT
trap 'rm -rf $OUT/fetch $OUT/write' EXIT
cd /tmp && export TMPDIR=/tmp
W="--steps 1 --warmup 0 --no-nar --no-roofline --no-train --no-cpu-baseline --no-traffic --no-config5 --no-perf-mode --no-rows64 --no-default-generate"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python3 $R/bench.py $W > $OUT/fetch.json 2> $OUT/fetch.err
python3 $R/tools/summarize_prof.py $OUT/fetch $OUT/pmc_beams_fetch_size.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py $W"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python3 $R/bench.py $W > $OUT/write.json 2> $OUT/write.err
python3 $R/tools/summarize_prof.py $OUT/write $OUT/pmc_beams_write_size.md "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python3 bench.py $W"
