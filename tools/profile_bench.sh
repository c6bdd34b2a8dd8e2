#!/bin/bash
# Round-end evidence run (on the GPU box through gpurun): the default bench line as the driver runs it, the same command
# under rocprofv3 --kernel-trace --stats, the two PMC passes for the dominant kernel's HBM traffic (separate passes, no
# tracing domain beside --kernel-trace; bench.py also measures the traffic itself through child processes), and the
# training step's kernel statistics and launch trace.  Summaries land in gpurun_out/prof_bench/ — copy them to profiles/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench
mkdir -p $OUT
# the raw profiler output is large (gpurun copies at most 64 MiB back): whatever path the script leaves by, only summaries stay
trap 'rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma $OUT/train $OUT/tt $OUT/beams $OUT/bf16 $OUT/bf16m $OUT/bf16l' EXIT
cd $R
PART=${1:-all}          # "1": the bench line + its kernel statistics + the PMC passes; "2": the beams / perf-mode / training passes
if [ "$PART" != "2" ]; then
timeout -k 10 700 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-traffic --no-config5 --no-beams --no-perf-mode --no-rows64"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py $P > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python3 $R/tools/summarize_prof.py $OUT/stats $OUT/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py $P"
Q="--steps 1 --warmup 0 --no-nar --no-roofline --no-train $P"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python3 $R/bench.py $Q > $OUT/fetch.json 2> $OUT/fetch.err
python3 $R/tools/summarize_prof.py $OUT/fetch $OUT/pmc_fetch_size.md "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py $Q"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python3 $R/bench.py $Q > $OUT/write.json 2> $OUT/write.err
python3 $R/tools/summarize_prof.py $OUT/write $OUT/pmc_write_size.md "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python3 bench.py $Q"
# MFMA pipe utilisation of the MFMA-bound kernels (prompt pass, NAR stage): one counter pass, kernel trace only
U="--steps 1 --warmup 0 --no-roofline --no-train $P --no-rows64"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o pmc --output-format csv -- python3 $R/bench.py $U > $OUT/mfma.json 2> $OUT/mfma.err
python3 $R/tools/summarize_prof.py $OUT/mfma $OUT/pmc_mfma_busy.md "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py $U"
fi
if [ "$PART" = "1" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-traffic --no-config5 --no-beams --no-perf-mode --no-rows64"
# round 5: the shared-prompt decode (the `beams` leg: generate() of one utterance with 32 beams) under the kernel trace ...
W="--steps 1 --warmup 0 --no-nar --no-roofline --no-train --no-cpu-baseline --no-traffic --no-config5 --no-perf-mode --no-rows64 --no-default-generate"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/beams -o stats --output-format csv -- python3 $R/bench.py $W > $OUT/beams.json 2> $OUT/beams.err
python3 $R/tools/summarize_prof.py $OUT/beams $OUT/beams_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py $W"
# ... and the perf-mode (bf16 MFMA) kernels: kernel statistics, MFMA-busy and LDS bank-conflict counters (separate passes)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bf16 -o stats --output-format csv -- python3 $R/tools/bench_bf16.py --reps 3 > $OUT/bench_bf16_under_rocprof.log 2> $OUT/bf16.err
python3 $R/tools/summarize_prof.py $OUT/bf16 $OUT/bf16_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_bf16.py --reps 3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/bf16m -o pmc --output-format csv -- python3 $R/tools/bench_bf16.py --reps 1 --skip-model > $OUT/bf16m.log 2> $OUT/bf16m.err
python3 $R/tools/summarize_prof.py $OUT/bf16m $OUT/pmc_bf16_mfma_busy.md "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/bench_bf16.py --reps 1 --skip-model"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/bf16l -o pmc --output-format csv -- python3 $R/tools/bench_bf16.py --reps 1 --skip-model > $OUT/bf16l.log 2> $OUT/bf16l.err
python3 $R/tools/summarize_prof.py $OUT/bf16l $OUT/pmc_bf16_lds.md "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/bench_bf16.py --reps 1 --skip-model"
timeout -k 10 200 python3 $R/tools/bench_bf16.py > $OUT/bench_bf16.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/train -o t --output-format csv -- python3 $R/tools/bench_train.py steps=4 > $OUT/train.log 2> $OUT/train.err
python3 $R/tools/summarize_prof.py $OUT/train $OUT/train_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py steps=4 (7 AR + 7 NAR steps of configs[3])"
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tt -o t --output-format csv -- python3 $R/tools/train_trace.py run > $OUT/tt.log 2> $OUT/tt.err
python3 $R/tools/train_trace.py report $OUT/tt > $OUT/train_trace.md
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma $OUT/train $OUT/tt $OUT/beams $OUT/bf16 $OUT/bf16m $OUT/bf16l
