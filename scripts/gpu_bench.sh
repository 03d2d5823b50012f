#!/bin/bash
# One GPU-box call: the default bench (roofline + CPU legs + live traffic) and the rocprofv3 kernel-trace summary of the
# same command.   scripts/gpu_bench.sh <tag> [bench args...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 400 python3 $R/bench.py "$@" > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err || { echo bench failed; tail -5 $R/gpurun_out/${tag}_bench.err; exit 1; }
cat $R/gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py "$@" --no-cpu-baseline --no-live-traffic > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_prof.err || { echo rocprof failed; tail -5 $R/gpurun_out/${tag}_prof.err; exit 1; }
f=$(find $R/gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv && head -5 $R/gpurun_out/${tag}_kernel_stats.csv
