#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (run on the GPU box).
# usage: tools/stats_run.sh <tag>  -> gpurun_out/stats_<tag>_kernel_stats.csv, gpurun_out/stats_<tag>_bench.json
set -e
# BENCH_ARGS (environment): extra bench.py arguments, e.g. "--filter usckf" or "--clones 31 --batch 512"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-x}
export TMPDIR=/tmp
OUT=/tmp/stats_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
grep '^{' $OUT/bench.log > gpurun_out/stats_${TAG}_bench.json
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/stats_${TAG}_kernel_stats.csv
cat gpurun_out/stats_${TAG}_kernel_stats.csv
