#!/bin/bash
# rocprofv3 kernel-trace + stats of the EKF update bench (run on the GPU box).
# usage: tools/stats_run_ekf.sh <tag>  -> gpurun_out/stats_ekf_<tag>_kernel_stats.csv, gpurun_out/stats_ekf_<tag>.log
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-x}
export TMPDIR=/tmp
OUT=/tmp/stats_ekf_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_ekf.py --no-cpu > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
grep 'EKF update' $OUT/bench.log > gpurun_out/stats_ekf_${TAG}.log
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/stats_ekf_${TAG}_kernel_stats.csv
cat gpurun_out/stats_ekf_${TAG}_kernel_stats.csv gpurun_out/stats_ekf_${TAG}.log
