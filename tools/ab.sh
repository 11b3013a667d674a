#!/bin/bash
# A/B builds of libslk_hip on the SAME GPU box (box-to-box noise is ~2 %):
#   tools/ab.sh rounds "bench args" libA.so libB.so ...
# prints the bench value of each library, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for L in "$@"; do
    v=$(SLK_HIP_LIB=$PWD/$L timeout -k 10 120 python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
    echo "$L [$ARGS]: $v"
  done
done
