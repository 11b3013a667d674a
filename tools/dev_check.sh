#!/bin/bash
# Development loop on the GPU box: parity of a development build (ab/<name>.so, N = 60 instantiation only) against the
# oracle on the tests that exercise that shape, then a same-box A/B of the given builds.
#   tools/dev_check.sh <lib-under-test> [other libs for the A/B ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1
SLK_HIP_LIB=$PWD/$T timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q \
  -k "8-8-64 or full_size or long_trajectory or batch_golden or unit_test_scenario_against_golden[8]" 2>&1 | tail -5 || exit 1
tools/ab.sh 3 "--steps 200 --warmup 20" "$@"
