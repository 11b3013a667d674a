#!/bin/bash
# Every measurement DESIGN.md / BASELINE.md / profiles/README.md quote, in one go on the GPU box (product build in-tree, a
# development stamps build ab/<stamps>.so for the phase tools).  usage: tools/record_round.sh <tag> [ab/<stamps>.so]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-rXX}; ST=${2:-}
O=gpurun_out
export TMPDIR=/tmp
python3 bench.py > $O/${T}_bench_final.json 2> $O/${T}_bench_final.err
python3 bench.py --filter usckf > $O/${T}_bench_usckf.json 2> /dev/null
python3 bench.py --gpus 2 --share-devices --steps 50 > $O/${T}_bench_2rank_shared_device.json 2> /dev/null
tools/stats_run.sh ${T}_msckf > /dev/null 2>&1
BENCH_ARGS="--filter usckf" tools/stats_run.sh ${T}_usckf > /dev/null 2>&1
BENCH_ARGS="--clones 0 --meas 3 --batch 1024" tools/stats_run.sh ${T}_cfg2 > /dev/null 2>&1
BENCH_ARGS="--clones 31 --batch 512" tools/stats_run.sh ${T}_cfg5 > /dev/null 2>&1
tools/stats_run_ekf.sh ${T} > /dev/null 2>&1
tools/pmc_run.sh ${T}_final > $O/${T}_pmc_final.txt 2>&1
BENCH_ARGS="--filter usckf" tools/pmc_run.sh ${T}_usckf > $O/${T}_pmc_usckf.txt 2>&1
tools/sweep_configs.sh > $O/${T}_bench_other_configs.log 2>&1
python3 tools/precision_sweep.py > $O/${T}_precision_sweep.log 2>&1
if [ -n "$ST" ]; then
  python3 tools/phase_profile_fast.py --lib $ST > $O/${T}_phase_fast_path.log 2>&1
  (cd /tmp && rm -rf /tmp/pp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d /tmp/pp -- python3 $OLDPWD/tools/pmc_phases.py --fast --lib $ST > /tmp/pp.log 2>&1)
  python3 tools/pmc_phases.py --fast --report /tmp/pp > $O/${T}_pmc_phases_fast_path.txt 2>&1
  python3 tools/fast_path_bailouts.py $ST 200 > $O/${T}_fast_path_bailouts.log 2>&1
fi
tail -c 600 $O/${T}_bench_final.json; echo; head -4 $O/stats_${T}_msckf_kernel_stats.csv
