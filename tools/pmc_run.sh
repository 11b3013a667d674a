#!/bin/bash
# PMC passes for the bench kernel (counters in their own runs, no trace domains mixed in).
# usage: tools/pmc_run.sh <tag>   (run on the GPU box; writes gpurun_out/pmc_<tag>_*.csv summaries)
set -e
# BENCH_ARGS (environment): extra bench.py arguments, e.g. "--filter usckf" or "--clones 31 --batch 512"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-x}
export TMPDIR=/tmp
OUT=/tmp/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "msckf_" not in k and "usckf_" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    print(name, k[:60])
    for c, v in acc[k].items():
        print(f"   {c:28s} per-dispatch {v / n[(k, c)]:.4g}")
PY
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU 2>&1 | tee gpurun_out/pmc_${TAG}_sq1.txt
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE 2>&1 | tee gpurun_out/pmc_${TAG}_sq2.txt
run sq3 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT 2>&1 | tee gpurun_out/pmc_${TAG}_sq3.txt
run sq4 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU 2>&1 | tee gpurun_out/pmc_${TAG}_sq4.txt
run tcc1 FETCH_SIZE 2>&1 | tee gpurun_out/pmc_${TAG}_fetch.txt
run tcc2 WRITE_SIZE 2>&1 | tee gpurun_out/pmc_${TAG}_write.txt
if [ "${2:-}" = "icache" ]; then
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES 2>&1 | tee gpurun_out/pmc_${TAG}_icache.txt
fi
