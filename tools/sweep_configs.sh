#!/bin/bash
# Other-shape sweep of the current build (run on the GPU box): bench.py lines for the BASELINE configs' shapes and the
# mid-size states, the Usckf shape, the EKF update.  usage: tools/sweep_configs.sh > gpurun_out/sweep.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
line() {  # clones meas batch
  python3 bench.py --clones $1 --meas $2 --batch $3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('N=%d m=%d B=%d: %.4g steps/s %.4f ms/step bound=%s frac=%.4f fp64_frac=%.4f hbm_frac=%.4f' % (d['config']['state_dim'], d['config']['meas_rows'], d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], r['bound'], r['frac'], r['fp64_frac'], r['hbm_frac']))"
}
line 0 3 1024; line 1 2 1024; line 0 3 16384; line 1 2 16384
line 8 8 1024; line 8 8 4096; line 8 8 16384
line 4 8 4096; line 6 8 4096
line 10 8 2048; line 12 8 2048; line 14 8 2048; line 15 8 2048; line 19 8 1024; line 24 8 1024
line 31 8 512; line 32 8 512
python3 tools/bench_usckf.py 2>/dev/null | grep -v amdgpu
python3 tools/bench_ekf.py 2>/dev/null | grep -v amdgpu
