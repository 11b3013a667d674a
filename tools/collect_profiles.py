import json, shutil, os
g='gpurun_out'; p='profiles'
cp={ 'r03_bench_final.json':'r03_bench_final.json','r03_bench_usckf.json':'r03_bench_usckf.json','r03_bench_2rank_shared_device.json':'r03_bench_2rank_shared_device.json',
 'stats_r03_msckf_kernel_stats.csv':'r03_rocprofv3_kernel_stats_final.csv','stats_r03_msckf_bench.json':'r03_rocprofv3_bench_final.json',
 'stats_r03_usckf_kernel_stats.csv':'r03_rocprofv3_kernel_stats_usckf.csv','stats_r03_cfg2_kernel_stats.csv':'r03_rocprofv3_kernel_stats_cfg2.csv',
 'stats_r03_cfg5_kernel_stats.csv':'r03_rocprofv3_kernel_stats_cfg5.csv','stats_ekf_r03_kernel_stats.csv':'r03_rocprofv3_kernel_stats_ekf.csv','stats_ekf_r03.log':'r03_bench_ekf.log',
 'r03_pmc_final.txt':'r03_pmc_final.txt','r03_pmc_usckf.txt':'r03_pmc_usckf.txt','r03_bench_other_configs.log':'r03_bench_other_configs.log',
 'r03_precision_sweep.log':'r03_precision_sweep.log','r03_phase_fast_path.log':'r03_phase_fast_path.log','r03_pmc_phases_fast_path.txt':'r03_pmc_phases_fast_path.txt',
 'r03_fast_path_bailouts.log':'r03_fast_path_bailouts.log','r03_phase_base.log':'r03_phase_general_body_start_of_round.log','r03_pmc_phases_base.txt':'r03_pmc_phases_general_body_start_of_round.txt',
 'r03_dev_fast1.log':'r03_ab_fast_path_v1.log','r03_dev_fast2.log':'r03_ab_fast_path_v2_no_scratch_in_mean_loop.log','r03_dev_fast3.log':'r03_ab_fast_path_v3_scans_on_two_waves.log',
 'r03_dev_fast4.log':'r03_ab_fast_path_v4_pair_table_roles.log','r03_dev_fast5.log':'r03_ab_fast_path_v5_branch_order.log','r03_phase_fast1.log':'r03_phase_fast_path_v1.log','r03_phase_fast2.log':'r03_phase_fast_path_v2.log'}
for a,b in cp.items():
    if os.path.exists(os.path.join(g,a)): shutil.copy(os.path.join(g,a), os.path.join(p,b))
    else: print('missing',a)
def kb(fn, ctr):
    out={}; cur=None
    for l in open(fn):
        if l.startswith('tcc'):
            cur=l.split(None,1)[1].strip()
        elif ctr in l and cur:
            out[cur]=float(l.split()[-1])
    return out
for tag,kern,wl,out in (('final','one filter step = msckf_predict_kernel + msckf_chol_kernel<4,8> + msckf_step_kernel<4,256,8,8> (exact-shape fast path)',{"state_dim":60,"meas_rows":8,"batch_per_gpu":4096},'pmc_traffic.json'),
                        ('usckf','one filter step = usckf_predict_kernel + usckf_kernel<3,128,true,true> (the factorisation runs inside the update kernel)',{"state_dim":48,"meas_rows":3,"batch_per_gpu":4096},'pmc_traffic_usckf.json')):
    f=kb(f'{g}/r03_pmc_{tag}.txt','FETCH_SIZE'); w=kb(f'{g}/r03_pmc_{tag}.txt','WRITE_SIZE')
    per={k.split('(')[0].replace('void slk::','').replace('slk::',''):{"fetch":f[k],"write":w.get(k)} for k in f}
    d={"source":f"profiles/r03_pmc_{tag}.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_run.sh)","kernel":kern,"workload":wl,
       "fetch_size_kb_per_launch":sum(f.values()),"write_size_kb_per_launch":sum(w.values()),"per_kernel_kb":per,
       "note":"gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024 as an upper estimate (our loads are 8 B/lane: uncalibrated); WRITE_SIZE x 1024 is exact for wide stores."}
    if tag=='final':
        d["note"]+=" The step is three launches: the factorisation kernel (one wave per filter) reads the lower triangle of P and writes the packed factor, the update kernel reads it back (L2 / Infinity Cache resident) and stores P+ as lower triangle + diagonal tiles (the strict upper triangle is completed on demand, slk_mirror_upper_kernel) -- about 1.28 x the algorithmic bytes over the whole step (round 2: 1.48 x); the step is bound by instruction issue and per-filter latency of the update kernel, not by memory (DESIGN.md section 3)."
    else:
        d["note"]+=" Usckf step in two launches: predict rewrites the lower-triangle part of the twelve rows / columns of state k+i, the update kernel factors the lower triangle in LDS (no factor round trip) and downdates the six lower tiles as a read-modify-write; the strict upper triangle is completed on demand: about 1.1 x the algorithmic bytes (start of the round: 2.4 x; DESIGN.md section 3, Usckf kernels)."
    json.dump(d,open(f'{p}/{out}','w'),indent=2)
    print(out, (2*d["fetch_size_kb_per_launch"]+d["write_size_kb_per_launch"])*1024/1e6,'MB')
