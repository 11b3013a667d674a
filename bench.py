"""bench.py -- filter predict+update steps/sec on MI355X (BASELINE.json metric).

Workload (N GPUs, weak scaling): per GPU a batch of B = 4096 independent Msckf filters with
k = 8 clones (state dim N = 60), m = 8 measurement rows (4 two-dimensional features), synthetic
inputs of SURVEY.md 8(d), resident in HBM before the timed region.  One "step" = one fused
predict + update (+ applyDelta) launch over the whole batch = B filter steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--clones k] [--meas m]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (public spec)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_flops(N, m, msckf=True):
    """F(N, m) = F_pred + F_upd per filter step."""
    n = 12
    S = 2 * N + 1
    f_pred = n ** 3 / 3 + 2 * n * n + 64 * (2 * n + 1) + 2 * (2 * n + 1) * (n + 30) + 4 * n * n * (2 * n + 1) \
        + (8 / 3) * n ** 3 + 2 * n ** 3
    f_upd = 2 * N ** 3 / 3 + 4 * N * N + 64 * S + S * m + 2 * m * m * S + 2 * N * m * S + S * N + (8 / 3) * m ** 3 \
        + 4 * N * m * m + 2 * N * N * m + 2 * N * m + 2 * S * N + 2 * N * N * S
    if not msckf:
        f_upd -= N ** 3 / 3 + 2 * N * N + 2 * S * N + 2 * N * N * S
    return f_pred + f_upd


def algorithmic_bytes(N, Nq, m, U=13):
    """each of {mean, P} read once and written once per step, plus step inputs."""
    return 8 * (2 * N * N + 2 * Nq + m + m * m + 144 + U)


def host_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota where there is one, and
    capped at 32 so that the bounded sample stays bounded on a box that shows every core of a shared host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, 32)


def cpu_baseline(k, m, seed, threads=1):
    """The CPU oracle (C restatement of the reference algorithm, NOT Eigen) on a bounded sample of the same
    workload: threads = 1 is the "reference single-thread" leg; threads > 1 spreads the filters of the sample over
    host threads (the reference itself has no threads: independent filter objects, one per thread)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as o
    import scenarios as sc
    N = 12 + 6 * k
    per_thread = max(8, int(64 * (60.0 / N) ** 3))          # ~10 s of CPU work per thread at any N
    Bc, steps = per_thread * threads, 1200 if threads == 1 else 900      # ~10 s of CPU work on this host per leg
    s = sc.synthetic_msckf(Bc, k, m=m, seed=seed)
    mean, P = s["mean"].copy(), s["P"].copy()
    o.lib()

    def shard(i):      # ctypes releases the GIL for the duration of the C call
        sl = slice(i * per_thread, (i + 1) * per_thread)
        st, _ = o.msckf_step_batch(k, m, steps, mean[sl], P[sl], np.ascontiguousarray(s["u"][sl]),
                                   np.ascontiguousarray(s["feat"][sl]), np.ascontiguousarray(s["z"][sl]), s["Q"], s["R"])
        return st

    t0 = time.perf_counter()
    if threads == 1:
        sts = [shard(0)]
    else:
        with ThreadPoolExecutor(threads) as ex:
            sts = list(ex.map(shard, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": Bc * steps / dt, "unit": "filter_steps/s", "cores": threads, "kind": "port",
            "sample": f"{Bc} filters x {steps} steps of the same workload (N={N}, m={m}), "
                      f"oracle/slk_oracle.c gcc -O3 -march=native, {threads} thread(s), {dt:.1f} s, status {max(sts)}"}


def dry_run(args, rank, world):
    """Rehearse the multi-rank control path on CPU (gloo): per-rank shard seeds, barrier, max-over-ranks
    timing, status reduction, rank-0 JSON.  No filter arithmetic happens here."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = torch.tensor([0x5EED0000 + rank], dtype=torch.int64)
    if world > 1:
        gathered = [torch.zeros_like(seeds) for _ in range(world)]
        dist.all_gather(gathered, seeds)
        dist.barrier()
    else:
        gathered = [seeds]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))          # uneven ranks: the reduction must take the slowest
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "filter predict+update steps/sec", "dry_run": True, "n_gpus": world,
                          "value": args.batch * world * args.steps / elapsed, "unit": "filter_steps/s",
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "scaling": "weak", "shard_seeds": [int(g.item()) for g in gathered]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="filters per GPU")
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--meas", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launch/rendezvous/reduction path (gloo, no GPU, no kernels); "
                         "its numbers are meaningless and marked as such")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from slkpkg import slk
    import scenarios as sc

    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available() or slk.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev)

    B, k, m = args.batch, args.clones, args.meas
    s = sc.synthetic_msckf(B, k, m=(m if m != 3 else 2), seed=0x5EED0000 + rank)
    if m == 3:
        s["R"] = 0.01 * np.eye(3)
    N, Nq = s["N"], s["Nq"]
    stream = torch.cuda.current_stream(dev)
    f = slk.Msckf.__new__(slk.Msckf)
    slk._FilterBatch.__init__(f, B, device=local_rank, stream=stream.cuda_stream, n_clones=k)
    f.set_state(s["mean"], s["P"])
    # step inputs resident in HBM (device pointers handed to the C ABI)
    d = {n: torch.from_numpy(np.ascontiguousarray(s[n])).to(dev) for n in ("u", "feat", "z")}
    d["feat"] = d["feat"].reshape(B, -1).contiguous()
    d["Q"] = torch.from_numpy(np.ascontiguousarray(s["Q"].T)).to(dev)
    d["R"] = torch.from_numpy(np.ascontiguousarray(s["R"].T)).to(dev)

    if m == 3:      # BASELINE cfg2 (N=12, m=3): position fix of pose 0 instead of image features
        d["pose"] = torch.zeros(1, dtype=torch.float64, device=dev)
        d["z3"] = torch.from_numpy(np.ascontiguousarray(s["mean"][:, 0:3] + 0.05)).to(dev)

    def step():
        if m == 3:
            f.step(slk.PM_DELTA_POSE, d["u"], d["Q"], d["z3"], slk.MM_POSE_POSITION, d["pose"], d["R"], gate=0)
        else:
            f.step(slk.PM_DELTA_POSE, d["u"], d["Q"], d["z"], slk.MM_FEATURE_PROJ, d["feat"], d["R"])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    f.timer_start()                       # HIP events on the stream the kernel is launched on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = f.timer_stop() / args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kt = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kernel_ms = float(kt.item())
    status = f.status()
    bad = int(np.count_nonzero(status & ~slk.ST_ALL_REJECTED))
    if world > 1:
        bt = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(bt, op=dist.ReduceOp.SUM)     # RCCL: the only collective, outside the timed region
        bad = int(bt.item())

    if rank == 0:
        flops = algorithmic_flops(N, m)
        achieved = flops * B / (kernel_ms * 1e-3) / 1e12
        hbm = algorithmic_bytes(N, Nq, m) * B / (kernel_ms * 1e-3) / 1e9
        traffic = None
        try:    # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                t = json.load(fh)
            w = t["workload"]
            if (w["state_dim"], w["meas_rows"], w["batch_per_gpu"]) == (N, m, B):
                traffic = (2.0 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024.0
        except (OSError, KeyError, ValueError):
            traffic = None
        # N >= 48: 83-85 % of the flops are the MFMA contractions and the arithmetic intensity (21 flop/B at N = 60) is
        # above the fp64 ridge -> priced against the fp64 matrix peak.  N <= 18 (BASELINE cfg2): tiny problems at
        # ~10 flop/B, priced against the HBM roof with the algorithmic bytes of SURVEY 8(d).
        mfma_bound = N >= 48
        roofline = {"bound": "mfma" if mfma_bound else "hbm",
                    "achieved": achieved if mfma_bound else hbm,
                    "peak": PEAK_FP64_MFMA_TFLOPS if mfma_bound else PEAK_HBM_GBS,
                    "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": achieved / PEAK_FP64_MFMA_TFLOPS if mfma_bound else hbm / PEAK_HBM_GBS,
                    "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/pmc_traffic.json",
                    "algorithmic_bytes_per_launch": algorithmic_bytes(N, Nq, m) * B,
                    "kernel": ("msckf_predict_kernel + msckf_chol_kernel + msckf_step_kernel (one filter step = three launches; "
                               "kernel_ms = their summed duration per step, HIP events on the launch stream)") if 32 < N <= 64
                              else "msckf_predict_kernel + msckf_step_kernel",
                    "kernel_ms": kernel_ms,
                    "flops_per_filter_step": flops, "fp64_TFLOPs": achieved, "fp64_frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                    "hbm_algorithmic_GBs": hbm, "hbm_frac": hbm / PEAK_HBM_GBS}
        out = {
            "metric": "filter predict+update steps/sec",
            "value": B * world * args.steps / elapsed,
            "unit": "filter_steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"Msckf N={N} (k={k} clones), m={m}, batch={B} per GPU, fused predict+update "
                                   f"(BASELINE.json configs[2]/[3])",
                       "state_dim": N, "meas_rows": m, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"independent filters sharded over {world} GPU(s), no data-path collective"},
            "roofline": roofline,
            "filters_with_numerical_status": bad,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(k, m, 0x5EED0000)
            out["speedup_vs_cpu_single_thread"] = out["value"] / out["cpu_baseline"]["value"]
            cores = host_cores()
            if cores > 1:
                out["cpu_baseline_all_cores"] = cpu_baseline(k, m, 0x5EED0000, threads=cores)
                out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
