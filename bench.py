"""bench.py -- filter predict+update steps/sec on MI355X (BASELINE.json metric).

Workload (N GPUs, weak scaling): per GPU a batch of B = 4096 independent Msckf filters with
k = 8 clones (state dim N = 60), m = 8 measurement rows (4 two-dimensional features), synthetic
inputs of SURVEY.md 8(d), resident in HBM before the timed region.  One "step" = one
predict + update (+ applyDelta) pass over the whole batch = B filter steps.
`--filter usckf` runs the batched Usckf step instead (N = 48, m = 3: BASELINE configs[0]/[1]'s shape).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--clones k] [--meas m] [--filter msckf|usckf]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process per GPU,
started before this process touches torch, HIP or the product library) and relays rank 0's line.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (public spec)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
SEED0 = 0x5EED0000             # rank r owns the filters seeded SEED0 + r
ST_ALL_REJECTED = 8            # include/slk.h SLK_ST_ALL_REJECTED: every measurement block failed the gate (not an error)


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
        f_pred += 8 * n ** 3 + 4 * n * n * (N - 36)
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


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def cpu_baseline(kind, k, m, seed, threads=1):
    """The CPU oracle (C restatement of the reference algorithm, NOT Eigen) on a bounded sample of the same
    workload: threads = 1 is the "reference single-thread" leg; threads > 1 spreads the filters of the sample over
    host threads (the reference itself has no threads: independent filter objects, one per thread)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as o
    import scenarios as sc
    o.lib()
    if kind == "usckf":
        N = 48
        per_thread, steps = 96, 1200 if threads == 1 else 900          # ~10 s of CPU work per thread
        Bc = per_thread * threads
        s = sc.synthetic_usckf(Bc, seed=seed + 0x1000)
        mean, P = s["mean"].copy(), s["P"].copy()

        def shard(i):
            sl = slice(i * per_thread, (i + 1) * per_thread)
            return o.usckf_step_batch(3, 9, steps, mean[sl], P[sl], np.ascontiguousarray(s["u"][sl]),
                                      np.ascontiguousarray(s["z"][sl]), s["Q"], s["R"])
    else:
        N = 12 + 6 * k
        per_thread = max(8, int(64 * (60.0 / N) ** 3))          # ~10 s of CPU work per thread at any N
        Bc, steps = per_thread * threads, 1200 if threads == 1 else 900
        s = sc.synthetic_msckf(Bc, k, m=m, seed=seed)
        mean, P = s["mean"].copy(), s["P"].copy()

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
            "sample": f"{Bc} filters x {steps} steps of the same workload ({kind} N={N}, m={m}), "
                      f"oracle/slk_oracle.c gcc -O3 -march=native, {threads} thread(s), {dt:.1f} s, status {max(sts)}"}


# ------------------------------------------------------------------------------------------------ self-launch (parent)
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set as torch.distributed.run would), relay rank 0's JSON line, exit with the ranks' status.  This
    parent never imports torch, never touches HIP and never loads the product library; nothing is re-exec'ed."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SLK_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()                     # drain rank 0's pipe while it runs (a full pipe would block the rank)
    rc = 0
    pending = set(range(n))
    while pending and rc == 0:
        for r in sorted(pending):
            c = procs[r].poll()
            if c is not None:
                pending.discard(r)
                if c != 0:
                    rc = c
                    sys.stderr.write(f"bench.py: rank {r} exited with status {c}; stopping the other ranks\n")
        if pending and rc == 0:
            time.sleep(0.05)
    if rc != 0:
        for r in pending:              # exactly the processes started above
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    reader.join(timeout=30)
    out = captured[0] if captured else ""
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    for l in lines:
        print(l, flush=True)
    return rc


# ------------------------------------------------------------------------------------------------ the workload seam
class Rehearsal:
    """--dry-run: a host stand-in for the filter batch (no GPU, no arithmetic, numbers meaningless and marked so).
    It exists so that the control path below -- rank seeding, rendezvous, barriers, the one all_gather of the
    per-rank counters, MAX over ranks, status sums, rank 0's line -- runs on CPU under gloo exactly as written."""
    device = "cpu"
    dry = True

    def __init__(self, args, rank, local_rank):
        import numpy as np
        self.rank, self.B = rank, args.batch
        self.N, self.Nq, self.m = 12 + 6 * args.clones, 13 + 7 * args.clones, args.meas
        self.seed = SEED0 + rank
        self.kind = args.filter
        self._np = np

    def step(self):
        time.sleep(0.001 * (1 + self.rank))          # uneven ranks: the reduction must take the slowest

    def sync(self):
        pass

    def timer_start(self):
        self._t = time.perf_counter()

    def timer_stop(self):
        return (time.perf_counter() - self._t) * 1e3

    def status(self):
        st = self._np.zeros(self.B, dtype=self._np.int32)
        st[:self.rank] = 1                            # rank r: r filters with a numerical status,
        st[self.rank:3 * self.rank] = ST_ALL_REJECTED  # 2r filters whose blocks were all gated out
        return st


class GpuBatch:
    """The filter batch of one rank on its MI355X: the product library behind the reference's filter interface
    (slam-localization_amd/slk.py -> include/slk.h).  No CPU fallback."""
    dry = False

    def __init__(self, args, rank, local_rank):
        import numpy as np
        import torch
        from slkpkg import slk
        import scenarios as sc
        ndev = slk.device_count() if torch.cuda.is_available() else 0
        if args.share_devices and ndev > 0:      # rehearsal of N ranks on fewer GPUs (one-GPU box): ranks share devices
            local_rank %= ndev
        if ndev <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} needs HIP device {local_rank} but {max(ndev, 0)} device(s) are visible "
                             f"(--gpus {args.gpus}); the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.device = dev = torch.device("cuda", local_rank)
        self.shared = bool(args.share_devices)
        self.torch, self.slk = torch, slk
        self.seed = SEED0 + rank
        self.kind = args.filter
        B, k, m = args.batch, args.clones, args.meas
        self.B, self.m = B, m
        stream = torch.cuda.current_stream(dev)
        if args.filter == "usckf":
            s = sc.synthetic_usckf(B, seed=self.seed + 0x1000)
            self.m = m = 3
            f = slk.Usckf.__new__(slk.Usckf)
            slk._FilterBatch.__init__(f, B, device=local_rank, stream=stream.cuda_stream, nfk=3, nfkl=9)
        else:
            s = sc.synthetic_msckf(B, k, m=(m if m != 3 else 2), seed=self.seed)
            if m == 3:
                s["R"] = 0.01 * np.eye(3)
            f = slk.Msckf.__new__(slk.Msckf)
            slk._FilterBatch.__init__(f, B, device=local_rank, stream=stream.cuda_stream, n_clones=k)
        f.set_state(s["mean"], s["P"])
        self.f, self.N, self.Nq = f, s["N"], s["Nq"]
        # step inputs resident in HBM (device pointers handed to the C ABI)
        d = {n: torch.from_numpy(np.ascontiguousarray(s[n])).to(dev) for n in ("u", "z")}
        d["Q"] = torch.from_numpy(np.ascontiguousarray(s["Q"].T)).to(dev)
        d["R"] = torch.from_numpy(np.ascontiguousarray(s["R"].T)).to(dev)
        if args.filter == "usckf":
            self.step = lambda: f.step(slk.PM_CONST_VELOCITY, d["u"], d["Q"], d["z"], slk.MM_VO_RELATIVE, None, d["R"])
        elif m == 3:      # BASELINE cfg2 (N=12, m=3): position fix of pose 0 instead of image features
            d["pose"] = torch.zeros(1, dtype=torch.float64, device=dev)
            d["z3"] = torch.from_numpy(np.ascontiguousarray(s["mean"][:, 0:3] + 0.05)).to(dev)
            self.step = lambda: f.step(slk.PM_DELTA_POSE, d["u"], d["Q"], d["z3"], slk.MM_POSE_POSITION, d["pose"], d["R"], gate=0)
        else:
            d["feat"] = torch.from_numpy(np.ascontiguousarray(s["feat"])).to(dev).reshape(B, -1).contiguous()
            self.step = lambda: f.step(slk.PM_DELTA_POSE, d["u"], d["Q"], d["z"], slk.MM_FEATURE_PROJ, d["feat"], d["R"])
        self._inputs = d

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def timer_start(self):                 # HIP events on the stream the kernels are launched on
        self.f.timer_start()

    def timer_stop(self):
        return self.f.timer_stop()

    def status(self):
        return self.f.status()


def make_workload(args, rank, local_rank):
    """The one place where the bench meets the product library (tests replace the filter batch here)."""
    return Rehearsal(args, rank, local_rank) if args.dry_run else GpuBatch(args, rank, local_rank)


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args, rank, local_rank, world, factory=make_workload):
    import numpy as np
    import torch
    import torch.distributed as dist

    w = factory(args, rank, local_rank)
    # counters travel on the device over RCCL; ranks that share a device (--share-devices) or have none use gloo / host
    dev = "cpu" if getattr(w, "shared", False) else w.device
    on_gpu = str(dev) != "cpu"
    if on_gpu:
        local_rank = dev.index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if on_gpu else {}
        dist.init_process_group(args.backend if on_gpu else "gloo", rank=rank, world_size=world, **kw)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        w.sync()

    for _ in range(args.warmup):
        w.step()
    barrier()
    w.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    kernel_ms_total = w.timer_stop()       # drains the stream: the events bracket exactly the K steps
    barrier()
    elapsed = time.perf_counter() - t0

    status = np.asarray(w.status())
    numerical = status & ~ST_ALL_REJECTED
    mine = [args.steps, int(elapsed * 1e9), int(kernel_ms_total * 1e6), int(np.bitwise_or.reduce(status)) if status.size else 0,
            int(np.count_nonzero(numerical)), int(np.count_nonzero(status & ST_ALL_REJECTED)), w.seed, w.B]
    if world > 1:          # the ONLY data collective (RCCL all_gather of 64 B per rank), after the timed region: SURVEY 8(e)
        t = torch.tensor(mine, dtype=torch.int64, device=dev)
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)
        per_rank = [[int(v) for v in x.cpu().tolist()] for x in g]
    else:
        per_rank = [mine]

    if rank == 0:
        report(args, w, world, per_rank)
    if world > 1:
        dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        dist.destroy_process_group()


def report(args, w, world, per_rank):
    keys = ("steps", "elapsed_ns", "kernel_ns", "status_or", "filters_with_numerical_status", "filters_all_rejected",
            "seed", "batch")
    ranks = [dict(zip(keys, r)) for r in per_rank]
    elapsed = max(r["elapsed_ns"] for r in ranks) * 1e-9          # MAX over ranks
    kernel_ms = max(r["kernel_ns"] for r in ranks) * 1e-6 / args.steps
    total_filters = sum(r["batch"] for r in ranks)
    B, N, Nq, m = w.B, w.N, w.Nq, w.m
    usckf = w.kind == "usckf"
    value = total_filters * args.steps / elapsed
    flops = algorithmic_flops(N, m, msckf=not usckf)
    abytes = algorithmic_bytes(N, Nq, m)
    achieved = flops * B / (kernel_ms * 1e-3) / 1e12
    hbm = abytes * B / (kernel_ms * 1e-3) / 1e9
    traffic = None
    try:    # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_usckf.json" if usckf else "pmc_traffic.json")) as fh:
            t = json.load(fh)
        wl = t["workload"]
        if (wl["state_dim"], wl["meas_rows"], wl["batch_per_gpu"]) == (N, m, B):
            traffic = (2.0 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024.0
    except (OSError, KeyError, ValueError):
        traffic = None
    # Msckf N >= 48: 83-85 % of the flops are the MFMA contractions and the arithmetic intensity (21 flop/B at N = 60) is
    # above the fp64 ridge -> priced against the fp64 matrix peak.  N <= 18 (BASELINE cfg2) and the Usckf step (no
    # applyDelta rebuild, 3.7 flop/B): priced against the HBM roof with the algorithmic bytes of 8(d).
    mfma_bound = N >= 48 and not usckf
    if usckf:
        kernel = "usckf_predict_kernel + usckf_kernel with the factorisation inside (one filter step = two launches)"
    elif 32 < N <= 64 and m == 8:
        kernel = "msckf_predict_kernel + msckf_step_kernel with the first factorisation inside (one filter step = two launches)"
    elif 32 < N <= 64:
        kernel = "msckf_predict_kernel + msckf_chol_kernel + msckf_step_kernel (one filter step = three launches)"
    else:
        kernel = "msckf_predict_kernel + msckf_step_kernel"
    roofline = {"bound": "mfma" if mfma_bound else "hbm",
                "achieved": achieved if mfma_bound else hbm,
                "peak": PEAK_FP64_MFMA_TFLOPS if mfma_bound else PEAK_HBM_GBS,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS if mfma_bound else hbm / PEAK_HBM_GBS,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/pmc_traffic*.json",
                "algorithmic_bytes_per_launch": abytes * B,
                "kernel": kernel + "; kernel_ms = their summed duration per step, HIP events on the launch stream, max over ranks",
                "kernel_ms": kernel_ms,
                "flops_per_filter_step": flops, "fp64_TFLOPs": achieved, "fp64_frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "hbm_algorithmic_GBs": hbm, "hbm_frac": hbm / PEAK_HBM_GBS}
    if usckf:
        workload = f"Usckf N={N} (36 + 3 + 9 features), m={m}, batch={B} per GPU, predict+update (BASELINE.json configs[0]'s shape batched)"
    else:
        workload = (f"Msckf N={N} (k={args.clones} clones), m={m}, batch={B} per GPU, predict+update "
                    f"(BASELINE.json configs[2]/[3])")
    out = {
        "metric": "filter predict+update steps/sec",
        "value": value,
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
        "config": {"workload": workload, "state_dim": N, "meas_rows": m, "batch_per_gpu": B, "global_batch": total_filters,
                   "parallelism": f"independent filters sharded over {world} GPU(s), no data-path collective"},
        "roofline": roofline,
        "filters_with_numerical_status": sum(r["filters_with_numerical_status"] for r in ranks),
        "filters_all_rejected": sum(r["filters_all_rejected"] for r in ranks),
        "status_or": int(np_or(r["status_or"] for r in ranks)),
        "per_rank": ranks,
    }
    if getattr(w, "shared", False):
        out["shared_devices"] = True          # NOT a scaling measurement: the ranks ran on the same GPU(s)
    if w.dry:
        out["dry_run"] = True
        out["shard_seeds"] = [r["seed"] for r in ranks]
    if args.rate_1gpu:
        out["weak_scaling_efficiency_vs_given_1gpu_rate"] = value / (world * args.rate_1gpu)
    if world == 1 and not args.no_cpu_baseline and not w.dry:
        out["cpu_baseline"] = cpu_baseline(w.kind, args.clones, m, SEED0)
        out["speedup_vs_cpu_single_thread"] = out["value"] / out["cpu_baseline"]["value"]
        cores = host_cores()
        if cores > 1:
            out["cpu_baseline_all_cores"] = cpu_baseline(w.kind, args.clones, m, SEED0, threads=cores)
            out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
    print(json.dumps(out), flush=True)


def np_or(values):
    acc = 0
    for v in values:
        acc |= int(v)
    return acc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="filters per GPU")
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--meas", type=int, default=8)
    ap.add_argument("--filter", choices=("msckf", "usckf"), default="msckf")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--rate-1gpu", type=float, default=0.0,
                    help="the N=1 rate of the same workload: adds value / (N * rate) to the line")
    ap.add_argument("--share-devices", action="store_true",
                    help="let ranks share GPUs (rank r on device r mod visible devices; counters gathered over gloo because "
                         "RCCL wants one device per rank): a multi-rank rehearsal on a one-GPU box, marked in the line")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launch/rendezvous/gather path (gloo, no GPU, no kernels); "
                         "its numbers are meaningless and marked as such")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, argv))          # nothing GPU-related has been imported or called yet
        rank = local_rank = 0
        world = 1
    else:
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.filter == "usckf":
        args.clones, args.meas = 0, 3
    run_rank(args, rank, local_rank, world)


if __name__ == "__main__":
    main()
