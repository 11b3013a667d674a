"""Throughput of the Msckf EKF update (slk_update_ekf, Msckf.hpp:284-349) with device-resident inputs.
Flop figure (documented in DESIGN.md): the reference's own steps --
  gate  H P H^T + R: 2mN^2 + 2m^2N, its inverse 2m^3;  Householder QR + thinQ: 4mN^2 - 4N^3/3;
  thinQ^T r, thinQ^T R thinQ: 2mN + 2m^2N + 2mN^2;  S, S^-1, K, K S K^T: 14 N^3."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def flops(N, m):
    return (2 * m * N * N + 2 * m * m * N) + 2 * m ** 3 + (4 * m * N * N - 4 * N ** 3 / 3) + (2 * m * N + 2 * m * m * N + 2 * m * N * N) + 14 * N ** 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--meas", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda")
    torch.cuda.init()
    from slkpkg import slk
    import scenarios as sc
    from oracle import oracle as o
    B, k, m = args.batch, args.clones, args.meas
    e = sc.synthetic_ekf(B, k, m, seed=99)
    N = e["N"]
    f = slk.Msckf(e["mean"], e["P"])
    z = torch.from_numpy(e["z"]).to(dev)
    zm = torch.from_numpy(e["zmean"]).to(dev)
    H = torch.from_numpy(np.ascontiguousarray(np.transpose(e["H"], (0, 2, 1)))).to(dev)
    R = torch.from_numpy(np.ascontiguousarray(np.transpose(e["R"], (0, 2, 1)))).to(dev)
    for _ in range(2):
        f.set_state(e["mean"], e["P"])
        f.update_ekf(z, zm, H, R)
    f.sync()
    tot = 0.0
    for _ in range(args.steps):
        f.set_state(e["mean"], e["P"])            # same well-conditioned problem every time (not timed)
        f.sync()
        f.timer_start()
        f.update_ekf(z, zm, H, R)
        tot += f.timer_stop()
    ms = tot / args.steps
    fl = flops(N, m)
    print(f"Msckf EKF update N={N} m={m} B={B}: {B / (ms * 1e-3):.4g} updates/s, {ms:.3f} ms per launch, "
          f"{fl / 1e6:.2f} Mflop per update -> {fl * B / (ms * 1e-3) / 1e12:.3f} TFLOP/s fp64 "
          f"({100 * fl * B / (ms * 1e-3) / 78.6e12:.2f} % of 78.6), outliers {int(f.outliers().sum())}, status {int((f.status() != 0).sum())}")
    if args.no_cpu:
        return
    # CPU oracle, single thread, bounded sample
    import time
    n = min(B, 8)
    t0 = time.perf_counter()
    for b in range(n):
        r = o.Msckf(k, e["mean"][b], e["P"][b])
        r.update_ekf(e["z"][b], e["zmean"][b], e["H"][b], e["R"][b])
    dt = time.perf_counter() - t0
    print(f"CPU oracle (1 thread, {n} filters): {n / dt:.4g} updates/s")


if __name__ == "__main__":
    main()
