"""Throughput of the batched Usckf path (BASELINE config 1 shape: N = 48, m = 3, SPD variant)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from slkpkg import slk
    import scenarios as sc
    from oracle import oracle as o
    batches = [int(x) for x in sys.argv[1:]] or [1024, 4096, 16384]
    for B in batches:
        s = sc.synthetic_usckf(B)
        f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
        dev = torch.device("cuda")
        u = torch.from_numpy(s["u"]).to(dev)
        z = torch.from_numpy(s["z"]).to(dev)
        Q = torch.from_numpy(np.ascontiguousarray(s["Q"].T)).to(dev)
        R = torch.from_numpy(np.ascontiguousarray(s["R"].T)).to(dev)
        for _ in range(5):
            f.step(slk.PM_CONST_VELOCITY, u, Q, z, slk.MM_VO_RELATIVE, None, R)
        f.sync()
        K = 100
        f.timer_start()
        for _ in range(K):
            f.step(slk.PM_CONST_VELOCITY, u, Q, z, slk.MM_VO_RELATIVE, None, R)
        ms = f.timer_stop() / K
        bad = int(np.count_nonzero(f.status()))
        print(f"Usckf N=48 m=3 B={B}: {B / ms * 1e3:.3e} filter-steps/s, {ms:.4f} ms/step, filters with status {bad}")
    s = sc.synthetic_usckf(64)
    t0 = time.perf_counter()
    n = 0
    for b in range(64):
        g = o.Usckf(nfk=3, nfkl=9, mean=s["mean"][b], P=s["P"][b])
        uu = s["u"][b]
        pm = o.pm_const_velocity(uu[0:3], uu[3:6], uu[6])
        for _ in range(20):
            g.predict(pm, s["Q"])
            g.update(s["z"][b], o.mm_vo_relative(), s["R"])
            n += 1
    dt = time.perf_counter() - t0
    print(f"CPU oracle (1 thread, includes ctypes call overhead): {n / dt:.1f} filter-steps/s")


if __name__ == "__main__":
    main()
