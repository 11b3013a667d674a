"""PCIe-inclusive rate of the headline workload: the same fused steps, but the per-step inputs (u, z, landmarks, Q, R)
are HOST buffers handed over the C ABI (SLK_HOST) every step -- the library stages them through its own device
buffers.  The state (mean, P) stays resident, as in a running filter.  Not the bench.py `value` (that one has the
inputs resident in HBM); quoted in DESIGN.md."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    from slkpkg import slk
    import scenarios as sc
    B, k, m = 4096, 8, 8
    s = sc.synthetic_msckf(B, k, m=m)
    f = slk.Msckf(s["mean"], s["P"])
    feat = np.ascontiguousarray(s["feat"].reshape(B, -1))
    args = (slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, feat, s["R"])
    for _ in range(10):
        f.step(*args)
    f.sync()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        f.step(*args)
    f.sync()
    dt = time.perf_counter() - t0
    nbytes = s["u"].nbytes + s["z"].nbytes + feat.nbytes + s["Q"].nbytes + s["R"].nbytes
    print(f"host inputs every step (N=60, m=8, B={B}): {B * K / dt:.4g} filter-steps/s, {1e3 * dt / K:.4f} ms/step, "
          f"{nbytes / 1024:.0f} KiB host->device per step, status {int((f.status() != 0).sum())}")
    # state round trip as well (set_state + step + get_state): the drop-in single-object pattern, batched
    t0 = time.perf_counter()
    for _ in range(20):
        f.set_state(s["mean"], s["P"])
        f.step(*args)
        f.getPk()
        f.muState()
    dt = time.perf_counter() - t0
    print(f"state up + step + state down: {B * 20 / dt:.4g} filter-steps/s, {1e3 * dt / 20:.3f} ms/step "
          f"({(s['mean'].nbytes + s['P'].nbytes) / 2**20:.1f} MiB each way)")


if __name__ == "__main__":
    main()
