"""Precision sweep of the covariance rebuild (BASELINE.json config 5): fp64 (parity path) vs fp32 MFMA vs
bf16-operand / fp32-accumulate MFMA.  Same inputs; errors after 3 steps against the fp64 GPU result (which itself agrees with the CPU
oracle to ~1e-14) and, for a few filters, against the oracle; times like bench.py (10 warm-up + 50 timed steps on one
handle): the fp64 row is the bench row of the same shape."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(slk, s, mode, steps, warmup=10, timed=50):
    """Errors: `steps` steps from the scenario's state on a fresh handle.  Time: like bench.py -- `warmup` untimed steps, then
    `timed` steps on the SAME handle between two HIP events on the launch stream (inputs resident in HBM)."""
    import torch
    dev = torch.device("cuda")
    d = {n: torch.from_numpy(np.ascontiguousarray(s[n])).to(dev) for n in ("u", "z")}
    d["feat"] = torch.from_numpy(np.ascontiguousarray(s["feat"].reshape(s["B"], -1))).to(dev)
    d["Q"] = torch.from_numpy(np.ascontiguousarray(s["Q"].T)).to(dev)
    d["R"] = torch.from_numpy(np.ascontiguousarray(s["R"].T)).to(dev)

    def step(h):
        h.step(slk.PM_DELTA_POSE, d["u"], d["Q"], d["z"], slk.MM_FEATURE_PROJ, d["feat"], d["R"])
    f = slk.Msckf(s["mean"], s["P"])
    f.set_rebuild_precision(mode)
    for _ in range(steps):
        step(f)
    f.sync()
    st = f.status()
    P, M = f.getPk(), f.muState()
    t = slk.Msckf(s["mean"], s["P"])
    t.set_rebuild_precision(mode)
    for _ in range(warmup):
        step(t)
    t.sync()
    t.timer_start()
    for _ in range(timed):
        step(t)
    ms = t.timer_stop() / timed
    return P, M, ms, int(np.count_nonzero(st & ~slk.ST_ALL_REJECTED))


def main():
    import torch  # noqa: F401
    from slkpkg import slk
    from oracle import oracle as o
    import scenarios as sc
    names = {0: "fp64", 1: "fp32", 2: "bf16/fp32acc"}
    print("config                mode          ms/step   max rel err P   max err mean   filters flagged")
    for (B, k, steps) in ((4096, 8, 3), (512, 31, 3), (512, 32, 3)):
        s = sc.synthetic_msckf(B, k, m=8)
        lay = o.layout(o.MULTI, k)
        ref = None
        for mode in (0, 1, 2):
            P, M, ms, bad = run(slk, s, mode, steps)
            if mode == 0:
                ref = (P, M)
                # oracle check of the parity path on 2 filters
                mean, Pc = s["mean"][:2].copy(), s["P"][:2].copy()
                o.msckf_step_batch(k, 8, steps, mean, Pc, np.ascontiguousarray(s["u"][:2]), np.ascontiguousarray(s["feat"][:2]),
                                   np.ascontiguousarray(s["z"][:2]), s["Q"], s["R"])
                N = s["N"]
                eo = max(np.abs(P[b] - Pc[b].reshape(N, N).T).max() / np.abs(Pc[b]).max() for b in range(2))
                print(f"N={s['N']:3d} B={B:5d}        fp64 vs CPU oracle (2 filters): max rel err P {eo:.2e}")
            ok = np.isfinite(P).all(axis=(1, 2))
            eP = max(np.abs(P[b] - ref[0][b]).max() / np.abs(ref[0][b]).max() for b in np.nonzero(ok)[0][:256])
            eM = max(np.abs(o.boxminus(lay, M[b], ref[1][b])).max() for b in np.nonzero(ok)[0][:64])
            print(f"N={s['N']:3d} B={B:5d}        {names[mode]:12s} {ms:8.3f}   {eP:13.2e}   {eM:12.2e}   {bad}")


if __name__ == "__main__":
    main()
