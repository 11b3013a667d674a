"""DIAGNOSTIC: per-phase cycle shares of the Msckf step kernel from in-kernel s_memtime stamps.

Uses the separate stamps build (libslk_hip_stamps.so, -DSLK_STAMPS); its run time is not a
benchmark number (cdna_hip_programming.md section 7, In-kernel stamps): read SHARES only.

    python tools/phase_profile.py [--batch 4096] [--clones 8] [--meas 8] [--steps 3]
"""
import argparse
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PHASES = ["load", "predict", "chol1", "Z=h(X)", "zbar/innov", "S + Pxz", "gate", "S^-1", "K,KS,delta", "downdate",
          "chol2", "mean loop", "final Drot", "MFMA rebuild", "acc->LDS", "store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--meas", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lib", default=None, help="a stamps build made elsewhere (build.py --dev NAME --stamps), e.g. ab/NAME.so")
    args = ap.parse_args()
    import torch
    spec = importlib.util.spec_from_file_location("slk_build", os.path.join(ROOT, "slam-localization_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = os.path.join(ROOT, args.lib) if args.lib else b.build(stamps=True)
    from slkpkg import slk
    import scenarios as sc
    lib = slk.load_library(so)
    slk._lib = lib                       # route the host mirror through the diagnostic library
    lib.slk_debug_set_stamps.argtypes = [C.c_void_p]
    B, k, m = args.batch, args.clones, args.meas
    s = sc.synthetic_msckf(B, k, m=(m if m != 3 else 2))
    f = slk.Msckf(s["mean"], s["P"])
    dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
    lib.slk_debug_set_stamps(dbg.data_ptr())
    for _ in range(args.steps):
        if m == 3:      # bench.py's config-2 workload: position fix of pose 0
            f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["mean"][:, 0:3] + 0.05, slk.MM_POSE_POSITION, np.zeros(1), 0.01 * np.eye(3),
                   gate=0)
        else:
            f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    f.sync()
    t = dbg.cpu().numpy()
    ok = t[:, 15] > 0
    d = np.diff(t[ok][:, :16].astype(np.float64), axis=1)
    tot = d.sum(axis=1)
    print(f"blocks with a full update: {ok.sum()} / {B}; mean-loop trips: "
          f"{np.bincount(t[ok][:, 20].astype(int))}")
    print(f"median cycles per filter (s_memtime ticks): {np.median(tot):.0f}")
    for i, name in enumerate(PHASES[1:] if False else PHASES[:15]):
        pass
    names = ["load", "predict", "chol1", "Z=h(X)", "zbar/innov", "S+Pxz", "gate", "S^-1", "K,KS,delta", "downdate",
             "chol2", "mean loop", "final Drot", "MFMA rebuild", "acc->LDS+store"]
    for i, name in enumerate(names):
        print(f"  {name:16s} {np.median(d[:, i]):10.0f}  {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")
    if (t[ok][:, 21] > 0).all():      # sub-stamps of the predict phase (wave 0)
        x = t[ok].astype(np.float64)
        print(f"  predict detail: 12x12 Cholesky {np.median(x[:, 21] - x[:, 1]):.0f}, sigma points + f {np.median(x[:, 22] - x[:, 21]):.0f}, "
              f"manifold mean {np.median(x[:, 23] - x[:, 22]):.0f} ({np.bincount(t[ok][:, 25].astype(int))} trips), "
              f"final deviations + covariance + copy {np.median(x[:, 2] - x[:, 23]):.0f}")
    if (t[ok][:, 16] > 0).all():      # sub-stamps of the moments phase (thread 0 = wave 0)
        x = t[ok].astype(np.float64)
        print(f"  moments detail (wave 0): stamp5->Pxz done {np.median(x[:, 16] - x[:, 5]):.0f}, "
              f"S tiles {np.median(x[:, 17] - x[:, 16]):.0f}, barrier wait {np.median(x[:, 18] - x[:, 17]):.0f}")


if __name__ == "__main__":
    main()
