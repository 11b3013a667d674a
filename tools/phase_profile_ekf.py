"""DIAGNOSTIC: per-phase cycle shares of the LDS-resident EKF update kernel (stamps build, -DSLK_STAMPS)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["P H^T, S0 (MFMA)", "Cholesky of S0 (m x m)", "information blocks (L0^-1 by block columns)", "outlier gate",
         "gather rows", "Householder sweep (panels + WY)", "Hr, thinQ in place", "rn, thinQ^T R thinQ", "U = Hr P, S",
         "Cholesky of S", "X = Ls^-1 U", "Pk update, delta, boxplus"]


def main():
    import torch
    torch.cuda.init()
    spec = importlib.util.spec_from_file_location("slk_build", os.path.join(ROOT, "slam-localization_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = os.path.join(ROOT, sys.argv[sys.argv.index("--lib") + 1]) if "--lib" in sys.argv else b.build(stamps=True)
    from slkpkg import slk
    import scenarios as sc
    lib = slk.load_library(so)
    slk._lib = lib
    lib.slk_debug_set_stamps.argtypes = [C.c_void_p]
    B, k, m = 1024, 8, 128
    e = sc.synthetic_ekf(B, k, m, seed=99)
    f = slk.Msckf(e["mean"], e["P"])
    dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
    lib.slk_debug_set_stamps(dbg.data_ptr())
    f.update_ekf(e["z"], e["zmean"], e["H"], e["R"])
    f.sync()
    t = dbg.cpu().numpy().astype(np.float64)
    d = np.diff(t[:, :13], axis=1)
    tot = np.median(d.sum(axis=1))
    print(f"EKF update N={e['N']} m={m}: median cycles per filter {tot:.0f}")
    for i, n in enumerate(NAMES):
        print(f"  {n:38s} {np.median(d[:, i]):10.0f}  {100 * np.median(d[:, i]) / tot:5.1f} %")
    if (t[:, 22] > 0).all():
        print(f"  Householder sweep: panel columns {np.median(t[:, 22]):.0f}, G / W {np.median(t[:, 23]):.0f}, T {np.median(t[:, 24]):.0f}, "
              f"Z + trailing update {np.median(t[:, 25]):.0f} cycles")
    if (t[:, 20] > 0).all():
        print(f"  diagonal tiles (all factorisations of one update): factor {np.median(t[:, 20]):.0f}, inverse {np.median(t[:, 21]):.0f} cycles")


if __name__ == "__main__":
    main()
