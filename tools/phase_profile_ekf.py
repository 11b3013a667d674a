"""DIAGNOSTIC: per-phase cycle shares of the LDS-resident EKF update kernel (stamps build, -DSLK_STAMPS)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["P H^T, S0 (MFMA)", "Cholesky of S0 (m x m)", "inverse factor", "information blocks, gate", "gather rows",
         "Householder sweep", "thinQ", "Hr, rn, R thinQ, thinQ^T R thinQ", "P Hr^T, S", "Cholesky of S", "K row solves",
         "Pk update, delta, boxplus"]


def main():
    import torch
    torch.cuda.init()
    spec = importlib.util.spec_from_file_location("slk_build", os.path.join(ROOT, "slam-localization_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = b.build(stamps=True)
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


if __name__ == "__main__":
    main()
