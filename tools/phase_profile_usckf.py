"""DIAGNOSTIC: per-phase cycles of the split Usckf update kernel (stamps build, -DSLK_STAMPS); shares only."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    s = sc.synthetic_usckf(B)
    f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
    f.predict(slk.PM_CONST_VELOCITY, s["u"], s["Q"])
    f.sync()
    lib.slk_debug_set_stamps(dbg.data_ptr())
    f.update(s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    f.sync()
    t = dbg.cpu().numpy().astype(np.float64)
    names = {0: "start", 3: "state + factor loaded", 4: "Z = h(X)", 5: "zbar, innovation", 6: "S, covXZ (MFMA)", 7: "chol(S)",
             8: "K, mahalanobis", 9: "delta, P -= covXZ K^T", 10: "boxplus, mean store"}
    keys = sorted(names)
    tot = np.median(t[:, 10] - t[:, 0])
    print(f"Usckf update N={s['N']} B={B}: median cycles per filter {tot:.0f}")
    for a, bb in zip(keys[:-1], keys[1:]):
        d = np.median(t[:, bb] - t[:, a])
        print(f"  {names[bb]:28s} {d:10.0f}  {100 * d / tot:5.1f} %")


if __name__ == "__main__":
    main()
