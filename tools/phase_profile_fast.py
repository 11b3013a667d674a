"""DIAGNOSTIC: per-phase cycles of the exact-shape fast path (slk_step_fast.hpp) from in-kernel stamps (a -DSLK_STAMPS build).
    python tools/phase_profile_fast.py --lib ab/NAME.so [--batch 4096] [--clones 8]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
STAMPS = [(0, "entry"), (3, "load + first factorisation (wave 0), mean, small arrays"), (4, "Z = h(X), zbar, innovation"), (6, "S (wave 0)"), (7, "gate"),
          (8, "x, b, delta (wave 0) | scan + columns (wave 3)"), (11, "factor update L M"), (12, "mean loop"),
          (13, "correction, odd / even split, mean out"), (14, "rebuild MFMA"), (15, "rank-2 + store")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lib", required=True)
    args = ap.parse_args()
    import torch
    from slkpkg import slk
    import scenarios as sc
    lib = slk.load_library(os.path.join(ROOT, args.lib))
    slk._lib = lib
    lib.slk_debug_set_stamps.argtypes = [C.c_void_p]
    B, k = args.batch, args.clones
    s = sc.synthetic_msckf(B, k, m=8)
    f = slk.Msckf(s["mean"], s["P"])
    dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
    lib.slk_debug_set_stamps(dbg.data_ptr())
    for _ in range(args.steps):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    f.sync()
    t = dbg.cpu().numpy().astype(np.float64)
    ok = t[:, 15] > 0
    print(f"filters through the fast path to the end: {int(ok.sum())} / {B}; mean-loop passes: {np.bincount(t[ok][:, 20].astype(int))}")
    tot = np.median(t[ok][:, 15] - t[ok][:, 0])
    print(f"median cycles per filter (s_memtime ticks): {tot:.0f}")
    prev = 0
    for i, name in STAMPS[1:]:
        d = np.median(t[ok][:, i] - t[ok][:, prev])
        print(f"  {name:52s} {d:9.0f}  {100 * d / tot:5.1f} %")
        prev = i
    x = t[ok]
    if (x[:, 23] > 0).all():
        m = lambda a, b: np.median(x[:, a] - x[:, b])
        print(f"  wave 3: its 20 scans done {m(21, 4):.0f} after stamp 4 (S: {m(6, 4):.0f}); past the park barrier {m(22, 7):.0f} after the gate; "
              f"parked sums + columns {m(23, 22):.0f}")
        print(f"  wave 0: x = S^-1 nu {m(24, 7):.0f} after the gate; b, delta = L b, reference {m(25, 24):.0f}; waits for wave 3 {m(8, 25):.0f}")
    # which waves left the direct SO(3) series in the first mean-loop pass (bit 0 exp, bit 1 log; wave 1: bits 2 / 3 = its second round)
    for w, slot in enumerate((26, 27, 29, 30)):
        print(f"  mean loop, wave {w}: angle-halving exp / log codes {np.bincount(x[:, slot].astype(int), minlength=4)}")


if __name__ == "__main__":
    main()
