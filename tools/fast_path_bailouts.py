"""DIAGNOSTIC: which filters leave the exact-shape fast path for the general body, and why (a -DSLK_STAMPS build: the fast path
notes the number of its `return false` in stamp slot 28).   python tools/fast_path_bailouts.py ab/NAME.so [steps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from slkpkg import slk
import scenarios as sc

lib = slk.load_library(os.path.join(ROOT, sys.argv[1]))
slk._lib = lib
lib.slk_debug_set_stamps.argtypes = [C.c_void_p]
B, k = 4096, 8
s = sc.synthetic_msckf(B, k, m=8, seed=0x5EED0000)
f = slk.Msckf(s["mean"], s["P"])
dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
lib.slk_debug_set_stamps(dbg.data_ptr())
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    dbg.zero_()
    f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    f.sync()
    t = dbg.cpu().numpy()
    if it % 5 == 0 or it < 8:
        print(it, "bail-out codes (0 = none):", np.bincount(t[:, 28].astype(int))[:12], "mean-loop passes:", np.bincount(t[:, 20].astype(int))[:6])
