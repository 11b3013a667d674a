"""DIAGNOSTIC: per-phase instruction counts of the Msckf step kernel.

Runs the stamps build (libslk_hip_stamps.so) with the kernel leaving after stamp s for a list of stop
points; under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA ...` every dispatch is
one row of the counter CSV, so the differences between consecutive stop points are the per-phase counts.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pp -- \
        python3 tools/pmc_phases.py
    python3 tools/pmc_phases.py --report /tmp/pp
"""
import argparse
import ctypes as C
import csv
import glob
import importlib.util
import os
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STOPS = [2, 3, 6, 8, 11, 12, 14, 0]
NAMES = {2: "load+predict", 3: "chol1", 6: "Z, moments (S, Pxz)", 8: "gate, S^-1, K", 11: "delta, downdate+chol2",
         12: "mean loop", 14: "rebuild MFMA", 0: "store"}


def report(d, B):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "msckf_step" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[-len(STOPS):]
    prev = collections.defaultdict(float)
    ctrs = sorted(rows[ids[0]])
    print("per FILTER wave-instructions by phase (B=%d)" % B)
    print("%-26s" % "phase" + "".join("%16s" % c.replace("SQ_INSTS_", "") for c in ctrs))
    for s, i in zip(STOPS, ids):
        print("%-26s" % NAMES[s] + "".join("%16.0f" % ((rows[i][c] - prev[c]) / B) for c in ctrs))
        prev = rows[i]
    print("%-26s" % "total" + "".join("%16.0f" % (prev[c] / B) for c in ctrs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--meas", type=int, default=8)
    ap.add_argument("--report", default=None)
    ap.add_argument("--fast", action="store_true", help="stop points / phase names of the exact-shape fast path (slk_step_fast.hpp)")
    ap.add_argument("--lib", default=None, help="a stamps build made elsewhere (build.py --dev NAME --stamps), e.g. ab/NAME.so")
    args = ap.parse_args()
    if args.fast:
        global STOPS, NAMES
        STOPS = [3, 4, 6, 8, 11, 12, 13, 14, 0]
        NAMES = {3: "load", 4: "Z = h(X), zbar", 6: "S | scans", 8: "gate, x b delta | columns", 11: "factor update",
                 12: "mean loop", 13: "correction, split", 14: "rebuild MFMA", 0: "rank-2, store"}
    if args.report:
        return report(args.report, args.batch)
    import torch
    spec = importlib.util.spec_from_file_location("slk_build", os.path.join(ROOT, "slam-localization_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = os.path.join(ROOT, args.lib) if args.lib else b.build(stamps=True)
    from slkpkg import slk
    import scenarios as sc
    lib = slk.load_library(so)
    slk._lib = lib
    lib.slk_debug_set_stamps.argtypes = [C.c_void_p]
    lib.slk_debug_set_stop.argtypes = [C.c_int]
    B, k, m = args.batch, args.clones, args.meas
    s = sc.synthetic_msckf(B, k, m=m)
    dbg = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
    lib.slk_debug_set_stamps(dbg.data_ptr())
    for stop in STOPS:
        f = slk.Msckf(s["mean"], s["P"])          # same state for every stop point
        lib.slk_debug_set_stop(stop)
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        f.sync()


if __name__ == "__main__":
    main()
