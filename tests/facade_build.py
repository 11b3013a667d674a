"""Build helper for the C++ header-facade scenario program (tests/cpp/facade_scenarios.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_scenarios.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(OUT_DIR, "facade_scenarios")


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    libdir = os.path.join(ROOT, "slam-localization_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L" + libdir, "-lslk_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def run(args=()):
    out = subprocess.run([build()] + [repr(float(a)) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) < 4:
            continue
        name, r, c = parts[0], int(parts[1]), int(parts[2])
        vals = [float(v) for v in parts[3:]]
        import numpy as np
        a = np.array(vals).reshape(c, r).T
        res[name] = a
    return res
