"""Build helper for the C++ header-facade scenario program (tests/cpp/facade_scenarios.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_scenarios.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(OUT_DIR, "facade_scenarios")


def build(name="facade_scenarios", link_hip=True, std="c++17"):
    """Compile tests/cpp/<name>.cpp against the header facade.  The programs that paste the reference's model functions
    (they use the `register` keyword, gone in C++17) are built as C++14."""
    os.makedirs(OUT_DIR, exist_ok=True)
    libdir = os.path.join(ROOT, "slam-localization_amd")
    exe = os.path.join(OUT_DIR, name)
    cmd = ["g++", "-std=" + std, "-O1", "-Wall", "-Werror", "-Wno-sign-compare", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe]
    if link_hip:
        cmd += ["-L" + libdir, "-lslk_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def run(args=(), name="facade_scenarios", std="c++17"):
    out = subprocess.run([build(name, std=std)] + [repr(float(a)) for a in args], capture_output=True, text=True, timeout=300)
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
