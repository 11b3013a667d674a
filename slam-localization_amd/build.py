"""Build the HIP shared library in-tree: slam-localization_amd/libslk_hip.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "slk_api.hip")
# translation units of the product build (compiled in parallel, then linked): the host side + most kernels, and the
# explicit instantiations of the largest step kernels
UNITS = [os.path.join(HERE, "csrc", f) for f in ("slk_api.hip", "slk_inst_big.hip", "slk_inst_mid.hip")]
DEPS = [os.path.join(HERE, "csrc", f) for f in ("slk_api.hip", "slk_inst_big.hip", "slk_inst_mid.hip", "slk_kernels.hpp", "slk_usckf.hpp",
                                                "slk_math.hpp", "slk_step_fast.hpp", "slk_usckf_fast.hpp", "slk_general.hpp", "slk_ekf.hpp", "slk_ekf_tiles.hpp", "slk_pose.hpp")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "slk.h"))
OUT = os.path.join(HERE, "libslk_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_dev(name, defines=(), verbose=False, stamps=False):
    """Development build for same-box A/B runs (tools/ab.sh): ab/<name>.so with only the headline instantiation
    (-DSLK_DEV_N60) plus the given defines.  Never the product."""
    out_dir = os.path.join(os.path.dirname(HERE), "ab")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSLK_DEV_N60", "-o", out, SRC]
    cmd[1:1] = ["-D" + d for d in defines]
    if stamps:
        cmd.insert(1, "-DSLK_STAMPS")
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False, stamps=False):
    """stamps=True builds the DIAGNOSTIC variant libslk_hip_stamps.so (in-kernel phase stamps; never
    the product: its run time is not quotable)."""
    out = OUT.replace(".so", "_stamps.so") if stamps else OUT
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in DEPS):
        return out
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    if stamps:
        flags.append("-DSLK_STAMPS")
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    obj_dir = os.path.join(HERE, "_build_stamps" if stamps else "_build")
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, os.path.basename(u).replace(".hip", ".o")) for u in UNITS]
    procs = [subprocess.Popen([hipcc()] + flags + ["-c", u, "-o", o]) for u, o in zip(UNITS, objs)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), "hipcc -c (one of %s)" % ", ".join(os.path.basename(u) for u in UNITS))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--dev" in sys.argv:
        i = sys.argv.index("--dev")
        print(build_dev(sys.argv[i + 1], [a[2:] for a in sys.argv if a.startswith("-D")], verbose="--verbose" in sys.argv,
                        stamps="--stamps" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, stamps="--stamps" in sys.argv))
