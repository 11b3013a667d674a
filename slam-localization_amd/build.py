"""Build the HIP shared library in-tree: slam-localization_amd/libslk_hip.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "slk_api.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("slk_api.hip", "slk_kernels.hpp", "slk_usckf.hpp", "slk_math.hpp", "slk_ekf.hpp", "slk_ekf_tiles.hpp", "slk_pose.hpp")]
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
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out, SRC]
    if stamps:
        cmd.insert(1, "-DSLK_STAMPS")
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--dev" in sys.argv:
        i = sys.argv.index("--dev")
        print(build_dev(sys.argv[i + 1], [a[2:] for a in sys.argv if a.startswith("-D")], verbose="--verbose" in sys.argv,
                        stamps="--stamps" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, stamps="--stamps" in sys.argv))
