"""Kernel resource table from a `-Rpass-analysis=kernel-resource-usage` build log (python slam-localization_amd/build.py --force
--verbose 2> log): one row per kernel -- VGPRs, SGPRs, spills, scratch bytes per lane, occupancy."""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    rows, cur = [], None
    for line in open(sys.argv[1]):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    seen = set()
    print("| kernel | VGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        if r["name"] in seen or "Occupancy [waves/SIMD]" not in r:
            continue
        seen.add(r["name"])
        name = demangle(r["name"]).replace("slk::", "").replace("(KArgs)", "").replace("void ", "")
        print(f"| `{name[:80]}` | {r.get('VGPRs', 0)} | {r.get('TotalSGPRs', 0)} | {r.get('SGPRs Spill', 0)} | {r.get('VGPRs Spill', 0)} | "
              f"{r.get('ScratchSize [bytes/lane]', 0)} | {r.get('Occupancy [waves/SIMD]', 0)} |")


if __name__ == "__main__":
    main()
