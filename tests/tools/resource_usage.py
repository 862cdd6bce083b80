#!/usr/bin/env python
"""Register / scratch / LDS usage of every kernel in csrc/ebm_kernels.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950 without a GPU).

    python tests/tools/resource_usage.py [-D...] > profiles/rNN_resource_usage.txt

One line per kernel instantiation: name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy.
Exit status 1 if any kernel of the shipped library uses scratch (register spills)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.environ.get("EBM_KERNEL_SRC") or os.path.join(ROOT, "energybalancemodel.jl_amd", "csrc", "ebm_kernels.hip")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True, check=True).stdout.splitlines()
    return [re.sub(r"\(ebm::\w+\)|\(ebm::\w+ const\)|\(.*\)$", "", n).replace("void ebm::", "") for n in out]


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
           "-c", SRC, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|"
                      r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" ")[0]] = val
    if not rows:
        sys.stderr.write(err)
        return 2
    names = demangle([r["name"] for r in rows])
    bad = 0
    print(f"# {' '.join(cmd[1:])}")
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s}")
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        sc = int(r.get("ScratchSize", 0))
        bad += sc > 0
        print(f"{n:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{sc:>15d} {r.get('Occupancy', '?'):>11s}")
    print(f"# kernels: {len(rows)}, with scratch: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
