#!/usr/bin/env python3
"""Register / LDS / occupancy figures of the library's kernels as the compiler reports them.

    python tools/kernel_resources.py bern_kernels.hip [substring ...]      (extra flags: -- -ffp-contract=off)

Compiles one translation unit of optimalbeziertrajectorygeneration_amd/csrc with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel whose demangled name contains every substring."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimalbeziertrajectorygeneration_amd", "csrc")


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    unit, subs = args[0], args[1:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-fno-gpu-rdc"] + extra + \
          ["-x", "hip", "-c", os.path.join(CSRC, unit), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    txt = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, universal_newlines=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].split()[0] for b in blocks]
    if not names:
        sys.stderr.write(txt[-3000:])
        raise SystemExit("no kernels reported: compilation failed?")
    dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, universal_newlines=True).stdout.splitlines()
    for b, d in zip(blocks, dem):
        if not all(s in d for s in subs):
            continue

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        print("%-100s VGPR %s AGPR %s spill %s scratch %s occ %s LDS %s" % (
            d.replace("obtg::", "").split("(")[0][:100], g("VGPRs"), g("AGPRs"), g("VGPR Spill"),
            g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
