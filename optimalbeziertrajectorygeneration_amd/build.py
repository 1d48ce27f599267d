"""In-tree build of libobtg_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m optimalbeziertrajectorygeneration_amd.build [--force]

The shared library lands next to this file (git-ignored, but it travels with the gpurun
snapshot).  gjk_kernels.hip is compiled with -ffp-contract=off (bit-exact branch decisions);
the Bernstein kernels keep the default contraction (parity bound there is 1e-9 relative).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libobtg_hip.so")
ARCH = "gfx950"

UNITS = [
    # (source, extra flags)
    ("bern_kernels.hip", []),
    ("gjk_kernels.hip", ["-ffp-contract=off"]),
    ("capi.cpp", []),
    ("tables.cpp", []),
    ("comm.cpp", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-fast-math",
          "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fgpu-rdc" if False else "-fno-gpu-rdc"]
HEADERS = ["obtg_internal.h", "gjk_device.h", "gjk_true.h", "bern_device.h", "libm_pow2.h", "libm_pow2_tables.h", os.path.join("..", "..", "include", "obtg.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    objs = []
    procs = []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _mtime(o) < max(_mtime(s), hdr_time, _mtime(__file__)):
            cmd = [_hipcc()] + COMMON + extra + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("libobtg_hip.so: compilation failed")
    if force or procs or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
