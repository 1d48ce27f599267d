"""Time the C3 one-launch step of ANY build of libobtg_hip.so through the entry points that exist since round 2
(ctypes, no binding table): for A/B runs of library builds on one box.
    python tools/ab_c3_step.py LIB [LIB ...]   ->  ms per step (HIP events around 300 steps), three interleaved repetitions"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import synth

vp, i32, f64 = C.c_void_p, C.c_int, C.c_double


def setup(path):
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    lib.obtg_ctx_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, vp, i32]
    lib.obtg_ctx_set_stream.argtypes = [vp, vp]
    lib.obtg_ctx_set_polygons.argtypes = [vp, vp, i32, vp, i32]
    lib.obtg_ctx_set_hull_pairs.argtypes = [vp, vp, vp, i32]
    lib.obtg_fd_view_begin.argtypes = [vp, vp, i32, f64, i32]
    lib.obtg_fd_view_end.argtypes = [vp]
    lib.obtg_constraint_sweep_dev.argtypes = [vp, vp, vp, i32, f64, vp, f64, i32, f64, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    cfg = synth.CONFIGS["C3"]
    N, d, n = cfg["N"], cfg["d"], cfg["n"]
    Y = synth.swarm_control_points(N, d, n)
    statics, pa, pb = synth.config_hull_sweep("C3")
    pts, off = synth.pack_polys(statics)
    h = vp()
    assert lib.obtg_ctx_create(C.byref(h), N, d, n, 0, 0, None, 0) == 0
    lib.obtg_ctx_set_stream(h, vp(torch.cuda.current_stream().cuda_stream))
    assert lib.obtg_ctx_set_polygons(h, pts.ctypes.data_as(vp), len(pts), off.ctypes.data_as(vp), len(off) - 1) == 0
    assert lib.obtg_ctx_set_hull_pairs(h, pa.ctypes.data_as(vp), pb.ctypes.data_as(vp), len(pa)) == 0
    B = N * d * (n - 1) + 1
    P, L, Ps = N * (N - 1) // 2, 2 * n + 1, len(pa)
    t = dict(d0=torch.from_numpy(Y).cuda(), tf=torch.full((B,), 10.0, dtype=torch.float64, device="cuda"),
             sep=torch.empty((B, P * L), dtype=torch.float64, device="cuda"), sp=torch.empty((B, N * L), dtype=torch.float64, device="cuda"),
             an=torch.empty((B, N * (4 * n + 1)), dtype=torch.float64, device="cuda"), flag=torch.empty((B, Ps), dtype=torch.int32, device="cuda"),
             p1=torch.empty((B, Ps, 3), dtype=torch.float64, device="cuda"), p2=torch.empty((B, Ps, 3), dtype=torch.float64, device="cuda"),
             dist=torch.empty((B, Ps), dtype=torch.float64, device="cuda"), st=torch.empty((B, Ps), dtype=torch.int32, device="cuda"))

    def step():
        assert lib.obtg_fd_view_begin(h, vp(t["d0"].data_ptr()), 1, synth.FD_STEP, B) == 0
        rc = lib.obtg_constraint_sweep_dev(h, None, vp(t["tf"].data_ptr()), B, 0.9, vp(t["sep"].data_ptr()), 5.0, 1, 1.0, vp(t["sp"].data_ptr()),
                                           vp(t["an"].data_ptr()), 128, 256, vp(t["flag"].data_ptr()), vp(t["p1"].data_ptr()),
                                           vp(t["p2"].data_ptr()), vp(t["dist"].data_ptr()), None, vp(t["st"].data_ptr()))
        assert rc == 0, rc
        lib.obtg_fd_view_end(h)
    return step, t


def main():
    libs = sys.argv[1:]
    stream = torch.cuda.Stream()            # (the default stream's handle is 0, which the library reads as "your own stream")
    torch.cuda.set_stream(stream)
    steps = [setup(os.path.abspath(p)) for p in libs]
    for s, _ in steps:
        for _ in range(3000):
            s()
    torch.cuda.synchronize()
    for rep in range(3):
        for path, (s, _) in zip(libs, steps):
            for _ in range(30):
                s()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(300):
                s()
            b.record()
            torch.cuda.synchronize()
            print("rep %d  %-60s %.4f ms per step" % (rep, path, a.elapsed_time(b) / 300), flush=True)
    ref = steps[0][1]
    for path, (_, t) in zip(libs[1:], steps[1:]):
        same = all(torch.equal(ref[k].view(torch.uint8), t[k].view(torch.uint8)) for k in ("sep", "sp", "an", "flag", "p1", "p2", "dist", "st"))
        print("outputs of %s identical to those of %s: %s" % (path, libs[0], same))


if __name__ == "__main__":
    main()
