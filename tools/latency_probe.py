"""Per-call latency of the host-buffer entry points at the sizes the reference's own examples use
(one evaluation row per call, as SciPy's SLSQP issues them).  python tools/latency_probe.py"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import _capi, synth


def probe(N, d, n, R, calls=3000):
    ctx = _capi.Context(N, d, n, R)
    Y = synth.swarm_control_points(N, d, n, seed=1)
    tf = np.array([10.0])
    res = {}
    fams = [("temporal_sep", lambda: ctx.temporal_sep(Y, 0.9))] if N > 1 else []
    fams += [("speed", lambda: ctx.speed(Y, tf, 5.0, True))]
    if d == 2:
        fams += [("ang_rate", lambda: ctx.ang_rate(Y, tf, 1.0))]
    for name, fn in fams:
        for _ in range(200):
            fn()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        res[name] = round((time.perf_counter() - t0) / calls * 1e6, 2)
    ctx.close()
    return res


if __name__ == "__main__":
    out = {}
    for tag, shape in (("example1 N=2 d=2 n=10 R=30", (2, 2, 10, 30)), ("C2 N=8 d=3 n=10 R=0", (8, 3, 10, 0)),
                       ("swarm N=36 d=3 n=5 R=0", (36, 3, 5, 0)), ("C3 row N=64 d=2 n=10 R=0", (64, 2, 10, 0))):
        out[tag] = probe(*shape)
        print(tag, out[tag], flush=True)
    print(json.dumps({"unit": "us per call, B = 1", "latency": out}))


def raw_probe(N=2, d=2, n=10, R=30, calls=5000):
    """The same call without the Python wrapper (pre-allocated arrays, bare ctypes): what the C side costs."""
    import ctypes as C
    ctx = _capi.Context(N, d, n, R)
    lib = ctx._lib
    Y = np.ascontiguousarray(synth.swarm_control_points(N, d, n, seed=1))
    out = _capi.pinned_empty((1, ctx.len_temporal_sep))
    outp = np.empty((1, ctx.len_temporal_sep))
    res = {}
    for tag, o in (("pinned result", out), ("pageable result", outp)):
        yp, op = Y.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p)
        for _ in range(200):
            lib.obtg_temporal_sep(ctx._h, yp, 1, C.c_double(0.9), op)
        t0 = time.perf_counter()
        for _ in range(calls):
            lib.obtg_temporal_sep(ctx._h, yp, 1, C.c_double(0.9), op)
        res[tag] = round((time.perf_counter() - t0) / calls * 1e6, 2)
    t0 = time.perf_counter()
    for _ in range(calls):
        lib.obtg_len_temporal_sep(ctx._h)
    res["bare ctypes call"] = round((time.perf_counter() - t0) / calls * 1e6, 2)
    ctx.close()
    return res


if __name__ == "__main__":
    print("raw obtg_temporal_sep, example1 shape:", raw_probe())
