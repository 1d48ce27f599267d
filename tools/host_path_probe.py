"""PCIe-inclusive rates of the host-buffer entry points (what an SLSQP callback pays)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, d, n = 64, 2, 10
Y = synth.swarm_control_points(N, d, n)
ctx = _capi.Context(N, d, n, 0)
for B in (1, 1153):
    Yb = synth.fd_batch(Y, B=B)
    tf = np.full(B, 10.0)
    for _ in range(3):
        ctx.temporal_sep(Yb, 0.9); ctx.speed(Yb, tf, 5.0, True); ctx.ang_rate(Yb, tf, 1.0)
    t = time.perf_counter(); K = 50 if B == 1 else 5
    for _ in range(K):
        ctx.temporal_sep(Yb, 0.9); ctx.speed(Yb, tf, 5.0, True); ctx.ang_rate(Yb, tf, 1.0)
    dt = (time.perf_counter() - t) / K
    print("B=%d: %.3f ms per (tsep+speed+ang) host call set -> %.1f evals/s" % (B, dt * 1e3, B / dt))
