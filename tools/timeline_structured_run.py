import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from optimalbeziertrajectorygeneration_amd import _capi as capi, synth
WL = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[WL]
N, n, R = cfg["N"], cfg["n"], cfg["R"]
Y = synth.swarm_control_points(N, 2, n, seed=1234)
statics, pa, pb = synth.config_hull_sweep(WL)
B = N * 2 * (n - 1) + 1
ctx = capi.Context(N, 2, n, R)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_polygons(*synth.pack_polys(statics)); ctx.set_hull_pairs(pa, pb)
d0 = torch.from_numpy(Y).cuda(); dtf = torch.full((B,), 10.0, dtype=torch.float64, device="cuda")
P, L, Ps = ctx.num_pairs, 2 * n + R + 1, len(pa)
f64 = torch.float64
sep = torch.empty((B, P * L), dtype=f64, device="cuda"); sp = torch.empty((B, ctx.len_speed), dtype=f64, device="cuda")
an = torch.empty((B, ctx.len_ang_rate), dtype=f64, device="cuda"); flag = torch.empty((B, Ps), dtype=torch.int32, device="cuda")
p1 = torch.empty((B, Ps, 3), dtype=f64, device="cuda"); p2 = torch.empty((B, Ps, 3), dtype=f64, device="cuda")
dist = torch.empty((B, Ps), dtype=f64, device="cuda"); st = torch.empty((B, Ps), dtype=torch.int32, device="cuda")
import time
for i in range({"C3": 600, "C5": 500}.get(WL, 40)):  # (the timeline, when asked for, is taken at launch OBTG_TIMELINE_AT)
    ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, synth.FD_STEP, dtf.data_ptr(), B, 0.9, sep.data_ptr(), 5.0, True, 1.0,
                                           sp.data_ptr(), an.data_ptr(), flag.data_ptr(), p1.data_ptr(), p2.data_ptr(), dist.data_ptr(), None, st.data_ptr(), 128, 256)
torch.cuda.synchronize()
