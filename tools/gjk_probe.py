"""Time the planar GJK sweep of a configuration: first launch (no trip-count history) and steady state.
    python tools/gjk_probe.py [C3] [reps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = synth.CONFIGS[name]
N, d, n, M = cfg['N'], cfg['d'], cfg['n'], cfg.get('n_poly', 8)
Y = synth.swarm_control_points(N, d, n); B = cfg.get('batch', N * d * (n - 1) + 1)
polys = synth.polygon_obstacles(M); ppts, poff = synth.pack_polys(polys); pa, pb = synth.swarm_pairs(N, M)
dev = torch.device('cuda'); f64 = torch.float64
c = _capi.Context(N, d, n, 0); c.set_polygons(ppts, poff); c.set_hull_pairs(pa, pb)
d0 = torch.from_numpy(Y).to(dev); dY = torch.empty((B, N * d, n + 1), dtype=f64, device=dev)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.fd_batch_dev(d0.data_ptr(), 1, 1.49e-8, B, dY.data_ptr()); torch.cuda.synchronize()
Ps = len(pa)
g_flag = torch.empty((B, Ps), dtype=torch.int32, device=dev); g_p1 = torch.empty((B, Ps, 3), dtype=f64, device=dev)
g_p2 = torch.empty((B, Ps, 3), dtype=f64, device=dev); g_dist = torch.empty((B, Ps), dtype=f64, device=dev)
f = lambda: c.gjk_swarm_dev(dY.data_ptr(), B, g_flag.data_ptr(), g_p1.data_ptr(), g_p2.data_ptr(), g_dist.data_ptr(), None, None, 128, 256)
def timed(k):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
c.set_gjk_history(False); f(); f()
print('%s B=%d pairs=%d  list order (history off): %.4f ms' % (name, B, Ps, timed(reps)))
c.set_gjk_history(True); f(); f()
print('history order, steady state: %.4f ms' % timed(reps))
