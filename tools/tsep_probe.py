"""Time the temporal-separation sweep alone: python tools/tsep_probe.py [C3]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
cfg = synth.CONFIGS[name]
N, d, n, R = cfg['N'], cfg['d'], cfg['n'], cfg['R']
Y = synth.swarm_control_points(N, d, n); B = int(sys.argv[2]) if len(sys.argv) > 2 else N * d * (n - 1) + 1
dev = torch.device('cuda'); f64 = torch.float64
c = _capi.Context(N, d, n, R); c.set_stream(torch.cuda.current_stream().cuda_stream)
P, L = c.num_pairs, 2 * n + R + 1
d0 = torch.from_numpy(Y).to(dev); dY = torch.empty((B, N * d, n + 1), dtype=f64, device=dev)
c.fd_batch_dev(d0.data_ptr(), 1, synth.FD_STEP, B, dY.data_ptr())
out = torch.empty((B, P * L), dtype=f64, device=dev)
f = lambda: c.temporal_sep_dev(dY.data_ptr(), B, 0.9, out.data_ptr())
for _ in range(20): f()
torch.cuda.synchronize(); t = time.perf_counter()
reps = 200
for _ in range(reps): f()
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / reps * 1e3
print('%s B=%d: temporal sweep %.4f ms, %.2f TB/s of output' % (name, B, ms, B * P * L * 8 / ms / 1e9))
