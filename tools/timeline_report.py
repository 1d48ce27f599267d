"""Report on a workgroup timeline written by libobtg_hip.so (OBTG_TIMELINE=<file>, one-launch pair sweep):
how many workgroups are resident over time, how long they last, what the tail of the launch looks like."""
import sys
import numpy as np

path = sys.argv[1]
hdr = open(path).readline().split()
sweep_blocks = int(hdr[hdr.index("sweep_blocks") + 1])
ts_blocks = int(hdr[hdr.index("ts_blocks") + 1]) if "ts_blocks" in hdr else 0
d = np.loadtxt(path, dtype=np.int64, ndmin=2)
blk, t0, t1, hw, xcc = d.T
ok = t1 > 0
us = 0.01
t0 = t0 * us; t1 = t1 * us
span = t1[ok].max()
sw = ok & (blk < sweep_blocks)
tso = ok & (blk >= sweep_blocks) & (blk < sweep_blocks + ts_blocks)
dy = ok & (blk >= sweep_blocks + ts_blocks)
if tso.any():
    print("separation-only workgroups %d: mean %.1f us, first starts %.1f, last ends %.1f" % (tso.sum(), (t1 - t0)[tso].mean(), t0[tso].min(), t1[tso].max()))
print(" ".join(hdr))
print("launch span %.1f us; sweep workgroups %d (duration mean %.1f, p10 %.1f, p90 %.1f, max %.1f us); dynamics groups %d (mean %.1f us)"
      % (span, sw.sum(), (t1 - t0)[sw].mean(), np.percentile((t1 - t0)[sw], 10), np.percentile((t1 - t0)[sw], 90),
         (t1 - t0)[sw].max(), dy.sum(), (t1 - t0)[dy].mean() if dy.any() else 0.0))
print("last sweep workgroup starts at %.1f us, last ends at %.1f; first dynamics group starts %.1f, last ends %.1f"
      % (t0[sw].max(), t1[sw].max(), t0[dy].min() if dy.any() else 0, t1[dy].max() if dy.any() else 0))
step = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
print(" t(us)  resident sweep wgs  resident dynamics groups  sweep wgs started in bin")
for a in np.arange(0.0, span + step, step):
    m = a + step / 2
    print("%6.1f  %6d  %6d  %6d   sep-only %d" % (a, (sw & (t0 <= m) & (t1 > m)).sum(), (dy & (t0 <= m) & (t1 > m)).sum(),
                                    (sw & (t0 >= a) & (t0 < a + step)).sum(), (tso & (t0 <= m) & (t1 > m)).sum()))
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
n_per = np.bincount(np.unique(key[sw], return_inverse=True)[1])
print("CUs seen %d; sweep workgroups per CU: min %d max %d" % (len(n_per), n_per.min(), n_per.max()))
for a in np.arange(0.0, span, 20.0):
    m = sw & (t0 >= a) & (t0 < a + 20.0)
    if m.any():
        print("started in [%5.0f, %5.0f) us: %5d sweep wgs, mean duration %.1f us" % (a, a + 20, m.sum(), (t1 - t0)[m].mean()))
