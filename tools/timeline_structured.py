"""Timeline report of the structured finite-difference step (kinds S / F / G / D of k_step_fd_structured: of every 16 block ids the header's `pat` says which kind each is: 0 S, 1 F, 2 G, 3 D)."""
import sys
import numpy as np
path = sys.argv[1]
hdr = open(path).readline().split()
print(" ".join(hdr))
pat = np.array([int(ch) for ch in hdr[hdr.index("pat") + 1]])
d = np.loadtxt(path, dtype=np.int64, ndmin=2)
blk, t0, t1 = d[:, 0], d[:, 1] * 0.01, d[:, 2] * 0.01
kind = pat[(blk + (blk >> 4)) & 15]      # the pattern is rotated by one slot per group of 16 (XCD balance)
live = (t1 - t0) > 0.5                      # blocks beyond their kind's count return at once
print("span %.1f us" % t1.max())
for k, name in enumerate("SFGD"):
    m = (kind == k) & live
    if m.any():
        print("%s: %5d workgroups, duration mean %.1f max %.1f us; first starts %.1f, last ends %.1f" %
              (name, m.sum(), (t1 - t0)[m].mean(), (t1 - t0)[m].max(), t0[m].min(), t1[m].max()))
step = 10.0
for a in np.arange(0, t1.max() + step, step):
    mid = a + step / 2
    print("%6.1f " % a + "  ".join("%s %4d" % (name, ((kind == k) & live & (t0 <= mid) & (t1 > mid)).sum()) for k, name in enumerate("SFGD")))
