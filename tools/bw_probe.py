"""Achievable HBM write / copy bandwidth on this box with stock kernels (context for roofline fractions)."""
import torch, time
n = 390_507_264 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / k
w = t(lambda: x.fill_(1.0)); c = t(lambda: y.copy_(x)); z = t(lambda: x.zero_())
print("fill  %.1f us  %.2f TB/s (write-only)" % (w * 1e6, n * 8 / w / 1e12))
print("zero  %.1f us  %.2f TB/s (write-only)" % (z * 1e6, n * 8 / z / 1e12))
print("copy  %.1f us  %.2f TB/s (read+write bytes)" % (c * 1e6, 2 * n * 8 / c / 1e12))
