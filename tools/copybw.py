import torch, time
n = 518**3
a = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("copy 1.11GB: %.3f ms, %.2f TB/s (read+write)" % (dt * 1e3, 2 * n * 8 / dt / 1e12))
t0 = time.perf_counter()
for _ in range(20): s = a.sum()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("sum  1.11GB: %.3f ms, %.2f TB/s (read)" % (dt * 1e3, n * 8 / dt / 1e12))
t0 = time.perf_counter()
for _ in range(20): b.fill_(1.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("fill 1.11GB: %.3f ms, %.2f TB/s (write)" % (dt * 1e3, n * 8 / dt / 1e12))
