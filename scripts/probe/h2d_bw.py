import torch, time
x = torch.empty(15_360_000 // 8, dtype=torch.float64).pin_memory()
y = torch.empty_like(x, device="cuda")
for _ in range(3): y.copy_(x, non_blocking=True)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(20): y.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/20
print("pinned H2D 15.36 MB: %.3f ms  %.1f GB/s" % (dt*1e3, 15.36e6/dt/1e9))
