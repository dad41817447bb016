import time, numpy as np, torch
x = np.random.randn(100_000_000).astype(np.float32)   # 400 MB pageable
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter(); d = torch.from_numpy(x).cuda(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("pageable H2D %.1f GB/s" % (x.nbytes / dt / 1e9))
p = torch.from_numpy(x).pin_memory()
for rep in range(3):
    t = time.perf_counter(); d = p.cuda(non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("pinned H2D %.1f GB/s" % (x.nbytes / dt / 1e9))
y = np.empty_like(x)
for rep in range(3):
    t = time.perf_counter(); np.copyto(y, x); dt = time.perf_counter() - t
    print("host memcpy %.1f GB/s" % (x.nbytes / dt / 1e9))
pp = p.numpy()
for rep in range(3):
    t = time.perf_counter(); np.copyto(pp, x); dt = time.perf_counter() - t
    print("host memcpy into pinned %.1f GB/s" % (x.nbytes / dt / 1e9))
import os; print("cores", os.cpu_count())
