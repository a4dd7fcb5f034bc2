import torch, time
x = torch.randn(54, 3, 600, 1000).pin_memory()
d = torch.empty_like(x, device='cuda')
for _ in range(2): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
print('H2D 54 x 7.2 MB fp32 pinned: %.2f ms  (%.1f GB/s)' % (t * 1e3, x.numel() * 4 / t / 1e9))
