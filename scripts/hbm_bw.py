"""Practical HBM bandwidth of the box through plain torch kernels (copy / fill / read / 2-read-1-write): the ceiling the
GroupNorm / gate / pooling passes are held against in DESIGN.md (measured on the pool's MI355X: 5.5 / 6.8 / 3.9 / 5.8 TB/s)."""
import torch, time
x = torch.empty(256*1024*1024, dtype=torch.float32, device='cuda')   # 1 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (('copy', lambda: y.copy_(x), 2*x.numel()*4), ('fill', lambda: y.fill_(1.0), x.numel()*4), ('sum-read', lambda: x.sum(), x.numel()*4), ('add3', lambda: torch.add(x, y, out=y), 3*x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print('%-9s %.2f TB/s' % (name, nbytes/dt/1e12))
