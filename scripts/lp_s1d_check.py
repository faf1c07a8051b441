"""Development aid for csrc/lowp_s1d.hip (run on the GPU box): correctness of the LDS-DMA stride-1 conv against torch's fp32 conv on the
same 16-bit-rounded operands and against the register-staged kernel (BTS_LP_S1D=0), then per-shape timings of both kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: F401,E402
from bts_amd import lowp, ops  # noqa: E402

DEV = torch.device('cuda', 0)


def run(code, tdt, x, wp, b, cout, new, out=None):
    os.environ['BTS_LP_S1D'] = '1' if new else '0'
    return lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=out)


def check(n, d, h, w, cin, cout, dtype='bfloat16', slab=False, seed=0):
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.3).to(DEV)
    if slab:
        xb = torch.zeros((n, d, h, w, cin + 16), dtype=tdt, device=DEV)
        xb[..., 16:] = x
        xin = xb[..., 16:]
    else:
        xin = x
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    ref = torch.nn.functional.conv3d(x.float().permute(0, 4, 1, 2, 3), wt.to(tdt).float().permute(4, 3, 0, 1, 2), b, padding=1).permute(0, 2, 3, 4, 1)
    out = None
    if slab:
        ob = torch.full((n, d, h, w, cout + 24), 7.0, dtype=tdt, device=DEV)
        out = ob[..., 8:8 + cout]
    y1 = run(code, tdt, xin, wp, b, cout, True, out=out).float().clone()
    y0 = run(code, tdt, xin, wp, b, cout, False).float()
    torch.cuda.synchronize()
    u = 2.0 ** -8 if dtype == 'bfloat16' else 2.0 ** -11
    e1 = float(((y1 - ref).abs() / (u * ref.abs() + 1e-3)).max())
    e0 = float(((y0 - ref).abs() / (u * ref.abs() + 1e-3)).max())
    same = float((y1 - y0).abs().max())
    ok = e1 <= 1.5
    if slab:
        ok = ok and bool((ob[..., :8] == 7.0).all()) and bool((ob[..., 8 + cout:] == 7.0).all())
    print('%s check n%d %dx%dx%d %d->%d %s slab=%d: new err %.3f (old %.3f) of bound, |new-old| max %.2e' %
          ('ok  ' if ok else 'FAIL', n, d, h, w, cin, cout, dtype, slab, e1, e0, same), flush=True)
    return ok


def bench(n, d, h, w, cin, cout, dtype='bfloat16', reps=5):
    code, tdt = lowp.DTYPES[dtype]
    x = torch.randn((n, d, h, w, cin), device=DEV).to(tdt)
    wt = torch.randn((3, 3, 3, cin, cout), device=DEV) * (2.0 / (27 * cin)) ** 0.5
    b = torch.zeros(cout, device=DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    out = torch.empty((n, d, h, w, cout), dtype=tdt, device=DEV)
    res = []
    for new in (False, True):
        for _ in range(2):
            run(code, tdt, x, wp, b, cout, new, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(code, tdt, x, wp, b, cout, new, out=out)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps)
    fl = 2.0 * 27 * cin * cout * n * d * h * w
    print('bench n%d %3dx%3dx%3d %4d->%4d %s: old %8.1f us %6.0f TF | new %8.1f us %6.0f TF (%.2fx)' %
          (n, d, h, w, cin, cout, dtype[:4], res[0] * 1e3, fl / res[0] / 1e9, res[1] * 1e3, fl / res[1] / 1e9, res[0] / res[1]), flush=True)


if __name__ == '__main__':
    allok = True
    for args in [(1, 4, 8, 32, 16, 32), (1, 8, 8, 32, 32, 32), (2, 8, 16, 64, 32, 32), (1, 4, 4, 32, 32, 64), (2, 9, 13, 40, 48, 64),
                 (1, 8, 8, 16, 64, 128), (2, 10, 12, 20, 32, 64), (1, 16, 16, 16, 256, 256), (1, 32, 32, 32, 64, 32), (1, 12, 20, 70, 16, 96),
                 (1, 40, 48, 40, 128, 128), (8, 16, 16, 16, 64, 64)]:
        allok &= check(*args)
        allok &= check(*args, dtype='float16', slab=True, seed=1)
    print('ALL OK' if allok else 'SOME FAILED', flush=True)
    if '--bench' in sys.argv:
        B = 8
        for shp in [(B, 128, 128, 128, 16, 32), (B, 128, 128, 128, 32, 32), (B, 128, 128, 128, 64, 32), (B, 128, 128, 128, 32, 64),
                    (B, 64, 64, 64, 64, 64), (B, 64, 64, 64, 128, 64), (B, 64, 64, 64, 192, 64), (B, 64, 64, 64, 64, 192),
                    (B, 32, 32, 32, 128, 128), (B, 32, 32, 32, 512, 128), (B, 32, 32, 32, 128, 512),
                    (B, 16, 16, 16, 256, 256), (B, 16, 16, 16, 768, 256),
                    (1, 160, 192, 160, 16, 32), (1, 160, 192, 160, 32, 32), (1, 160, 192, 160, 64, 32),
                    (1, 80, 96, 80, 64, 64), (1, 80, 96, 80, 192, 64), (1, 40, 48, 40, 128, 128), (1, 40, 48, 40, 512, 128),
                    (1, 20, 24, 20, 256, 256), (1, 20, 24, 20, 768, 256)]:
            bench(*shp, dtype='bfloat16' if shp[0] > 1 else 'float16')
