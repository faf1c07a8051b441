"""old (8 gather launches) vs new (lp_up_kernel) transposed-form conv, per shape; run on the GPU box"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
for (n, d, cin, cout, dt) in [(8, 64, 64, 32, 'bfloat16'), (8, 32, 128, 64, 'bfloat16'), (8, 16, 256, 128, 'bfloat16'), (8, 64, 32, 16, 'bfloat16'),
                              (1, 80, 64, 32, 'float16'), (1, 40, 128, 64, 'float16'), (1, 20, 256, 128, 'float16')]:
    code, tdt = lowp.DTYPES[dt]
    if dt == 'float16' and d in (80, 40, 20):
        shp = (n, d, d * 12 // 10, d, cin)
    else:
        shp = (n, d, d, d, cin)
    x = torch.randn(shp, device=DEV).to(tdt)
    wt = torch.randn((3, 3, 3, cout, cin), device=DEV) * 0.05
    wp = lowp.pack(ops.K3S2T, code, wt, cin, cout)
    y = torch.zeros((shp[0], 2 * shp[1], 2 * shp[2], 2 * shp[3], cout), dtype=tdt, device=DEV)
    res = []
    for new in ('0', '1'):
        os.environ['BTS_LP_UP'] = new
        for _ in range(2): lowp.conv(ops.K3S2T, code, tdt, x, wp, None, cout, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lowp.conv(ops.K3S2T, code, tdt, x, wp, None, cout, out=y)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * 27 * cin * cout * x.numel() / cin
    print('up n%d %s %4d->%4d: old %8.1f us (%5.0f TF) | new %8.1f us (%5.0f TF)  %.2fx' % (n, tuple(shp[1:4]), cin, cout, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[0] / res[1]), flush=True)
