"""old (gather) vs new (streaming) 1x1x1 kernel, per shape; run on the GPU box"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
code, tdt = lowp.DTYPES['bfloat16']
for (n, d, cin, cout, acc) in [(8, 128, 16, 32, 0), (8, 128, 64, 32, 0), (8, 128, 32, 64, 1), (8, 128, 32, 32, 1), (8, 64, 64, 64, 0), (8, 64, 192, 64, 0), (8, 64, 64, 192, 1),
                               (8, 32, 512, 128, 0), (8, 16, 768, 256, 0), (1, 160, 64, 32, 0)]:
    x = torch.randn((n, d, d, d, cin), device=DEV).to(tdt)
    wt = torch.randn((1, 1, 1, cin, cout), device=DEV) * 0.1
    wp = lowp.pack(ops.K1, code, wt, cin, cout)
    wpb = lowp.pack(ops.K1, code, wt, cout, cin) if acc else None
    y = torch.zeros((n, d, d, d, cout), dtype=tdt, device=DEV)
    res = []
    for new in ('0', '1'):
        os.environ['BTS_LP_K1'] = new
        def run():
            if acc:
                lowp.conv_bwd_data(ops.K1, code, x, wp, y, True) if False else lowp.conv(ops.K1, code, tdt, x, wp, None, cout, out=y)
            else:
                lowp.conv(ops.K1, code, tdt, x, wp, None, cout, out=y)
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    gb = n * d ** 3 * (cin + cout) * 2 / 1e9
    print('k1 n%d %3d^3 %4d->%4d: old %8.1f us (%5.2f TB/s) | new %8.1f us (%5.2f TB/s)' % (n, d, cin, cout, res[0], gb / res[0] * 1e3, res[1], gb / res[1] * 1e3), flush=True)
