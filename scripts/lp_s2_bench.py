#!/usr/bin/env python3
"""stride-2 3x3x3 conv (downsample.py:30-48) on 16-bit storage, per shape: plain gather kernel vs the whole-row-load form"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
code, tdt = lowp.DTYPES['bfloat16']
CASES = [(1, (80, 96, 80), 64, 64, 128), (1, (40, 48, 40), 128, 128, 256), (1, (160, 192, 160), 32, 32, 64), (8, 128, 32, 32, 64), (8, 128, 32, 32, 32), (8, 128, 32, 64, 64), (8, 64, 64, 64, 128), (8, 64, 64, 64, 64), (8, 32, 128, 128, 256), (1, 160, 32, 32, 64)]
for (n, d, cin, cout, ldx) in CASES:
    dims = d if isinstance(d, tuple) else (d, d, d)
    slab = torch.randn((n,) + dims + (ldx,), device=DEV).to(tdt)
    x = slab[..., :cin]
    wt = torch.randn((3, 3, 3, cin, cout), device=DEV) * 0.05
    b = torch.zeros(cout, device=DEV)
    wp = lowp.pack(ops.K3S2, code, wt, cin, cout)
    res, outs = [], []
    for q in ('0', '1'):
        os.environ['BTS_LP_GATHERQ'] = q
        run = lambda: lowp.conv(ops.K3S2, code, tdt, x, wp, b, cout)
        for _ in range(2):
            y = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
        outs.append(y.float())
    vin = dims[0] * dims[1] * dims[2]
    fl = 2.0 * 27 * cin * cout * n * vin / 8
    gb = (n * vin * cin + n * vin // 8 * cout) * 2 / 1e9
    print('s2 n%d %s %3d->%3d ldx %3d: plain %7.1f us (%4.0f TF) | whole rows %7.1f us (%4.0f TF, %.2f TB/s)  maxdiff %.3g' %
          (n, 'x'.join(map(str, dims)), cin, cout, ldx, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, gb / res[1] * 1e3, float((outs[0] - outs[1]).abs().max())), flush=True)
