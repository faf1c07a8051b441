#!/usr/bin/env python3
"""Stride-1 3x3x3 conv with few channels on 16-bit storage: the z-marching kernel (lowp_s1z.hip) against the tiled LDS-DMA kernel
(BTS_LP_S1Z=0) on the CLI model's top-level layers.  lp_s1z_bench.py [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import lowp, ops  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bfloat16'
code, tdt = lowp.DTYPES[dtype]
D = torch.device('cuda:0')
for shape, cin, cout, ldx in (((8, 128, 128, 128), 32, 32, 32), ((8, 128, 128, 128), 16, 32, 16), ((8, 128, 128, 128), 32, 32, 64),
                              ((1, 160, 192, 160), 32, 32, 32), ((1, 160, 192, 160), 16, 32, 16), ((8, 64, 64, 64), 32, 32, 32)):
    n, d, h, w = shape
    xs = torch.randn((n, d, h, w, ldx), device=D).to(tdt)
    x = xs[..., :cin]
    wt = torch.randn((3, 3, 3, cin, cout), device=D) * (2.0 / (27 * cin)) ** 0.5
    b = torch.zeros(cout, device=D)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    y = torch.empty((n, d, h, w, cout), dtype=tdt, device=D)
    res = {}
    for mode in ('1', '0'):
        os.environ['BTS_LP_S1Z'] = mode
        for _ in range(2):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=y)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 5, y.clone())
    fl = 2.0 * 27 * cin * cout * n * d * h * w
    gb = (cin + cout) * 2.0 * n * d * h * w / 1e9
    diff = float((res['1'][1].float() - res['0'][1].float()).abs().max())
    print('%s %2d->%2d (ld %d): streaming %.3f ms (%.0f TF, %.2f TB/s compulsory)  tiled %.3f ms (%.0f TF)  max diff %.2e' %
          (shape, cin, cout, ldx, res['1'][0], fl / res['1'][0] / 1e9, gb / res['1'][0], res['0'][0], fl / res['0'][0] / 1e9, diff))
    del xs, y
