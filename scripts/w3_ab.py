#!/usr/bin/env python3
"""A/B of the two Winograd forms on the training shapes: python scripts/w3_ab.py  (toggles BTS_W3 per call)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import ops  # noqa: E402

D = torch.device('cuda:0')
K = ops.K3S1
shapes = [(128, 32, 32), (64, 64, 64), (32, 128, 128), (16, 256, 256), (128, 64, 32), (64, 128, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for d, cin, cout in shapes:
    x = torch.randn((1, d, d, d, cin), device=D)
    w = torch.randn((3, 3, 3, cin, cout), device=D) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, device=D)
    res = {}
    for mode in ('0', '1'):
        os.environ['BTS_W3'] = mode
        wp = ops.conv_pack(K, ops.ROLE_FWD, w, cin, cout)   # (the image holds the enabled form's part only)
        y = ops.conv_fwd(K, x, wp, b, cout)
        for _ in range(3):
            ops.conv_fwd(K, x, wp, b, cout, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_fwd(K, x, wp, b, cout, out=y)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 10 * 1e3, y.clone())
    os.environ['BTS_W3'] = '0'
    os.environ['BTS_WINO'] = '0'
    yd = ops.conv_fwd(K, x, wp, b, cout)
    del os.environ['BTS_WINO']
    torch.cuda.synchronize()
    fl = 2.0 * 27 * cin * cout * d ** 3
    t0, t1 = res['0'][0], res['1'][0]
    print('%4d^3 %3d->%3d  2-D %8.1f us (%6.1f TF alg, exec %.3f)   3-D %8.1f us (%6.1f TF alg, exec %.3f)   err vs direct: 2-D %.2e  3-D %.2e  |y| %.2f'
          % (d, cin, cout, t0, fl / t0 / 1e6, fl * 12 / 27 / t0 / 1e6 / 157.3, t1, fl / t1 / 1e6, fl * 8 / 27 / t1 / 1e6 / 157.3,
             float((res['0'][1] - yd).abs().max()), float((res['1'][1] - yd).abs().max()), float(yd.abs().max())))
