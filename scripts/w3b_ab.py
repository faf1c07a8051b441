#!/usr/bin/env python3
"""A/B of w3_kernel (default T, and T = 1) against the two-waves-per-SIMD experiment w3b_kernel (BTS_W3B=1, one item per workgroup)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import ops  # noqa: E402

D = torch.device('cuda:0')
K = ops.K3S1
shapes = [(128, 32, 32), (64, 64, 64), (32, 128, 128), (128, 64, 32), (64, 128, 64)]
for d, cin, cout in shapes:
    x = torch.randn((1, d, d, d, cin), device=D)
    w = torch.randn((3, 3, 3, cin, cout), device=D) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, device=D)
    wp = ops.conv_pack(K, ops.ROLE_FWD, w, cin, cout)
    res = {}
    for mode, env in (('w3', {}), ('w3 T=1', {'BTS_W3_T': '1'}), ('w3b', {'BTS_W3B': '1'})):
        for k in ('BTS_W3_T', 'BTS_W3B'):
            os.environ.pop(k, None)
        os.environ.update(env)
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
    for k in ('BTS_W3_T', 'BTS_W3B'):
        os.environ.pop(k, None)
    print('%4d^3 %3d->%3d  ' % (d, cin, cout) + '  '.join('%s %7.1f us' % (m, r[0]) for m, r in res.items()) +
          '   |w3b - w3| max %.2e' % float((res['w3b'][1] - res['w3'][1]).abs().max()))
