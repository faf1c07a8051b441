#!/usr/bin/env python3
"""Is lp_s1d power-limited?  The same 64 -> 64 convolution at 8 x 64^3 on random operands, on an all-zero input, and on all-zero input
AND weights (same instructions, same memory traffic, no toggling in the matrix pipe's datapath): if the part lowers its clock for
power under this kernel, the zero runs are faster."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
os.environ['BTS_LP_S1Z'] = '0'
for shape, cin, cout in (((8, 64, 64, 64), 64, 64), ((8, 32, 32, 32), 128, 128), ((8, 128, 128, 128), 32, 32)):
    for name, xs, ws in (('random', 1.0, 0.02), ('zero input', 0.0, 0.02), ('zero input and weights', 0.0, 0.0)):
        if cin == 32:
            os.environ['BTS_LP_S1Z'] = '1'
        x = (torch.randn(shape + (cin,), device=D) * xs).to(tdt)
        wt = torch.randn((3, 3, 3, cin, cout), device=D) * ws
        b = torch.zeros(cout, device=D)
        wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
        for _ in range(5):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * 27 * cin * cout * shape[0] * shape[1] * shape[2] * shape[3]
        print(shape, '%d -> %d %-24s %.1f us  %.0f TF' % (cin, cout, name, ms * 1e3, fl / ms / 1e9), flush=True)
