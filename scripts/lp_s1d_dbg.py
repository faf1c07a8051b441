#!/usr/bin/env python3
"""lp_s1d under its timing switches (a -DBTS_TIMING_EXPERIMENTS build, BTS_HIP_LIB=...): BTS_S1D_DBG 0 = normal, 1 = no output stores,
2 = no halo traffic, 4 = no matrix instructions, 8 = no fragment reads from LDS, 16 = no weight traffic (sums combine)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
for shape, cin, cout in (((8, 64, 64, 64), 64, 64), ((8, 128, 128, 128), 64, 32), ((8, 32, 32, 32), 128, 128)):
    x = torch.randn(shape + (cin,), device=D).to(tdt)
    wt = torch.randn((3, 3, 3, cin, cout), device=D) * 0.02
    b = torch.zeros(cout, device=D)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    os.environ['BTS_LP_S1Z'] = '0'
    for dbg in ('0', '1', '2', '4', '3', '8', '12', '16', '18', '19', '27', '31'):
        os.environ['BTS_S1D_DBG'] = dbg
        for _ in range(3):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * 27 * cin * cout * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3]
        print(shape, cin, cout, 'dbg', dbg, '%.1f us  %.0f TF' % (ms * 1e3, fl / ms / 1e9), flush=True)
