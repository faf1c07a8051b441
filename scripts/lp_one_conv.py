#!/usr/bin/env python3
"""Run one 16-bit stride-1 3x3x3 conv shape repeatedly (for rocprofv3 passes): lp_one_conv.py N D H W Cin Cout dtype reps [old]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import lowp, ops  # noqa: E402

n, d, h, w, cin, cout = (int(v) for v in sys.argv[1:7])
dtype, reps = sys.argv[7], int(sys.argv[8])
if len(sys.argv) > 9 and sys.argv[9] == 'old':
    os.environ['BTS_LP_S1D'] = '0'
code, tdt = lowp.DTYPES[dtype]
D = torch.device('cuda:0')
x = torch.randn((n, d, h, w, cin), device=D).to(tdt)
wt = torch.randn((3, 3, 3, cin, cout), device=D) * (2.0 / (27 * cin)) ** 0.5
b = torch.zeros(cout, device=D)
wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
out = torch.empty((n, d, h, w, cout), dtype=tdt, device=D)
for _ in range(reps):
    lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=out)
torch.cuda.synchronize()
