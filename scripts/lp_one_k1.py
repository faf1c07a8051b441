#!/usr/bin/env python3
"""a few launches of the streaming 1x1x1 conv at one shape (for the counter passes of scripts/pmc_mem.sh): lp_one_k1.py n d cin cout"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
n, d, cin, cout = (int(v) for v in sys.argv[1:5])
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
x = torch.randn((n, d, d, d, cin), device=D).to(tdt)
wt = torch.randn((1, 1, 1, cin, cout), device=D) * 0.1
wp = lowp.pack(ops.K1, code, wt, cin, cout)
y = torch.zeros((n, d, d, d, cout), dtype=tdt, device=D)
for _ in range(4):
    lowp.conv(ops.K1, code, tdt, x, wp, None, cout, out=y)
torch.cuda.synchronize()
print('algorithmic bytes per launch: %.1f MB' % (n * d ** 3 * (cin + cout) * 2 / 1e6))
