#!/usr/bin/env python3
"""a few launches of the 16-bit stride-2 conv at one shape (for the counter passes of scripts/pmc_prog.sh / pmc_mem.sh):
lp_one_s2.py n d cin cout ldx"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
n, d, cin, cout, ldx = (int(v) for v in sys.argv[1:6])
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
slab = torch.randn((n, d, d, d, ldx), device=D).to(tdt)
x = slab[..., :cin]
wt = torch.randn((3, 3, 3, cin, cout), device=D) * 0.05
wp = lowp.pack(ops.K3S2, code, wt, cin, cout)
b = torch.zeros(cout, device=D)
for _ in range(4):
    lowp.conv(ops.K3S2, code, tdt, x, wp, b, cout)
torch.cuda.synchronize()
