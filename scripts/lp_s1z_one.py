#!/usr/bin/env python3
"""one 32->32 @128^3 x8 launch of the stride-1 conv, repeated (for rocprofv3 passes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
x = torch.randn((8, 128, 128, 128, 32), device=D).to(tdt)
wt = torch.randn((3, 3, 3, 32, 32), device=D) * 0.05
b = torch.zeros(32, device=D)
wp = lowp.pack(ops.K3S1, code, wt, 32, 32)
y = torch.empty((8, 128, 128, 128, 32), dtype=tdt, device=D)
for _ in range(6):
    lowp.conv(ops.K3S1, code, tdt, x, wp, b, 32, out=y)
torch.cuda.synchronize()
