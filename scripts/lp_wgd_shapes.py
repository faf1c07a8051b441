#!/usr/bin/env python3
"""lowp_wgd.hip on volumes of equal size and different row pitch (debug aid: how much of the stage time is DRAM locality)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
for shape in ((8, 128, 128, 128), (32, 128, 128, 32), (16, 128, 64, 128), (8, 128, 32, 512)):
    n, d, h, w = shape
    x = torch.randn((n, d, h, w, 32), device=D).to(tdt)
    dy = torch.randn((n, d, h, w, 32), device=D).to(tdt)
    dw = torch.zeros((3, 3, 3, 32, 32), device=D)
    for _ in range(2):
        lowp.conv_bwd_weight(ops.K3S1, code, x, dy, dw, None, accumulate=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lowp.conv_bwd_weight(ops.K3S1, code, x, dy, dw, None, accumulate=False)
    e1.record()
    torch.cuda.synchronize()
    print(shape, '%.3f ms' % (e0.elapsed_time(e1) / 5))
    del x, dy
