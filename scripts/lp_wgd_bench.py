#!/usr/bin/env python3
"""Stride-1 3x3x3 weight gradient with few output channels on 16-bit operands: the streaming kernel (lowp_wgd.hip) against lowp.hip's
general one (BTS_LP_WGD=0) on the CLI model's 128^3 layers at batch 8.  lp_wgd_bench.py [N] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import lowp, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bfloat16'
code, tdt = lowp.DTYPES[dtype]
D = torch.device('cuda:0')
for d, cin, cout in ((128, 32, 32), (128, 64, 32), (128, 16, 32), (64, 64, 64), (64, 128, 64), (32, 128, 128), (32, 256, 128)):
    x = torch.randn((N, d, d, d, cin), device=D).to(tdt)
    dy = torch.randn((N, d, d, d, cout), device=D).to(tdt)
    dw = torch.zeros((3, 3, 3, cin, cout), device=D)
    res = {}
    for mode in ('1', '0'):
        os.environ['BTS_LP_WGD'] = mode
        for _ in range(2):
            lowp.conv_bwd_weight(ops.K3S1, code, x, dy, dw, None, accumulate=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lowp.conv_bwd_weight(ops.K3S1, code, x, dy, dw, None, accumulate=False)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 5, dw.clone())
    fl = 2.0 * 27 * cin * cout * N * d ** 3
    gb = (cin + cout) * 2.0 * N * d ** 3 / 1e9
    rel = float((res['1'][1] - res['0'][1]).norm() / res['0'][1].norm())
    print('%d^3 x%d %3d->%3d: streaming %.3f ms (%.0f TF, %.2f TB/s of compulsory bytes)  general %.3f ms (%.0f TF)  rel diff %.1e' %
          (d, N, cin, cout, res['1'][0], fl / res['1'][0] / 1e9, gb / res['1'][0], res['0'][0], fl / res['0'][0] / 1e9, rel))
    del x, dy
