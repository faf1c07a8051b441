#!/usr/bin/env python3
"""host enqueue time of one 16-bit forward (160x192x160, fp16) against its GPU time: is the forward launch-bound?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
from bts_amd.model import Model
os.environ['BTS_STEP_FENCE'] = '0'
m = Model(base_filters=32, reduction=8, depth=4, groups=8)
m.build((1, 128, 128, 128, 2))
x = torch.randn(1, 160, 192, 160, 2).cuda()
run = lowp.LowPrecisionForward(m, 'float16')
for _ in range(5):
    run(x)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    run(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms per forward; total %.2f ms per forward (GPU-bound if total > enqueue)' % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
