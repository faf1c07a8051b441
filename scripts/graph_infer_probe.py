#!/usr/bin/env python3
"""Does replaying the 16-bit forward as a hipGraph (torch.cuda.CUDAGraph around the same host code) beat enqueueing its ~170 launches
one by one?  160x192x160, fp16 storage."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['BTS_STEP_FENCE'] = '0'
import bts_amd  # noqa
from bts_amd import lowp, ops
from bts_amd.model import Model
m = Model(base_filters=32, reduction=8, depth=4, groups=8)
m.build((1, 128, 128, 128, 2))
x = torch.randn(1, 160, 192, 160, 2).cuda()
run = lowp.LowPrecisionForward(m, 'float16')
def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for side in (False, True):
    ops.enable_side_streams(side)
    y_ref = run(x).clone()
    print('side streams %s: eager %.3f ms' % (side, timeit(lambda: run(x))), flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            for _ in range(2):
                run(x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                y = run(x)
        torch.cuda.synchronize()
        ms = timeit(g.replay)
        print('side streams %s: graph %.3f ms, max |dy| vs eager %.2e' % (side, ms, float((y - y_ref).abs().max())), flush=True)
    except Exception as e:
        print('side streams %s: capture failed: %s: %s' % (side, type(e).__name__, str(e)[:300]), flush=True)
