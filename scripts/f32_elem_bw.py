#!/usr/bin/env python3
"""the element-wise passes of the fp32 engine, one call each at a level's shape: microseconds and the HBM rate of their algorithmic bytes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import ops

D = torch.device('cuda:0')
SHAPES = [(1, 128, 128, 128, 32), (1, 64, 64, 64, 64), (1, 32, 32, 32, 128), (1, 16, 16, 16, 256)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for shape in SHAPES:
    n, d, h, w, c = shape
    v = d * h * w
    G = 8
    tb = n * v * c * 4
    x, y, z = (torch.randn(shape, device=D) for _ in range(3))
    gamma = torch.rand(c, device=D) + 0.5
    beta = torch.randn(c, device=D) * 0.1
    mode = ops.GN_SLAB
    rows = []
    mean, rstd = ops.gn_stats(x, G, mode)
    rows.append(('gn_stats', timed(lambda: ops.gn_stats(x, G, mode)), tb))
    out = torch.empty_like(x)
    rows.append(('gn_apply', timed(lambda: ops.gn_apply(x, gamma, beta, mean, rstd, G, mode, True, out=out)), 2 * tb))
    wsp = torch.randn(c, device=D) * 0.1
    ch = torch.rand((n, c), device=D)
    rows.append(('block_epilogue_fwd', timed(lambda: ops.block_epilogue_fwd(x, y, out, wsp, ch, gamma, beta, mean, rstd, G, mode)), 3 * tb + n * v * 4))
    dg, db = torch.zeros(c, device=D), torch.zeros(c, device=D)
    rows.append(('gn_bwd (reduce+apply)', timed(lambda: ops.gn_bwd(x, y, gamma, beta, mean, rstd, dg, db, G, mode, True, True)), 5 * tb))
    red = max(c // 8, 1)
    w1 = torch.randn((c, red), device=D) * 0.1
    w2 = torch.randn((red, c), device=D) * 0.1
    gap = torch.rand((n, c), device=D)
    hb, chh = ops.se_mlp_fwd(gap, w1, w2)
    dw1, dw2, dwsp = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(wsp)
    spv = torch.rand(n * v, device=D)
    rows.append(('se_bwd (reduce+apply)', timed(lambda: ops.se_bwd(z, x, spv, gap, hb, chh, w1, w2, wsp, dw1, dw2, dwsp, True)), 4 * tb + 3 * n * v * 4))
    rows.append(('colsum', timed(lambda: ops.colsum(x)), tb))
    print('float32 %s  (one tensor = %.0f MB)' % (shape, tb / 1e6))
    for name, us, by in rows:
        print('   %-26s %8.1f us   %5.2f TB/s' % (name, us, by / us / 1e6))
