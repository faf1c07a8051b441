#!/usr/bin/env python3
"""the element-wise passes of the 16-bit engine, one launch each at a level's shape: microseconds and the HBM rate their ALGORITHMIC
bytes amount to (16-bit tensors read + written; the small fp32 side outputs counted too).  Held against scripts/hbm_bw.py's
practical ceilings (read-only ~3.9 TB/s, 2 reads + 1 write ~5.9 TB/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops

D = torch.device('cuda:0')
SHAPES = [('bfloat16', (8, 128, 128, 128, 32)), ('bfloat16', (8, 64, 64, 64, 64)), ('bfloat16', (8, 32, 32, 32, 128)),
          ('float16', (1, 160, 192, 160, 32)), ('float16', (1, 80, 96, 80, 64)), ('float16', (1, 40, 48, 40, 128))]
if len(sys.argv) > 1:
    import ast
    SHAPES = ast.literal_eval(sys.argv[1])


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dtype, shape in SHAPES:
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, c = shape
    v = d * h * w
    G = 8
    tb = n * v * c * 2            # bytes of one 16-bit tensor
    x = torch.randn(shape, device=D).to(tdt)
    y = torch.randn(shape, device=D).to(tdt)
    z = torch.randn(shape, device=D).to(tdt)
    gamma = torch.rand(c, device=D) + 0.5
    beta = torch.randn(c, device=D) * 0.1
    mode = ops.GN_SLAB
    rows = []
    mean, rstd = lowp.gn_stats(code, x, G, mode, 1e-5)
    rows.append(('gn_stats', timed(lambda: lowp.gn_stats(code, x, G, mode, 1e-5)), tb))
    out = torch.empty_like(x)
    rows.append(('gn_apply', timed(lambda: lowp.gn_apply(code, x, gamma, beta, mean, rstd, G, mode, True, out=out)), 2 * tb))
    wsp = torch.randn(c, device=D) * 0.1
    ch = torch.rand((n, c), device=D)
    sp = torch.empty(n * v, dtype=torch.float32, device=D)
    rows.append(('block_epilogue', timed(lambda: lowp.block_epilogue(code, x, y, out, wsp, ch, gamma, beta, mean, rstd, G, mode)), 3 * tb))
    rows.append(('block_epilogue+sp', timed(lambda: lowp.block_epilogue(code, x, y, out, wsp, ch, gamma, beta, mean, rstd, G, mode, sp_out=sp)),
                 3 * tb + n * v * 4))
    dg, db = torch.zeros(c, device=D), torch.zeros(c, device=D)
    dbias = torch.zeros(c, device=D)
    r = lowp.gn_bwd(code, tdt, x, y, gamma, beta, mean, rstd, dg, db, G, True, want_f32=False, dbias=dbias)
    if r is not None:
        rows.append(('gn_bwd (reduce+apply)', timed(lambda: lowp.gn_bwd(code, tdt, x, y, gamma, beta, mean, rstd, dg, db, G, True, want_f32=False,
                                                                       dbias=dbias)), 5 * tb))
    red = max(c // 8, 1)
    w1 = torch.randn((c, red), device=D) * 0.1
    w2 = torch.randn((red, c), device=D) * 0.1
    gap = torch.rand((n, c), device=D)
    hb, chh = ops.se_mlp_fwd(gap, w1, w2)
    dw1, dw2, dwsp = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(wsp)
    spv = torch.rand(n * v, device=D)
    dbp, dbc = torch.zeros(c, device=D), torch.zeros(c, device=D)
    r = lowp.block_bwd(code, tdt, z, x, y, spv, gap, hb, chh, w1, w2, wsp, gamma, beta, mean, rstd, G, dw1, dw2, dwsp, dg, db, dbp, dbc)
    if r is not None:
        rows.append(('block_bwd (reduce+apply)', timed(lambda: lowp.block_bwd(code, tdt, z, x, y, spv, gap, hb, chh, w1, w2, wsp, gamma, beta, mean, rstd,
                                                                             G, dw1, dw2, dwsp, dg, db, dbp, dbc)), 8 * tb + 3 * n * v * 4))
    print('%s %s  (one tensor = %.0f MB)' % (dtype, shape, tb / 1e6))
    for name, us, by in rows:
        print('   %-26s %8.1f us   %5.2f TB/s' % (name, us, by / us / 1e6))
