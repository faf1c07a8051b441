#!/usr/bin/env python3
"""the deepest level of the full inference volume (256 -> 256 at 20x24x20): tiled LDS-DMA kernel (x extent 16 or 32) vs the
register-staged one, timed in one process"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['float16']
D = torch.device('cuda:0')
VARIANTS = (('old', {'BTS_LP_S1D': '0'}), ('s1d tx16', {'BTS_LP_S1D': '1', 'BTS_LP_S1D_FLOOR': '0', 'BTS_LP_S1D_TXL': '4'}),
            ('s1d tx32', {'BTS_LP_S1D': '1', 'BTS_LP_S1D_FLOOR': '0', 'BTS_LP_S1D_TXL': '5'}))
import ast
SHAPES = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else (((1, 20, 24, 20), 256, 256), ((1, 20, 24, 20), 512, 256), ((1, 16, 16, 16), 256, 256), ((2, 16, 16, 16), 256, 256))
for shape, cin, cout in SHAPES:
    x = torch.randn(shape + (cin,), device=D).to(tdt)
    wt = torch.randn((3, 3, 3, cin, cout), device=D) * 0.02
    b = torch.zeros(cout, device=D)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    ref = None
    for name, env in VARIANTS:
        for k in ('BTS_LP_S1D', 'BTS_LP_S1D_FLOOR', 'BTS_LP_S1D_TXL'):
            os.environ.pop(k, None)
        os.environ.update(env)
        ops.profile_enable(True)
        y = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        ops.profile_enable(False)
        syms = [r[0] for r in ops.profile_records()]
        if ref is None:
            ref = y.float()
        err = (y.float() - ref).abs().max().item()
        for _ in range(3):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * 27 * cin * cout * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3]
        print(shape, cin, cout, '%-9s' % name, syms, '%.1f us  %.0f TF  maxdiff %.3g' % (ms * 1e3, fl / ms / 1e9, err))
