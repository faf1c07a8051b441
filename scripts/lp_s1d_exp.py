"""Timing experiments on lp_s1d_kernel (a -DBTS_TIMING_EXPERIMENTS build via BTS_HIP_LIB; BTS_S1D_DBG bits: 1 no output stores,
2 no halo traffic, 4 no matrix instructions): per-shape time of the new kernel under the current environment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: F401,E402
from bts_amd import lowp, ops  # noqa: E402

DEV = torch.device('cuda', 0)
SHAPES = [(8, 128, 128, 128, 16, 32), (8, 128, 128, 128, 32, 32), (8, 128, 128, 128, 32, 64), (8, 64, 64, 64, 64, 64),
          (8, 32, 32, 32, 512, 128)]
code, tdt = lowp.DTYPES['bfloat16']
out = []
for (n, d, h, w, cin, cout) in SHAPES:
    x = torch.randn((n, d, h, w, cin), device=DEV).to(tdt)
    wt = torch.randn((3, 3, 3, cin, cout), device=DEV) * (2.0 / (27 * cin)) ** 0.5
    b = torch.zeros(cout, device=DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    y = torch.empty((n, d, h, w, cout), dtype=tdt, device=DEV)
    for _ in range(2):
        lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout, out=y)
    e1.record()
    torch.cuda.synchronize()
    out.append('%d->%d@%d: %7.1f us' % (cin, cout, d, e0.elapsed_time(e1) / 5 * 1e3))
print('DBG=%s  ' % os.environ.get('BTS_S1D_DBG', '-') + ' | '.join(out), flush=True)
