"""Timing experiments on lp_s1d_kernel at the INFERENCE volume's levels (fp16; a -DBTS_TIMING_EXPERIMENTS build via BTS_HIP_LIB;
BTS_S1D_DBG bits: 1 no output stores, 2 no halo traffic, 4 no matrix instructions, 8 no fragment reads, 16 no weight traffic): per-shape
time of conv (plain) and conv + fused GroupNorm partial sums, to see what an item's fixed cost is made of."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: F401,E402
from bts_amd import lowp, ops  # noqa: E402
from bts_amd.layers.group_norm import GroupNormalization  # noqa: E402

DEV = torch.device('cuda', 0)
SHAPES = [(1, 80, 96, 80, 32, 64), (1, 80, 96, 80, 64, 64), (1, 80, 96, 80, 192, 64), (1, 40, 48, 40, 128, 128), (1, 40, 48, 40, 512, 128)]
code, tdt = lowp.DTYPES['float16']
out = []
for (n, d, h, w, cin, cout) in SHAPES:
    x = torch.randn((n, d, h, w, cin), device=DEV).to(tdt)
    wt = torch.randn((3, 3, 3, cin, cout), device=DEV) * (2.0 / (27 * cin)) ** 0.5
    b = torch.zeros(cout, device=DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    norm = GroupNormalization(groups=8, axis=-1)
    norm.build((None, None, None, None, cout))
    res = []
    for fn in (lambda: lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout), lambda: lowp.conv_gn(code, tdt, x, wp, b, cout, norm)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    out.append('%d->%d@%d: %6.1f / %6.1f us' % (cin, cout, d, res[0], res[1]))
print('DBG=%-3s ' % os.environ.get('BTS_S1D_DBG', '-') + ' | '.join(out), flush=True)
