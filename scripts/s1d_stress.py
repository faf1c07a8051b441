"""determinism stress of the 16-bit stride-1 conv kernels: the same launch repeated, outputs compared bitwise with the first"""
import sys
import torch
import bts_amd  # noqa: F401
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
shapes = [(2, 16, 16, 16, 32, 32), (2, 16, 16, 16, 96, 32), (2, 16, 16, 16, 64, 64), (1, 16, 24, 40, 32, 32), (2, 32, 32, 32, 32, 16)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for dt in ('float16', 'bfloat16'):
    code, tdt = lowp.DTYPES[dt]
    for (n, d, h, w, cin, cout) in shapes:
        g = torch.Generator().manual_seed(1)
        x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
        wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * 0.05).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
        ops.profile_enable(True)
        first = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout).clone()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        syms = sorted(set(s for s, _, _ in ops.profile_records()))
        bad = 0
        worst = 0.0
        for r in range(reps):
            # other work in between, so that LDS / registers of the CUs hold something else
            junk = torch.randn((1 << 20,), device=DEV).sin_()
            y = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
            torch.cuda.synchronize()
            if not torch.equal(y, first):
                bad += 1
                worst = max(worst, float((y.float() - first.float()).abs().max()))
        print('%-9s %s %s: %d / %d runs differ (max |d| %.3e)' % (dt, (n, d, h, w, cin, cout), syms, bad, reps, worst), flush=True)
