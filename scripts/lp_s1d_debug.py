import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
def run(n, d, h, w, cin, cout, dtype='bfloat16', seed=0, reps=3):
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.3).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    ref = torch.nn.functional.conv3d(x.float().permute(0, 4, 1, 2, 3), wt.to(tdt).float().permute(4, 3, 0, 1, 2), b, padding=1).permute(0, 2, 3, 4, 1)
    for rep in range(reps):
        os.environ['BTS_LP_S1D'] = '1'
        y = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout).float()
        torch.cuda.synchronize()
        e = (y - ref).abs()
        bad = e > 0.05 + 0.02 * ref.abs()
        print('shape', (n, d, h, w, cin, cout), dtype, 'rep', rep, 'bad', int(bad.sum()), 'of', bad.numel())
        if bad.any():
            print(' bad per n', bad.sum(dim=(1, 2, 3, 4)).tolist())
            print(' bad per z', bad.sum(dim=(0, 2, 3, 4)).tolist())
            print(' bad per y', bad.sum(dim=(0, 1, 3, 4)).tolist())
            print(' bad per x', bad.sum(dim=(0, 1, 2, 4)).tolist())
            print(' bad per c', bad.sum(dim=(0, 1, 2, 3)).tolist())
for a in [(2, 10, 12, 20, 32, 64), (1, 12, 20, 70, 16, 96), (2, 9, 13, 40, 48, 64), (1, 8, 8, 40, 16, 32), (1, 8, 8, 48, 16, 32)]:
    run(*a)
