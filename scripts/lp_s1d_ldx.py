#!/usr/bin/env python3
"""Does lp_s1d's time depend on how its halo requests coalesce?  The same 32 -> 64 convolution at 8 x 64^3 on an input that is dense
(voxel stride 32 channels: a k-step's 32 bytes are half of a 64-byte row) or a 32-channel view of a 64 / 128 / 256-channel slab (32 of
128 / 256 / 512 bytes: one L2 request per voxel and k-step either way, but 2x / 4x / 8x the lines touched per useful byte)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
os.environ['BTS_LP_S1Z'] = '0'
for shape, cin, cout in (((8, 64, 64, 64), 32, 64), ((8, 64, 64, 64), 64, 64), ((8, 32, 32, 32), 128, 128)):
    wt = torch.randn((3, 3, 3, cin, cout), device=D) * 0.02
    b = torch.zeros(cout, device=D)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    for ld in (cin, 2 * cin, 4 * cin, 8 * cin):
        slab = torch.randn(shape + (ld,), device=D).to(tdt)
        x = slab[..., :cin]
        for _ in range(3):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * 27 * cin * cout * shape[0] * shape[1] * shape[2] * shape[3]
        print(shape, '%d -> %d, voxel stride %4d channels: %.1f us  %.0f TF' % (cin, cout, ld, ms * 1e3, fl / ms / 1e9), flush=True)
        del slab, x
