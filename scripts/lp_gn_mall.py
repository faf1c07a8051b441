#!/usr/bin/env python3
"""Does the second pass of the 16-bit GroupNorm backward run faster when its operands still sit in the 256 MB memory-side cache?
gn_bwd (reduce + apply) on tensors of growing size: time per GB of compulsory traffic (reduce 2 reads, apply 2 reads + 1 write)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
from bts_amd.layers.group_norm import GroupNormalization
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
for n, d, c in ((1, 64, 64), (2, 64, 64), (4, 64, 64), (8, 64, 64), (1, 128, 32), (2, 128, 32), (8, 128, 32)):
    x = torch.randn((n, d, d, d, c), device=D).to(tdt)
    dy = torch.randn((n, d, d, d, c), device=D).to(tdt)
    norm = GroupNormalization(groups=8, axis=-1)
    norm.build((None, None, None, None, c))
    mean, rstd = lowp.gn_stats(code, x, 8, norm._mode, norm.epsilon)
    dg, db = torch.zeros(c, device=D), torch.zeros(c, device=D)
    for _ in range(2):
        lowp.gn_bwd(code, tdt, x, dy, norm.gamma.t, norm.beta.t, mean, rstd, dg, db, 8, True, want_f32=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lowp.gn_bwd(code, tdt, x, dy, norm.gamma.t, norm.beta.t, mean, rstd, dg, db, 8, True, want_f32=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    T = x.numel() * 2 / 1e9
    print('N=%d %d^3 x %d ch: tensor %.0f MB, gn_bwd %.3f ms = %.2f TB/s over 5 tensor passes' % (n, d, c, T * 1e3, ms, 5 * T / ms))
