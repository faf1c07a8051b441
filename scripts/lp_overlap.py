#!/usr/bin/env python3
"""do an HBM-bound pass and a matrix-bound kernel of the 16-bit engine run side by side on two streams?  Each alone, then together:
lp_overlap.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import lowp, ops
code, tdt = lowp.DTYPES['bfloat16']
D = torch.device('cuda:0')
G = 8
# matrix-bound: stride-1 weight gradient 64 -> 64 at 8 x 64^3 (lp_wgd) and the conv itself (lp_s1d)
xs = torch.randn((8, 64, 64, 64, 64), device=D).to(tdt)
dys = torch.randn((8, 64, 64, 64, 64), device=D).to(tdt)
dw = torch.zeros((3, 3, 3, 64, 64), device=D)
wt = torch.randn((3, 3, 3, 64, 64), device=D) * 0.02
wp = lowp.pack(ops.K3S1, code, wt, 64, 64)
ys = torch.empty_like(xs)
# HBM-bound: GroupNorm backward at 8 x 128^3 x 32
xe = torch.randn((8, 128, 128, 128, 32), device=D).to(tdt)
de = torch.randn((8, 128, 128, 128, 32), device=D).to(tdt)
gamma = torch.rand(32, device=D) + 0.5
beta = torch.zeros(32, device=D)
mean, rstd = lowp.gn_stats(code, xe, G, ops.GN_SLAB, 1e-5)
dg, db = torch.zeros(32, device=D), torch.zeros(32, device=D)


def mm_wgd():
    lowp.conv_bwd_weight(ops.K3S1, code, xs, dys, dw, None, 0, 0, True)


def mm_conv():
    lowp.conv(ops.K3S1, code, tdt, xs, wp, None, 64, out=ys)


def hbm():
    lowp.gn_bwd(code, tdt, xe, de, gamma, beta, mean, rstd, dg, db, G, True, want_f32=False)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, na, fb, nb, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(na):
                fa()
        with torch.cuda.stream(s2):
            for _ in range(nb):
                fb()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for name, mm, nm in (('lp_wgd 64->64 @8x64^3', mm_wgd, 6), ('lp_s1d 64->64 @8x64^3', mm_conv, 4)):
    for _ in range(2):
        mm(); hbm()
    ta = timed(mm, nm, lambda: None, 0)
    tb = timed(lambda: None, 0, hbm, 2)
    tab = timed(mm, nm, hbm, 2)
    print('%s x%d alone %.2f ms | GroupNorm backward 8x128^3x32 x2 alone %.2f ms | together on two streams %.2f ms (sum %.2f, max %.2f)' %
          (name, nm, ta, tb, tab, ta + tb, max(ta, tb)))


# other tenants: torch's own element-wise kernels (no LDS, few registers, several loads in flight per thread) on a tensor pair of the same
# size -- is it THIS engine's pass that cannot overlap, or any HBM-bound kernel next to these matrix kernels?
src = torch.randn((8, 128, 128, 128, 32), device=D).to(tdt)
dst = torch.empty_like(src)


def t_copy():
    dst.copy_(src)
    dst.copy_(src)
    dst.copy_(src)


def t_add():
    torch.add(src, xe, out=dst)
    torch.add(src, xe, out=dst)


for tname, ten in (('torch copy x3 (3.2 GB r + 3.2 GB w)', t_copy), ('torch add x2 (4.3 GB r + 2.1 GB w)', t_add)):
    for name, mm, nm in (('lp_wgd', mm_wgd, 6), ('lp_s1d', mm_conv, 4)):
        for _ in range(2):
            mm(); ten()
        ta = timed(mm, nm, lambda: None, 0)
        tb = timed(lambda: None, 0, ten, 1)
        tab = timed(mm, nm, ten, 1)
        print('%s x%d alone %.2f ms | %s alone %.2f ms | together %.2f ms (sum %.2f, max %.2f)' % (name, nm, ta, tname, tb, tab, ta + tb, max(ta, tb)))
