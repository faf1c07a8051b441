#!/usr/bin/env python3
"""Per-layer-shape timing of the three conv kernels (fwd / bwd_data / bwd_weight) at the CLI-default 128^3 shapes.
Usage on the GPU box: python scripts/conv_microbench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import ops  # noqa: E402

SHAPES = [  # kind, D, Cin, Cout, count(label)
    (1, 128, 2, 32, 'enc.L0 conv1'), (1, 128, 32, 32, '128^3 32->32 (x7)'), (1, 128, 64, 32, 'dec.L0 conv1'),
    (1, 128, 32, 2, 'vae.out'),
    (1, 64, 32, 64, 'L1 b0 conv1'), (1, 64, 64, 64, '64^3 64->64'), (1, 64, 192, 64, 'dec.L1 conv1'),
    (1, 32, 128, 128, '32^3 128->128'), (1, 32, 256, 128, 'L2 b2 conv1(folded)'), (1, 32, 512, 128, 'dec.L2 conv1'),
    (1, 16, 256, 256, '16^3 256->256'), (1, 16, 768, 256, 'L3 b3 conv1(folded)'),
    (0, 128, 64, 32, 'dec.L0 ptwise'), (0, 128, 32, 32, '128^3 ptwise'), (0, 16, 768, 256, 'L3 b3 ptwise'),
    (2, 128, 32, 32, 'down L0'), (2, 64, 128, 64, 'down L1'), (2, 32, 384, 128, 'down L2'), (2, 16, 1024, 16, 'vae down'),
    (3, 64, 64, 32, 'up ->128^3'), (3, 32, 128, 64, 'up ->64^3'), (3, 16, 1024, 128, 'up ->32^3'), (3, 8, 1, 256, 'vae up 1->256'),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    D = torch.device('cuda:0')
    print('%-26s %5s %5s %5s | %9s %7s | %9s %7s | %9s %7s' % ('layer', 'D', 'Cin', 'Cout', 'fwd ms', 'TF', 'bwdD ms', 'TF',
                                                              'wgrad ms', 'TF'))
    for kind, d, cin, cout, label in SHAPES:
        x = torch.randn((1, d, d, d, cin), device=D)
        if os.environ.get('BTS_BENCH_ZERO'):  # data-dependent power: all-zero operands let the clocks stay up
            x.zero_()
        k = 1 if kind == 0 else 3
        w = torch.randn((k, k, k, cout, cin) if kind == 3 else (k, k, k, cin, cout), device=D) * 0.1
        b = torch.randn(cout, device=D)
        wp = ops.conv_pack(kind, ops.ROLE_FWD, w, cin, cout)
        wpb = ops.conv_pack(kind, ops.ROLE_BWD, w, cin, cout)
        y = ops.conv_fwd(kind, x, wp, b, cout)
        dy = torch.randn_like(y)
        if os.environ.get('BTS_BENCH_ZERO'):
            dy.zero_()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        fl = ops.conv_flops(kind, 1, d, d, d, cin, cout)
        tf = timeit(lambda: ops.conv_fwd(kind, x, wp, b, cout, out=y), reps)
        tb = timeit(lambda: ops.conv_bwd_data(kind, dy, wpb, dx, False), reps)
        tw = timeit(lambda: ops.conv_bwd_weight(kind, x, dy, dw, None if kind == 3 else db), reps)
        print('%-26s %5d %5d %5d | %9.3f %7.1f | %9.3f %7.1f | %9.3f %7.1f' % (label, d, cin, cout, tf, fl / tf / 1e9, tb,
                                                                              fl / tb / 1e9, tw, fl / tw / 1e9))


if __name__ == '__main__':
    main()
