#!/usr/bin/env python3
"""conv2 of the top inference level (32 -> 32 @ 160x192x160, fp16) three ways: gn_apply + conv_gn, the fused form
(bts_lp_conv3d_gnin_fwd_gn), and the plain conv alone -- HIP-event times per call.  With BTS_HIP_LIB pointing at a timing-experiment
build of lowp_s1z.hip (make variantf FILE=lowp_s1z NAME=gna_nolds EXTRA=-DS1Z_EXP_GNA_NOLDS / ..._NOMATH) the fused time shows what the
in-LDS transform costs without its LDS traffic / its arithmetic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import lowp, ops  # noqa: E402


def main(shape=(1, 160, 192, 160), c=32, dtype='float16', reps=20):
    code, tdt = lowp.DTYPES[dtype]
    D = torch.device('cuda:0')
    n, d, h, w = shape
    c1 = torch.randn(shape + (c,), device=D).to(tdt)
    wt = torch.randn((3, 3, 3, c, c), device=D) * 0.05
    b = torch.zeros(c, device=D)
    wp = lowp.pack(ops.K3S1, code, wt, c, c)

    class _P(object):
        def __init__(self, t):
            self.t = t

    class _N(object):
        groups, epsilon, _mode = 8, 1e-5, ops.GN_SLAB
        gamma, beta = _P(torch.ones(c, device=D)), _P(torch.zeros(c, device=D))
    m1, r1 = lowp.gn_stats(code, c1, 8, ops.GN_SLAB, 1e-5)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def unfused():
        a = lowp.gn_apply(code, c1, _N.gamma.t, _N.beta.t, m1, r1, 8, ops.GN_SLAB, True)
        return lowp.conv_gn(code, tdt, a, wp, b, c, _N)
    t_plain = timed(lambda: lowp.conv_gn(code, tdt, c1, wp, b, c, _N))
    t_unf = timed(unfused)
    t_fus = timed(lambda: lowp.conv_gn_normed_input(code, tdt, c1, _N, m1, r1, True, wp, b, c, _N))
    print('%s: conv_gn alone %.1f us | gn_apply + conv_gn %.1f us | fused %.1f us' % (os.environ.get('BTS_HIP_LIB', 'product')[-24:], t_plain, t_unf, t_fus))


if __name__ == '__main__':
    main()
