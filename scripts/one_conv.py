#!/usr/bin/env python3
"""Run one conv kernel variant repeatedly (for rocprofv3 --pmc passes). Usage: one_conv.py which kind D Cin Cout reps
which in {fwd, bwd, wgrad}"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import ops  # noqa: E402

which, kind, d, cin, cout, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
D = torch.device('cuda:0')
x = torch.randn((1, d, d, d, cin), device=D)
k = 1 if kind == 0 else 3
w = torch.randn((k, k, k, cout, cin) if kind == 3 else (k, k, k, cin, cout), device=D) * 0.1
b = torch.randn(cout, device=D)
wp = ops.conv_pack(kind, ops.ROLE_FWD, w, cin, cout)
wpb = ops.conv_pack(kind, ops.ROLE_BWD, w, cin, cout)
y = ops.conv_fwd(kind, x, wp, b, cout)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
db = torch.empty_like(b)
for _ in range(reps):
    if which == 'fwd':
        ops.conv_fwd(kind, x, wp, b, cout, out=y)
    elif which == 'bwd':
        ops.conv_bwd_data(kind, dy, wpb, dx, False)
    else:
        ops.conv_bwd_weight(kind, x, dy, dw, None if kind == 3 else db)
torch.cuda.synchronize()
