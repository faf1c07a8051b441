#!/usr/bin/env python3
"""Bitwise run-to-run reproducibility of the full-size training step (CLI-default model, 2ch x 128^3): two fresh models with
identical weights, RNG counters and data must hold identical parameters after two optimiser steps (no float atomics,
fixed-order reductions everywhere)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd.layers import _base  # noqa: E402
from bts_amd.model import Model  # noqa: E402
from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

dev = torch.device('cuda', 0)
crop = (128, 128, 128)
x, y, _, _ = R.synthetic_batch(1, crop, latent=128, seed=5)
x, y = x.to(dev), y.to(dev)
outs = []
for run in range(2):
    _base.set_seed(1234)
    m = Model(base_filters=32, reduction=8, depth=4, groups=8)
    m.build((1,) + crop + (2,))
    opt = ScheduledOptim(1e-3)
    opt(epoch=0)
    lf, df = DiceVAELoss(), DiceCoefficient()
    for _ in range(2):
        loss, macro, micro = train_step(m, opt, lf, df, x, y)
    torch.cuda.synchronize()
    outs.append((m.flat_params.clone(), float(loss), float(macro)))
    del m, opt
same = torch.equal(outs[0][0], outs[1][0])
print('loss', outs[0][1], outs[1][1], 'dice', outs[0][2], outs[1][2])
print('bitwise identical parameters after 2 steps:', same)
assert same and outs[0][1] == outs[1][1]
