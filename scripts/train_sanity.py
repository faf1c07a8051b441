#!/usr/bin/env python3
"""End-to-end sanity on the GPU box: 40 optimiser steps of the CLI-default model on two fixed synthetic 64^3 volumes must
drive the loss down without NaNs (not a parity test -- those are in tests/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd.model import Model  # noqa: E402
from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

dev = torch.device('cuda', 0)
crop = (64, 64, 64)
model = Model(base_filters=32, reduction=8, depth=4, groups=8)
model.build((1,) + crop + (2,))
data = []
for s in (1, 2):
    x, y, _, _ = R.synthetic_batch(1, crop, latent=128, seed=s)
    data.append((x.to(dev), y.to(dev)))
opt = ScheduledOptim(1e-3)
opt(epoch=0)
lf, df = DiceVAELoss(), DiceCoefficient()
hist = []
for it in range(40):
    x, y = data[it % 2]
    loss, macro, micro = train_step(model, opt, lf, df, x, y)
    hist.append(float(loss))
    if it % 5 == 0 or it == 39:
        print('step %2d loss %.4f macro dice %.4f' % (it, hist[-1], float(macro)))
assert all(h == h for h in hist), 'NaN in the loss'
assert sum(hist[-4:]) / 4 < 0.9 * sum(hist[:4]) / 4, 'loss did not decrease: %r -> %r' % (hist[:4], hist[-4:])
print('ok: %.4f -> %.4f' % (sum(hist[:4]) / 4, sum(hist[-4:]) / 4))
