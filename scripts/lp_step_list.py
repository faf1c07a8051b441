#!/usr/bin/env python3
"""every profiled launch of one 16-bit training step (bf16, batch 8, 128^3): symbol, algorithmic GFLOP, microseconds, TFLOP/s;
argument: a substring filter on the symbol"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd import ops
from bts_amd.model import Model
from bts_amd.util import DiceCoefficient, ScheduledOptim
from bts_amd.lowp_train import LowPrecisionTrainer
from oracle import torch_ref as R
dev = torch.device('cuda', 0)
flt = sys.argv[1] if len(sys.argv) > 1 else ''
model = Model(base_filters=32, reduction=8, depth=4, groups=8)
model.build((1, 128, 128, 128, 2))
x, y, _, _ = R.synthetic_batch(8, (128, 128, 128), latent=128, seed=1)
x, y = x.to(dev), y.to(dev)
opt = ScheduledOptim(1e-4)
opt(epoch=0)
df = DiceCoefficient()
tr = LowPrecisionTrainer(model, 'bfloat16')
ops.enable_side_streams(False)
for _ in range(2):
    tr.step(opt, df, x, y)
torch.cuda.synchronize()
ops.profile_enable(True)
tr.step(opt, df, x, y)
torch.cuda.synchronize()
ops.profile_enable(False)
tot = {}
for sym, fl, ms in ops.profile_records():
    tot[sym] = tot.get(sym, 0.0) + ms
    if flt in sym:
        print('%-24s %9.2f GF %9.1f us %8.1f TF' % (sym, fl / 1e9, ms * 1e3, fl / ms / 1e9))
print({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
