#!/usr/bin/env python3
"""A/B inside one process (the pool's boxes wander by several % between runs): the batch-8 bf16 step with the block backward's gate /
GroupNorm-2 passes separate, with the apply passes fused, and with both pairs fused; interleaved rounds, median ms per step."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa
from bts_amd.data import synthetic_batch
from bts_amd.lowp_train import LowPrecisionTrainer
from bts_amd.model import Model
from bts_amd.util import DiceCoefficient, ScheduledOptim
dev = torch.device('cuda', 0)
m = Model(base_filters=32, reduction=8, depth=4, groups=8)
m.build((8, 128, 128, 128, 2))
x, y, _, _ = synthetic_batch(8, (128, 128, 128), latent=128, seed=1234)
x, y = x.to(dev), y.to(dev)
opt = ScheduledOptim(1e-4); opt(epoch=0)
tr = LowPrecisionTrainer(m, 'bfloat16')
df = DiceCoefficient()
modes = {'separate': (False, None), 'apply fused': (True, '0'), 'both fused': (True, '1')}
res = {k: [] for k in modes}
for _ in range(3):
    tr.step(opt, df, x, y)
for rnd in range(6):
    for name, (fuse, red) in modes.items():
        tr.fuse_block_bwd = fuse
        if red is None:
            os.environ.pop('BTS_LP_FUSE_BLOCK_BWD_REDUCE', None)
        else:
            os.environ['BTS_LP_FUSE_BLOCK_BWD_REDUCE'] = red
        tr.step(opt, df, x, y)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            tr.step(opt, df, x, y)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / 4 * 1e3)
for k, v in res.items():
    print('%-12s median %.2f ms  (%s)' % (k, statistics.median(v), ' '.join('%.1f' % t for t in v)))
