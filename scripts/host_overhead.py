#!/usr/bin/env python3
"""How long does the HOST need to enqueue one training step (Python + ctypes + launch calls)?  If this approaches the GPU
time per step the engine becomes launch-bound.  Usage on the GPU box: python scripts/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd.model import Model  # noqa: E402
from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

dev = torch.device('cuda', 0)
crop = (128, 128, 128)
MODE = sys.argv[1] if len(sys.argv) > 1 else 'f32'       # f32 | bf16 (batch 8) | infer (fp16, 160x192x160)
model = Model(base_filters=32, reduction=8, depth=4, groups=8)
model.build((1,) + crop + (2,))
batch = 8 if MODE == 'bf16' else 1
x, y, _, _ = R.synthetic_batch(batch, crop, latent=128, seed=1)
x, y = x.to(dev), y.to(dev)
opt = ScheduledOptim(1e-4)
opt(epoch=0)
lf, df = DiceVAELoss(), DiceCoefficient()
if MODE == 'bf16':
    from bts_amd.lowp_train import LowPrecisionTrainer
    trainer = LowPrecisionTrainer(model, 'bfloat16')
    train_step = lambda m_, o_, l_, d_, x_, y_: trainer.step(o_, d_, x_, y_)   # noqa: E731,F811
elif MODE == 'infer':
    from bts_amd import lowp
    run = lowp.LowPrecisionForward(model, 'float16')
    xv = torch.randn((1, 160, 192, 160, 2), device=dev)
    train_step = lambda m_, o_, l_, d_, x_, y_: run(xv)   # noqa: E731,F811
for _ in range(3):
    train_step(model, opt, lf, df, x, y)
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(model, opt, lf, df, x, y)
    t1 = time.perf_counter()          # everything enqueued (the step never reads a value back)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
print('host enqueue per step: %.1f ms (min %.1f)   step wall: %.1f ms' % (1e3 * sum(host) / len(host), 1e3 * min(host),
                                                                        1e3 * sum(total) / len(total)))
