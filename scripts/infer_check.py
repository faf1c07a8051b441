#!/usr/bin/env python3
"""BASELINE configs[4] shape: full-volume inference (155x190x147 zero-padded to 160x192x160, test.py:164-178), inference=True
(VAE skipped), CLI-default model.  usage: infer_check.py [fp32|float16|bfloat16]   -> ms per volume, peak memory"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd import lowp  # noqa: E402
from bts_amd.model import Model  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
model = Model(base_filters=32, reduction=8)
x = torch.randn((1, 160, 192, 160, 2))
x[:, 155:] = 0
x[:, :, 190:] = 0
x[:, :, :, 147:] = 0
x = x.cuda()
model.build((1, 128, 128, 128, 2))     # weights are built for the training crop (the VAE is tied to it, vae.py:101-111)
run = (lambda: model(x, training=False, inference=True)[0].t) if mode == 'fp32' else lowp.LowPrecisionForward(model, mode)
fwd = run if mode == 'fp32' else (lambda: run(x))
for _ in range(2):
    y = fwd()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    y = fwd()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('inference 160x192x160 %s: %.2f ms/volume = %.2f volumes/s ; y_pred %s mean %.4f ; peak memory %.2f GB' %
      (mode, 1e3 * dt, 1 / dt, tuple(y.shape), float(y.mean()), torch.cuda.max_memory_allocated() / 2 ** 30))
