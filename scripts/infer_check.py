#!/usr/bin/env python3
"""BASELINE configs[4] shape check: full-volume inference (155x190x147 zero-padded to 160x192x160, test.py:164-178),
inference=True (VAE skipped), fp32 (the fp16 storage path is not built this round). Prints volumes/s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd.model import Model  # noqa: E402

model = Model(base_filters=32, reduction=8)
x = torch.randn((1, 160, 192, 160, 2))
x[:, 155:] = 0
x[:, :, 190:] = 0
x[:, :, :, 147:] = 0
x = x.cuda()
model.build((1, 128, 128, 128, 2))     # weights are built for the training crop (the VAE is tied to it, vae.py:101-111)
for _ in range(2):
    y = model(x, training=False, inference=True)[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y = model(x, training=False, inference=True)[0]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('inference 160x192x160 fp32: %.1f ms/volume = %.2f volumes/s ; y_pred %s mean %.4f' % (1e3 * dt, 1 / dt, tuple(y.shape), float(y.t.mean())))
print('max memory allocated: %.2f GB' % (torch.cuda.max_memory_allocated() / 2 ** 30))
