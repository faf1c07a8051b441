#!/usr/bin/env python3
"""fp32 train step (CLI model, 128^3, batch 1): main chain on a HIGH-priority stream or the default one, weight-gradient kernels with
their planned workgroup count or more, shorter workgroups (BTS_WGRAD_WGS) -- does finer-grained weight-gradient work let the main
chain's HBM-bound passes in sooner?  ms per step, wall clock over `--steps` steps."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: E402,F401
from bts_amd.data import synthetic_batch  # noqa: E402
from bts_amd.model import Model  # noqa: E402
from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--high', type=int, default=0)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    m = Model(base_filters=32, reduction=8, depth=4, groups=8)
    m.build((1, 128, 128, 128, 2))
    x, y, _, _ = synthetic_batch(1, (128,) * 3, latent=128, seed=1)
    x, y = x.to(dev), y.to(dev)
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    lf, df = DiceVAELoss(), DiceCoefficient()
    s = torch.cuda.Stream(priority=-1) if a.high else torch.cuda.current_stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            train_step(m, opt, lf, df, x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            train_step(m, opt, lf, df, x, y)
        torch.cuda.synchronize()
    print('high=%d BTS_WGRAD_WGS=%s: %.3f ms per step' % (a.high, os.environ.get('BTS_WGRAD_WGS', '-'), (time.perf_counter() - t0) / a.steps * 1e3))


if __name__ == '__main__':
    main()
