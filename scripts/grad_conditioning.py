#!/usr/bin/env python3
"""How far is the engine's gradient from the fp64 oracle at the CLI model / 64^3, per conv form, next to torch-fp32's own
deviation?  (Diagnostic behind the tolerances of tests/test_model_gpu.py and tests/test_fullsize_gpu.py.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import bts_amd  # noqa: E402,F401
from bts_amd.model import Model  # noqa: E402
from bts_amd.tape import GradientTape  # noqa: E402
from bts_amd.util import DiceVAELoss, reduce_sum  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
import test_model_gpu as TM  # noqa: E402

crop = tuple(int(v) for v in (sys.argv[1:4] or (64, 64, 64)))
kw = dict(base_filters=32, groups=8, reduction=8, depth=4)
cfg = R.default_config(**kw)
x, y, mask, eps = R.synthetic_batch(1, crop, latent=128, seed=1234)
P = TM.randomised_params(cfg, crop, seed=7)
_, loss_r, g64 = TM.run_oracle(cfg, P, x, y, mask, eps)
_, _, g32 = TM.run_oracle(cfg, P, x, y, mask, eps, dtype=torch.float32)


def engine(direct):
    for k in ('BTS_WINO', 'BTS_WGW', 'BTS_K1W'):
        if direct:
            os.environ[k] = '0'
        else:
            os.environ.pop(k, None)
    model = Model(**kw)
    model.build((1,) + crop + (2,))
    model.set_weights_from(P)
    model.encoder.set_dropout_mask(mask)
    model.vae.set_eps(eps)
    lf = DiceVAELoss()
    with GradientTape() as tape:
        out = model(x, training=True, inference=False)
        loss = lf(x, y, *out) + reduce_sum(model.losses)
    grads = tape.gradient(loss, model.trainable_variables)
    torch.cuda.synchronize()
    return {model.oracle_name(p): g.detach().cpu().double() for p, g in zip(model.trainable_variables, grads)}, float(loss)


def table(name, g):
    rows = []
    for k, ref in g64.items():
        s = float(ref.abs().max()) + 1e-30
        rows.append((float((g[k].double() - ref).abs().max()) / s, float((g[k].double() - ref).norm() / (ref.norm() + 1e-30)), k, s))
    rows.sort(reverse=True)
    print('== %s: worst max-abs-relative deviations from fp64 (and relative L2 of the same variable)' % name)
    for r in rows[:8]:
        print('   %-36s %.3e  (L2 %.3e, scale %.3e)' % (r[2], r[0], r[1], r[3]))
    print('   median over variables: %.3e ; > 1e-3: %d of %d' % (sorted(r[0] for r in rows)[len(rows) // 2],
                                                                  sum(r[0] > 1e-3 for r in rows), len(rows)))


table('torch fp32', g32)
ga, la = engine(False)
table('engine, default forms (loss %.7f vs %.7f)' % (la, float(loss_r)), ga)
gd, ld = engine(True)
table('engine, direct forms (loss %.7f)' % ld, gd)
