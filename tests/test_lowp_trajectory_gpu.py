"""-m gpu: bf16 STORAGE must TRAIN like fp32, not just match it for one step (round-3 verdict, missing item 5; the reference's own
evidence that its step trains is the Dice curve of README.md:81-89).

scripts/lp_trajectory.py: the CLI-default model (42,174,773 parameters) at 64^3, 8 synthetic training volumes whose label spheres
carry a per-class intensity contrast (so there is something to learn) + 2 held-out ones, 200 Adam steps at batch 1 and the reference's
learning rate from the same initial weights and the same dropout / eps draws, once through the fp32 engine and once through
bts_amd.lowp_train (bf16 storage, fp32 sums and master weights).  Stated bounds: the held-out macro Dice of the two runs ends within
0.01 (measured 0.0025 at 0.90), the mean training loss of the last 25 steps within 4 % (measured 1.7 % where the loss still falls 6 % per 25 steps), and the fp32 run must actually have learnt something (held-out Dice up
by >= 0.05 over the run; otherwise equal Dice would mean nothing).  The curves are printed (pytest -s) and stored under gpurun_out/."""
import importlib.util
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script():
    spec = importlib.util.spec_from_file_location('lp_trajectory', os.path.join(ROOT, 'scripts', 'lp_trajectory.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bf16_storage_follows_the_fp32_trajectory_for_200_steps():
    res = _script().run('bfloat16', steps=200, crop=64, n_train=8, n_heldout=2, lr=1e-4, every=25)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'lp_trajectory_bf16.json'), 'w'), indent=1)
    except OSError:
        pass
    assert res['dice_gain_f32'] >= 0.05, res['checkpoints']
    assert res['final_dice_gap'] <= 0.01, res['checkpoints']
    assert res['final_loss_gap_rel'] <= 0.04, res['checkpoints']


def test_float16_trainer_scales_its_loss():
    """float16 activation gradients need loss scaling (class docstring of LowPrecisionTrainer); 30 steps at 32^3: no skipped step after
    the scale has settled, a gradient that agrees with fp32's, and an overflow (forced by an absurd scale) skips the step and halves
    the scale instead of writing infinities into the weights"""
    import torch
    import bts_amd  # noqa: F401
    from bts_amd.data import synthetic_batch
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    kw = dict(base_filters=16, groups=8, reduction=2, depth=3)
    crop = (32, 32, 32)
    m = Model(**kw)
    m.build((2,) + crop + (2,))
    x, y, mask, eps = synthetic_batch(2, crop, latent=32, seed=9)
    start = m.flat_params.clone()
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    train_step(m, opt, DiceVAELoss(), DiceCoefficient(), x, y)
    g32 = m.flat_grads.clone()
    m.flat_params.copy_(start)
    bump_weights_epoch()
    with pytest.raises(ValueError):
        LowPrecisionTrainer(m, 'float16', loss_scale=1000.0)
    tr = LowPrecisionTrainer(m, 'float16')
    assert tr.loss_scale == 2.0 ** 16 and tr.dynamic_scale
    o = ScheduledOptim(1e-4)
    o(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    tr.step(o, DiceCoefficient(), x, y)
    g16 = m.flat_grads.clone() / tr.last_grad_scale      # (the buffer keeps the loss scale: Adam un-scales as it reads)
    tr.settle()
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    rel = float((g16 - g32).norm() / g32.norm())
    print('float16 with loss scale 2^16: gradient cosine %.6f, relative L2 %.3e, skipped %d' % (cos, rel, tr.skipped_steps))
    assert tr.skipped_steps == 0 and cos >= 0.999 and rel <= 0.05
    # forced overflow
    before = m.flat_params.clone()
    tr.loss_scale = 2.0 ** 60
    it = o.iterations
    tr.step(o, DiceCoefficient(), x, y)
    assert o.iterations == it + 1                       # the skip happened on the device; the host has not read the flag yet
    tr.settle()
    assert tr.skipped_steps == 1 and tr.loss_scale == 2.0 ** 59 and o.iterations == it
    assert torch.equal(m.flat_params, before) and bool(torch.isfinite(m.flat_params).all())
