#!/usr/bin/env python3
"""Does 16-bit STORAGE train like fp32?  (round-3 verdict, missing item 5; the reference's evidence is a 300-epoch Dice curve,
README.md:81-89, on data that is not available here.)

Two copies of the CLI-default model (42,174,773 parameters) start from the same weights and see the same volumes, dropout masks and
reparameterisation draws for `--steps` Adam steps at batch 1 (the reference's batch size, train.py:60-76): one through the fp32
engine (bts_amd.util.train_step), one through bts_amd.lowp_train.LowPrecisionTrainer.  Data: `--train` synthetic training volumes
and `--heldout` held-out ones at `--crop`^3.  The generator of SURVEY 8(d) draws pure noise inside the brain ellipsoid, from which
nothing can be learnt; here the three nested label spheres also carry a per-class, per-channel intensity offset (like the contrast
of enhancing core / oedema on T1ce / FLAIR), so the Dice of a held-out volume measures what the network has learnt.

Printed / returned: the two loss curves (mean over each window of `--every` steps) and the held-out macro Dice (util.DiceCoefficient
on the forward with training=False, each model through its own engine) at every checkpoint.  tests/test_lowp_trajectory_gpu.py
bounds the final Dice gap and the final loss gap.

usage (GPU box): python scripts/lp_trajectory.py [--dtype bfloat16] [--steps 200] [--crop 64] [--lr 1e-4] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRAST = ((0.8, 0.3), (0.2, 1.0), (1.4, -0.7))     # class k adds CONTRAST[k] to the two channels of its voxels


def volumes(n, crop, seed):
    import torch
    from bts_amd.data import synthetic_batch
    x, y, _, _ = synthetic_batch(n, crop, latent=128, seed=seed)
    c = torch.tensor(CONTRAST, dtype=torch.float32)
    return (x + y @ c).contiguous(), y


def run(dtype='bfloat16', steps=200, crop=64, n_train=8, n_heldout=2, lr=1e-4, every=25, seed=77, log=print):
    import torch
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    from bts_amd.layers import _base
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.model import Model
    from bts_amd.tape import Tensor, bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    dev = torch.device('cuda', torch.cuda.current_device())
    kw = dict(base_filters=32, reduction=8, depth=4, groups=8)
    shape = (1,) + (crop,) * 3 + (2,)
    _base.set_seed(seed)
    m32 = Model(**kw)
    m32.build(shape)
    m16 = Model(**kw)
    m16.build(shape)
    m16.flat_params.copy_(m32.flat_params)
    bump_weights_epoch()
    for a, b in ((m32.encoder, m16.encoder), (m32.vae, m16.vae)):      # the same dropout / eps counters -> the same draws
        b._seed = a._seed
    xt, yt = volumes(n_train, (crop,) * 3, seed)
    xh, yh = volumes(n_heldout, (crop,) * 3, seed + 1)
    xt, yt, xh, yh = xt.to(dev), yt.to(dev), xh.to(dev), yh.to(dev)
    o32, o16 = ScheduledOptim(lr), ScheduledOptim(lr)
    o32(epoch=0)
    o16(epoch=0)
    tr = LowPrecisionTrainer(m16, dtype)
    fwd16 = lowp.LowPrecisionForward(m16, dtype)
    lf = DiceVAELoss()

    def heldout():
        d32, d16 = [], []
        for k in range(n_heldout):
            x, y = xh[k:k + 1], yh[k:k + 1]
            yp = m32(x, training=False, inference=True)[0]
            d32.append(float(DiceCoefficient()(Tensor(y, requires_grad=False), yp)[0]))
            yp16 = fwd16(x)
            d16.append(float(DiceCoefficient()(Tensor(y, requires_grad=False), Tensor(yp16, requires_grad=False))[0]))
        return sum(d32) / len(d32), sum(d16) / len(d16)

    out = {'dtype': dtype, 'steps': steps, 'crop': crop, 'lr': lr, 'n_train': n_train, 'n_heldout': n_heldout, 'checkpoints': []}
    w32, w16 = [], []
    d = heldout()
    out['checkpoints'].append({'step': 0, 'heldout_macro_dice_f32': d[0], 'heldout_macro_dice_16': d[1]})
    log('step    0: held-out macro Dice fp32 %.4f  %s %.4f' % (d[0], dtype, d[1]))
    for s in range(steps):
        k = s % n_train
        x, y = xt[k:k + 1], yt[k:k + 1]
        l32, _, _ = train_step(m32, o32, lf, DiceCoefficient(), x, y)
        l16, _, _ = tr.step(o16, DiceCoefficient(), x, y)
        w32.append(float(l32))
        w16.append(float(l16))
        if (s + 1) % every == 0 or s + 1 == steps:
            d = heldout()
            a, b = sum(w32) / len(w32), sum(w16) / len(w16)
            out['checkpoints'].append({'step': s + 1, 'train_loss_f32': a, 'train_loss_16': b, 'heldout_macro_dice_f32': d[0],
                                       'heldout_macro_dice_16': d[1]})
            log('step %4d: train loss (window mean) fp32 %.4f  %s %.4f | held-out macro Dice fp32 %.4f  %s %.4f'
                % (s + 1, a, dtype, b, d[0], dtype, d[1]))
            w32, w16 = [], []
    last = out['checkpoints'][-1]
    out['final_dice_gap'] = abs(last['heldout_macro_dice_f32'] - last['heldout_macro_dice_16'])
    out['final_loss_gap_rel'] = abs(last['train_loss_f32'] - last['train_loss_16']) / abs(last['train_loss_f32'])
    out['dice_gain_f32'] = last['heldout_macro_dice_f32'] - out['checkpoints'][0]['heldout_macro_dice_f32']
    out['skipped_steps'] = tr.skipped_steps
    out['param_rel_l2'] = float((m16.flat_params - m32.flat_params).norm() / (m32.flat_params - _initial(m32, kw, shape, seed)).norm())
    log('final: Dice gap %.4f, loss gap %.2e relative, fp32 Dice gain over the run %.4f, |p16 - p32| / |p32 - p0| = %.3f, skipped steps %d'
        % (out['final_dice_gap'], out['final_loss_gap_rel'], out['dice_gain_f32'], out['param_rel_l2'], out['skipped_steps']))
    return out


def _initial(model, kw, shape, seed):
    """the starting point again (same seed -> same initialiser draws), to express the 16-bit run's drift in units of the distance travelled"""
    from bts_amd.layers import _base
    from bts_amd.model import Model
    _base.set_seed(seed)
    m0 = Model(**kw)
    m0.build(shape)
    return m0.flat_params


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bfloat16')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--crop', type=int, default=64)
    ap.add_argument('--train', type=int, default=8)
    ap.add_argument('--heldout', type=int, default=2)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--every', type=int, default=25)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    res = run(a.dtype, a.steps, a.crop, a.train, a.heldout, a.lr, a.every)
    if a.json:
        json.dump(res, open(a.json, 'w'), indent=1)
