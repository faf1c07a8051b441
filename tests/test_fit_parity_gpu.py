"""-m gpu: SURVEY 8 f-1 -- the training-loop shell against the oracle's restatement of the reference loop.

bts_amd.train.fit() (train.py:116-216) runs 2 epochs x 3 training volumes + 2 validation volumes on the HIP engine; the
oracle (oracle/torch_ref.fit: train.py:133-181 on explicit draws, fp64) runs the same epochs on the same weights, volumes,
dropout masks and eps.  Compared per epoch: the learning rate set by optimizer(epoch) (train.py:136), the three training
means and the three validation means (forward with training=False, inference=False: train.py:167), and the final parameters.

Tolerances.  A multi-step trajectory compounds what the single-step tests bound: Adam's first steps are ~lr*sign(g) where
|g| is small, so a gradient rounding difference moves such an element by up to 2*lr per step.  The yardstick is therefore
the oracle itself evaluated in fp32 (torch-CPU, same graph): the engine may deviate from the fp64 trajectory by at most
max(stated floor, 4 x the deviation of that fp32 evaluation).  Floors: loss 2e-5 relative, Dice 1e-4, parameters 1e-5.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

KW = dict(base_filters=8, groups=4, reduction=2, depth=3)
CROP = (16, 16, 16)
LR = 1e-4
SCHEDULE_EPOCHS = 4          # steep schedule so that the per-epoch LR re-set is visible in 2 epochs
N_EPOCHS = 2
KEYS = ('train_loss', 'train_macro_dice', 'train_micro_dice', 'val_loss', 'val_macro_dice', 'val_micro_dice')


def _volumes():
    latent = KW['base_filters'] * 2 ** (KW['depth'] - 2)
    train = [R.synthetic_batch(1, CROP, latent=latent, seed=500 + i) for i in range(3)]
    val = [R.synthetic_batch(1, CROP, latent=latent, seed=600 + i) for i in range(2)]
    return train, val


def _start_params(cfg):
    P = R.build_params(cfg, CROP, seed=3)
    g = torch.Generator().manual_seed(5)
    for k in P:
        if k.endswith('gn2_g'):          # gamma_2 = 0 at init would hide the conv branch
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64)
        P[k] = P[k].float().double()
    return P


def _oracle(cfg, dtype):
    train, val = _volumes()
    P = _start_params(cfg)
    for k in P:
        P[k] = P[k].to(dtype)
    tr = [(x.to(dtype), y.to(dtype), m.to(dtype), e.to(dtype)) for x, y, m, e in train]
    va = [(x.to(dtype), y.to(dtype), e.to(dtype)) for x, y, m, e in val]
    rows, _, step = R.fit(P, cfg, tr, va, N_EPOCHS, LR, schedule_epochs=SCHEDULE_EPOCHS)
    assert step == N_EPOCHS * len(tr)
    return rows, P


def test_fit_trajectory_matches_the_oracle_loop(tmp_path):
    import bts_amd  # noqa: F401
    from bts_amd import train as T
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    cfg = R.default_config(**KW)
    rows64, P64 = _oracle(cfg, torch.float64)
    rows32, P32 = _oracle(cfg, torch.float32)

    dev = torch.device('cuda', 0)
    train, val = _volumes()
    model = Model(**KW)
    model.build((1,) + CROP + (2,))
    model.set_weights_from(_start_params(cfg))
    opt = ScheduledOptim(LR, n_epochs=SCHEDULE_EPOCHS)
    lf, df = DiceVAELoss(), DiceCoefficient()
    tr_draws = {id_: (m, e) for id_, (_, _, m, e) in enumerate(train)}
    va_draws = {id_: e for id_, (_, _, _, e) in enumerate(val)}
    count = {'t': 0, 'v': 0}

    def tstep(x, y):                       # product train_step with the oracle's draws injected (one-shot per call)
        m, e = tr_draws[count['t'] % len(train)]
        count['t'] += 1
        model.encoder.set_dropout_mask(m)
        model.vae.set_eps(e)
        return train_step(model, opt, lf, df, x, y)

    def estep(x, y):                       # product eval_step: no dropout when training=False, eps still drawn
        model.vae.set_eps(va_draws[count['v'] % len(val)])
        count['v'] += 1
        return T.eval_step(model, lf, df, x, y)

    hist = T.fit(model, opt, lf, df, [(x.to(dev), y.to(dev)) for x, y, _, _ in train],
                 [(x.to(dev), y.to(dev)) for x, y, _, _ in val], n_epochs=N_EPOCHS, patience=5,
                 save_folder=str(tmp_path), train_step_fn=tstep, eval_step_fn=estep, log=lambda s: None)
    torch.cuda.synchronize()
    assert count == {'t': 6, 'v': 4} and opt.iterations == 6 and len(hist) == N_EPOCHS
    assert model.encoder._mask is None and model.vae._eps is None        # every injection was consumed

    for h, r64, r32 in zip(hist, rows64, rows32):
        assert h['epoch'] == r64['epoch']
        assert h['lr'] == r64['lr'], (h['lr'], r64['lr'])                # float32(lr0*(1-epoch/n)^0.9), train.py:136
        for k in KEYS:
            floor = 2e-5 * max(1.0, abs(float(r64[k]))) if k.endswith('loss') else 1e-4
            tol = max(floor, 4.0 * abs(float(r32[k]) - float(r64[k])))
            d = abs(float(h[k]) - float(r64[k]))
            print('epoch %d %-16s engine %.7f oracle %.7f |d| %.2e (fp32-torch %.2e)' %
                  (h['epoch'], k, float(h[k]), float(r64[k]), d, abs(float(r32[k]) - float(r64[k]))))
            assert d <= tol, (h['epoch'], k, float(h[k]), float(r64[k]), tol)
    assert float(hist[1]['lr']) < float(hist[0]['lr'])                    # the schedule was re-set for epoch 1
    worst, worst32 = 0.0, 0.0
    for p in model.trainable_variables:
        name = model.oracle_name(p)
        worst = max(worst, float((p.t.cpu().double() - P64[name]).abs().max()))
        worst32 = max(worst32, float((P32[name].double() - P64[name]).abs().max()))
    print('final parameters: engine vs fp64 oracle max |d| %.3e ; torch-fp32 vs fp64 %.3e' % (worst, worst32))
    assert worst <= max(1e-5, 4.0 * worst32)
    # the log the loop wrote is the history, formatted like the reference's rows (train.py:186-193)
    lines = open(str(tmp_path / 'train.log')).read().strip().split('\n')
    assert lines[0] == T.LOG_HEADER and len(lines) == 1 + N_EPOCHS
    assert lines[2] == T.log_row(1, hist[1]['lr'], *[hist[1][k] for k in KEYS])
    assert np.float32(lines[2].split(',')[1]) == np.float32(LR * (1 - 1.0 / SCHEDULE_EPOCHS) ** 0.9)
