"""-m gpu: one training step with 16-bit storage (bts_amd.lowp_train; BASELINE configs[2] is bf16, batch 8) against the fp32
engine's step on the same weights, volumes, dropout mask and eps.  The reference has no such mode: the fp32
engine is the parity reference, and what is stated is how far 16-bit storage moves the step:
loss, Dice, label map, the gradient (whole-buffer relative L2 and cosine; per-variable cosine) and the parameters after Adam."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(base_filters=16, groups=8, reduction=2, depth=3)
CROP = (32, 32, 32)
N = 2


def _setup(seed=3, **model_kw):
    import bts_amd  # noqa: F401
    from bts_amd.data import synthetic_batch
    from bts_amd.layers import _base
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    _base.set_seed(seed)
    m = Model(**dict(KW, **model_kw))
    m.build((N,) + CROP + (2,))
    g = torch.Generator().manual_seed(seed + 1)
    for p in m.trainable_variables:
        if p.name.endswith('gamma'):
            p.t.copy_((1.0 + 0.3 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
        elif p.name.endswith('beta') or p.t.dim() == 1:
            p.t.copy_((0.1 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
    bump_weights_epoch()
    latent = KW['base_filters'] * 2 ** (KW['depth'] - 2)
    x, y, mask, eps = synthetic_batch(N, CROP, latent=latent, seed=99)
    return m, x, y, mask, eps


@pytest.mark.parametrize('dtype,lim', [('bfloat16', dict(loss=5e-3, l2=0.12, cos=0.99, var_cos=0.97)),
                                       ('float16', dict(loss=5e-4, l2=0.04, cos=0.999, var_cos=0.995))])
def test_step_against_the_fp32_engine(dtype, lim):
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    m, x, y, mask, eps = _setup()
    start = m.flat_params.clone()
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    df32 = DiceCoefficient()
    loss32, macro32, _ = train_step(m, opt, DiceVAELoss(), df32, x, y)
    torch.cuda.synchronize()
    g32, p32, lab32 = m.flat_grads.clone(), m.flat_params.clone(), df32.last_labels.clone()
    # same start, same draws, 16-bit storage
    from bts_amd.tape import bump_weights_epoch
    m.flat_params.copy_(start)
    bump_weights_epoch()
    opt2 = ScheduledOptim(1e-4)
    opt2(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    tr = LowPrecisionTrainer(m, dtype)
    df16 = DiceCoefficient()
    loss16, macro16, _ = tr.step(opt2, df16, x, y)
    torch.cuda.synchronize()
    g16, p16 = m.flat_grads.clone() / tr.last_grad_scale, m.flat_params.clone()      # (fp16: the buffer keeps the loss scale, Adam un-scales)
    dl = abs(float(loss16) - float(loss32)) / abs(float(loss32))
    rel = float((g16 - g32).norm() / g32.norm())
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    mism = float((df16.last_labels != lab32).float().mean())
    rows = []
    for p in m.trainable_variables:
        off = (p._gview.data_ptr() - m.flat_grads.data_ptr()) // 4
        a, b = g16[off:off + p._gview.numel()], g32[off:off + p._gview.numel()]
        if float(b.norm()) > 1e-12:
            rows.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), p.name, float(b.norm()), float(a.norm())))
    rows.sort()
    tot = float(g32.norm())
    for r in rows[:6]:
        print('   cosine %.4f  %-34s |g32| %.3e (%.2f %% of the gradient norm) |g16| %.3e' % (r[0], r[1], r[2], 100 * r[2] / tot, r[3]))
    # small gradients that are sums of cancelling terms over every voxel (a block's spatial-gate vector, the SE MLP of a deep
    # block) are dominated by the rounding of the stored activations: the per-variable bound covers the variables that carry
    # the gradient (>= 2 % of its norm), the whole-buffer figures cover the rest
    heavy = [r for r in rows if r[2] >= 0.02 * tot]
    worst = heavy[0][:2]
    print('   %d of %d variables carry >= 2 %% of the gradient norm; worst cosine among them %.4f (%s)' % (len(heavy), len(rows), worst[0], worst[1]))
    moved = float((p32 - start).abs().max())
    dpar = float((p16 - p32).abs().max())
    print('%s: loss %.6f vs %.6f (rel %.2e), macro Dice %.5f vs %.5f, label changes %.3f %%; gradient rel L2 %.3e cosine %.6f; '
          'worst variable cosine %.4f (%s); parameters moved %.2e, differ by %.2e' %
          (dtype, float(loss16), float(loss32), dl, float(macro16), float(macro32), 100 * mism, rel, cos, worst[0], worst[1], moved, dpar))
    assert dl <= lim['loss'] and abs(float(macro16) - float(macro32)) <= 5e-3 and mism <= 1e-2
    assert rel <= lim['l2'] and cos >= lim['cos'] and worst[0] >= lim['var_cos']
    assert dpar <= 2.0 * moved            # Adam's first step is ~lr*sign(g): a flipped sign moves a parameter by 2 lr at most


def test_bf16_training_tracks_the_fp32_trajectory():
    """six optimiser steps on two fixed volumes: the bf16-storage run must learn (loss falls) and stay next to the fp32 run"""
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    m, x, y, mask, eps = _setup(seed=8)
    start = m.flat_params.clone()
    runs = {}
    for mode in ('fp32', 'bf16'):
        m.flat_params.copy_(start)
        bump_weights_epoch()
        opt = ScheduledOptim(1e-3)
        opt(epoch=0)
        tr = LowPrecisionTrainer(m, 'bfloat16') if mode == 'bf16' else None
        df = DiceCoefficient()
        losses = []
        for _ in range(6):
            m.encoder.set_dropout_mask(mask)
            m.vae.set_eps(eps)
            loss = tr.step(opt, df, x, y)[0] if tr else train_step(m, opt, DiceVAELoss(), df, x, y)[0]
            losses.append(float(loss))
        torch.cuda.synchronize()
        runs[mode] = losses
    print('fp32 losses', ['%.5f' % v for v in runs['fp32']])
    print('bf16 losses', ['%.5f' % v for v in runs['bf16']])
    assert runs['bf16'][-1] < runs['bf16'][0] - 0.01 and runs['fp32'][-1] < runs['fp32'][0] - 0.01
    # (lr 1e-3, ten times the default: Adam's early steps are ~lr*sign(g), so rounding noise in small gradients moves the two runs
    #  apart by up to ~1 % of the loss on the way down; measured: 1.64430/1.64448 ... 1.19224/1.18136 ... 1.02990/1.02862)
    assert all(abs(a - b) <= 2e-2 * abs(b) for a, b in zip(runs['bf16'], runs['fp32']))


# ---- the non-default options of args.py:121-141 on the 16-bit step ---------------------------------------------------
OPTIONS = [dict(data_format='channels_first'), dict(downsampling='max', upsampling='linear'),
           dict(data_format='channels_first', downsampling='max', upsampling='linear')]


def _public(t, cf):
    return t.permute(0, 4, 1, 2, 3).contiguous() if cf else t


# fp16 storage pins the GRAPH of every option (the same bounds as the default graph's fp16 case).  bf16 storage is bounded like the
# default graph where the activations stay normalised (channels_first; the linear up-sampler).  Max pooling is followed by no
# normalisation: the untrained net's encoder activations grow to ~2e2 by the top level (scripts/lp_opts_debug.py prints them), bf16's 8
# bits leave ~3 % of that at single voxels, and the gradient of that ill-conditioned net moves accordingly -- measured rel L2 0.10
# (max + linear) and 1.1-1.7 (channels_first + max), against 0.02 in fp16 on the same graph: stated, bounded only where it is small.
STEP_LIMITS = {'float16': dict(loss=5e-4, l2=0.04, cos=0.999, var_cos=0.99, labels=1e-2),
               'float16-max': dict(loss=5e-4, l2=0.07, cos=0.998, var_cos=0.9, labels=1e-2),
               'bfloat16': dict(loss=5e-3, l2=0.12, cos=0.99, var_cos=0.97, labels=1e-2),
               'bfloat16-max': dict(loss=5e-3, l2=0.15, cos=0.99, var_cos=0.90, labels=2e-2),
               'bfloat16-cf-max': dict(loss=5e-3, l2=None, cos=None, var_cos=None, labels=2e-2)}


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('opts', OPTIONS, ids=lambda o: '-'.join('%s' % v for v in o.values()))
def test_options_step_against_the_fp32_engine(opts, dtype):
    """channels_first (NCDHW volumes in, GroupNorm over true channel groups, per-class Dice), MaxPooling3D down-sampling and
    1x1x1-conv + repeat up-sampling through the 16-bit-storage step"""
    key = dtype
    if opts.get('downsampling') == 'max':
        key = dtype + ('-cf-max' if (dtype == 'bfloat16' and opts.get('data_format') == 'channels_first') else '-max')
    lim = STEP_LIMITS[key]
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    m, x, y, mask, eps = _setup(seed=5, **opts)
    df = opts.get('data_format', 'channels_last')
    cf = df == 'channels_first'
    x, y, mask = _public(x, cf), _public(y, cf), _public(mask, cf)
    start = m.flat_params.clone()
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    d32 = DiceCoefficient(data_format=df)
    loss32, macro32, micro32 = train_step(m, opt, DiceVAELoss(data_format=df), d32, x, y)
    torch.cuda.synchronize()
    g32, p32, lab32 = m.flat_grads.clone(), m.flat_params.clone(), d32.last_labels.clone()
    m.flat_params.copy_(start)
    bump_weights_epoch()
    opt2 = ScheduledOptim(1e-4)
    opt2(epoch=0)
    m.encoder.set_dropout_mask(mask)
    m.vae.set_eps(eps)
    tr = LowPrecisionTrainer(m, dtype)
    d16 = DiceCoefficient(data_format=df)
    loss16, macro16, micro16 = tr.step(opt2, d16, x, y)
    torch.cuda.synchronize()
    g16, p16 = m.flat_grads.clone() / tr.last_grad_scale, m.flat_params.clone()      # (fp16: the buffer keeps the loss scale, Adam un-scales)
    dl = abs(float(loss16) - float(loss32)) / abs(float(loss32))
    rel = float((g16 - g32).norm() / g32.norm())
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    mism = float((d16.last_labels != lab32).float().mean())
    worst = (2.0, '')
    tot = float(g32.norm())
    for p in m.trainable_variables:
        off = (p._gview.data_ptr() - m.flat_grads.data_ptr()) // 4
        a, b = g16[off:off + p._gview.numel()], g32[off:off + p._gview.numel()]
        if float(b.norm()) >= 0.02 * tot:
            worst = min(worst, (float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), p.name))
    moved = float((p32 - start).abs().max())
    dpar = float((p16 - p32).abs().max())
    print('%s %s: loss %.6f vs %.6f (rel %.2e), macro Dice %.5f vs %.5f, micro %.5f vs %.5f, label changes %.3f %%; gradient rel L2 %.3e '
          'cosine %.6f; worst heavy variable cosine %.4f (%s)' % (opts, dtype, float(loss16), float(loss32), dl, float(macro16), float(macro32),
                                                                  float(micro16), float(micro32), 100 * mism, rel, cos, worst[0], worst[1]))
    assert dl <= lim['loss'] and abs(float(macro16) - float(macro32)) <= 5e-3 and mism <= lim['labels']
    if lim['l2'] is not None:
        assert rel <= lim['l2'] and cos >= lim['cos'] and worst[0] >= lim['var_cos']
    assert dpar <= 2.0 * moved
    # every parameter of the optional layers received a gradient (none was skipped by the explicit backward)
    for p in m.trainable_variables:
        off = (p._gview.data_ptr() - m.flat_grads.data_ptr()) // 4
        n32 = float(g32[off:off + p._gview.numel()].norm())
        if n32 > 1e-10:
            assert float(g16[off:off + p._gview.numel()].norm()) > 0.0, p.name


def test_bf16_channels_first_max_pooling_gradient_on_a_conditioned_net():
    """The one option combination whose bf16 gradient STEP_LIMITS leaves unbounded (channels_first + max pooling: the untrained net's
    activations reach ~2e2 behind the un-normalised pooling and bf16 moves that ill-conditioned gradient by O(1)) -- bounded here on
    the same graph with conv kernels at 0.4 x their initial scale, where the activations stay O(1): a fault in the channel-group
    GroupNorm path or in bts_lp_maxpool2_bwd under bf16 would show as a wrong gradient, not as rounding.  Stated: whole-gradient
    relative L2 <= 0.25, cosine >= 0.97, every variable carrying >= 2 % of the norm at cosine >= 0.9."""
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    opts = dict(data_format='channels_first', downsampling='max')
    m, x, y, mask, eps = _setup(seed=5, **opts)
    for p in m.trainable_variables:
        if p.t.dim() > 1:
            p.t.mul_(0.4)
    bump_weights_epoch()
    x, y, mask = _public(x, True), _public(y, True), _public(mask, True)
    start = m.flat_params.clone()
    grads = []
    for dtype in (None, 'bfloat16'):
        m.flat_params.copy_(start)
        bump_weights_epoch()
        opt = ScheduledOptim(1e-4)
        opt(epoch=0)
        m.encoder.set_dropout_mask(mask)
        m.vae.set_eps(eps)
        d = DiceCoefficient(data_format='channels_first')
        if dtype is None:
            train_step(m, opt, DiceVAELoss(data_format='channels_first'), d, x, y)
            grads.append(m.flat_grads.clone())
        else:
            tr = LowPrecisionTrainer(m, dtype)
            tr.step(opt, d, x, y)
            grads.append(m.flat_grads.clone() / tr.last_grad_scale)
        torch.cuda.synchronize()
    g32, g16 = grads
    rel = float((g16 - g32).norm() / g32.norm())
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    worst, tot = (2.0, ''), float(g32.norm())
    for p in m.trainable_variables:
        off = (p._gview.data_ptr() - m.flat_grads.data_ptr()) // 4
        a, b = g16[off:off + p._gview.numel()], g32[off:off + p._gview.numel()]
        if float(b.norm()) >= 0.02 * tot:
            worst = min(worst, (float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), p.name))
    print('bf16, channels_first + max pooling, kernels x 0.4: gradient rel L2 %.3e cosine %.6f; worst heavy variable cosine %.4f (%s)'
          % (rel, cos, worst[0], worst[1]))
    assert rel <= 0.25 and cos >= 0.97 and worst[0] >= 0.9


# (the single-voxel maximum behind max pooling moves with every rounding-level change of the element-wise passes: 0.12 .. 0.17 seen in fp16)
@pytest.mark.parametrize('dtype,tol', [('float16', (2e-2, 1e-3, 0.3, 1e-3)), ('bfloat16', (1.6e-1, 4e-3, 0.9, 8e-3))])
@pytest.mark.parametrize('opts', OPTIONS, ids=lambda o: '-'.join('%s' % v for v in o.values()))
def test_options_forward_against_the_fp32_engine(opts, dtype, tol):
    """inference graph (lowp.LowPrecisionForward) under the options: (max, mean) |dy_pred| bounds -- the default graph's where the
    activations stay normalised, wider single-voxel bounds behind max pooling (see STEP_LIMITS above; the mean stays small)"""
    from bts_amd.lowp import LowPrecisionForward
    m, x, y, mask, eps = _setup(seed=6, **opts)
    cf = opts.get('data_format', 'channels_last') == 'channels_first'
    x = _public(x, cf)
    ref = m(x, training=False, inference=True)[0].public()
    got = LowPrecisionForward(m, dtype)(x)
    assert tuple(got.shape) == tuple(ref.shape)
    err, mean = float((got - ref).abs().max()), float((got - ref).abs().mean())
    print('%s %s: |dy_pred| max %.3e mean %.3e' % (opts, dtype, err, mean))
    mx_tol, mean_tol = tol[2:] if opts.get('downsampling') == 'max' else tol[:2]
    assert err <= mx_tol and mean <= mean_tol


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
def test_step_without_the_normalised_tensor_is_the_same_step(dtype, monkeypatch):
    """at the levels the streaming kernels take (here: all of a 64^3 crop's first level), conv2's forward and weight gradient normalise
    conv1's raw output themselves and relu(GN1(c1)) is never written (lowp_train fuse_gn1_apply): same arithmetic element for element --
    loss, label map and EVERY gradient bit-equal to the step with the separate apply pass (BTS_LP_FUSE_GN1_APPLY=0)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    from bts_amd.data import synthetic_batch
    from bts_amd.layers import _base
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, ScheduledOptim
    crop = (64, 64, 64)
    x, y, mask, eps = synthetic_batch(1, crop, latent=KW['base_filters'] * 2 ** (KW['depth'] - 2), seed=7)
    taken = []
    real = lowp.conv_bwd_weight_normed_input

    def spy(*a, **k):
        taken.append(1)
        return real(*a, **k)
    monkeypatch.setattr(lowp, 'conv_bwd_weight_normed_input', spy)

    def run(switch):
        monkeypatch.setenv('BTS_LP_FUSE_GN1_APPLY', switch)
        _base.set_seed(5)
        m = Model(**KW)
        m.build((1,) + crop + (2,))
        g = torch.Generator().manual_seed(6)
        for p in m.trainable_variables:
            if p.name.endswith('gamma'):
                p.t.copy_((1.0 + 0.3 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
        bump_weights_epoch()
        opt = ScheduledOptim(1e-4)
        opt(epoch=0)
        m.encoder.set_dropout_mask(mask)
        m.vae.set_eps(eps)
        tr = LowPrecisionTrainer(m, dtype)
        df = DiceCoefficient()
        loss, _, _ = tr.step(opt, df, x, y)
        torch.cuda.synchronize()
        return float(loss), m.flat_grads.clone() / tr.last_grad_scale, df.last_labels.clone()
    l1, g1, lab1 = run('1')
    assert len(taken) >= 3, 'the fused route was not taken by the first-level blocks'
    n_taken = len(taken)
    l0, g0, lab0 = run('0')
    assert len(taken) == n_taken
    assert l1 == l0 and torch.equal(lab1, lab0)
    assert torch.equal(g1, g0), float((g1 - g0).abs().max())
