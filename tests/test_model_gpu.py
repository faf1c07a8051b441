"""-m gpu: full-model forward / backward / Adam parity of the HIP engine against the fp64 oracle, on the same seeded
inputs and weights.  Tolerances: y_pred max-abs <= 1e-4, gradients <= 1e-3 relative to the gradient's
max-abs (a ReLU mask that flips at a pre-activation within fp32 rounding of zero moves a weight gradient by one term), argmax label map bit-exact, loss and Dice within 1e-5 / 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

CONFIGS = {
    'micro': (dict(base_filters=4, groups=2, reduction=2, depth=2), (8, 8, 8), 1),
    'micro_n2': (dict(base_filters=4, groups=2, reduction=2, depth=3), (8, 16, 8), 2),
    'tiny': (dict(base_filters=16, groups=8, reduction=2, depth=3), (32, 32, 32), 1),
    'cli_small': (dict(base_filters=32, groups=8, reduction=8, depth=4), (16, 16, 16), 1),
    'cli_32': (dict(base_filters=32, groups=8, reduction=8, depth=4), (32, 32, 32), 1),
    # the largest grid the fp64 oracle finishes in test time (~1 min of host CPU for its fp64 + fp32 evaluations): at 64^3 the
    # dispatcher already picks the big-grid conv forms for the top two levels
    'cli_64': (dict(base_filters=32, groups=8, reduction=8, depth=4), (64, 64, 64), 1),
}


def randomised_params(cfg, crop, seed):
    """oracle ParamSet with every gamma/beta/bias randomised (gamma_2 = 0 at init would hide the conv branch: F6)"""
    P = R.build_params(cfg, crop, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in P:
        if k.endswith('_b') or k.endswith('gn_b') or k.endswith('gn1_b') or k.endswith('gn2_b'):
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.1
        if k.endswith('_g'):
            P[k] = 1.0 + torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.3
    for k in P:  # keep fp32-representable values so both sides see identical inputs
        P[k] = P[k].float().double()
    return P


def run_oracle(cfg, P, x, y, mask, eps, dtype=torch.float64):
    x, y, mask, eps = x.to(dtype), y.to(dtype), mask.to(dtype), eps.to(dtype)
    leaves = {k: t.clone().to(dtype).requires_grad_(True) for k, t in P.items()}
    PP = R.ParamSet()
    PP.update(leaves)
    PP.l2 = P.l2
    out = R.model(x, PP, cfg, training=True, inference=False, mask=mask, eps=eps)
    loss = R.dice_vae_loss(x, y, *out, cfg['data_format']) + R.l2_regularisation(PP)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return out, loss.detach(), dict(zip(leaves.keys(), grads))


@pytest.mark.parametrize('name', list(CONFIGS))
def test_train_step_parity(name):
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.tape import GradientTape
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, reduce_sum
    kw, crop, n = CONFIGS[name]
    cfg = R.default_config(**kw)
    latent = cfg['base_filters'] * 2 ** (cfg['depth'] - 2)
    x, y, mask, eps = R.synthetic_batch(n, crop, latent=latent, seed=1234)
    P = randomised_params(cfg, crop, seed=7)
    (yp_r, yv_r, zm_r, zl_r), loss_r, grads_r = run_oracle(cfg, P, x, y, mask, eps)
    # conditioning yardstick: the same graph evaluated by torch in fp32.  GroupNorms over 2-element groups (tiny crops)
    # amplify fp32 rounding ~1e3x; the engine may deviate from fp64 as much as a plain fp32 evaluation does, not more.
    _, _, grads_32 = run_oracle(cfg, P, x, y, mask, eps, dtype=torch.float32)

    model = Model(**kw)
    model.build((n,) + crop + (2,))
    assert model.n_params == sum(t.numel() for t in P.values())
    model.set_weights_from(P)
    model.encoder.set_dropout_mask(mask)
    model.vae.set_eps(eps)
    loss_fn, dice_fn = DiceVAELoss(), DiceCoefficient()
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(x, training=True, inference=False)
        loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)
        loss = loss + reduce_sum(model.losses)
    macro, micro = dice_fn(y, y_pred)
    grads = tape.gradient(loss, model.trainable_variables)
    torch.cuda.synchronize()

    def maxerr(a, b):
        return float((a.detach().double().cpu() - b.detach().double()).abs().max())

    e = maxerr(y_pred.t, yp_r)
    assert e <= 1e-4, 'y_pred max-abs err %.3e' % e
    e = maxerr(y_vae.t, yv_r)
    assert e <= 1e-4 * max(1.0, float(yv_r.detach().abs().max())), 'y_vae err %.3e' % e
    assert maxerr(z_mean.t, zm_r) <= 1e-4 and maxerr(z_logvar.t, zl_r) <= 1e-4
    assert abs(float(loss) - float(loss_r)) <= 1e-5 * max(1.0, abs(float(loss_r))), (float(loss), float(loss_r))
    # metric + bit-exact label map (near-threshold voxels are counted and must be zero on this fixture)
    macro_r, micro_r, labels_r = R.dice_coefficient(y.double(), yp_r, cfg['data_format'])
    top2 = yp_r.topk(2, dim=-1).values
    ambiguous = ((yp_r.max(dim=-1).values - 0.5).abs() < 1e-5) | ((top2[..., 0] - top2[..., 1]).abs() < 1e-5)
    n_amb = int(ambiguous.sum())
    print('near-threshold voxels (|p-0.5|<1e-5 or top-2 gap<1e-5):', n_amb, 'of', ambiguous.numel())
    assert n_amb <= 1e-2 * ambiguous.numel()
    lab = dice_fn.last_labels.cpu().long()
    assert torch.equal(lab[~ambiguous], labels_r.long()[~ambiguous]), 'argmax label map differs'
    assert abs(float(macro) - float(macro_r)) <= 1e-4 and abs(float(micro) - float(micro_r)) <= 1e-4
    # gradients
    worst = (0.0, None)
    # global conditioning of this configuration: worst fp32-vs-fp64 deviation of torch itself over all variables.  A variable
    # may deviate as much as torch-fp32 does on ITS worst variable: at the CLI model's 64^3 case torch-fp32 is off by > 1e-3 on
    # 22 of 260 variables (worst 7e-3), the engine on 9 (worst 5e-3, a GroupNorm beta of a 16^3 level: 4096 terms per channel,
    # one ReLU within rounding of zero moves it by a whole term) -- scripts/grad_conditioning.py prints the table
    gdev = max(maxerr(grads_32[k], grads_r[k]) / (float(grads_r[k].abs().max()) + 1e-12) for k in grads_r)
    for p, g in zip(model.trainable_variables, grads):
        assert g is not None, p.name
        gr = grads_r[model.oracle_name(p)]
        scale = float(gr.abs().max()) + 1e-12
        err = maxerr(g, gr) / scale
        if err > worst[0]:
            worst = (err, p.name)
        dev32 = maxerr(grads_32[model.oracle_name(p)], gr) / scale
        assert err <= max(1e-3, 4 * dev32, gdev) or maxerr(g, gr) <= 1e-9, \
            'grad %s rel err %.3e (fp32-torch deviates %.3e; scale %.3e)' % (p.name, err, dev32, scale)
    print('worst grad rel err', worst)
    # aggregate bound, insensitive to WHICH pre-activations sit within rounding of zero: the whole gradient in relative L2
    num = sum(float((g.detach().double().cpu() - grads_r[model.oracle_name(p)]).pow(2).sum())
              for p, g in zip(model.trainable_variables, grads)) ** 0.5
    num32 = sum(float((grads_32[k].double() - grads_r[k]).pow(2).sum()) for k in grads_r) ** 0.5
    den = sum(float(grads_r[k].pow(2).sum()) for k in grads_r) ** 0.5
    print('whole-gradient relative L2 error: engine %.3e, torch-fp32 %.3e' % (num / den, num32 / den))
    assert num / den <= max(1e-4, 2.0 * num32 / den)
    # one Adam step (TF form)
    opt = ScheduledOptim(learning_rate=1e-4)
    opt(epoch=0)
    before = {p.name: p.t.detach().cpu().double().clone() for p in model.trainable_variables}
    opt.apply_gradients(zip(grads, model.trainable_variables), model=model)
    torch.cuda.synchronize()
    for p, g in zip(model.trainable_variables, grads):
        # the Adam kernel itself: expected update from the ENGINE's gradient in fp64 (gradient parity is checked above;
        # Adam's first step ~ lr*g/(|g|+3e-6) is ill-conditioned in g for tiny |g|, so mixing the two would test neither)
        ge = g.detach().cpu().double()
        exp, _, _ = R.adam_tf_step(before[p.name], ge, torch.zeros_like(ge), torch.zeros_like(ge), 1, 1e-4)
        d = (p.t.detach().cpu().double() - exp).abs()
        assert float(d.max()) <= 1e-7 + 1e-6 * float(exp.abs().max()), (p.name, float(d.max()))


def test_inference_mode_skips_vae_and_build_call():
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    model = Model(base_filters=4, groups=2, reduction=2, depth=2)
    x = torch.zeros((1, 8, 8, 8, 2))
    out = model(x)                         # train.py:95 builds with zeros, training=None
    assert out[0].shape == (1, 8, 8, 8, 3) and out[1].shape == (1, 8, 8, 8, 2)
    y_pred, a, b, c = model(x, training=False, inference=True)
    assert a is None and b is None and c is None
    with pytest.raises(AssertionError):
        model(x, training=True, inference=True)
    # fresh block: gamma_2 == 0 so the conv branch contributes exactly 0 (SURVEY F6 / KAT 3)
    from bts_amd.layers.resnet import ResnetBlock
    blk = ResnetBlock(8, groups=2, reduction=2)
    xx = torch.randn((1, 4, 4, 4, 6))
    o1 = blk(xx).t.cpu()
    P = {k: None for k in ()}
    res = R.conv3d(xx.double(), blk.ptwise_k.t.cpu().double(), blk.ptwise_b.t.cpu().double())
    gap = res.mean(dim=(1, 2, 3))
    ch = torch.sigmoid(torch.relu(gap @ blk.se_w1.t.cpu().double()) @ blk.se_w2.t.cpu().double())
    sp = torch.sigmoid(res @ blk.spatial_k.t.cpu().double().reshape(-1, 1))
    ref = res * (sp + ch.reshape(1, 1, 1, 1, -1))
    assert float((o1.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize('name', ['micro_n2', 'tiny'])
def test_channels_first_public_layout_parity(name):
    """data_format='channels_first' (the reference's --gpu default, args.py:121-123): raw NCDHW in/out,
    true channel-group GroupNorm (F1), Dice over all spatial axes (F8 intended form), same Keras-layout weights.
    Forward, loss, metric, labels and every gradient against the fp64 oracle evaluated in channels_first."""
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.tape import GradientTape
    from bts_amd.util import DiceCoefficient, DiceVAELoss, reduce_sum
    kw, crop, n = CONFIGS[name]
    kw = dict(kw, data_format='channels_first')
    cfg = R.default_config(**kw)
    latent = cfg['base_filters'] * 2 ** (cfg['depth'] - 2)
    x, y, mask, eps = R.synthetic_batch(n, crop, latent=latent, seed=4321)
    cf = lambda t: t.permute(0, 4, 1, 2, 3).contiguous()
    x, y, mask = cf(x), cf(y), cf(mask)
    P = randomised_params(cfg, crop, seed=9)
    (yp_r, yv_r, zm_r, zl_r), loss_r, grads_r = run_oracle(cfg, P, x, y, mask, eps)
    _, _, grads_32 = run_oracle(cfg, P, x, y, mask, eps, dtype=torch.float32)
    assert yp_r.shape[1] == 3 and yv_r.shape[1] == 2      # oracle really ran channels_first

    model = Model(**kw)
    out0 = model(torch.zeros_like(x))                      # train.py:95-96 build call with an NCDHW zeros tensor
    assert out0[0].numpy().shape == tuple(yp_r.shape)
    model.set_weights_from(P)
    model.encoder.set_dropout_mask(mask)
    model.vae.set_eps(eps)
    loss_fn, dice_fn = DiceVAELoss(data_format='channels_first'), DiceCoefficient(data_format='channels_first')
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(x, training=True, inference=False)
        loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)
        loss = loss + reduce_sum(model.losses)
    macro, micro = dice_fn(y, y_pred)
    grads = tape.gradient(loss, model.trainable_variables)
    torch.cuda.synchronize()

    def maxerr(a, b):
        return float((a.detach().double().cpu() - b.detach().double()).abs().max())

    assert y_pred.cf and tuple(y_pred.public().shape) == tuple(yp_r.shape)
    assert maxerr(y_pred.public(), yp_r) <= 1e-4
    assert maxerr(y_vae.public(), yv_r) <= 1e-4 * max(1.0, float(yv_r.detach().abs().max()))
    assert maxerr(z_mean.t, zm_r) <= 1e-4 and maxerr(z_logvar.t, zl_r) <= 1e-4
    assert abs(float(loss) - float(loss_r)) <= 1e-5 * max(1.0, abs(float(loss_r)))
    # metric kernel in its channels_first form (one Dice cell per class): the oracle formula on the ENGINE's probabilities,
    # so near-tie voxels (this fixture has top-2 gaps below the 1e-4 forward tolerance) cannot blur the comparison
    macro_e, micro_e, labels_e = R.dice_coefficient(y.double(), y_pred.public().detach().cpu().double(), 'channels_first')
    assert abs(float(macro) - float(macro_e)) <= 1e-6 and abs(float(micro) - float(micro_e)) <= 1e-6
    assert torch.equal(dice_fn.last_labels.cpu().long(), labels_e.long())
    # label map against the oracle's own forward: bit-exact outside the voxels whose decision is within the forward tolerance
    _, _, labels_r = R.dice_coefficient(y.double(), yp_r, 'channels_first')
    top2 = yp_r.topk(2, dim=1).values
    ambiguous = ((yp_r.max(dim=1).values - 0.5).abs() < 2e-4) | ((top2[:, 0] - top2[:, 1]).abs() < 2e-4)
    lab = dice_fn.last_labels.cpu().long()
    assert torch.equal(lab[~ambiguous], labels_r.long()[~ambiguous]), 'argmax label map differs'
    print('voxels within 2e-4 of a decision boundary:', int(ambiguous.sum()), 'of', ambiguous.numel())
    gdev = max(maxerr(grads_32[k], grads_r[k]) / (float(grads_r[k].abs().max()) + 1e-12) for k in grads_r)
    # A conv bias in front of a GroupNormalization has the exact gradient 0 (2.7e-15 in fp64 here): what any fp32 evaluation returns is
    # the rounding residue of ~256 gradient values of the layer that cancel -- 2e-8 or 6e-5 depending on the last bits upstream (the
    # layer's kernel gradient is 12.6).  Such entries are held to 1e-5 of the model's largest gradient instead of to their own size.
    gmax = max(float(grads_r[k].abs().max()) for k in grads_r)
    for p, g in zip(model.trainable_variables, grads):
        gr = grads_r[model.oracle_name(p)]
        scale = float(gr.abs().max()) + 1e-12
        err = maxerr(g, gr) / scale
        dev32 = maxerr(grads_32[model.oracle_name(p)], gr) / scale
        assert err <= max(1e-3, 4 * dev32, gdev) or maxerr(g, gr) <= max(1e-9, 1e-5 * gmax if scale < 1e-9 * gmax else 0.0), \
            'grad %s rel err %.3e (fp32-torch deviates %.3e)' % (p.name, err, dev32)


@pytest.mark.parametrize('samplers', [dict(downsampling='max'), dict(upsampling='linear'), dict(downsampling='max', upsampling='linear')])
def test_non_default_samplers_parity(samplers):
    """--downsampling max (MaxPooling3D 2/2, keeps every channel) and --upsampling linear (1x1x1 conv +
    nearest repeat) end to end: forward, loss and every gradient against the fp64 oracle."""
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.tape import GradientTape
    from bts_amd.util import DiceVAELoss, reduce_sum
    kw = dict(base_filters=8, groups=2, reduction=2, depth=3, **samplers)
    crop, n = (16, 16, 16), 1
    cfg = R.default_config(**kw)
    latent = cfg['base_filters'] * 2 ** (cfg['depth'] - 2)
    x, y, mask, eps = R.synthetic_batch(n, crop, latent=latent, seed=77)
    P = randomised_params(cfg, crop, seed=3)
    (yp_r, yv_r, zm_r, zl_r), loss_r, grads_r = run_oracle(cfg, P, x, y, mask, eps)
    _, _, grads_32 = run_oracle(cfg, P, x, y, mask, eps, dtype=torch.float32)
    model = Model(**kw)
    model.build((n,) + crop + (2,))
    assert model.n_params == sum(t.numel() for t in P.values())
    assert sorted(model.oracle_name(p) for p in model.trainable_variables) == sorted(P.keys())
    model.set_weights_from(P)
    model.encoder.set_dropout_mask(mask)
    model.vae.set_eps(eps)
    loss_fn = DiceVAELoss()
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(x, training=True, inference=False)
        loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)
        loss = loss + reduce_sum(model.losses)
    grads = tape.gradient(loss, model.trainable_variables)
    torch.cuda.synchronize()

    def maxerr(a, b):
        return float((a.detach().double().cpu() - b.detach().double()).abs().max())

    assert maxerr(y_pred.t, yp_r) <= 1e-4
    assert maxerr(y_vae.t, yv_r) <= 1e-4 * max(1.0, float(yv_r.detach().abs().max()))
    assert abs(float(loss) - float(loss_r)) <= 1e-5 * max(1.0, abs(float(loss_r)))
    gdev = max(maxerr(grads_32[k], grads_r[k]) / (float(grads_r[k].abs().max()) + 1e-12) for k in grads_r)
    for p, g in zip(model.trainable_variables, grads):
        gr = grads_r[model.oracle_name(p)]
        scale = float(gr.abs().max()) + 1e-12
        err = maxerr(g, gr) / scale
        dev32 = maxerr(grads_32[model.oracle_name(p)], gr) / scale
        assert err <= max(1e-3, 4 * dev32, gdev) or maxerr(g, gr) <= 1e-9, \
            'grad %s rel err %.3e (fp32-torch deviates %.3e)' % (p.name, err, dev32)
