"""Golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py; restatement-derived, see its header).
CPU: both oracle restatements still reproduce them.  GPU (-m gpu): the HIP engine, through the C ABI, reproduces them
within the stated fp32 tolerances without running the oracle at all."""
import os

import numpy as np
import pytest
import torch

from oracle import np_ref as NP
from oracle import torch_ref as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


def test_oracles_reproduce_ops_vectors():
    z = np.load(os.path.join(G, 'ops_vectors.npz'))
    x = _t(z['x'])
    for name, stride in (('conv_k3s1', 1), ('conv_k3s2', 2), ('conv_k1', 1)):
        y = R.conv3d(x, _t(z[name + '_w']), _t(z[name + '_b']), stride).numpy()
        assert np.abs(y - z[name + '_y']).max() <= 1e-12
    assert np.abs(NP.conv3d(z['x'].astype(np.float64), z['conv_k3s2_w'], z['conv_k3s2_b'], 2) - z['conv_k3s2_y']).max() <= 1e-10
    yt = R.conv3d_transpose(x[:, :3, :4, :4], _t(z['convT_w']), _t(z['convT_b'])).numpy()
    assert np.abs(yt - z['convT_y']).max() <= 1e-12
    ys = R.group_norm(_t(z['gn_x']), _t(z['gn_gamma']), _t(z['gn_beta']), 8, -1).numpy()
    assert np.abs(ys - z['gn_slab_y']).max() <= 1e-12
    assert np.abs(NP.group_norm_channel(z['gn_x'], z['gn_gamma'], z['gn_beta'], 8) - z['gn_channel_y']).max() <= 1e-10


def _micro():
    z = np.load(os.path.join(G, 'model_micro.npz'))
    kw = dict(base_filters=4, groups=2, reduction=2, depth=2)
    P = R.ParamSet()
    for k in z.files:
        if k.startswith('P/'):
            P[k[2:]] = _t(z[k])
    ref = R.build_params(R.default_config(**kw), (8, 8, 8))
    P.l2 = ref.l2
    return z, kw, P


def test_oracle_reproduces_model_vector():
    z, kw, P = _micro()
    cfg = R.default_config(**kw)
    out = R.model(_t(z['x']), P, cfg, training=True, inference=False, mask=_t(z['mask']), eps=_t(z['eps']))
    assert float((out[0] - _t(z['y_pred'])).abs().max()) <= 1e-12
    assert float((out[1] - _t(z['y_vae'])).abs().max()) <= 1e-12
    loss = R.dice_vae_loss(_t(z['x']), _t(z['y']), *out) + R.l2_regularisation(P)
    assert abs(float(loss) - float(z['loss'])) <= 1e-12
    _, _, labels = R.dice_coefficient(_t(z['y']), out[0])
    assert np.array_equal(labels.numpy().astype(np.uint8), z['labels'])


@pytest.mark.gpu
def test_engine_reproduces_ops_vectors():
    import bts_amd  # noqa: F401
    from bts_amd import ops
    D = torch.device('cuda:0')
    z = np.load(os.path.join(G, 'ops_vectors.npz'))
    x = torch.from_numpy(z['x']).float().to(D)
    for name, kind, cin, cout in (('conv_k3s1', ops.K3S1, 5, 7), ('conv_k3s2', ops.K3S2, 5, 4), ('conv_k1', ops.K1, 5, 6)):
        w = torch.from_numpy(z[name + '_w']).float().to(D)
        b = torch.from_numpy(z[name + '_b']).float().to(D)
        y = ops.conv_fwd(kind, x, ops.conv_pack(kind, ops.ROLE_FWD, w, cin, cout), b, cout).cpu().double().numpy()
        assert np.abs(y - z[name + '_y']).max() <= 2e-5 * np.abs(z[name + '_y']).max(), name
    w = torch.from_numpy(z['convT_w']).float().to(D)
    b = torch.from_numpy(z['convT_b']).float().to(D)
    xt = x[:, :3, :4, :4].contiguous()
    y = ops.conv_fwd(ops.K3S2T, xt, ops.conv_pack(ops.K3S2T, ops.ROLE_FWD, w, 5, 4), b, 4).cpu().double().numpy()
    assert np.abs(y - z['convT_y']).max() <= 2e-5 * np.abs(z['convT_y']).max()
    xg = torch.from_numpy(z['gn_x']).float().to(D)
    gam, bet = torch.from_numpy(z['gn_gamma']).float().to(D), torch.from_numpy(z['gn_beta']).float().to(D)
    for mode, key in ((ops.GN_SLAB, 'gn_slab_y'), (ops.GN_CHANNEL, 'gn_channel_y')):
        mean, rstd = ops.gn_stats(xg, 8, mode)
        y = ops.gn_apply(xg, gam, bet, mean, rstd, 8, mode, False).cpu().double().numpy()
        assert np.abs(y - z[key]).max() <= 2e-5 * max(1.0, np.abs(z[key]).max()), key


@pytest.mark.gpu
def test_engine_reproduces_model_vector():
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.tape import GradientTape
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, reduce_sum
    z, kw, P = _micro()
    model = Model(**kw)
    model.build((2, 8, 8, 8, 2))
    model.set_weights_from(P)
    model.encoder.set_dropout_mask(z['mask'])
    model.vae.set_eps(z['eps'])
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['y'])
    loss_fn, dice_fn = DiceVAELoss(), DiceCoefficient()
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(x, training=True, inference=False)
        loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)
        loss = loss + reduce_sum(model.losses)
    macro, micro = dice_fn(y, y_pred)
    grads = tape.gradient(loss, model.trainable_variables)
    assert np.abs(y_pred.numpy() - z['y_pred']).max() <= 1e-4          # stated fp32 tolerance
    assert np.abs(y_vae.numpy() - z['y_vae']).max() <= 1e-4 * max(1.0, np.abs(z['y_vae']).max())
    assert np.abs(z_mean.numpy() - z['z_mean']).max() <= 1e-4 and np.abs(z_logvar.numpy() - z['z_logvar']).max() <= 1e-4
    assert abs(float(loss) - float(z['loss'])) <= 1e-5
    assert np.array_equal(dice_fn.last_labels.cpu().numpy(), z['labels']), 'argmax label map must be bit-exact'
    assert abs(float(macro) - float(z['macro'])) <= 1e-4 and abs(float(micro) - float(z['micro'])) <= 1e-4
    for p, g in zip(model.trainable_variables, grads):
        ref = z['G/' + model.oracle_name(p)]
        sc = np.abs(ref).max() + 1e-12
        assert np.abs(g.cpu().double().numpy() - ref).max() <= 5e-4 * sc + 1e-9, p.name
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    opt.apply_gradients(zip(grads, model.trainable_variables), model=model)
    for p in model.trainable_variables:
        ref = z['A/' + model.oracle_name(p)]
        big = np.abs(z['G/' + model.oracle_name(p)]) > 1e-3 * (np.abs(z['G/' + model.oracle_name(p)]).max() + 1e-30)
        d = np.abs(p.t.cpu().double().numpy() - ref)
        assert (d[big].max() if big.any() else 0.0) <= 5e-6, p.name
