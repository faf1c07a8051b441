"""CPU: analytic known-answer tests that pin the oracle without TensorFlow (SURVEY 8c list (1)-(13)).
The reference has no tests or golden vectors (parity unpinned), so these closed-form cases are the anchor."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ref as NP
from oracle import torch_ref as R


def test_gn_slab_constant_chunks_gives_beta_pattern():
    # (1) input constant within each contiguous 1/G chunk => x_hat = 0 => output = beta[g*C/G + c mod C/G]
    n, d, h, w, c, g = 2, 4, 2, 2, 8, 4
    x = torch.zeros(n, d * h * w * c, dtype=torch.float64)
    L = x.shape[1] // g
    for k in range(g):
        x[:, k * L:(k + 1) * L] = float(k + 1)
    x = x.reshape(n, d, h, w, c)
    gamma = torch.arange(1, c + 1, dtype=torch.float64)
    beta = torch.arange(10, 10 + c, dtype=torch.float64)
    y = R.group_norm(x, gamma, beta, g, -1).reshape(n, -1)
    cg = c // g
    for k in range(g):
        cidx = (torch.arange(k * L, (k + 1) * L) % c) % cg
        assert torch.allclose(y[0, k * L:(k + 1) * L], beta[k * cg + cidx])


def test_gn_slab_is_not_channel_groupnorm_but_channels_first_is():
    # (1)/(2) SURVEY F1: channels_last path == contiguous-chunk norm, channels_first path == textbook GroupNorm
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 4, 4, 8, generator=g, dtype=torch.float64)
    gamma = torch.randn(8, generator=g, dtype=torch.float64)
    beta = torch.randn(8, generator=g, dtype=torch.float64)
    y_last = R.group_norm(x, gamma, beta, 4, -1)
    xcf = x.permute(0, 4, 1, 2, 3).contiguous()
    y_first = R.group_norm(xcf, gamma, beta, 4, 1)
    textbook = F.group_norm(xcf, 4, gamma, beta, eps=1e-5)
    assert torch.allclose(y_first, textbook, atol=1e-10)
    assert not torch.allclose(y_last.permute(0, 4, 1, 2, 3), textbook, atol=1e-3)
    chunks = y_last.reshape(2, 4, -1)  # before affine each chunk has mean 0 / var 1: check with gamma=1, beta=0
    y0 = R.group_norm(x, torch.ones(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64), 4, -1).reshape(2, 4, -1)
    assert torch.allclose(y0.mean(-1), torch.zeros(2, 4, dtype=torch.float64), atol=1e-10)
    assert torch.allclose((y0 ** 2).mean(-1), torch.ones(2, 4, dtype=torch.float64), atol=1e-3)
    assert chunks.shape == (2, 4, 128)


def test_fresh_resnet_block_is_gated_shortcut_only():
    # (3) gamma_2 = 0 at init (resnet.py:104-110) => out == res*(sigmoid(sp)+ch) exactly
    cfg = R.default_config(base_filters=8, groups=4, reduction=2, depth=2)
    P = R.build_params(cfg, (8, 8, 8), seed=1)
    pre = 'encoder/L0/B0/'
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 4, 4, 2, generator=g, dtype=torch.float64)
    out = R.resnet_block(x, P, pre, cfg)
    res = R.conv3d(x, P[pre + 'ptwise_k'], P[pre + 'ptwise_b'])
    ch = torch.sigmoid(torch.relu(res.mean(dim=(1, 2, 3)) @ P[pre + 'se_w1']) @ P[pre + 'se_w2'])
    sp = torch.sigmoid(res @ P[pre + 'spatial_k'][0, 0, 0])
    assert torch.allclose(out, res * (sp + ch.reshape(1, 1, 1, 1, -1)), rtol=1e-12, atol=1e-14)


def test_stride2_conv_window_is_2o_to_2o_plus_2_with_end_padding():
    # (4) SURVEY F7: TF SAME, k=3, s=2, even size: y[o] = sum_k x[2o+k] w[k]; x[n] == 0
    x = torch.zeros(1, 8, 1, 1, 1, dtype=torch.float64)
    w = torch.tensor([1.0, 10.0, 100.0], dtype=torch.float64).reshape(3, 1, 1, 1, 1)
    kern = torch.zeros(3, 3, 3, 1, 1, dtype=torch.float64)
    kern[:, 0, 0, 0, 0] = w.flatten()   # with H=W=1 'same' pads (1,1)->centre... use tap index 1 on the unit axes
    kern = torch.zeros(3, 3, 3, 1, 1, dtype=torch.float64)
    kern[:, 1, 1, 0, 0] = torch.tensor([1.0, 10.0, 100.0], dtype=torch.float64)
    for pos in range(8):
        x.zero_()
        x[0, pos, 0, 0, 0] = 1.0
        y = R.conv3d(x, kern, None, stride=2).flatten()
        exp = torch.zeros(4, dtype=torch.float64)
        for o in range(4):
            k = pos - 2 * o
            if 0 <= k <= 2:
                exp[o] = [1.0, 10.0, 100.0][k]
        assert torch.equal(y, exp), (pos, y, exp)
    # H=W=1 are odd sizes: pad (1,1) => the centre tap (index 1) is the one aligned with the voxel, as used above
    ynp = NP.conv3d(x.numpy(), kern.numpy(), None, 2).flatten()
    assert np.allclose(ynp, R.conv3d(x, kern, None, 2).flatten().numpy())


def test_transposed_conv_impulse():
    # (5) x = delta_i => y[2i:2i+3] = w, index 2n cropped
    n = 4
    kern = torch.zeros(3, 3, 3, 1, 1, dtype=torch.float64)
    kern[:, 0, 0, 0, 0] = torch.tensor([1.0, 10.0, 100.0], dtype=torch.float64)
    for i in range(n):
        x = torch.zeros(1, n, 1, 1, 1, dtype=torch.float64)
        x[0, i] = 1.0
        y = R.conv3d_transpose(x, kern, None)[0, :, 0, 0, 0]
        exp = torch.zeros(2 * n + 1, dtype=torch.float64)
        exp[2 * i:2 * i + 3] = torch.tensor([1.0, 10.0, 100.0], dtype=torch.float64)
        assert torch.equal(y, exp[:2 * n])
        assert y.shape[0] == 2 * n


def test_loss_known_answers():
    # (6) perfect prediction, perfect reconstruction, mu=0, logvar=0 => 0 ; (7) mu=1 => KL term contributes 0.1
    g = torch.Generator().manual_seed(2)
    y = (torch.rand(1, 4, 4, 4, 3, generator=g) > 0.5).double()
    x = torch.randn(1, 4, 4, 4, 2, generator=g, dtype=torch.float64)
    z0 = torch.zeros(1, 8, dtype=torch.float64)
    assert float(R.dice_vae_loss(x, y, y, x, z0, z0)) == pytest.approx(0.0, abs=1e-15)
    assert float(R.dice_vae_loss(x, y, y, x, torch.ones(1, 8, dtype=torch.float64), z0)) == pytest.approx(0.1, abs=1e-15)
    assert NP.dice_vae_loss(x.numpy(), y.numpy(), y.numpy(), x.numpy(), np.ones((1, 8)), np.zeros((1, 8))) == pytest.approx(0.1)


def test_dice_coefficient_reduces_axes_012_only():
    # (8) SURVEY F8: channels_last reduces (N,D,H): one Dice cell per (w, c). Hand-computed 1x1x2x2x1-like example.
    yp = torch.zeros(1, 1, 2, 2, 2, dtype=torch.float64)
    yt = torch.zeros(1, 1, 2, 2, 2, dtype=torch.float64)
    yp[0, 0, 0, 0] = torch.tensor([0.9, 0.1])   # -> class 0 on
    yp[0, 0, 0, 1] = torch.tensor([0.2, 0.8])   # -> class 1 on
    yp[0, 0, 1, 0] = torch.tensor([0.4, 0.3])   # below threshold -> nothing
    yp[0, 0, 1, 1] = torch.tensor([0.6, 0.7])   # -> class 1 on
    yt[0, 0, 0, 0, 0] = 1
    yt[0, 0, 0, 1, 0] = 1
    yt[0, 0, 1, 1, 1] = 1
    macro, micro, labels = R.dice_coefficient(yt, yp)
    # cells (w,c): w=0: c0 I=1,P=1,T=1 -> 3/3 ; c1 I=0,P=0,T=0 -> 1 ; w=1: c0 I=0,P=0,T=1 -> 1/2 ; c1 I=1,P=2,T=1 -> 3/4
    assert float(macro) == pytest.approx((1.0 + 1.0 + 0.5 + 0.75) / 4)
    assert float(micro) == pytest.approx(2.0 / (3.0 + 3.0))
    assert labels.flatten().tolist() == [1, 2, 0, 2]
    m2, mi2, l2 = NP.dice_coefficient(yt.numpy(), yp.numpy())
    assert m2 == pytest.approx(float(macro)) and mi2 == pytest.approx(float(micro)) and l2.flatten().tolist() == [1, 2, 0, 2]


def test_lr_schedule_and_adam_first_step():
    # (9) util.py:82-84 ; (10) Keras Adam closed form: dw = -lr*g/(|g| + eps/sqrt(1-b2))
    assert R.scheduled_lr(1e-4, 0) == pytest.approx(1e-4)
    assert R.scheduled_lr(1e-4, 150) == pytest.approx(1e-4 * 0.5 ** 0.9)
    assert R.scheduled_lr(1e-4, 150) == pytest.approx(5.3589e-5, rel=1e-4)
    assert R.scheduled_lr(1e-4, 299) == pytest.approx(1e-4 * (1 / 300.0) ** 0.9)
    for gval, exp in ((1.0, -9.99997e-5), (1e-3, -9.96848e-5), (1e-6, -2.40253e-5)):
        p, m, v = R.adam_tf_step(torch.zeros(1, dtype=torch.float64), torch.tensor([gval], dtype=torch.float64),
                                 torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), 1, 1e-4)
        assert float(p) == pytest.approx(exp, rel=1e-5)
        closed = -1e-4 * gval / (abs(gval) + 1e-7 / math.sqrt(1 - 0.999))
        assert float(p) == pytest.approx(closed, rel=1e-9)


def test_parameter_counts_and_shape_chain():
    # (11) 42,174,773 (CLI defaults) / 10,636,061 (ctor defaults); (12) unproj 512 units, flatten 8192
    cli = R.build_params(R.default_config(base_filters=32, reduction=8), (128, 128, 128))
    assert sum(t.numel() for t in cli.values()) == 42174773
    assert sum(t.numel() for k, t in cli.items() if k.startswith('encoder')) == 30567328
    assert sum(t.numel() for k, t in cli.items() if k.startswith('decoder')) == 6640387
    assert sum(t.numel() for k, t in cli.items() if k.startswith('vae')) == 4967058
    assert tuple(cli['vae/proj_k'].shape) == (8192, 256) and tuple(cli['vae/unproj_k'].shape) == (128, 512)
    assert tuple(cli['encoder/L3/B3/conv1_k'].shape) == (3, 3, 3, 1024, 256)   # F4: [o2, o0, o1, o2]
    assert tuple(cli['decoder/L0/res/conv1_k'].shape) == (3, 3, 3, 64, 32)
    ctor = R.build_params(R.default_config(), (128, 128, 128))
    assert sum(t.numel() for t in ctor.values()) == 10636061
    # L2 mask (A.9): no bias, no down/up-sample GN, no transposed-conv kernel; VAE extra downsample fixed 1e-5
    assert cli.l2['encoder/L0/B0/ptwise_b'] == 0 and cli.l2['encoder/L0/down/gn_g'] == 0
    assert cli.l2['decoder/L2/up/conv_k'] == 0 and cli.l2['encoder/L0/B0/gn2_g'] == 1e-5
    odd = R.build_params(R.default_config(l2_scale=3e-4, depth=2, base_filters=8), (8, 8, 8))
    assert odd.l2['vae/down/conv_k'] == 1e-5 and odd.l2['vae/proj_k'] == 3e-4


def test_resnet_block_rejects_bad_reduction():
    with pytest.raises(ValueError):
        R.build_params(R.default_config(base_filters=6, reduction=4, depth=2, groups=2), (8, 8, 8))
