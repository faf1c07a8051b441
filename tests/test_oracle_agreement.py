"""CPU: the two independent restatements (torch composite vs explicit-index numpy) must agree to <= 1e-6 relative
before any vector derived from them is trusted."""
import numpy as np
import pytest
import torch

from oracle import np_ref as NP
from oracle import torch_ref as R


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('stride,shape,cin,cout,k', [(1, (4, 5, 6), 3, 4, 3), (2, (4, 6, 8), 2, 3, 3), (1, (3, 3, 3), 5, 2, 1),
                                                     (2, (5, 6, 7), 2, 2, 3)])
def test_conv3d(stride, shape, cin, cout, k):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2,) + shape + (cin,), generator=g, dtype=torch.float64)
    w = torch.randn((k, k, k, cin, cout), generator=g, dtype=torch.float64)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    assert _rel(NP.conv3d(x.numpy(), w.numpy(), b.numpy(), stride), R.conv3d(x, w, b, stride).numpy()) < 1e-12


def test_conv3d_transpose():
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 3, 4, 2, 3), generator=g, dtype=torch.float64)
    w = torch.randn((3, 3, 3, 5, 3), generator=g, dtype=torch.float64)   # (k,k,k,Cout,Cin)
    b = torch.randn(5, generator=g, dtype=torch.float64)
    assert _rel(NP.conv3d_transpose(x.numpy(), w.numpy(), b.numpy()), R.conv3d_transpose(x, w, b).numpy()) < 1e-12


def test_group_norm_both_modes():
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 4, 2, 3, 8), generator=g, dtype=torch.float64)
    gamma = torch.randn(8, generator=g, dtype=torch.float64)
    beta = torch.randn(8, generator=g, dtype=torch.float64)
    assert _rel(NP.group_norm_slab(x.numpy(), gamma.numpy(), beta.numpy(), 4), R.group_norm(x, gamma, beta, 4, -1).numpy()) < 1e-10
    cf = R.group_norm(x.permute(0, 4, 1, 2, 3), gamma, beta, 4, 1).permute(0, 2, 3, 4, 1)
    assert _rel(NP.group_norm_channel(x.numpy(), gamma.numpy(), beta.numpy(), 4), cf.numpy()) < 1e-10


def test_resnet_block_loss_metric_adam():
    cfg = R.default_config(base_filters=4, groups=2, reduction=2, depth=2)
    P = R.build_params(cfg, (8, 8, 8), seed=3)
    g = torch.Generator().manual_seed(3)
    for k in P:
        if k.endswith('_g') or k.endswith('_b'):
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64)
    x = torch.randn((1, 4, 4, 4, 2), generator=g, dtype=torch.float64)
    Pn = {k: v.numpy() for k, v in P.items()}
    a = NP.resnet_block(x.numpy(), Pn, 'encoder/L0/B0/', 2)
    b = R.resnet_block(x, P, 'encoder/L0/B0/', cfg).numpy()
    assert _rel(a, b) < 1e-10
    yp = torch.rand((2, 3, 4, 5, 3), generator=g, dtype=torch.float64)
    y = (torch.rand((2, 3, 4, 5, 3), generator=g) > 0.6).double()
    xx = torch.randn((2, 3, 4, 5, 2), generator=g, dtype=torch.float64)
    yv = torch.randn((2, 3, 4, 5, 2), generator=g, dtype=torch.float64)
    zm, zl = torch.randn((2, 6), generator=g, dtype=torch.float64), torch.randn((2, 6), generator=g, dtype=torch.float64)
    l1 = NP.dice_vae_loss(xx.numpy(), y.numpy(), yp.numpy(), yv.numpy(), zm.numpy(), zl.numpy())
    assert abs(l1 - float(R.dice_vae_loss(xx, y, yp, yv, zm, zl))) < 1e-12
    m1, mi1, lab1 = NP.dice_coefficient(y.numpy(), yp.numpy())
    m2, mi2, lab2 = R.dice_coefficient(y, yp)
    assert abs(m1 - float(m2)) < 1e-12 and abs(mi1 - float(mi2)) < 1e-12 and np.array_equal(lab1, lab2.numpy())
    p, gg = torch.randn(10, generator=g, dtype=torch.float64), torch.randn(10, generator=g, dtype=torch.float64)
    m, v = torch.zeros(10, dtype=torch.float64), torch.zeros(10, dtype=torch.float64)
    pn, mn, vn = p.numpy(), m.numpy(), v.numpy()
    for t in (1, 2, 3):
        p, m, v = R.adam_tf_step(p, gg, m, v, t, 1e-4)
        pn, mn, vn = NP.adam_tf_step(pn, gg.numpy(), mn, vn, t, 1e-4)
    assert _rel(pn, p.numpy()) < 1e-12
