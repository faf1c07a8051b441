#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run from the repository root: python tests/golden/make_golden.py).

RESTATEMENT-DERIVED: the reference needs tensorflow==2.0.0-alpha0 (absent here) and ships no vectors of its own, so
these are produced by oracle/torch_ref.py in fp64 from fp32-representable seeded inputs, and only written after the
independent explicit-index numpy restatement (oracle/np_ref.py) agrees to <= 1e-6 relative."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_ref as NP  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def f32(t):
    return t.float().double()


def agree(a, b, what):
    rel = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    assert rel <= 1e-6, (what, rel)


def ops_vectors():
    g = torch.Generator().manual_seed(2024)
    d = {}
    x = f32(torch.randn((1, 6, 8, 8, 5), generator=g))
    for name, k, cin, cout, stride in (('conv_k3s1', 3, 5, 7, 1), ('conv_k3s2', 3, 5, 4, 2), ('conv_k1', 1, 5, 6, 1)):
        w = f32(torch.randn((k, k, k, cin, cout), generator=g) * 0.3)
        b = f32(torch.randn(cout, generator=g))
        y = R.conv3d(x, w, b, stride)
        agree(NP.conv3d(x.numpy(), w.numpy(), b.numpy(), stride), y.numpy(), name)
        d[name + '_w'], d[name + '_b'], d[name + '_y'] = w.numpy(), b.numpy(), y.numpy()
    wt = f32(torch.randn((3, 3, 3, 4, 5), generator=g) * 0.3)
    bt = f32(torch.randn(4, generator=g))
    xt = x[:, :3, :4, :4]
    yt = R.conv3d_transpose(xt, wt, bt)
    agree(NP.conv3d_transpose(xt.numpy(), wt.numpy(), bt.numpy()), yt.numpy(), 'convT')
    d['convT_w'], d['convT_b'], d['convT_y'] = wt.numpy(), bt.numpy(), yt.numpy()
    d['x'] = x.numpy()
    xg = f32(torch.randn((2, 4, 4, 4, 16), generator=g) * 2 + 0.5)
    gam, bet = f32(torch.randn(16, generator=g)), f32(torch.randn(16, generator=g))
    ys = R.group_norm(xg, gam, bet, 8, -1)
    agree(NP.group_norm_slab(xg.numpy(), gam.numpy(), bet.numpy(), 8), ys.numpy(), 'gn slab')
    yc = R.group_norm(xg.permute(0, 4, 1, 2, 3), gam, bet, 8, 1).permute(0, 2, 3, 4, 1)
    agree(NP.group_norm_channel(xg.numpy(), gam.numpy(), bet.numpy(), 8), yc.numpy(), 'gn channel')
    d.update(gn_x=xg.numpy(), gn_gamma=gam.numpy(), gn_beta=bet.numpy(), gn_slab_y=ys.numpy(), gn_channel_y=yc.numpy())
    np.savez_compressed(os.path.join(OUT, 'ops_vectors.npz'), **d)


def model_vectors():
    kw = dict(base_filters=4, groups=2, reduction=2, depth=2)
    crop, n = (8, 8, 8), 2
    cfg = R.default_config(**kw)
    x, y, mask, eps = R.synthetic_batch(n, crop, latent=4, seed=1234, dtype=torch.float64)
    P = R.build_params(cfg, crop, seed=7)
    g = torch.Generator().manual_seed(8)
    for k in P:
        if k.endswith('_b'):
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.1
        if k.endswith('_g'):
            P[k] = 1.0 + torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.3
        P[k] = f32(P[k])
    # block-level agreement of the two restatements on the first encoder block with these weights
    xb = R.dropout(x, mask, cfg['dropout'])
    agree(NP.resnet_block(xb.numpy(), {k: v.numpy() for k, v in P.items()}, 'encoder/L0/B0/', cfg['groups']),
          R.resnet_block(xb, P, 'encoder/L0/B0/', cfg).numpy(), 'resnet block')
    leaves = {k: t.clone().requires_grad_(True) for k, t in P.items()}
    PP = R.ParamSet(); PP.update(leaves); PP.l2 = P.l2
    y_pred, y_vae, zm, zl = R.model(x, PP, cfg, training=True, inference=False, mask=mask, eps=eps)
    loss_main = R.dice_vae_loss(x, y, y_pred, y_vae, zm, zl)
    l2 = R.l2_regularisation(PP)
    loss = loss_main + l2
    grads = torch.autograd.grad(loss, list(leaves.values()))
    macro, micro, labels = R.dice_coefficient(y, y_pred.detach())
    agree(np.array(NP.dice_vae_loss(x.numpy(), y.numpy(), y_pred.detach().numpy(), y_vae.detach().numpy(), zm.detach().numpy(),
                                   zl.detach().numpy())), loss_main.detach().numpy(), 'loss')
    m2, mi2, lab2 = NP.dice_coefficient(y.numpy(), y_pred.detach().numpy())
    assert abs(m2 - float(macro)) < 1e-12 and np.array_equal(lab2, labels.numpy())
    d = dict(x=x.numpy().astype(np.float32), y=y.numpy().astype(np.float32), mask=mask.numpy().astype(np.uint8),
             eps=eps.numpy().astype(np.float32), y_pred=y_pred.detach().numpy(), y_vae=y_vae.detach().numpy(),
             z_mean=zm.detach().numpy(), z_logvar=zl.detach().numpy(), loss=float(loss), loss_main=float(loss_main),
             l2=float(l2), macro=float(macro), micro=float(micro), labels=labels.numpy().astype(np.uint8))
    for k, t in P.items():
        d['P/' + k] = t.numpy().astype(np.float32)
    for (k, _), gr in zip(leaves.items(), grads):
        d['G/' + k] = gr.numpy()
    # one TF-form Adam step at lr 1e-4
    for (k, t), gr in zip(P.items(), grads):
        p1, _, _ = R.adam_tf_step(t, gr, torch.zeros_like(t), torch.zeros_like(t), 1, 1e-4)
        d['A/' + k] = p1.numpy()
    np.savez_compressed(os.path.join(OUT, 'model_micro.npz'), **d)


if __name__ == '__main__':
    ops_vectors()
    model_vectors()
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(OUT, f)), 'bytes')
