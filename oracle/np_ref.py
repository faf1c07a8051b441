"""ORACLE (test infrastructure only) -- second, independent restatement in explicit-index numpy (float64).

PARITY UNPINNED (no TensorFlow here, no reference tests/golden vectors; see oracle/torch_ref.py header).
This file re-derives every op from its index formula with plain loops, so that it shares
no code path with torch_ref.py (which leans on torch's conv/pad/reshape).  Small shapes only (8^3..16^3).
channels_last layout only: arrays are [N,D,H,W,C].
"""
import numpy as np


def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def conv3d(x, kernel, bias=None, stride=1):
    """Conv3D 'same' (resnet.py:80-87, downsample.py:28-35): y[o] = b + sum_k x[o*s + k - pad_before] W[k]."""
    N, D, H, W, Cin = x.shape
    k = kernel.shape[0]
    Cout = kernel.shape[4]
    (Do, pz), (Ho, py), (Wo, px) = _same_pads(D, k, stride), _same_pads(H, k, stride), _same_pads(W, k, stride)
    y = np.zeros((N, Do, Ho, Wo, Cout), np.float64)
    for oz in range(Do):
        for oy in range(Ho):
            for ox in range(Wo):
                acc = np.zeros((N, Cout), np.float64)
                for a in range(k):
                    iz = oz * stride + a - pz
                    if iz < 0 or iz >= D:
                        continue
                    for b in range(k):
                        iy = oy * stride + b - py
                        if iy < 0 or iy >= H:
                            continue
                        for c in range(k):
                            ix = ox * stride + c - px
                            if ix < 0 or ix >= W:
                                continue
                            acc += x[:, iz, iy, ix, :] @ kernel[a, b, c]
                y[:, oz, oy, ox, :] = acc
    if bias is not None:
        y += bias
    return y


def conv3d_transpose(x, kernel, bias=None):
    """Conv3DTranspose k3 s2 'same' (upsample.py:28-33): scatter y[2i+k] += x[i] W[k] (W: k,k,k,Cout,Cin), crop to 2n."""
    N, D, H, W, Cin = x.shape
    Cout = kernel.shape[3]
    y = np.zeros((N, 2 * D + 1, 2 * H + 1, 2 * W + 1, Cout), np.float64)
    for iz in range(D):
        for iy in range(H):
            for ix in range(W):
                v = x[:, iz, iy, ix, :]
                for a in range(3):
                    for b in range(3):
                        for c in range(3):
                            y[:, 2 * iz + a, 2 * iy + b, 2 * ix + c, :] += v @ kernel[a, b, c].T
    y = y[:, :2 * D, :2 * H, :2 * W, :]
    if bias is not None:
        y = y + bias
    return y


def group_norm_slab(x, gamma, beta, groups, eps=1e-5):
    """channels_last GroupNormalization (group_norm.py:83-124) written from the flat-index view."""
    N = x.shape[0]
    C = x.shape[-1]
    cg = C // groups
    flat = x.reshape(N, -1).astype(np.float64)
    E = flat.shape[1]
    L = E // groups
    out = np.empty_like(flat)
    cidx = np.arange(E) % C
    for n in range(N):
        for g in range(groups):
            seg = flat[n, g * L:(g + 1) * L]
            mu = seg.sum() / L
            var = ((seg - mu) ** 2).sum() / L
            xh = (seg - mu) / np.sqrt(var + eps)
            aff = g * cg + (cidx[g * L:(g + 1) * L] % cg)
            out[n, g * L:(g + 1) * L] = xh * gamma[aff] + beta[aff]
    return out.reshape(x.shape)


def group_norm_channel(x, gamma, beta, groups, eps=1e-5):
    """channels_first semantics (true GroupNorm) evaluated on an NDHWC array."""
    N = x.shape[0]
    C = x.shape[-1]
    cg = C // groups
    out = np.empty(x.shape, np.float64)
    for n in range(N):
        for g in range(groups):
            seg = x[n, ..., g * cg:(g + 1) * cg].astype(np.float64)
            mu = seg.mean()
            var = ((seg - mu) ** 2).mean()
            out[n, ..., g * cg:(g + 1) * cg] = (seg - mu) / np.sqrt(var + eps) * gamma[g * cg:(g + 1) * cg] + beta[g * cg:(g + 1) * cg]
    return out


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def resnet_block(x, P, pre, groups):
    """resnet.py:116-138 (channels_last)."""
    res = conv3d(x, P[pre + 'ptwise_k'], P[pre + 'ptwise_b'])
    gap = res.mean(axis=(1, 2, 3))
    ch = sigmoid(np.maximum(gap @ P[pre + 'se_w1'], 0.0) @ P[pre + 'se_w2'])
    sp = sigmoid(res @ P[pre + 'spatial_k'][0, 0, 0])
    res = res * (sp + ch[:, None, None, None, :])
    h = x
    for k in ('1', '2'):
        h = conv3d(h, P[pre + 'conv%s_k' % k], P[pre + 'conv%s_b' % k])
        h = np.maximum(group_norm_slab(h, P[pre + 'gn%s_g' % k], P[pre + 'gn%s_b' % k], groups), 0.0)
    return res + h


def dice_vae_loss(x, y, y_pred, y_vae, z_mean, z_logvar):
    """util.py:13-24 (channels_last)."""
    l2 = ((x - y_vae) ** 2).sum() / x.size
    kl = (z_mean ** 2 + np.exp(z_logvar) - z_logvar - 1.0).sum() / z_mean.size
    C = y.shape[-1]
    d = 0.0
    for c in range(C):
        i = (y_pred[..., c] * y[..., c]).sum()
        p = (y_pred[..., c] ** 2).sum()
        t = (y[..., c] ** 2).sum()
        d += 1.0 - (2.0 * i + 1.0) / (p + t + 1.0)
    return d / C + 0.1 * l2 + 0.1 * kl


def dice_coefficient(y_true, y_pred):
    """util.py:35-57 for channels_last: reduces axes (0,1,2) only, i.e. one Dice cell per (w, c)."""
    N, D, H, W, C = y_pred.shape
    I = np.zeros((W, C)); Pp = np.zeros((W, C)); T = np.zeros((W, C))
    labels = np.zeros((N, D, H, W), np.int64)
    for n in range(N):
        for d in range(D):
            for h in range(H):
                for w in range(W):
                    p = y_pred[n, d, h, w]
                    a = int(np.argmax(p))
                    on = p[a] > 0.5
                    labels[n, d, h, w] = a + 1 if on else 0
                    for c in range(C):
                        ph = 1.0 if (on and c == a) else 0.0
                        I[w, c] += ph * y_true[n, d, h, w, c]
                        Pp[w, c] += ph
                        T[w, c] += y_true[n, d, h, w, c]
    macro = ((2 * I + 1) / (Pp + T + 1)).mean()
    micro = I.sum() / (Pp.sum() + T.sum())
    return macro, micro, labels


def adam_tf_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
    return p, m, v
