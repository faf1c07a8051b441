"""ORACLE (test infrastructure only) -- CPU restatement of the reference's training step in torch-CPU ops.

PARITY UNPINNED: the reference (vliu15/3d-brain-tumor-segmentation) delegates all arithmetic to
tensorflow==2.0.0-alpha0 (requirements.txt:2), which is not installable here, and it ships no tests, golden
vectors or fixtures.  This file restates the reference line by line with the TF op
semantics written out; it is cross-checked against an independent explicit-index numpy
restatement (oracle/np_ref.py) and analytic known-answer tests (tests/test_oracle_kat.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(3d-brain-tumor-segmentation_amd/) never does.

Tensors are torch tensors in the reference's public layout ([N,D,H,W,C] for 'channels_last', [N,C,D,H,W] for
'channels_first'); parameters live in an ordered dict name -> tensor in the reference's Keras layouts:
Conv3D (kd,kh,kw,Cin,Cout), Conv3DTranspose (kd,kh,kw,Cout,Cin), Dense (in,out).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# primitive ops (TF semantics)
# --------------------------------------------------------------------------------------------------
def _to_ncdhw(x, data_format):
    return x.permute(0, 4, 1, 2, 3) if data_format == 'channels_last' else x


def _from_ncdhw(x, data_format):
    return x.permute(0, 2, 3, 4, 1) if data_format == 'channels_last' else x


def conv3d(x, kernel, bias=None, stride=1, data_format='channels_last'):
    """tf.keras.layers.Conv3D(padding='same') -- resnet.py:30-37,80-87; downsample.py:28-35.

    k=1: no padding. k=3,s=1: pad 1 each side. k=3,s=2: out=ceil(in/2), pad_total=max((out-1)*2+3-in,0),
    pad_before=pad_total//2 -> even sizes pad (0,1), odd sizes pad (1,1)."""
    k = kernel.shape[0]
    xc = _to_ncdhw(x, data_format)
    w = kernel.permute(4, 3, 0, 1, 2)
    pads = []
    for n in reversed(xc.shape[2:]):  # F.pad wants last dim first
        out = -(-n // stride)
        total = max((out - 1) * stride + k - n, 0)
        pads += [total // 2, total - total // 2]
    xc = F.pad(xc, pads)
    y = F.conv3d(xc, w, bias, stride=stride)
    return _from_ncdhw(y, data_format)


def conv3d_transpose(x, kernel, bias=None, data_format='channels_last'):
    """tf.keras.layers.Conv3DTranspose(k=3, s=2, 'same') -- upsample.py:28-33.

    y[2i+k] += x[i]*W[k]; output size 2*in (the tap landing on index 2*in is cropped)."""
    xc = _to_ncdhw(x, data_format)
    w = kernel.permute(4, 3, 0, 1, 2)  # (Cin, Cout, kd, kh, kw)
    y = F.conv_transpose3d(xc, w, bias, stride=2, padding=0)
    d, h, ww = xc.shape[2:]
    y = y[:, :, :2 * d, :2 * h, :2 * ww]
    return _from_ncdhw(y, data_format)


def group_norm(x, gamma, beta, groups, axis, eps=1e-5):
    """GroupNormalization.call, literal restatement of group_norm.py:83-124 (note SURVEY F1 for axis=-1)."""
    input_shape = list(x.shape)
    nd = len(input_shape)
    ax = axis if axis >= 0 else nd + axis
    broadcast_shape = [1] * nd
    broadcast_shape[ax] = input_shape[ax] // groups
    broadcast_shape.insert(1, groups)                      # :89-91
    group_axes = list(input_shape)
    group_axes[ax] = input_shape[ax] // groups
    group_axes.insert(1, groups)                           # :93-95
    group_shape = [group_axes[0], groups] + group_axes[2:]  # :98
    g = x.reshape(group_shape)                              # :100 raw row-major reshape
    red = list(range(2, len(group_axes)))                   # :102-103
    mean = g.mean(dim=red, keepdim=True)                    # :105 tf.nn.moments -> population variance
    var = ((g - mean) ** 2).mean(dim=red, keepdim=True)
    g = (g - mean) / torch.sqrt(var + eps)                  # :107
    if gamma is not None:
        g = g * gamma.reshape(broadcast_shape)              # :115-116
    if beta is not None:
        g = g + beta.reshape(broadcast_shape)               # :119-120
    return g.reshape(input_shape)                           # :122


def dropout(x, mask, rate):
    """tf.keras.layers.Dropout in training (encoder.py:39,71): x*mask/(1-rate); mask = (u >= rate)."""
    return x * mask / (1.0 - rate)


def sample(z_mean, z_logvar, eps):
    """vae.py:9-13 with eps injected."""
    return z_mean + torch.exp(0.5 * z_logvar) * eps


# --------------------------------------------------------------------------------------------------
# parameter construction (initialisers: SURVEY A.11; regulariser set: SURVEY A.9)
# --------------------------------------------------------------------------------------------------
def _trunc_normal(gen, shape, std):
    # Keras truncated_normal: resample outside 2 sigma; VarianceScaling divides stddev by .87962566103423978
    t = torch.empty(shape, dtype=torch.float64)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std / 0.87962566103423978, a=-2 * std / 0.87962566103423978,
                                b=2 * std / 0.87962566103423978, generator=gen)
    return t


def _fans(shape, transposed=False):
    rf = 1
    for s in shape[:-2]:
        rf *= s
    cin, cout = (shape[-1], shape[-2]) if transposed else (shape[-2], shape[-1])
    return rf * cin, rf * cout


class ParamSet(OrderedDict):
    """name -> tensor, plus .l2[name] = regulariser coefficient (0 when unregularised)."""

    def __init__(self):
        super().__init__()
        self.l2 = OrderedDict()


def _add(P, gen, name, shape, init, l2, transposed=False):
    if init == 'zeros':
        t = torch.zeros(shape, dtype=torch.float64)
    elif init == 'ones':
        t = torch.ones(shape, dtype=torch.float64)
    elif init == 'he_normal':
        fi, _ = _fans(shape, transposed)
        t = _trunc_normal(gen, shape, math.sqrt(2.0 / fi))
    elif init == 'glorot_normal':
        fi, fo = _fans(shape, transposed)
        t = _trunc_normal(gen, shape, math.sqrt(2.0 / (fi + fo)))
    elif init == 'glorot_uniform':
        fi, fo = _fans(shape, transposed)
        lim = math.sqrt(6.0 / (fi + fo))
        t = (torch.rand(shape, dtype=torch.float64, generator=gen) * 2 - 1) * lim
    else:
        raise ValueError(init)
    P[name] = t
    P.l2[name] = float(l2)


def _add_resblock(P, gen, pre, cin, f, reduction, l2):
    if f % reduction != 0:  # resnet.py:39-42
        raise ValueError('Reduction ratio, {}, must be a factor of number of channels, {}.'.format(reduction, f))
    _add(P, gen, pre + 'ptwise_k', (1, 1, 1, cin, f), 'he_normal', l2)
    _add(P, gen, pre + 'ptwise_b', (f,), 'zeros', 0)
    _add(P, gen, pre + 'se_w1', (f, f // reduction), 'he_normal', l2)
    _add(P, gen, pre + 'se_w2', (f // reduction, f), 'he_normal', l2)
    _add(P, gen, pre + 'spatial_k', (1, 1, 1, f, 1), 'he_normal', l2)
    _add(P, gen, pre + 'conv1_k', (3, 3, 3, cin, f), 'he_normal', l2)
    _add(P, gen, pre + 'conv1_b', (f,), 'zeros', 0)
    _add(P, gen, pre + 'gn1_g', (f,), 'ones', l2)
    _add(P, gen, pre + 'gn1_b', (f,), 'zeros', l2)
    _add(P, gen, pre + 'conv2_k', (3, 3, 3, f, f), 'he_normal', l2)
    _add(P, gen, pre + 'conv2_b', (f,), 'zeros', 0)
    _add(P, gen, pre + 'gn2_g', (f,), 'zeros', l2)  # resnet.py:104-110
    _add(P, gen, pre + 'gn2_b', (f,), 'zeros', l2)


def _add_down(P, gen, pre, cin, f, l2):
    _add(P, gen, pre + 'conv_k', (3, 3, 3, cin, f), 'he_normal', l2)  # downsample.py:28-35
    _add(P, gen, pre + 'conv_b', (f,), 'zeros', 0)
    _add(P, gen, pre + 'gn_g', (f,), 'ones', 0)                      # no regulariser: downsample.py:36-38
    _add(P, gen, pre + 'gn_b', (f,), 'zeros', 0)


def _add_up(P, gen, pre, cin, f):
    _add(P, gen, pre + 'conv_k', (3, 3, 3, f, cin), 'glorot_uniform', 0, transposed=True)  # upsample.py:28-33 (F10)
    _add(P, gen, pre + 'conv_b', (f,), 'zeros', 0)
    _add(P, gen, pre + 'gn_g', (f,), 'ones', 0)
    _add(P, gen, pre + 'gn_b', (f,), 'zeros', 0)


def _add_lin_up(P, gen, pre, cin, f, l2):
    _add(P, gen, pre + 'ptwise_k', (1, 1, 1, cin, f), 'he_normal', l2)   # upsample.py:62-69 (LinearUpsample)
    _add(P, gen, pre + 'ptwise_b', (f,), 'zeros', 0)


def default_config(**kw):
    cfg = dict(data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, dropout=0.2, downsampling='conv',
               upsampling='conv', base_filters=16, depth=4, in_ch=2, out_ch=3)  # model.py:9-20
    cfg.update(kw)
    return cfg


def build_params(cfg, crop, seed=0):
    """All trainable variables of Model(**cfg) built for a (D,H,W) crop (vae.py:101-111 ties the VAE to it)."""
    gen = torch.Generator().manual_seed(seed)
    P = ParamSet()
    bf, depth, l2, red = cfg['base_filters'], cfg['depth'], cfg['l2_scale'], cfg['reduction']
    cin = cfg['in_ch']
    level_out = []
    for i in range(depth):                       # encoder.py:43-67
        f = bf * 2 ** i
        for j in range(i + 1):
            bcin = cin if j == 0 else (j + 1) * f  # [inputs] + cache with inputs == cache[-1] (encoder.py:83-87, F4)
            _add_resblock(P, gen, 'encoder/L%d/B%d/' % (i, j), bcin, f, red, l2)
        cout = f if i == 0 else (i + 1) * f
        level_out.append(cout)
        if i < depth - 1:
            if cfg['downsampling'] == 'max':     # MaxPooling3D has no variables and keeps every channel (downsample.py:51-70)
                cin = cout
            else:
                _add_down(P, gen, 'encoder/L%d/down/' % i, cout, f, l2)
                cin = f
    linear = cfg['upsampling'] == 'linear'
    add_up = (lambda pre, ci, fo: _add_lin_up(P, gen, pre, ci, fo, l2)) if linear else (lambda pre, ci, fo: _add_up(P, gen, pre, ci, fo))
    c = level_out[-1]
    for i in range(depth - 2, -1, -1):           # decoder.py:38-53
        f = bf * 2 ** i
        add_up('decoder/L%d/up/' % i, c, f)
        _add_resblock(P, gen, 'decoder/L%d/res/' % i, level_out[i] + f, f, red, l2)
        c = f
    _add(P, gen, 'decoder/out_k', (1, 1, 1, c, cfg['out_ch']), 'glorot_normal', l2)  # decoder.py:55-63
    _add(P, gen, 'decoder/out_b', (cfg['out_ch'],), 'zeros', 0)
    # VAE (vae.py:53-111)
    c = level_out[-1]
    _add_down(P, gen, 'vae/down/', c, bf // 2, 1e-5)  # l2 fixed at the ConvDownsample default (F10)
    sp = [s // 2 ** (depth - 1) for s in crop]         # spatial dims of the VAE input
    flat = (sp[0] // 2) * (sp[1] // 2) * (sp[2] // 2) * (bf // 2)
    units = bf * 2 ** (depth - 1)
    _add(P, gen, 'vae/proj_k', (flat, units), 'he_normal', l2)
    _add(P, gen, 'vae/proj_b', (units,), 'zeros', 0)
    latent = bf * 2 ** (depth - 2)
    un = sp[0] * sp[1] * sp[2] // 8
    _add(P, gen, 'vae/unproj_k', (latent, un), 'he_normal', l2)
    _add(P, gen, 'vae/unproj_b', (un,), 'zeros', 0)
    add_up('vae/up/', 1, units)
    c = units
    for i in range(depth - 2, -1, -1):
        f = bf * 2 ** i
        add_up('vae/L%d/up/' % i, c, f)
        _add_resblock(P, gen, 'vae/L%d/res/' % i, f, f, red, l2)
        c = f
    _add(P, gen, 'vae/out_k', (3, 3, 3, c, cfg['in_ch']), 'he_normal', l2)
    _add(P, gen, 'vae/out_b', (cfg['in_ch'],), 'zeros', 0)
    return P


# --------------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------------
def _caxis(df):
    return -1 if df == 'channels_last' else 1


def resnet_block(x, P, pre, cfg):
    """ResnetBlock.call, resnet.py:116-138."""
    df, G = cfg['data_format'], cfg['groups']
    ax = _caxis(df)
    res = conv3d(x, P[pre + 'ptwise_k'], P[pre + 'ptwise_b'], 1, df)               # :118
    sp_dims = (1, 2, 3) if df == 'channels_last' else (2, 3, 4)
    chse = res.mean(dim=sp_dims)                                                    # :121 GlobalAveragePooling3D
    chse = torch.relu(chse @ P[pre + 'se_w1'])                                      # :122
    chse = torch.sigmoid(chse @ P[pre + 'se_w2'])                                   # :123
    chse = chse.reshape((-1, 1, 1, 1, chse.shape[-1]) if df == 'channels_last' else (-1, chse.shape[-1], 1, 1, 1))  # :124
    spse = torch.sigmoid(conv3d(res, P[pre + 'spatial_k'], None, 1, df))            # :127
    res = res * (spse + chse)                                                       # :130
    h = x
    for k in ('1', '2'):                                                            # :133-136
        h = conv3d(h, P[pre + 'conv%s_k' % k], P[pre + 'conv%s_b' % k], 1, df)
        h = group_norm(h, P[pre + 'gn%s_g' % k], P[pre + 'gn%s_b' % k], G, ax)
        h = torch.relu(h)
    return res + h                                                                  # :137


def conv_downsample(x, P, pre, cfg):
    """ConvDownsample.__call__, downsample.py:41-45."""
    df = cfg['data_format']
    h = conv3d(x, P[pre + 'conv_k'], P[pre + 'conv_b'], 2, df)
    return torch.relu(group_norm(h, P[pre + 'gn_g'], P[pre + 'gn_b'], cfg['groups'], _caxis(df)))


def conv_upsample(x, P, pre, cfg):
    """ConvUpsample.__call__, upsample.py:39-43."""
    df = cfg['data_format']
    h = conv3d_transpose(x, P[pre + 'conv_k'], P[pre + 'conv_b'], df)
    return torch.relu(group_norm(h, P[pre + 'gn_g'], P[pre + 'gn_b'], cfg['groups'], _caxis(df)))


def max_downsample(x, cfg):
    """MaxDownsample.__call__, downsample.py:51-70: MaxPooling3D(pool 2, stride 2, 'same') -- even sizes, no padding"""
    df = cfg['data_format']
    return _from_ncdhw(F.max_pool3d(_to_ncdhw(x, df), 2, 2), df)


def linear_upsample(x, P, pre, cfg):
    """LinearUpsample.__call__, upsample.py:76-79: 1x1x1 conv then UpSampling3D(size 2) = nearest-neighbour repeat"""
    df = cfg['data_format']
    h = _to_ncdhw(conv3d(x, P[pre + 'ptwise_k'], P[pre + 'ptwise_b'], 1, df), df)
    h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    return _from_ncdhw(h, df)


def upsample(x, P, pre, cfg):
    return linear_upsample(x, P, pre, cfg) if cfg['upsampling'] == 'linear' else conv_upsample(x, P, pre, cfg)


def encoder(x, P, cfg, training, mask):
    """Encoder.call, encoder.py:69-101."""
    ax = _caxis(cfg['data_format'])
    if training and cfg['dropout'] > 0:
        x = dropout(x, mask, cfg['dropout'])                     # :71
    residuals = []
    inputs = x
    for i in range(cfg['depth']):
        cache = []
        for j in range(i + 1):
            if j > 0:
                inputs = torch.cat([inputs] + cache, dim=ax)     # :85  (duplicates cache[-1])
            inputs = resnet_block(inputs, P, 'encoder/L%d/B%d/' % (i, j), cfg)
            cache.append(inputs)
        if i > 0:
            inputs = torch.cat(cache, dim=ax)                    # :90-91
        residuals.append(inputs)
        if i < cfg['depth'] - 1:
            if cfg['downsampling'] == 'max':
                inputs = max_downsample(inputs, cfg)
            else:
                inputs = conv_downsample(inputs, P, 'encoder/L%d/down/' % i, cfg)  # :97-98
    return residuals


def decoder(x, residuals, P, cfg):
    """Decoder.call, decoder.py:65-83."""
    df = cfg['data_format']
    ax = _caxis(df)
    lv = list(range(cfg['depth'] - 2, -1, -1))
    for i, residual in zip(lv, residuals[::-1]):
        x = upsample(x, P, 'decoder/L%d/up/' % i, cfg)
        x = torch.cat([residual, x], dim=ax)                     # :75
        x = resnet_block(x, P, 'decoder/L%d/res/' % i, cfg)
    return torch.sigmoid(conv3d(x, P['decoder/out_k'], P['decoder/out_b'], 1, df))  # :55-63,80


def vae(x, P, cfg, eps):
    """VariationalAutoencoder.call, vae.py:114-143."""
    df = cfg['data_format']
    h = conv_downsample(x, P, 'vae/down/', cfg)                  # :116
    n = h.shape[0]
    if df == 'channels_first':
        # Keras Flatten(data_format='channels_first') transposes to channels_last before flattening
        hf = h.permute(0, 2, 3, 4, 1).reshape(n, -1)
    else:
        hf = h.reshape(n, -1)                                    # :119
    p = hf @ P['vae/proj_k'] + P['vae/proj_b']                   # :120
    latent = cfg['base_filters'] * 2 ** (cfg['depth'] - 2)
    z_mean, z_logvar = p[:, :latent], p[:, latent:]              # :123-124
    z = sample(z_mean, z_logvar, eps)                            # :125
    u = torch.relu(z @ P['vae/unproj_k'] + P['vae/unproj_b'])    # :128
    sd = x.shape[1:4] if df == 'channels_last' else x.shape[2:5]
    if df == 'channels_last':
        u = u.reshape(n, sd[0] // 2, sd[1] // 2, sd[2] // 2, 1)  # :110-111,129
    else:
        u = u.reshape(n, 1, sd[0] // 2, sd[1] // 2, sd[2] // 2)
    u = upsample(u, P, 'vae/up/', cfg)                           # :132
    for i in range(cfg['depth'] - 2, -1, -1):                    # :135-138
        u = upsample(u, P, 'vae/L%d/up/' % i, cfg)
        u = resnet_block(u, P, 'vae/L%d/res/' % i, cfg)
    y = conv3d(u, P['vae/out_k'], P['vae/out_b'], 1, df)         # :141
    return y, z_mean, z_logvar


def model(x, P, cfg, training=None, inference=None, mask=None, eps=None):
    """Model.call, model.py:58-71."""
    assert (not inference or not training), 'Cannot run training and inference modes simultaneously.'
    res = encoder(x, P, cfg, training, mask)
    y_pred = decoder(res[-1], res[:-1], P, cfg)
    if inference:
        return (y_pred, None, None, None)
    y_vae, z_mean, z_logvar = vae(res[-1], P, cfg, eps)
    return (y_pred, y_vae, z_mean, z_logvar)


# --------------------------------------------------------------------------------------------------
# loss / metric / optimiser
# --------------------------------------------------------------------------------------------------
def dice_vae_loss(x, y, y_pred, y_vae, z_mean, z_logvar, data_format='channels_last'):
    """DiceVAELoss.__call__, util.py:13-24."""
    axis = (0, 1, 2, 3) if data_format == 'channels_last' else (0, 2, 3, 4)     # :11
    l2_loss = ((x - y_vae) ** 2).mean()                                           # :14
    kld_loss = (z_mean ** 2 + torch.exp(z_logvar) - z_logvar - 1.0).mean()        # :15
    inter = (y_pred * y).sum(dim=axis)                                            # :18
    pred = (y_pred ** 2).sum(dim=axis)                                            # :19
    true = (y ** 2).sum(dim=axis)                                                 # :20
    dice_loss = (1.0 - (2.0 * inter + 1.0) / (pred + true + 1.0)).mean()          # :22
    return dice_loss + 0.1 * l2_loss + 0.1 * kld_loss                             # :24


def l2_regularisation(P):
    """sum(model.losses), train.py:146: Keras l2(l) = l*sum(w^2) on the A.9 set."""
    tot = 0.0
    for name, t in P.items():
        if P.l2[name] > 0:
            tot = tot + P.l2[name] * (t ** 2).sum()
    return tot


def dice_coefficient(y_true, y_pred, data_format='channels_last'):
    """DiceCoefficient.__call__, util.py:35-57 (note the axes quirk). Returns (macro, micro, labels)."""
    dice_axes = (0, 1, 2) if data_format == 'channels_last' else (0, 2, 3, 4)    # :36
    oh_axis = -1 if data_format == 'channels_last' else 1                          # :37
    mask = (y_pred.max(dim=oh_axis, keepdim=True).values > 0.5).to(y_pred.dtype)   # :40-41
    out_ch = y_pred.shape[oh_axis]
    arg = y_pred.argmax(dim=oh_axis)                                               # :45
    oh = F.one_hot(arg, out_ch).to(y_pred.dtype)                                   # :46
    if data_format != 'channels_last':
        oh = oh.permute(0, 4, 1, 2, 3)
    yp = oh * mask                                                                 # :47
    inter = (yp * y_true).sum(dim=dice_axes)                                       # :50
    pred = yp.sum(dim=dice_axes)                                                   # :51
    true = y_true.sum(dim=dice_axes)                                               # :52
    macro = ((2.0 * inter + 1.0) / (pred + true + 1.0)).mean()                     # :54
    micro = (yp * y_true).sum() / (yp.sum() + y_true.sum())                        # :55
    labels = (arg + 1) * mask.squeeze(oh_axis).to(arg.dtype)
    return macro, micro, labels


def scheduled_lr(init_lr, epoch, n_epochs=300):
    """ScheduledOptim.__call__, util.py:82-84."""
    return init_lr * ((1.0 - epoch / float(n_epochs)) ** 0.9)


def adam_tf_step(p, g, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """Keras optimizer_v2 Adam (util.py:60-78). t is the 1-based step count. Returns new (p, m, v)."""
    m = beta_1 * m + (1 - beta_1) * g
    v = beta_2 * v + (1 - beta_2) * g * g
    lr_t = lr * math.sqrt(1 - beta_2 ** t) / (1 - beta_1 ** t)
    p = p - lr_t * m / (torch.sqrt(v) + epsilon)
    return p, m, v


def train_step(P, cfg, x, y, mask, eps, state, lr, step):
    """One iteration of train.py:140-152. state: dict name -> (m, v). Returns (loss, macro, micro, grads, outputs)."""
    leaves = OrderedDict((k, t.detach().clone().requires_grad_(True)) for k, t in P.items())
    PP = ParamSet()
    PP.update(leaves)
    PP.l2 = P.l2
    y_pred, y_vae, z_mean, z_logvar = model(x, PP, cfg, training=True, inference=False, mask=mask, eps=eps)
    loss = dice_vae_loss(x, y, y_pred, y_vae, z_mean, z_logvar, cfg['data_format'])
    loss = loss + l2_regularisation(PP)
    macro, micro, _ = dice_coefficient(y, y_pred.detach(), cfg['data_format'])
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    gd = OrderedDict()
    for (k, t), g in zip(leaves.items(), grads):
        g = torch.zeros_like(t) if g is None else g
        gd[k] = g
        m, v = state.get(k, (torch.zeros_like(t), torch.zeros_like(t)))
        np_, m, v = adam_tf_step(t.detach(), g, m, v, step, lr)
        state[k] = (m, v)
        P[k] = np_
    return loss.detach(), macro, micro, gd, (y_pred.detach(), y_vae.detach(), z_mean.detach(), z_logvar.detach())


def eval_step(P, cfg, x, y, eps):
    """One validation iteration, train.py:166-173: forward with training=False (the input Dropout is the identity,
    encoder.py:71; the VAE still draws eps, vae.py:9-13,127), loss incl. the regularisers, Dice.  No gradient."""
    with torch.no_grad():
        y_pred, y_vae, z_mean, z_logvar = model(x, P, cfg, training=False, inference=False, mask=None, eps=eps)
        loss = dice_vae_loss(x, y, y_pred, y_vae, z_mean, z_logvar, cfg['data_format']) + l2_regularisation(P)
        macro, micro, _ = dice_coefficient(y, y_pred, cfg['data_format'])
    return loss, macro, micro


def fit(P, cfg, train, val, n_epochs, init_lr, schedule_epochs=300, start_epoch=0, state=None, step=0):
    """The epoch loop of train.py:133-181 on explicit draws.  train: list of (x, y, mask, eps); val: list of (x, y, eps).
    Per epoch: lr = scheduled_lr(init_lr, epoch) (train.py:136, util.py:82-84; n_epochs=300 is ScheduledOptim's default
    because train.py:105 does not pass it), one train_step per training batch, one eval_step per validation batch; the
    six metrics are tf.keras.metrics.Mean = float32 running total / count (train.py:109-114,155-157,174-176).
    Returns (rows, state, step): rows = [{epoch, lr, train_loss, train_macro_dice, ..., val_micro_dice}]."""
    import numpy as np
    state = {} if state is None else state
    rows = []
    for epoch in range(start_epoch, n_epochs):
        lr = scheduled_lr(init_lr, epoch, schedule_epochs)
        acc = {k: [np.float32(0.0), 0] for k in ('train_loss', 'train_macro_dice', 'train_micro_dice', 'val_loss',
                                                 'val_macro_dice', 'val_micro_dice')}

        def upd(key, v):
            acc[key][0] = np.float32(acc[key][0] + np.float32(float(v)))
            acc[key][1] += 1

        for x, y, mask, eps in train:
            step += 1
            loss, macro, micro, _, _ = train_step(P, cfg, x, y, mask, eps, state, lr, step)
            upd('train_loss', loss), upd('train_macro_dice', macro), upd('train_micro_dice', micro)
        for x, y, eps in val:
            loss, macro, micro = eval_step(P, cfg, x, y, eps)
            upd('val_loss', loss), upd('val_macro_dice', macro), upd('val_micro_dice', micro)
        row = {'epoch': epoch, 'lr': np.float32(lr)}
        row.update({k: np.float32(t / np.float32(max(n, 1))) for k, (t, n) in acc.items()})
        rows.append(row)
    return rows, state, step


# --------------------------------------------------------------------------------------------------
# synthetic data of SURVEY 8(d)
# --------------------------------------------------------------------------------------------------
def synthetic_batch(n, crop, in_ch=2, out_ch=3, seed=1234, dropout_rate=0.2, latent=128, dtype=torch.float32):
    """x ~ N(0,1) inside a centred ellipsoid, 3 nested-sphere labels, Bernoulli keep mask, eps ~ N(0,1)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    D, H, W = crop
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing='ij')
    ax = np.array([56.0, 60.0, 52.0]) * np.array(crop) / 128.0
    inside = (((zz - D / 2) / ax[0]) ** 2 + ((yy - H / 2) / ax[1]) ** 2 + ((xx - W / 2) / ax[2]) ** 2) <= 1.0
    x = rng.standard_normal((n, D, H, W, in_ch)).astype(np.float32) * inside[None, ..., None]
    y = np.zeros((n, D, H, W, out_ch), np.float32)
    radii = np.array([36.0, 24.0, 12.0]) * min(crop) / 128.0
    for b in range(n):
        c = np.array([D / 2, H / 2, W / 2]) + rng.uniform(-0.15, 0.15, 3) * np.array(crop)
        r2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
        lab = np.zeros((D, H, W), np.int64)
        for k, r in enumerate(radii[:out_ch]):
            lab[r2 <= r * r] = k + 1
        for k in range(out_ch):
            y[b, ..., k] = (lab == k + 1)
    mask = (rng.random((n, D, H, W, in_ch)) >= dropout_rate).astype(np.float32)
    eps = rng.standard_normal((n, latent)).astype(np.float32)
    tt = lambda a: torch.from_numpy(a).to(dtype)
    return tt(x), tt(y), tt(mask), tt(eps)


# ---- full-volume inference wrapper (test.py:78-178,259-261) -------------------------------------------
def pad_to_spatial_res(res, x, mask):
    """test.py:164-178: append `res - (size % res)` zeros per spatial axis (a full block when already a multiple)"""
    shape = list(x.shape[:-1])
    pad = [res - (s % res) for s in shape]
    xp = torch.nn.functional.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    mp = torch.nn.functional.pad(mask, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    return xp, mp, shape


def tta_augment_axes(spatial_tta=True):
    """test.py:95-103 verbatim (channels_last: spatial axes 1,2,3 of the batched tensor)"""
    spatial_axes = [1, 2, 3]
    if not spatial_tta:
        return [[]]
    out = [list(spatial_axes), []]
    for axis in spatial_axes:
        pairs = list(spatial_axes)
        pairs.remove(axis)
        out.append([axis])
        out.append(pairs)
    return out


def tta_predict(x, bmask, P, cfg, mean, std, spatial_tta=True):
    """test.py:109-155 without channel TTA: normalise, for every flip set run the model (training=False,
    inference=True), un-flip, average, multiply by the brain mask.  x: (D,H,W,C), bmask: (D,H,W,1) -> (D,H,W,out_ch)"""
    xn = ((x - mean) / std).unsqueeze(0)
    ys = []
    for flip in tta_augment_axes(spatial_tta):
        aug = torch.flip(xn, dims=flip) if flip else xn
        y = model(aug, P, cfg, training=False, inference=True)[0]
        ys.append(torch.flip(y, dims=flip) if flip else y)
    y = torch.cat(ys, dim=0).mean(dim=0, keepdim=True)
    return (y * bmask.unsqueeze(0))[0]


def tta_labels(y, bmask, threshold=0.5):
    """the label map test.py:157-158,259-261 intends: argmax + 1, values >= 3 -> 4, 0 where masked / below threshold"""
    best, arg = y.max(dim=-1)
    lab = arg + 1
    lab = torch.where(lab >= 3, torch.full_like(lab, 4), lab)
    lab = torch.where((bmask[..., 0] == 0) | (best < threshold), torch.zeros_like(lab), lab)
    return lab.to(torch.uint8)


# ---- training-time augmentation (train.py:14-49) ------------------------------------------------------
def augment_example(x, y, crop_size, out_ch, shift, scale, offsets, flips):
    """`parse_example` after the proto parse, with the random draws as arguments.  x: (h,w,d,c), y: (h,w,d,1)."""
    c = x.shape[-1]
    var = x.var(dim=(0, 1, 2), unbiased=False, keepdim=True)                       # tf.nn.moments (:18)
    x = x + torch.as_tensor(shift, dtype=x.dtype).reshape(1, 1, 1, c) * torch.sqrt(var)   # :21
    x = x * torch.as_tensor(scale, dtype=x.dtype).reshape(1, 1, 1, c)                     # :22
    xy = torch.cat([x, y.to(x.dtype)], dim=-1)                                      # :25
    o = offsets
    xy = xy[o[0]:o[0] + crop_size[0], o[1]:o[1] + crop_size[1], o[2]:o[2] + crop_size[2]]   # :26
    for axis in (0, 1, 2):                                                          # :29-33
        if flips[axis]:
            xy = torch.flip(xy, dims=[axis])
    x, y = xy[..., :c], xy[..., c:]
    lab = y[..., 0].to(torch.int64)                                                 # :38 (cast truncates)
    onehot = torch.nn.functional.one_hot(lab.clamp(0, out_ch), out_ch + 1).to(x.dtype)   # :39
    return x, onehot[..., 1:]                                                       # :40
