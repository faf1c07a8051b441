"""VariationalAutoencoder -- drop-in for layers/vae.py of the reference (sample :9-13, constructor :17-99,
build :101-111, call :114-143).

    down(1024->bf/2, s2) -> Flatten -> Dense proj (-> 2*latent) -> split mu | logvar -> z = mu + exp(.5 logvar)*eps
    -> Dense unproj (ReLU) -> reshape (d/2,h/2,w/2,1) -> up(1->units) -> depth-1 x [up, ResnetBlock] -> conv3 -> in_ch
`unproj` is created in build() from the VAE *input* spatial shape, so the model is tied to the crop size.
The extra downsample always uses l2=1e-5 (the ConvDownsample default; vae.py:53-57 passes an unused kwarg: F10)."""
import torch

from .. import ops
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape
from .downsample import get_downsampling
from .resnet import ResnetBlock, _wgrad
from .upsample import get_upsampling


def sample(inputs, eps=None):
    """z_mean + exp(0.5*z_logvar)*eps on raw [N,L] torch tensors (vae.py:9-13); eps ~ N(0,1) drawn on-device if None"""
    z_mean, z_logvar = inputs
    proj = torch.cat([z_mean, z_logvar], dim=1).contiguous()
    if eps is None:
        eps = ops.normal(tuple(z_mean.shape), 0x5A17, proj.device)
    return ops.vae_sample_fwd(proj, eps)


class VariationalAutoencoder(Layer):
    def __init__(self, data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, downsampling='conv',
                 upsampling='conv', base_filters=16, depth=4, out_ch=2, name='vae'):
        super(VariationalAutoencoder, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.data_format = data_format
        self.l2_scale = l2_scale
        self.config = {'groups': groups, 'reduction': reduction, 'downsampling': downsampling, 'upsampling': upsampling,
                       'base_filters': base_filters, 'depth': depth, 'out_ch': out_ch}
        Downsample = get_downsampling(downsampling)
        Upsample = get_upsampling(upsampling)
        self.out_ch = out_ch
        self.downsample = self.track(Downsample(filters=base_filters // 2, groups=groups, data_format=data_format,
                                                name=self.name + '/down'))
        self.units = base_filters * (2 ** (depth - 1))
        self.latent_size = base_filters * (2 ** (depth - 2))
        self.upsample = self.track(Upsample(filters=self.units, groups=groups, data_format=data_format,
                                            l2_scale=l2_scale, name=self.name + '/up'))
        self.levels = []
        for i in range(depth - 2, -1, -1):
            up = self.track(Upsample(filters=base_filters * (2 ** i), groups=groups, data_format=data_format,
                                     l2_scale=l2_scale, name='%s/L%d/up' % (self.name, i)))
            conv = self.track(ResnetBlock(filters=base_filters * (2 ** i), groups=groups, reduction=reduction,
                                          data_format=data_format, l2_scale=l2_scale,
                                          name='%s/L%d/res' % (self.name, i)))
            self.levels.append([up, conv])
        self._eps = None
        self._seed = 0xE95

    def build(self, input_shape):
        n, d, h, w, c = input_shape
        self.downsample.build(tuple(input_shape))
        ds = self.downsample.compute_output_shape(tuple(input_shape))
        flat = ds[1] * ds[2] * ds[3] * ds[4]
        self.proj_k = self.add_weight('proj_k', (flat, self.units), 'he_normal', self.l2_scale)
        self.proj_b = self.add_weight('proj_b', (self.units,), 'zeros')
        un = d * h * w * 1 // 8                                                    # vae.py:105-106
        self.unproj_k = self.add_weight('unproj_k', (self.latent_size, un), 'he_normal', self.l2_scale)
        self.unproj_b = self.add_weight('unproj_b', (un,), 'zeros')
        self._unflat = (d // 2, h // 2, w // 2, 1)                                 # vae.py:110-111
        shp = (n,) + self._unflat
        self.upsample.build(shp)
        shp = self.upsample.compute_output_shape(shp)
        for up, conv in self.levels:
            up.build(shp)
            shp = up.compute_output_shape(shp)
            conv.build(shp)
            shp = shp[:4] + (conv.filters,)
        self.out_k = self.add_weight('out_k', (3, 3, 3, shp[-1], self.out_ch), 'he_normal', self.l2_scale)
        self.out_b = self.add_weight('out_b', (self.out_ch,), 'zeros')
        self.built = True

    @property
    def trainable_variables(self):
        out = list(self.downsample.trainable_variables) + self._params[:2]
        out += list(self.upsample.trainable_variables)
        for up, conv in self.levels:
            out += list(up.trainable_variables) + list(conv.trainable_variables)
        out += self._params[4:6]   # out conv
        out += self._params[2:4]   # unproj is created last in the reference (vae.py:105)
        return out

    def set_eps(self, eps):
        """inject the N(0,1) draw of `sample` (vae.py:12) for the next call (parity runs)"""
        self._eps = eps

    def call(self, inputs, training=None):
        x = as_tensor(inputs)
        dev = x.t.device
        hdn = self.downsample(x)                                                   # vae.py:116 (no training arg)
        n = hdn.shape[0]
        flat = hdn.t.reshape(n, -1)                                                # :119 row-major (d,h,w,c)
        proj = ops.dense_fwd(flat, self.proj_k.t, self.proj_b.t, False)            # :120
        L = self.latent_size
        if self._eps is not None:
            eps = torch.as_tensor(self._eps, dtype=torch.float32).to(dev).contiguous()
            self._eps = None
        else:
            self._seed += 1
            eps = ops.normal((n, L), self._seed, dev)
        z = ops.vae_sample_fwd(proj, eps)                                          # :123-125
        u = ops.dense_fwd(z, self.unproj_k.t, self.unproj_b.t, True)               # :128
        u5 = Tensor(u.reshape((n,) + self._unflat))                                # :129
        projT = _Proj(proj)
        z_mean = Tensor(proj[:, :L], base=projT, c0=0)
        z_logvar = Tensor(proj[:, L:], base=projT, c0=L)
        tape = current_tape()
        if tape is not None:
            def backward():
                du5 = u5.grad
                dproj = projT.grad_full()
                if du5 is not None:
                    g = ops.relu_bwd(u, du5.reshape(u.shape))
                    dz = torch.empty_like(z)
                    _dense_wgrad(z, self.unproj_k, self.unproj_b, g, dz)
                    ops.vae_sample_bwd(proj, eps, dz, dproj)
                dflat = None
                if hdn.requires_grad:
                    buf, acc = hdn.grad_slot()
                    dflat = buf.reshape(n, -1)
                _dense_wgrad(flat, self.proj_k, self.proj_b, dproj, dflat, acc if dflat is not None else False)
            tape.record(backward)
        y = self.upsample(u5)                                                      # :132
        for up, conv in self.levels:                                               # :135-138
            y = up(y, training=training)
            y = conv(y, training=training)
        cin = y.shape[-1]
        wp = self.packed('out_f', ops.K3S1, ops.ROLE_FWD, self.out_k, cin, self.out_ch)
        out = Tensor(ops.conv_fwd(ops.K3S1, y.t, wp, self.out_b.t, self.out_ch))   # :141
        if tape is not None:
            yin = y

            def backward_out():
                dy = out.grad
                if dy is None:
                    return
                if yin.requires_grad:
                    dx, acc = yin.grad_slot()
                    wpb = self.packed('out_b', ops.K3S1, ops.ROLE_BWD, self.out_k, cin, self.out_ch)
                    ops.conv_bwd_data(ops.K3S1, dy, wpb, dx, acc)
                _wgrad(ops.K3S1, yin.t, dy, self.out_k, self.out_b)
            tape.record(backward_out)
        return out, z_mean, z_logvar

    def get_config(self):
        cfg = dict(self.config)
        cfg.update({'data_format': self.data_format, 'l2_scale': self.l2_scale})
        return cfg


class _Proj(object):
    """grad holder for the (N, 2*latent) projection that z_mean / z_logvar are column views of"""

    def __init__(self, t):
        self.t = t
        self.g = None

    def grad_full(self):
        if self.g is None:
            self.g = torch.empty_like(self.t)
            ops.fill(self.g, 0.0)
        return self.g

    def grad_or_none(self):
        return self.g


def _dense_wgrad(x, kparam, bparam, g, dx, accumulate_dx=False):
    dw, aw = kparam.grad_slot()
    db, ab = bparam.grad_slot()
    if aw != ab:
        if not aw:
            ops.fill(dw, 0.0)
        if not ab:
            ops.fill(db, 0.0)
        aw = True
    ops.dense_bwd(x, kparam.t, g, dx, dw, db, accumulate_dx=accumulate_dx, accumulate_params=aw)
