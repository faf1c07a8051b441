"""Decoder -- drop-in for layers/decoder.py of the reference (constructor :9-63, call :65-83).

Per level: ConvUpsample -> Concatenate([residual, up]) (:75) -> ResnetBlock; then Conv3D 1x1x1 + sigmoid (:55-63).
The concatenation is virtual: the up-sampled tensor is written into the spare channels of the encoder level's slab
when the encoder reserved them, otherwise into a fresh slab next to a copy of the residual."""
from .. import ops
from ..tape import Slab
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape
from .resnet import ResnetBlock, _wgrad
from .upsample import get_upsampling


class Decoder(Layer):
    def __init__(self, data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, upsampling='conv',
                 base_filters=16, depth=4, out_ch=3, name='decoder'):
        super(Decoder, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'data_format': data_format, 'groups': groups, 'reduction': reduction, 'l2_scale': l2_scale,
                       'upsampling': upsampling, 'base_filters': base_filters, 'depth': depth, 'out_ch': out_ch}
        Upsample = get_upsampling(upsampling)
        self.base_filters = base_filters
        self.depth = depth
        self.out_ch = out_ch
        self.l2_scale = l2_scale
        self.levels = []
        for i in range(depth - 2, -1, -1):
            upsample = self.track(Upsample(filters=base_filters * (2 ** i), groups=groups, data_format=data_format,
                                           l2_scale=l2_scale, name='%s/L%d/up' % (self.name, i)))
            conv = self.track(ResnetBlock(filters=base_filters * (2 ** i), groups=groups, reduction=reduction,
                                          data_format=data_format, l2_scale=l2_scale,
                                          name='%s/L%d/res' % (self.name, i)))
            self.levels.append([upsample, conv])

    def build(self, input_shape):
        top, res_shapes = input_shape
        shp = tuple(top)
        for (upsample, conv), rs in zip(self.levels, list(res_shapes)[::-1]):
            upsample.build(shp)
            shp = upsample.compute_output_shape(shp)
            conv.build(shp[:4] + (rs[-1] + shp[-1],))
            shp = shp[:4] + (conv.filters,)
        self.out_k = self.add_weight('out_k', (1, 1, 1, shp[-1], self.out_ch), 'glorot_normal', self.l2_scale)
        self.out_b = self.add_weight('out_b', (self.out_ch,), 'zeros')
        self.built = True

    @property
    def trainable_variables(self):
        out = []
        for upsample, conv in self.levels:
            out.extend(upsample.trainable_variables)
            out.extend(conv.trainable_variables)
        return out + list(self._params)

    def call(self, inputs, training=None):
        import torch
        x, residuals = inputs
        x = as_tensor(x)
        for (upsample, conv), residual in zip(self.levels, residuals[::-1]):
            residual = as_tensor(residual)
            f = upsample.filters
            cres = residual.shape[-1]
            slab = residual.base
            if slab is not None and residual.c0 == 0 and slab.used == cres and slab.t.shape[-1] >= cres + f:
                up_view = slab.view(cres, cres + f)
                slab.used = cres + f
            else:  # generic path: materialise [residual | up] once
                n, d, h, w, _ = residual.shape
                slab = Slab(n, d, h, w, cres + f, residual.t.device)
                rv = slab.view(0, cres)
                ops.add_strided(rv.t, residual.t, False)
                tape = current_tape()
                if tape is not None and residual.requires_grad:
                    def backward(rv=rv, residual=residual):
                        if rv.grad is None:
                            return
                        buf, acc = residual.grad_slot()
                        ops.add_strided(buf, rv.grad, acc)
                    tape.record(backward)
                up_view = slab.view(cres, cres + f)
                slab.used = cres + f
            upsample(x, training=training, out=up_view)                           # decoder.py:72
            x = conv(slab.view(0, cres + f), training=training)                   # decoder.py:75-78
        # output conv 1x1x1 + sigmoid (decoder.py:55-63,80)
        cin = x.shape[-1]
        wp = self.packed('out_f', ops.K1, ops.ROLE_FWD, self.out_k, cin, self.out_ch)
        yt = ops.conv_fwd(ops.K1, x.t, wp, self.out_b.t, self.out_ch, sigmoid=True)
        y = Tensor(yt)
        tape = current_tape()
        if tape is not None:
            xin = x

            def backward():
                dy = y.grad
                if dy is None:
                    return
                dpre = ops.sigmoid_bwd(yt, dy)
                if xin.requires_grad:
                    dx, acc = xin.grad_slot()
                    wpb = self.packed('out_b', ops.K1, ops.ROLE_BWD, self.out_k, cin, self.out_ch)
                    ops.conv_bwd_data(ops.K1, dpre, wpb, dx, acc)
                _wgrad(ops.K1, xin.t, dpre, self.out_k, self.out_b)
            tape.record(backward)
        return y

    def get_config(self):
        return self.config
