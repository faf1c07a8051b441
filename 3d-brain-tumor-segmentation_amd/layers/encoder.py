"""Encoder -- drop-in for layers/encoder.py of the reference (constructor :9-67, call :69-101).

Level i runs (i+1) ResnetBlocks with "dense" connections: block j>0 consumes Concatenate([inputs] + cache) where
`inputs` IS cache[-1] (encoder.py:83-87), i.e. [o_{j-1}, o_0, ..., o_{j-1}].  The engine keeps one
NDHWC slab per level, [o_0 | o_1 | ... | o_i | (spare for the decoder's up-sampled tensor)], hands block j the view
[0, j*F) with the duplicate slice folded into its weights, and lets it write its output straight into [j*F, (j+1)*F):
no Concatenate ever copies (encoder.py:52-58,85,91; decoder.py:75)."""
from .. import ops
from ..tape import Slab
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape
from .downsample import get_downsampling
from .resnet import ResnetBlock


class Encoder(Layer):
    def __init__(self, data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, dropout=0.2,
                 downsampling='conv', base_filters=16, depth=4, name='encoder', _reserve_for_decoder=False):
        super(Encoder, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'data_format': data_format, 'groups': groups, 'reduction': reduction, 'l2_scale': l2_scale,
                       'downsampling': downsampling, 'base_filters': base_filters, 'depth': depth}
        Downsample = get_downsampling(downsampling)
        self.dropout_rate = dropout
        self.depth = depth
        self.base_filters = base_filters
        self._reserve = _reserve_for_decoder
        self.levels = []
        for i in range(depth):
            convs = []
            for j in range(i + 1):
                conv = self.track(ResnetBlock(filters=base_filters * (2 ** i), groups=groups, reduction=reduction,
                                              data_format=data_format, l2_scale=l2_scale,
                                              name='%s/L%d/B%d' % (self.name, i, j)))
                convs.append(conv)
            downsample = self.track(Downsample(filters=base_filters * (2 ** i), groups=groups, data_format=data_format,
                                               l2_scale=l2_scale, name='%s/L%d/down' % (self.name, i))) \
                if i < depth - 1 else None
            self.levels.append([convs, downsample])
        self._mask = None
        self._seed = 0x5EED

    def build(self, input_shape):
        shp = tuple(input_shape)
        for i, (convs, downsample) in enumerate(self.levels):
            f = self.base_filters * 2 ** i
            for j, conv in enumerate(convs):
                if j == 0:
                    conv.build(shp)
                else:
                    conv.build(shp[:4] + (j * f,), fold=((j - 1) * f, f))
            out_c = f * (i + 1)
            shp = shp[:4] + (out_c,)
            if downsample is not None:
                downsample.build(shp)
                shp = downsample.compute_output_shape(shp)
        self.built = True

    def set_dropout_mask(self, mask):
        """inject the Bernoulli keep-mask (uint8/bool/float, in the layer's public layout: [N,D,H,W,C], or [N,C,D,H,W] for
        channels_first) for the next training call (parity runs)"""
        self._mask = mask

    def call(self, inputs, training=None):
        import torch
        x = as_tensor(inputs)
        dev = x.t.device
        if training and self.dropout_rate > 0:                                    # encoder.py:71
            if self._mask is not None:
                m = torch.as_tensor(self._mask)
                if self.data_format == 'channels_first' and m.dim() == 5:
                    m = m.permute(0, 2, 3, 4, 1)
                m = (m != 0).to(torch.uint8).to(dev).contiguous()
                self._mask = None
            else:
                self._seed += 1
                m = ops.dropout_mask(x.shape, self.dropout_rate, self._seed, dev)
            xin = x
            x = Tensor(ops.dropout_apply(x.t.contiguous(), m, self.dropout_rate), requires_grad=x.requires_grad)
            tape = current_tape()
            if tape is not None and xin.requires_grad:
                xo = x

                def backward():
                    if xo.grad is None:
                        return
                    buf, acc = xin.grad_slot()
                    ops.add_strided(buf, ops.dropout_apply(xo.grad, m, self.dropout_rate), acc)
                tape.record(backward)
        residuals = []
        cur = x
        n = x.shape[0]
        for i, (convs, downsample) in enumerate(self.levels):
            d, h, w = cur.shape[1:4]
            f = self.base_filters * 2 ** i
            nb = len(convs)
            spare = f if (self._reserve and i < self.depth - 1) else 0
            slab = Slab(n, d, h, w, nb * f + spare, dev)
            slab.used = nb * f
            for j, conv in enumerate(convs):
                out = slab.view(j * f, (j + 1) * f)
                if j == 0:
                    conv(cur, training=training, out=out)
                else:
                    conv(slab.view(0, j * f), training=training, out=out, fold=((j - 1) * f, f))  # encoder.py:83-87
            level_out = slab.view(0, nb * f)                                       # encoder.py:90-91
            residuals.append(level_out)
            if downsample is not None:
                cur = downsample(level_out, training=training)                    # encoder.py:97-98
        return residuals

    def get_config(self):
        return self.config
