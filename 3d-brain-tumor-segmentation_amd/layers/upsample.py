"""Up-sampling layers -- drop-in for layers/upsample.py of the reference.

ConvUpsample (:14-46): Conv3DTranspose k3 s2 'same' (Glorot-uniform, bias, NO L2: l2_scale is accepted but unused,
SURVEY F10) -> GroupNorm -> ReLU.  The transposed conv runs in gather form over the 8 output-parity classes so that
every output voxel is written exactly once (deterministic, no atomics).
LinearUpsample (:49-79) is a non-default option: SURVEY 8(f-4) "next" row."""
from .. import ops
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape, gn_mode_of
from .group_norm import GroupNormalization, group_norm_backward
from .resnet import _wgrad


def get_upsampling(upsampling):
    if upsampling == 'linear':
        return LinearUpsample
    elif upsampling == 'conv':
        return ConvUpsample


class ConvUpsample(Layer):
    def __init__(self, filters, groups=8, data_format='channels_last', l2_scale=1e-5, name=None, **kwargs):
        super(ConvUpsample, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'filters': filters, 'data_format': data_format, 'groups': groups, 'l2_scale': l2_scale}
        self.filters = filters
        self.groups = groups
        self.norm = self.track(GroupNormalization(groups=groups, axis=-1, name=self.name + '/gn', semantics=gn_mode_of(data_format)))

    def build(self, input_shape):
        cin = input_shape[-1]
        self.cin = cin
        self.conv_k = self.add_weight('conv_k', (3, 3, 3, self.filters, cin), 'glorot_uniform', 0.0, transposed=True)
        self.conv_b = self.add_weight('conv_b', (self.filters,), 'zeros')
        self.norm.build((None, None, None, None, self.filters))
        self.built = True

    def compute_output_shape(self, s):
        return (s[0], s[1] * 2, s[2] * 2, s[3] * 2, self.filters)

    def call(self, inputs, training=None, out=None):
        x = as_tensor(inputs, data_format=self.data_format)
        f, g = self.filters, self.groups
        wp = self.packed('f', ops.K3S2T, ops.ROLE_FWD, self.conv_k, self.cin, f)
        c = ops.conv_fwd(ops.K3S2T, x.t, wp, self.conv_b.t, f)
        mean, rstd = ops.gn_stats(c, g, self.norm._mode, self.norm.epsilon)
        yt = ops.gn_apply(c, self.norm.gamma.t, self.norm.beta.t, mean, rstd, g, self.norm._mode, True,
                          out=None if out is None else out.t)
        y = out if out is not None else Tensor(yt)
        tape = current_tape()
        if tape is not None:
            def backward():
                dy = y.grad
                if dy is None:
                    return
                dc = group_norm_backward(self.norm, c, dy, self.norm.gamma.t, self.norm.beta.t, mean, rstd, True)
                if x.requires_grad:
                    dx, acc = x.grad_slot()
                    wpb = self.packed('b', ops.K3S2T, ops.ROLE_BWD, self.conv_k, self.cin, f)
                    ops.conv_bwd_data(ops.K3S2T, dc, wpb, dx, acc)
                _wgrad(ops.K3S2T, x.t, dc, self.conv_k, self.conv_b)
            tape.record(backward)
        return y

    def get_config(self):
        return self.config


class LinearUpsample(Layer):
    """Conv3D 1x1x1 (he_normal, L2) followed by UpSampling3D(size=2), i.e. nearest-neighbour repetition -- "linear" is the
    reference's name for it (upsample.py:49-79); the non-default `--upsampling linear` option."""

    def __init__(self, filters, data_format='channels_last', l2_scale=1e-5, name=None, **kwargs):
        super(LinearUpsample, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'filters': filters, 'data_format': data_format, 'l2_scale': l2_scale}
        self.filters = filters
        self.l2_scale = l2_scale

    def build(self, input_shape):
        cin = input_shape[-1]
        self.cin = cin
        self.ptwise_k = self.add_weight('ptwise_k', (1, 1, 1, cin, self.filters), 'he_normal', self.l2_scale)
        self.ptwise_b = self.add_weight('ptwise_b', (self.filters,), 'zeros')
        self.built = True

    def compute_output_shape(self, s):
        return (s[0], s[1] * 2, s[2] * 2, s[3] * 2, self.filters)

    def call(self, inputs, training=None, out=None):
        x = as_tensor(inputs, data_format=self.data_format)
        f = self.filters
        wp = self.packed('f', ops.K1, ops.ROLE_FWD, self.ptwise_k, self.cin, f)
        c = ops.conv_fwd(ops.K1, x.t, wp, self.ptwise_b.t, f)
        yt = ops.upsample2_fwd(c, out=None if out is None else out.t)
        y = out if out is not None else Tensor(yt)
        tape = current_tape()
        if tape is not None:
            def backward():
                dy = y.grad
                if dy is None:
                    return
                dc = ops.upsample2_bwd(dy)
                if x.requires_grad:
                    dx, acc = x.grad_slot()
                    wpb = self.packed('b', ops.K1, ops.ROLE_BWD, self.ptwise_k, self.cin, f)
                    ops.conv_bwd_data(ops.K1, dc, wpb, dx, acc)
                _wgrad(ops.K1, x.t, dc, self.ptwise_k, self.ptwise_b)
            tape.record(backward)
        return y

    def get_config(self):
        return self.config
