"""Down-sampling layers -- drop-in for layers/downsample.py of the reference.

ConvDownsample (:14-48): Conv3D k3 s2 'same' (he_normal, L2) -> GroupNorm (no regulariser) -> ReLU.
TF 'same' for k=3,s=2 on even sizes pads (0,1): output o reads inputs 2o..2o+2, index n reads zero.
MaxDownsample (:51-70): MaxPooling3D 2/2, the non-default `--downsampling max` option."""
from .. import ops
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape, gn_mode_of
from .group_norm import GroupNormalization, group_norm_backward
from .resnet import _wgrad


def get_downsampling(downsampling):
    if downsampling == 'max':
        return MaxDownsample
    elif downsampling == 'conv':
        return ConvDownsample
    # like the reference (:7-11), any other string (e.g. argparse's 'avg') yields None


class ConvDownsample(Layer):
    def __init__(self, filters, data_format='channels_last', groups=8, l2_scale=1e-5, name=None, **kwargs):
        super(ConvDownsample, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'filters': filters, 'data_format': data_format, 'groups': groups, 'l2_scale': l2_scale}
        self.filters = filters
        self.groups = groups
        self.l2_scale = l2_scale
        self.norm = self.track(GroupNormalization(groups=groups, axis=-1, name=self.name + '/gn', semantics=gn_mode_of(data_format)))

    def build(self, input_shape):
        cin = input_shape[-1]
        self.cin = cin
        self.conv_k = self.add_weight('conv_k', (3, 3, 3, cin, self.filters), 'he_normal', self.l2_scale)
        self.conv_b = self.add_weight('conv_b', (self.filters,), 'zeros')
        self.norm.build((None, None, None, None, self.filters))
        self.built = True

    def compute_output_shape(self, s):
        return (s[0], s[1] // 2, s[2] // 2, s[3] // 2, self.filters)

    def call(self, inputs, training=None, out=None):
        x = as_tensor(inputs, data_format=self.data_format)
        if any(s % 2 for s in x.shape[1:4]):
            raise ValueError('ConvDownsample needs even spatial sizes (TF SAME pads (0,1) only then), got %s' % (x.shape,))
        f, g = self.filters, self.groups
        wp = self.packed('f', ops.K3S2, ops.ROLE_FWD, self.conv_k, self.cin, f)
        if self.norm._mode == ops.GN_SLAB:
            c, mean, rstd = ops.conv_fwd_gn(ops.K3S2, x.t, wp, self.conv_b.t, f, g, self.norm.epsilon)
        else:
            c = ops.conv_fwd(ops.K3S2, x.t, wp, self.conv_b.t, f)
            mean, rstd = ops.gn_stats(c, g, self.norm._mode, self.norm.epsilon)
        yt = ops.gn_apply(c, self.norm.gamma.t, self.norm.beta.t, mean, rstd, g, self.norm._mode, True,
                          out=None if out is None else out.t)
        y = out if out is not None else Tensor(yt)
        y.cf = self.data_format == 'channels_first'
        tape = current_tape()
        if tape is not None:
            def backward():
                dy = y.grad
                if dy is None:
                    return
                dc = group_norm_backward(self.norm, c, dy, self.norm.gamma.t, self.norm.beta.t, mean, rstd, True)
                if x.requires_grad:
                    dx, acc = x.grad_slot()
                    wpb = self.packed('b', ops.K3S2, ops.ROLE_BWD, self.conv_k, self.cin, f)
                    ops.conv_bwd_data(ops.K3S2, dc, wpb, dx, acc)
                _wgrad(ops.K3S2, x.t, dc, self.conv_k, self.conv_b)
            tape.record(backward)
        return y

    def get_config(self):
        return self.config


class MaxDownsample(Layer):
    """MaxPooling3D(pool_size=2, strides=2, padding='same') (downsample.py:51-70); extra constructor arguments of the
    conv variant are accepted and ignored like the reference's **kwargs.  The channel count is unchanged."""

    def __init__(self, data_format='channels_last', name=None, **kwargs):
        super(MaxDownsample, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'data_format': data_format}

    def compute_output_shape(self, s):
        return (s[0], s[1] // 2, s[2] // 2, s[3] // 2, s[4])

    def call(self, inputs, training=None):
        x = as_tensor(inputs, data_format=self.data_format)
        if any(s % 2 for s in x.shape[1:4]):
            raise ValueError('MaxDownsample needs even spatial sizes, got %s' % (x.shape,))
        yt, idx = ops.maxpool2_fwd(x.t)
        y = Tensor(yt)
        tape = current_tape()
        if tape is not None and x.requires_grad:
            def backward():
                dy = y.grad
                if dy is None:
                    return
                dx, acc = x.grad_slot()
                ops.maxpool2_bwd(dy, idx, dx, acc)
            tape.record(backward)
        return y

    def get_config(self):
        return self.config
