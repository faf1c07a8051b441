"""GroupNormalization -- drop-in for layers/group_norm.py of the reference (constructor :10-22, build :42-81,
call :83-124).  axis=-1 reproduces the reference's channels_last behaviour exactly: groups are contiguous 1/G
chunks of each sample's flattened (D,H,W,C) memory, handled by the BTS_GN_SLAB kernels."""
from .. import ops
from ._base import Layer, Tensor, as_tensor, current_tape


class GroupNormalization(Layer):
    def __init__(self, groups=8, axis=-1, epsilon=1e-5, center=True, scale=True, beta_initializer='zeros',
                 gamma_initializer='ones', beta_regularizer=None, gamma_regularizer=None, beta_constraint=None,
                 gamma_constraint=None, **kwargs):
        super(GroupNormalization, self).__init__(name=kwargs.pop('name', None))
        self._semantics = kwargs.pop('semantics', None)   # engine layers: GN mode on internal NDHWC shapes (axis stays -1)
        self.supports_masking = True
        self.groups = groups
        self.axis = axis
        self.epsilon = epsilon
        self.center = center
        self.scale = scale
        self.beta_initializer = beta_initializer
        self.gamma_initializer = gamma_initializer
        # regularisers are l2 coefficients (float) or None; the reference passes tf.keras.regularizers.l2(l)
        self.beta_regularizer = beta_regularizer
        self.gamma_regularizer = gamma_regularizer
        self.beta_constraint = beta_constraint
        self.gamma_constraint = gamma_constraint
        self.gamma = None
        self.beta = None

    def build(self, input_shape):
        dim = input_shape[self.axis]
        if dim is None:
            raise ValueError('Axis ' + str(self.axis) + ' of input tensor should have a defined dimension '
                             'but the layer received an input with shape ' + str(input_shape) + '.')
        if dim < self.groups:
            raise ValueError('Number of groups (' + str(self.groups) + ') cannot be more than the number of channels (' +
                             str(dim) + ').')
        if dim % self.groups != 0:
            raise ValueError('Number of groups (' + str(self.groups) + ') must be a multiple of the number of channels (' +
                             str(dim) + ').')
        nd = len(input_shape)
        ax = self.axis if self.axis >= 0 else nd + self.axis
        if self._semantics is not None:      # built by an engine layer on internal NDHWC shapes
            self._mode = self._semantics
        elif ax == nd - 1:
            self._mode = ops.GN_SLAB
        elif ax == 1:                        # standalone channels_first use: raw NCDHW in, true GroupNorm
            self._mode = ops.GN_CHANNEL
        else:
            raise ValueError('GroupNormalization supports axis=-1 (channels_last) or axis=1 (channels_first)')
        if self.scale:
            self.gamma = self.add_weight('gamma', (dim,), self.gamma_initializer, l2=self.gamma_regularizer or 0.0)
        if self.center:
            self.beta = self.add_weight('beta', (dim,), self.beta_initializer, l2=self.beta_regularizer or 0.0)
        self.built = True

    def call(self, inputs, training=None, relu=False, out=None, **kwargs):
        cf = self._semantics is None and self.axis == 1
        x = as_tensor(inputs, data_format='channels_first' if cf else 'channels_last')
        y, _ = group_norm_forward(self, x, relu=relu, out=out)
        y.cf = x.cf
        return y

    def get_config(self):
        return {'groups': self.groups, 'axis': self.axis, 'epsilon': self.epsilon, 'center': self.center,
                'scale': self.scale, 'beta_initializer': self.beta_initializer,
                'gamma_initializer': self.gamma_initializer, 'beta_regularizer': self.beta_regularizer,
                'gamma_regularizer': self.gamma_regularizer, 'beta_constraint': self.beta_constraint,
                'gamma_constraint': self.gamma_constraint}


def group_norm_forward(norm, x, relu, out=None, record=True):
    """x: dense Tensor [N,D,H,W,C]. Returns (y Tensor, saved) and records the backward on the active tape."""
    import torch
    if not x.t.is_contiguous():
        raise RuntimeError('GroupNormalization input must be a dense tensor')
    c = x.t.shape[-1]
    dev = x.t.device
    gamma = norm.gamma.t if norm.gamma is not None else ops.fill(torch.empty(c, device=dev), 1.0)
    beta = norm.beta.t if norm.beta is not None else ops.fill(torch.empty(c, device=dev), 0.0)
    mean, rstd = ops.gn_stats(x.t, norm.groups, norm._mode, norm.epsilon)
    yt = ops.gn_apply(x.t, gamma, beta, mean, rstd, norm.groups, norm._mode, relu, out=None if out is None else out.t)
    y = out if out is not None else Tensor(yt)
    tape = current_tape()
    if tape is not None and record:
        def backward():
            dy = y.grad
            if dy is None:
                return
            dx = group_norm_backward(norm, x.t, dy, gamma, beta, mean, rstd, relu)
            if x.requires_grad:
                buf, acc = x.grad_slot()
                ops.add_strided(buf, dx, acc)
        tape.record(backward)
    return y, (gamma, beta, mean, rstd)


def group_norm_backward(norm, xt, dy, gamma, beta, mean, rstd, relu):
    """-> dx (dense); writes dgamma/dbeta into the parameters' grad slots"""
    import torch
    c = xt.shape[-1]
    dg = db = None
    accs = []
    for p in (norm.gamma, norm.beta):
        if p is not None:
            buf, acc = p.grad_slot()
            accs.append((buf, acc))
        else:
            accs.append((torch.empty(c, device=xt.device), False))
    (dg, ag), (db, ab) = accs
    if ag != ab:  # mixed first-write state: fall back to accumulate on zeroed buffers
        if not ag:
            ops.fill(dg, 0.0)
        if not ab:
            ops.fill(db, 0.0)
        ag = ab = True
    return ops.gn_bwd(xt, dy, gamma, beta, mean, rstd, dg, db, norm.groups, norm._mode, relu, accumulate_params=ag)
