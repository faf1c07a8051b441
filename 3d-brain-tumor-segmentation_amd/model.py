"""Model -- drop-in for model.py of the reference (constructor :9-56, call :58-71).

Model(**model_args)(x, training, inference) -> (y_pred, y_vae, z_mean, z_logvar), or (y_pred, None, None, None) when
inference=True (the VAE branch is skipped, :67-68).  After the first call all variables are re-homed into ONE flat
fp32 parameter buffer (plus flat gradient / Adam-moment buffers) so that the optimiser, the L2 regulariser and the
RCCL gradient all-reduce each touch a single contiguous range.
"""
import numpy as np
import torch

from . import ops, parallel
from .layers._base import Layer, Tensor, as_tensor, check_data_format, current_tape, mark_public_layout, to_internal
from .layers.decoder import Decoder
from .layers.encoder import Encoder
from .layers.vae import VariationalAutoencoder
from .tape import bump_weights_epoch


class Model(Layer):
    def __init__(self, data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, dropout=0.2,
                 downsampling='conv', upsampling='conv', base_filters=16, depth=4, in_ch=2, out_ch=3):
        super(Model, self).__init__(name='model')
        self.data_format = check_data_format(data_format)
        self.epoch = _EpochVariable()                                              # model.py:29
        self.in_ch = in_ch
        self.encoder = self.track(Encoder(data_format=data_format, groups=groups, reduction=reduction, l2_scale=l2_scale,
                                          dropout=dropout, downsampling=downsampling, base_filters=base_filters,
                                          depth=depth, _reserve_for_decoder=True))
        self.decoder = self.track(Decoder(data_format=data_format, groups=groups, reduction=reduction, l2_scale=l2_scale,
                                          upsampling=upsampling, base_filters=base_filters, depth=depth, out_ch=out_ch))
        self.vae = self.track(VariationalAutoencoder(data_format=data_format, groups=groups, reduction=reduction,
                                                     l2_scale=l2_scale, upsampling=upsampling,
                                                     base_filters=base_filters, depth=depth, out_ch=in_ch))
        self.flat_params = None
        self.flat_grads = None
        self._l2_ranges = None

    # ---- build: weights for a given input shape, no kernels launched ----
    def build(self, input_shape):
        shp = tuple(input_shape)
        self.encoder.build(shp)
        res_shapes = []
        s = shp
        for i in range(self.encoder.depth):
            f = self.encoder.base_filters * 2 ** i
            s = s[:4] + (f * (i + 1),)
            res_shapes.append(s)
            if i < self.encoder.depth - 1:
                s = tuple(self.encoder.levels[i][1].compute_output_shape(s))   # conv: f channels; max pooling keeps them all
        self.decoder.build((res_shapes[-1], res_shapes[:-1]))
        self.vae.build(res_shapes[-1])
        self.built = True
        self._flatten_parameters()

    def _backward_groups(self):
        """The variables grouped by layer call, in the order the backward pass FINISHES their gradients: the tape replays the forward's
        layer calls in reverse (model.py:63-68: encoder, decoder, vae -> vae, decoder, encoder level 3 .. 0).  The flat buffers follow
        this order so that the gradient buckets of parallel.GradSync ("reverse-layer order") complete front to back while
        the backward pass is still running, and the last bucket holds only the shallow encoder levels."""
        enc, dec, vae = self.encoder, self.decoder, self.vae
        own = lambda lay, names: [getattr(lay, n) for n in names if getattr(lay, n, None) is not None]
        groups = [own(vae, ['out_k', 'out_b'])]
        for up, blk in reversed(vae.levels):
            groups += [blk.trainable_variables, up.trainable_variables]
        groups += [vae.upsample.trainable_variables, own(vae, ['unproj_k', 'unproj_b', 'proj_k', 'proj_b']), vae.downsample.trainable_variables]
        groups.append(own(dec, ['out_k', 'out_b']))
        for up, blk in reversed(dec.levels):
            groups += [blk.trainable_variables, up.trainable_variables]
        for convs, down in reversed(enc.levels):
            if down is not None:
                groups.append(down.trainable_variables)
            for blk in reversed(convs):
                groups.append(blk.trainable_variables)
        groups = [g for g in groups if g]
        seen = [id(p) for g in groups for p in g]
        if len(seen) != len(set(seen)) or set(seen) != set(id(p) for p in self.trainable_variables):
            raise RuntimeError('backward grouping does not cover the variables exactly once')
        return groups

    def _flatten_parameters(self):
        """one contiguous buffer in backward-completion order; inside a layer group [regularised (by coefficient) | unregularised], so
        the regulariser touches one short list of ranges (ops.l2_reg_*: up to 128) and Adam / the all-reduce one contiguous buffer"""
        groups = self._backward_groups()
        total = sum(p.t.numel() for g in groups for p in g)
        pad = (-total) % 4
        dev = groups[0][0].t.device
        flat = torch.empty(total + pad, dtype=torch.float32, device=dev)
        grads = torch.zeros(total + pad, dtype=torch.float32, device=dev)
        if pad:
            flat[total:] = 0
        off = 0
        ranges, spans = [], []
        for grp in groups:
            start = off
            coefs = sorted(set(p.l2 for p in grp if p.l2 > 0), reverse=True)
            for cls in coefs + [0.0]:
                c0 = off
                for p in grp:
                    if (p.l2 if p.l2 > 0 else 0.0) != cls:
                        continue
                    n = p.t.numel()
                    flat[off:off + n].copy_(p.t.reshape(-1))
                    p.t = flat[off:off + n].view(p.t.shape)
                    p._gview = grads[off:off + n].view(p.t.shape)
                    off += n
                if cls > 0 and off > c0:
                    if ranges and ranges[-1][2] == cls and ranges[-1][0] + ranges[-1][1] == c0:
                        ranges[-1] = (ranges[-1][0], ranges[-1][1] + off - c0, cls)       # (adjacent ranges of one coefficient merge)
                    else:
                        ranges.append((c0, off - c0, cls))
            spans.append((start, off))
        if len(ranges) > 128:
            raise RuntimeError('more than 128 regulariser ranges')
        self.flat_params, self.flat_grads, self._l2_ranges = flat, grads, ranges
        self._group_spans = spans          # (start, end) of every layer group, backward-completion order
        self.n_params = total
        bump_weights_epoch()

    def call(self, inputs, training=None, inference=None):
        assert (not inference or not training), 'Cannot run training and inference modes simultaneously.'
        x = as_tensor(inputs, data_format=self.data_format)
        residuals = self.encoder(x, training=training)
        y_pred = self.decoder((residuals[-1], residuals[:-1]), training=training)
        if inference:
            return (y_pred, None, None, None)
        y_vae, z_mean, z_logvar = self.vae(residuals[-1], training=training)
        return (y_pred, y_vae, z_mean, z_logvar)

    def __call__(self, inputs, training=None, inference=None):
        cf = self.data_format == 'channels_first'
        if cf:
            inputs = to_internal(inputs)     # raw NCDHW -> engine Tensor (NDHWC memory)
        if not self.built:
            self.build(tuple(inputs.shape))
        out = self.call(inputs, training=training, inference=inference)
        return mark_public_layout(out) if cf else out

    @property
    def trainable_variables(self):
        return self.encoder.trainable_variables + self.decoder.trainable_variables + self.vae.trainable_variables

    @property
    def losses(self):
        """[sum of all L2 regularisers] as a single fused term (train.py:146 reduces the list with a sum anyway)"""
        if self.flat_params is None or not self._l2_ranges:
            return Layer.losses.fget(self)
        val = Tensor(ops.l2_reg_fwd(self.flat_params, self._l2_ranges))
        tape = current_tape()
        if tape is not None:
            gen = tape.gen

            self._l2_val, self._l2_val_gen = val, tape.gen

            def backward():
                g = val.grad
                if g is None:
                    return
                from .tape import current_or_replaying_tape
                tp = current_or_replaying_tape()
                if tp is not None and tp.grad_sync is not None:
                    return  # the gradient-sync hook applies the regulariser per bucket before each all-reduce
                # every parameter's gradient has been written by its layer by now (this node replays last)
                ops.join_side_stream()
                for p in self.trainable_variables:
                    if p._gen != gen:
                        ops.fill(p._gview, 0.0)
                        p._gen = gen
                k = parallel.l2_grad_scale()  # rank-identical term: pre-divide so the summing all-reduce restores it
                ops.l2_reg_bwd(self.flat_params, self.flat_grads, [(o, l, c * k) for o, l, c in self._l2_ranges], g)
            tape.nodes.insert(0, backward)  # replay after all layer nodes regardless of where .losses was read
        return [val]

    # ---- weights I/O (train.py:100,201 use Keras HDF5; h5py is absent here, so a flat .npz keyed by variable name) ----
    def save_weights(self, path):
        d = {p.name: p.t.detach().cpu().numpy() for p in self.trainable_variables}
        d['epoch'] = np.asarray(self.epoch.value().numpy())
        np.savez(path if str(path).endswith('.npz') else str(path) + '.npz', **d)

    def load_weights(self, path):
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        for p in self.trainable_variables:
            p.t.copy_(torch.from_numpy(z[p.name]).to(p.t.device).reshape(p.t.shape))
        if 'epoch' in z:
            self.epoch.assign(int(z['epoch']))
        bump_weights_epoch()

    # ---- Keras-order weight lists (importer keyed by the reference's variable order) ----
    def get_weights(self):
        """tf.keras.Model.get_weights(): `model.weights` order = trainable variables in layer-tracking order (model.py:29-56:
        encoder, decoder, vae; inside a layer its sub-layers in constructor order, kernel before bias, gamma before beta --
        group_norm.py:63-80; `vae.unproj` last because vae.py:105 creates it in build()), then the non-trainable `epoch`
        variable (model.py:29).  Arrays keep the Keras layouts ((kd,kh,kw,Cin,Cout), transposed convs (kd,kh,kw,Cout,Cin),
        Dense (in,out))."""
        return [p.t.detach().cpu().numpy().copy() for p in self.trainable_variables] + [np.asarray(self.epoch.numpy(), dtype=np.int32)]

    def set_weights(self, weights):
        """tf.keras.Model.set_weights(): the list get_weights() returns -- e.g. exported from the reference with
        `np.savez(path, *model.get_weights())` where h5py / Keras HDF5 (train.py:100,201) are available.  The trailing epoch
        entry is optional.  Shapes are checked one by one: a mismatch names the variable."""
        ps = self.trainable_variables
        weights = list(weights)
        if len(weights) not in (len(ps), len(ps) + 1):
            raise ValueError('expected %d (or %d with epoch) arrays, got %d' % (len(ps), len(ps) + 1, len(weights)))
        for p_, w in zip(ps, weights):
            w = np.asarray(w)
            if tuple(w.shape) != tuple(p_.t.shape):
                raise ValueError('weight for %s has shape %s, expected %s' % (p_.name, tuple(w.shape), tuple(p_.t.shape)))
        for p_, w in zip(ps, weights):
            p_.t.copy_(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(p_.t.device))
        if len(weights) == len(ps) + 1:
            self.epoch.assign(int(np.asarray(weights[-1])))
        bump_weights_epoch()

    def load_weights_keras_order(self, path):
        """weights exported as positional arrays (`np.savez(path, *keras_model.get_weights())` -> arr_0, arr_1, ...)"""
        z = np.load(path)
        self.set_weights([z['arr_%d' % i] for i in range(len(z.files))])

    def set_weights_from(self, named):
        """named: dict name -> array (oracle ParamSet naming, 'encoder/L0/B0/ptwise_k' ...) for parity runs"""
        for p in self.trainable_variables:
            key = p.name.replace('/gn1/gamma', '/gn1_g').replace('/gn1/beta', '/gn1_b').replace('/gn2/gamma', '/gn2_g') \
                .replace('/gn2/beta', '/gn2_b').replace('/gn/gamma', '/gn_g').replace('/gn/beta', '/gn_b')
            v = named[key]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            p.t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(p.t.device).reshape(p.t.shape))
        bump_weights_epoch()

    def oracle_name(self, p):
        return p.name.replace('/gn1/gamma', '/gn1_g').replace('/gn1/beta', '/gn1_b').replace('/gn2/gamma', '/gn2_g') \
            .replace('/gn2/beta', '/gn2_b').replace('/gn/gamma', '/gn_g').replace('/gn/beta', '/gn_b')


class _EpochVariable(object):
    """tf.Variable(0, name='epoch', trainable=False) stand-in (model.py:29; train.py:133-135)"""

    def __init__(self):
        self._v = 0

    def assign(self, v):
        self._v = int(v)

    def value(self):
        return self

    def numpy(self):
        return self._v
