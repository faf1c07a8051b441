"""Keras-like Layer protocol for the engine (the drop-in boundary): objects constructed with config
kwargs, weights created in build(input_shape), invoked as layer(inputs, training=...), exposing get_config(),
.trainable_variables and .losses.  Initialisers follow SURVEY A.11 (Keras VarianceScaling / Glorot)."""
import math
import weakref

import torch

from .. import ops
from ..tape import Param, Tensor, as_tensor, bump_weights_epoch, current_tape, default_device, weights_epoch

_gen = torch.Generator()
_gen.manual_seed(0)


def set_seed(seed):
    """seed the host-side weight initialiser (the reference seeds nothing: SURVEY F11)"""
    _gen.manual_seed(int(seed))


def _fans(shape, transposed=False):
    rf = 1
    for s in shape[:-2]:
        rf *= s
    if len(shape) < 2:
        return shape[0], shape[0]
    cin, cout = (shape[-1], shape[-2]) if transposed else (shape[-2], shape[-1])
    return rf * cin, rf * cout


def initialise(shape, init, transposed=False):
    """host-side draw in fp64, rounded to fp32"""
    if init == 'zeros':
        return torch.zeros(shape, dtype=torch.float32)
    if init == 'ones':
        return torch.ones(shape, dtype=torch.float32)
    fi, fo = _fans(shape, transposed)
    if init in ('he_normal', 'glorot_normal'):
        std = math.sqrt(2.0 / fi) if init == 'he_normal' else math.sqrt(2.0 / (fi + fo))
        s = std / 0.87962566103423978  # Keras truncated-normal variance correction
        t = torch.empty(shape, dtype=torch.float64)
        torch.nn.init.trunc_normal_(t, 0.0, s, -2 * s, 2 * s, generator=_gen)
        return t.float()
    if init == 'glorot_uniform':
        lim = math.sqrt(6.0 / (fi + fo))
        return ((torch.rand(shape, dtype=torch.float64, generator=_gen) * 2 - 1) * lim).float()
    raise ValueError('unknown initializer %r' % (init,))


class Layer(object):
    def __init__(self, name=None):
        self.built = False
        self.name = name or self.__class__.__name__
        self._params = []
        self._sublayers = []
        self._packs = {}

    # ---- Keras protocol ----
    def add_weight(self, name, shape, initializer, l2=0.0, transposed=False):
        t = initialise(tuple(int(s) for s in shape), initializer, transposed).to(default_device())
        p = Param(t, '%s/%s' % (self.name, name), l2=l2, init=initializer)
        p.owner = self
        self._params.append(p)
        return p

    def track(self, layer):
        self._sublayers.append(layer)
        return layer

    def build(self, input_shape):
        self.built = True

    def compute_output_shape(self, input_shape):
        return tuple(input_shape)

    def call(self, inputs, training=None):
        raise NotImplementedError

    def __call__(self, inputs, training=None, **kwargs):
        cf = getattr(self, 'data_format', 'channels_last') == 'channels_first'
        if cf:
            inputs = to_internal(inputs)   # raw NCDHW tensors -> engine Tensors (NDHWC memory)
        if not self.built:
            self.build(_shape_of(inputs))
            self.built = True
        out = self.call(inputs, training=training, **kwargs)
        return mark_public_layout(out) if cf else out

    @property
    def trainable_variables(self):
        out = list(self._params)
        for l in self._sublayers:
            out.extend(l.trainable_variables)
        return out

    @property
    def losses(self):
        """regularisation terms, one 1-element Tensor per regularised variable (Keras: layer.losses)"""
        out = []
        for p in self.trainable_variables:
            if p.l2 > 0:
                out.append(_l2_term(p))
        return out

    def get_config(self):
        return dict(getattr(self, 'config', {}))

    # ---- packed weight images for the implicit-GEMM kernel, rebuilt when parameters change ----
    def packed(self, key, kind, role, param, cin_ref, cout, cin_slab=None, dup_start=0, dup_shift=0):
        """Packed image of `param` for (kind, role).  Images live in persistent buffers; the first request after the
        parameters changed (optimiser step, assign, load) re-packs EVERY registered image of the process in one launch
        (bts_conv_pack_batch) instead of one ~5 us launch per layer and role."""
        cin_slab = cin_ref if cin_slab is None else cin_slab
        sig = (kind, role, cin_ref, cout, cin_slab, dup_start, dup_shift)
        ent = self._packs.get(key)
        if ent is None or ent.sig != sig or ent.param is not param:
            ent = _PackEntry(self, sig, param, ops.conv_pack(kind, role, param.t, cin_ref, cout, cin_slab, dup_start, dup_shift))
            self._packs[key] = ent
            _pack_registry.append(ent)
        elif ent.epoch != weights_epoch():
            _repack_all()
        if role == ops.ROLE_BWD:
            _order_behind_bwd_pack()
        return ent.wp


class _PackEntry(object):
    __slots__ = ('layer', 'sig', 'param', 'wp', 'epoch')

    def __init__(self, layer, sig, param, wp):
        self.layer = weakref.ref(layer)
        self.sig, self.param, self.wp = sig, param, wp
        self.epoch = weights_epoch()


_pack_registry = []
_pack_tables = {}
_bwd_pack_event = {}      # device index -> (event recorded on the side stream after the data-gradient-role pack, streams already ordered behind it)


def _order_behind_bwd_pack():
    """Data-gradient-role images are packed on the weight-gradient stream (below).  Whoever is handed one must run behind that
    pack: tape.gradient joins the side streams before its first node, but a caller outside a tape replay (lowp_train's head /
    VAE-output data gradients) would otherwise read the previous step's image -- or a half-written one."""
    if not torch.cuda.is_available():
        return
    ent = _bwd_pack_event.get(torch.cuda.current_device())
    if ent is None:
        return
    ev, done = ent
    cur = torch.cuda.current_stream()
    if cur.cuda_stream not in done:
        cur.wait_event(ev)
        done.add(cur.cuda_stream)


def _repack_all():
    live = []
    for e in _pack_registry:
        lay = e.layer()
        if lay is not None and any(v is e for v in lay._packs.values()):
            live.append(e)
    _pack_registry[:] = live
    ep = weights_epoch()
    todo = [e for e in live if e.epoch != ep]
    batch = [e for e in todo if e.param.t.is_contiguous()]
    for e in todo:
        if not e.param.t.is_contiguous():
            kind, role, cin_ref, cout, cin_slab, dup_start, dup_shift = e.sig
            e.wp = ops.conv_pack(kind, role, e.param.t, cin_ref, cout, cin_slab, dup_start, dup_shift)
    # forward-role images are needed at once; data-gradient-role images only when the backward starts (tape.gradient joins the
    # side streams first): their half of the pack runs on the weight-gradient stream, idle during the forward
    by_dev = {}
    for e in batch:
        by_dev.setdefault((e.wp.device, e.sig[1] == ops.ROLE_BWD), []).append(e)
    for (dev, bwd), es in by_dev.items():
        entries = [(e.sig[0], e.sig[1], e.param.t, e.wp) + tuple(e.sig[2:]) for e in es]
        side = ops.side_stream('wgrad') if (bwd and dev.type == 'cuda') else None
        table = _pack_tables.setdefault((dev, bwd), ops.PackTable())
        if side is None:
            table.run(entries)
        else:
            side.wait_stream(torch.cuda.current_stream())     # the optimiser step that changed the parameters
            with torch.cuda.stream(side):
                table.run(entries)
                ev = torch.cuda.Event()
                ev.record(side)
            _bwd_pack_event[dev.index if dev.index is not None else torch.cuda.current_device()] = (ev, {side.cuda_stream})
    for e in todo:
        e.epoch = ep


def to_internal(x):
    """channels_first boundary: every raw 5-D tensor / array in a (possibly nested) input becomes an engine Tensor"""
    if isinstance(x, (tuple, list)):
        return type(x)(to_internal(e) for e in x)
    if isinstance(x, Tensor) or x is None:
        return x
    return as_tensor(x, data_format='channels_first')


def mark_public_layout(out):
    """flag 5-D result Tensors of a channels_first layer so that .numpy() / .public() export NCDHW"""
    if isinstance(out, (tuple, list)):
        return type(out)(mark_public_layout(e) for e in out)
    if isinstance(out, Tensor) and out.t.dim() == 5:
        out.cf = True
    return out


def _l2_term(p):
    flat = p.t.reshape(-1)
    val = Tensor(ops.l2_reg_fwd(flat, [(0, flat.numel(), p.l2)]))
    tape = current_tape()
    if tape is not None:
        def backward():
            g = val.grad
            if g is None:
                return
            buf, acc = p.grad_slot()
            if not acc:
                ops.fill(buf, 0.0)
            ops.l2_reg_bwd(flat, buf.reshape(-1), [(0, flat.numel(), p.l2)], g)
        tape.record(backward)
    return val


def _shape_of(x):
    if isinstance(x, (tuple, list)) and len(x) and not isinstance(x[0], int):
        return [_shape_of(e) for e in x]
    if isinstance(x, Tensor):
        return tuple(x.shape)
    if hasattr(x, 'shape'):
        return tuple(x.shape)
    return tuple(x)


def check_data_format(data_format):
    """channels_last | channels_first (the reference's --gpu layout, args.py:121-123).  The engine's memory is NDHWC either
    way; channels_first changes the PUBLIC layout of raw inputs / exported outputs, GroupNorm to true channel-group
    semantics and the Dice metric to the intended all-spatial reduction (F8)."""
    if data_format not in ('channels_last', 'channels_first'):
        raise ValueError('unknown data_format %r' % (data_format,))
    return data_format


def gn_mode_of(data_format):
    return ops.GN_SLAB if data_format == 'channels_last' else ops.GN_CHANNEL


__all__ = ['Layer', 'Param', 'Tensor', 'as_tensor', 'bump_weights_epoch', 'check_data_format', 'current_tape', 'gn_mode_of', 'mark_public_layout', 'to_internal',
           'initialise', 'set_seed']
