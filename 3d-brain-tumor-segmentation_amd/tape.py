"""Gradient tape and tensor handles (the engine's replacement for tf.GradientTape, train.py:142-151).

The reference records every TF op and lets autodiff replay them.  Here each *layer call* records one backward
closure that launches hand-written HIP gradient kernels; `GradientTape.gradient` replays the closures in reverse.
Gradients are accumulated in place by the kernels themselves (accumulate flags), so no framework add-kernels run:
  * a plain Tensor's grad buffer is written by its first consumer (overwrite) and accumulated by later ones;
  * a channel-slice view of a Slab (virtual Concatenate) accumulates into the slab's zero-initialised grad buffer;
  * a Param's grad lives in the model's flat gradient buffer and uses the same first-write rule per tape.
torch only provides device memory here.
"""
import numpy as np
import torch

from . import ops

_current = None
_generation = 0
_weights_epoch = 0


def weights_epoch():
    return _weights_epoch


def bump_weights_epoch():
    """call after any in-place parameter update so cached packed weight images are rebuilt"""
    global _weights_epoch
    _weights_epoch += 1


def current_tape():
    return _current


def default_device():
    if torch.cuda.is_available():
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


class Tensor(object):
    """Handle on a device buffer: `.t` is the torch tensor (possibly a channel-slice view of a Slab)."""
    __slots__ = ('t', 'base', 'c0', 'requires_grad', '_grad', 'name', 'cf')

    def __init__(self, t, base=None, c0=0, requires_grad=True, name=None):
        self.t = t
        self.cf = False   # True: the PUBLIC layout of this tensor is channels_first (memory is NDHWC regardless)
        self.base = base
        self.c0 = c0
        self.requires_grad = requires_grad
        self._grad = None
        self.name = name

    @property
    def shape(self):
        return tuple(self.t.shape)

    def numpy(self):
        """host copy in the public layout (NCDHW when the producing layer was built with data_format='channels_first')"""
        t = self.t.detach()
        if self.cf and t.dim() == 5:
            t = t.permute(0, 4, 1, 2, 3)
        return t.cpu().numpy()

    def public(self):
        """device tensor in the public layout (a permuted view for channels_first)"""
        return self.t.permute(0, 4, 1, 2, 3) if (self.cf and self.t.dim() == 5) else self.t

    def __float__(self):
        return float(self.t.reshape(-1)[0].item())

    @property
    def grad(self):
        if self.base is not None:
            if hasattr(self.base, 'grad_read'):
                return self.base.grad_read(self.c0, self.c0 + self.t.shape[-1])
            g = self.base.grad_or_none()
            return None if g is None else g[..., self.c0:self.c0 + self.t.shape[-1]]
        return self._grad

    def grad_slot(self):
        """-> (buffer, accumulate): where a consumer's backward kernel must put d(loss)/d(self)"""
        if self.base is not None:
            if hasattr(self.base, 'grad_write'):
                return self.base.grad_write(self.c0, self.c0 + self.t.shape[-1])
            g = self.base.grad_full()
            return g[..., self.c0:self.c0 + self.t.shape[-1]], True
        if self._grad is None:
            self._grad = torch.empty(self.t.shape, dtype=torch.float32, device=self.t.device)
            return self._grad, False
        return self._grad, True

    def seed_grad(self, value=1.0):
        g, _ = self.grad_slot()
        ops.fill(g, value)

    # scalar arithmetic used by the training step: loss += sum(model.losses)  (train.py:145-146)
    def __add__(self, other):
        if other == 0 and not isinstance(other, Tensor):
            return self
        if not isinstance(other, Tensor):
            raise TypeError('only Tensor + Tensor (1-element) is supported')
        out = Tensor(ops.scalar_lincomb(self.t, other.t, 1.0, 1.0))
        tape = current_tape()
        if tape is not None:
            a, b = self, other

            def backward():
                g = out.grad
                if g is None:
                    return
                for src in (a, b):
                    if src.requires_grad:
                        buf, acc = src.grad_slot()
                        if acc:
                            ops.axpy(buf, g, 1.0)
                        else:
                            ops.add_strided(buf.reshape(1, 1), g.reshape(1, 1), False)
            tape.record(backward)
        return out

    __radd__ = __add__


class Slab(object):
    """A dense [N,D,H,W,Ctot] buffer whose channel ranges are handed out as Tensors: the engine's Concatenate
    (encoder.py:52-58,85,91; decoder.py:44-45,75) never copies."""

    def __init__(self, n, d, h, w, ctot, device):
        self.t = torch.empty((n, d, h, w, ctot), dtype=torch.float32, device=device)
        self.g = None
        self._cov = []
        self.used = 0

    def view(self, c0, c1, requires_grad=True):
        return Tensor(self.t[..., c0:c1], base=self, c0=c0, requires_grad=requires_grad)

    # The gradient buffer is NOT zero-filled: the first contribution to a channel range is written (accumulate = False), later ones
    # accumulate; `_cov` holds the channel intervals that have been written.  A range that is only partly covered (or read before
    # anything was written into it) has its uncovered channels zero-filled on demand.  In a training step the decoder block's input
    # view covers a level's whole slab and is its first writer, so nothing is filled at all (round 2 filled every slab gradient and
    # read it back in the first accumulation: ~1 GB of traffic per 128^3 step).
    def _uncovered(self, c0, c1):
        out, pos = [], c0
        for a, b in self._cov:
            if b <= pos or a >= c1:
                continue
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if pos < c1:
            out.append((pos, c1))
        return out

    def _cover(self, c0, c1):
        iv = sorted(self._cov + [(c0, c1)])
        merged = [iv[0]]
        for a, b in iv[1:]:
            if a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        self._cov = merged

    def _buffer(self):
        if self.g is None:
            self.g = torch.empty_like(self.t)
            self._cov = []
        return self.g

    def grad_write(self, c0, c1):
        """-> (gradient view of channels [c0, c1), accumulate) for a backward kernel about to contribute to it"""
        g = self._buffer()
        gaps = self._uncovered(c0, c1)
        if len(gaps) == 1 and gaps[0] == (c0, c1):      # untouched: this contribution is written
            self._cover(c0, c1)
            return g[..., c0:c1], False
        for a, b in gaps:                               # partly covered: zero what is missing, then accumulate
            g[..., a:b].zero_()
        if gaps:
            self._cover(c0, c1)
        return g[..., c0:c1], True

    def grad_read(self, c0, c1):
        if self.g is None:
            return None
        for a, b in self._uncovered(c0, c1):            # (never the case in a training step)
            self.g[..., a:b].zero_()
        self._cover(c0, c1)
        return self.g[..., c0:c1]

    def grad_full(self):
        g = self._buffer()
        for a, b in self._uncovered(0, g.shape[-1]):
            g[..., a:b].zero_()
        self._cov = [(0, g.shape[-1])]
        return g

    def grad_or_none(self):
        return self.g if self.g is None else self.grad_full()


class Param(Tensor):
    """Trainable variable in the reference's Keras layout. `l2` is its regulariser coefficient (0 = none)."""
    __slots__ = ('l2', 'init', '_gview', '_gen', 'owner')

    def __init__(self, t, name, l2=0.0, init=None):
        Tensor.__init__(self, t, requires_grad=True, name=name)
        self.l2 = float(l2)
        self.init = init
        self._gview = None
        self._gen = -1
        self.owner = None

    @property
    def grad(self):
        return self._gview if self._gen == _generation_of_last_tape() else None

    def grad_slot(self):
        tape = current_or_replaying_tape()
        if self._gview is None:
            self._gview = torch.empty(self.t.shape, dtype=torch.float32, device=self.t.device)
        gen = tape.gen if tape is not None else -2
        if tape is not None and tape._touched is not None:
            tape._touched.append(self)   # the replay loop reports these to the gradient-sync hook after the node
        if self._gen != gen:
            self._gen = gen
            return self._gview, False
        return self._gview, True

    def assign(self, value):
        v = torch.as_tensor(np.asarray(value), dtype=torch.float32).reshape(self.t.shape)
        self.t.copy_(v.to(self.t.device))
        bump_weights_epoch()


_replaying = None
_last_gen = -1


def current_or_replaying_tape():
    return _replaying if _replaying is not None else _current


def _generation_of_last_tape():
    return _last_gen


class GradientTape(object):
    """with GradientTape() as tape: ...forward...;  grads = tape.gradient(loss, model.trainable_variables)"""

    def __init__(self, persistent=False):
        self.nodes = []
        self.persistent = persistent
        self.gen = None
        self._prev = None
        self._touched = None   # list while a gradient-sync hook is attached to the replay
        self.grad_sync = None

    def __enter__(self):
        global _current, _generation
        self._prev = _current
        _current = self
        _generation += 1
        self.gen = _generation
        return self

    def __exit__(self, *exc):
        global _current
        _current = self._prev
        return False

    def record(self, fn):
        self.nodes.append(fn)

    def gradient(self, target, sources, grad_sync=None):
        """grad_sync (parallel.GradSync, optional): told after every node which parameters' gradients that node wrote, so
        that finished buckets of the flat gradient buffer are all-reduced while the rest of the backward still runs"""
        global _replaying, _last_gen
        if not isinstance(target, Tensor) or target.t.numel() != 1:
            raise ValueError('target must be a 1-element Tensor (the loss)')
        target.seed_grad(1.0)
        _replaying = self
        _last_gen = self.gen
        self.grad_sync = grad_sync
        self._touched = [] if grad_sync is not None else None
        try:
            ops.join_side_stream()      # data-gradient weight images packed on the side stream during the forward (layers/_base.py)
            if grad_sync is not None:
                grad_sync.begin(self)
            nodes = self.nodes if self.persistent else None
            seq = self.nodes
            self.nodes_replayed = 0
            for i in range(len(seq) - 1, -1, -1):
                seq[i]()
                self.nodes_replayed += 1
                if nodes is None:
                    seq[i] = None  # release saved activations as soon as they are consumed
                if self._touched:
                    grad_sync.params_written(self._touched)
                    self._touched = []
            ops.join_side_stream()      # weight gradients enqueued on the side stream (layers/resnet.py:_wgrad)
            if grad_sync is not None:
                grad_sync.finish()
        finally:
            _replaying = None
            self._touched = None
            self.grad_sync = None
        if not self.persistent:
            self.nodes = []
        return [s.grad for s in sources]


def as_tensor(x, requires_grad=False, data_format='channels_last'):
    """wrap user input (Tensor | torch tensor | numpy) as an engine Tensor on the default device.  Raw 5-D inputs of a
    channels_first layer are NCDHW and are re-laid out to the engine's NDHWC memory here (one HBM pass at the boundary)"""
    if isinstance(x, Tensor):
        return x
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not torch.is_tensor(x):
        raise TypeError('expected Tensor, torch.Tensor or numpy array, got %r' % type(x))
    if x.dtype != torch.float32:
        x = x.float()
    dev = default_device()
    if dev.type != 'cuda':
        raise RuntimeError('no MI355X visible: the engine has no CPU execution path')
    if not x.is_cuda:
        x = x.to(dev)
    if data_format == 'channels_first' and x.dim() == 5:
        out = Tensor(x.permute(0, 2, 3, 4, 1).contiguous(), requires_grad=requires_grad)
        out.cf = True
        return out
    return Tensor(x.contiguous() if not _is_slice_ok(x) else x, requires_grad=requires_grad)


def _is_slice_ok(x):
    return x.dim() != 5 or x.stride(-1) == 1
