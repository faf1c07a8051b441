"""Data-parallel glue: one process per GPU, torch.distributed (backend 'nccl' == RCCL over xGMI).

The reference is single-device (train.py:138); sharding the batch by sample needs exactly three exchanges:
  C1  gradient all-reduce (sum) over the model's flat gradient buffer, in a few large buckets sized for xGMI's
      point-to-point links (7 x ~153 GB/s per GPU: ring all-reduce is per-link bound, so few large messages win);
  C2  parameter broadcast from rank 0 after initialisation / load;
  C3  all-reduce of the raw Dice/MSE/KL sums (3*out_ch+4 doubles) and of the Dice-metric table, because the
      reference sums those over the batch axis (util.py:11,18-20).
Gradient scaling: the loss already is the GLOBAL-batch loss on every rank (C3), and each rank back-propagates only its
own samples' contribution, so per-rank activations' gradients SUM to the global gradient (no 1/world factor); the
weight-only L2 term is identical on every rank, so its gradient is pre-divided by world before the summing all-reduce.
"""
import os

import torch

# C1 bucket size.  64 MB: three buckets for the 168.7 MB fp32 gradient of the CLI model.  What to expect on a full-mesh xGMI node
# (7 links x ~153 GB/s per GPU, point to point): a RING all-reduce moves 2 (N-1)/N x bytes over ONE link pair per hop -- 168.7 MB on 8
# GPUs = 295 MB per link at <= 153 GB/s = 1.9 ms; a direct / mesh (one-shot reduce-scatter + all-gather over all 7 links) needs the same
# bytes over 7 links = 0.3 ms.  RCCL picks by message size and topology; NCCL_ALGO / RCCL's channel count are the
# knobs to A/B on the first 8-GPU run (bench.py prints per-bucket times and the exposed wait for that), the bucket size is this one:
# `bench.py --bucket-mb`, BTS_DP_BUCKET_MB, or set_bucket_bytes().
BUCKET_BYTES = int(float(os.environ.get('BTS_DP_BUCKET_MB', '64')) * (1 << 20))


def set_bucket_bytes(n):
    """C1 bucket size for GradSync objects built from now on (and for all_reduce_flat's default)"""
    global BUCKET_BYTES
    BUCKET_BYTES = int(n)


def world():
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_world_size()
    return 1


def rank():
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_rank()
    return 0


def active():
    """a process group exists: the exchanges C1-C3 run (also on a 1-rank group, which is how the RCCL path is smoke-tested
    on a single-GPU box: BTS_FORCE_PG=1)"""
    d = torch.distributed
    return d.is_available() and d.is_initialized()


def init_from_env(backend=None):
    """torchrun-style bring-up: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT"""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not os.environ.get('BTS_FORCE_PG')) or torch.distributed.is_initialized():
        return
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        backend = backend or 'nccl'
    else:
        backend = backend or 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.distributed.init_process_group(backend=backend, rank=int(os.environ.get('RANK', '0')), world_size=ws)


class _Done(object):
    """handle of an exchange that already completed"""

    def wait(self):
        return True


def _sum_over_ranks(t, async_op=False):
    """in-place sum of `t` over the group.  RCCL reduces device buffers directly (the production path, one GPU per rank).
    A gloo group given a device buffer -- ranks sharing ONE GPU, which RCCL refuses: tests/test_dp_gpu.py and
    `bench.py --share-gpu` -- is served through a host bounce buffer after the producing stream has drained."""
    d = torch.distributed
    if t.is_cuda and d.get_backend() == 'gloo':
        torch.cuda.current_stream(t.device).synchronize()
        h = t.detach().to('cpu')
        d.all_reduce(h, op=d.ReduceOp.SUM)
        t.copy_(h)
        return _Done()
    h = d.all_reduce(t, op=d.ReduceOp.SUM, async_op=async_op)
    return h if async_op else _Done()


def all_reduce_sum(t):
    """in-place sum over ranks (C3); no-op on one rank"""
    if active():
        _sum_over_ranks(t)
    return t


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (a collective when a process group exists; [obj] otherwise)"""
    if not active():
        return [obj]
    out = [None] * world()
    torch.distributed.all_gather_object(out, obj)
    return out


def bucket_ranges(n, elem_bytes=4, bucket_bytes=None):
    bucket_bytes = BUCKET_BYTES if bucket_bytes is None else bucket_bytes
    per = max(1, bucket_bytes // elem_bytes)
    return [(o, min(per, n - o)) for o in range(0, n, per)]


def all_reduce_flat(flat, bucket_bytes=None):
    """bucketed in-place sum of a flat buffer (C1)"""
    if not active():
        return flat
    handles = []
    for off, ln in bucket_ranges(flat.numel(), flat.element_size(), bucket_bytes):
        handles.append(_sum_over_ranks(flat[off:off + ln], async_op=True))
    for h in handles:
        h.wait()
    return flat


def l2_grad_scale():
    """factor applied to the (rank-identical) regulariser gradient before the summing all-reduce"""
    return 1.0 / world()


def all_reduce_gradients(model):
    """C1 on the model's flat gradient buffer. Returns the scale the optimiser must apply to the summed gradient."""
    if active():
        all_reduce_flat(model.flat_grads)
    return 1.0


def broadcast_parameters(model, src=0):
    """C2"""
    if active():
        fp = model.flat_params
        if fp.is_cuda and torch.distributed.get_backend() == 'gloo':    # ranks sharing one GPU (see _sum_over_ranks)
            h = fp.detach().to('cpu')
            torch.distributed.broadcast(h, src=src)
            fp.copy_(h)
        else:
            torch.distributed.broadcast(fp, src=src)
        from .tape import bump_weights_epoch
        bump_weights_epoch()
        decorrelate_rng(model)


def decorrelate_rng(model):
    """Weights are replicated, random draws are not: move each rank's dropout / reparameterisation counters
    (encoder.py:39, vae.py:12 draw fresh numbers per example) to its own stream, once per model."""
    if getattr(model, '_rng_rank', None) == rank():
        return
    for name in ('encoder', 'vae'):
        lay = getattr(model, name, None)
        if lay is not None and hasattr(lay, '_seed'):
            lay._seed = (int(lay._seed) + rank() * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
    model._rng_rank = rank()


class GradSync(object):
    """C1 overlapped with the backward pass: the flat gradient buffer is cut into buckets (BUCKET_BYTES, sized for xGMI's
    per-link-bound rings); the tape reports which parameters each node wrote, and as soon as every parameter of a bucket is
    final the bucket gets its share of the (rank-identical, pre-divided) L2 term and is all-reduced asynchronously on
    RCCL's stream while the remaining backward kernels keep the compute stream busy.  Parameters never written (unused in
    this step) are zero-filled at the end, exactly like the non-overlapped path.  Results are bit-identical to
    all_reduce_gradients() after a plain backward (same kernels on the same values, only earlier)."""

    def __init__(self, model, bucket_bytes=None):
        self.default_sized = bucket_bytes is None      # (grad_sync() rebuilds those when the default changes; explicit sizes are kept)
        bucket_bytes = BUCKET_BYTES if bucket_bytes is None else bucket_bytes
        self.bucket_bytes = bucket_bytes
        self.timing = False          # bench.py: events around finish()'s waits -> how long the compute stream stood still for the exchange
        self.exposed_ms = []
        self.model = model
        ps = model.trainable_variables
        base = model.flat_grads.data_ptr()
        self.spans = {}
        for p_ in ps:
            off = (p_._gview.data_ptr() - base) // 4
            self.spans[id(p_)] = (off, p_._gview.numel())
        self.buckets = []   # (start, length, [param ids])
        n = model.flat_grads.numel()
        for off, ln in self._cuts(model, n, bucket_bytes):
            members = [id(p_) for p_ in ps if self.spans[id(p_)][0] < off + ln and self.spans[id(p_)][0] + self.spans[id(p_)][1] > off]
            self.buckets.append((off, ln, members))
        self.by_param = {}
        for bi, (_, _, members) in enumerate(self.buckets):
            for pid in members:
                self.by_param.setdefault(pid, []).append(bi)
        self.params = {id(p_): p_ for p_ in ps}
        self.launch_log = []     # per step: (bucket, tape nodes replayed when it was launched, bytes) -- tests read it

    @staticmethod
    def _cuts(model, n, bucket_bytes):
        """Bucket boundaries.  The flat buffers are laid out in backward-completion order, one span per layer call
        (model._group_spans): a bucket is a run of whole layer groups of at least `bucket_bytes` (few large messages: xGMI rings are
        per-link bound), closed early so that the groups the backward pass finishes LAST -- the shallow encoder levels, a few MB --
        form a small bucket of their own: everything before it is in flight while the 128^3 levels are still being back-propagated."""
        spans = getattr(model, '_group_spans', None)
        if not spans or bucket_bytes < (1 << 20):
            return bucket_ranges(n, 4, bucket_bytes)      # (tests ask for many tiny buckets: plain equal cuts)
        per = max(1, bucket_bytes // 4)
        tail_elems = (4 << 20) // 4
        # index of the first group of the small tail: trailing groups that together stay under 4 MB
        t, acc = len(spans), 0
        while t > 0 and acc + (spans[t - 1][1] - spans[t - 1][0]) <= tail_elems:
            acc += spans[t - 1][1] - spans[t - 1][0]
            t -= 1
        cuts, start = [], 0
        for gi, (a, b) in enumerate(spans):
            if gi == t and a > start:
                cuts.append((start, a - start))
                start = a
            if b - start >= per and gi + 1 < len(spans) and gi + 1 != t:
                cuts.append((start, b - start))
                start = b
        if n > start:
            cuts.append((start, n - start))
        return cuts

    def begin(self, tape, l2_grad=None, prefilled=False):
        """tape: anything with .gen, .nodes and .nodes_replayed (tape.GradientTape; the explicit 16-bit step of lowp_train passes its own
        progress record).  l2_grad: the regulariser's upstream gradient (1-element tensor) when the caller, not the tape's L2 node, owns
        it.  prefilled: the flat gradient buffer was zeroed before the backward, parameters no node wrote need no fill"""
        self.tape = tape
        self.l2_grad = l2_grad
        self.prefilled = prefilled
        self.launch_log = []
        self.nodes_total = len(tape.nodes)
        self.done = set()
        self.pending = [len(m) for _, _, m in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []

    def params_written(self, params):
        seen = set()
        for p_ in params:
            pid = id(p_)
            if pid in seen or pid not in self.by_param:
                continue
            seen.add(pid)
            if pid in self.done:
                # a second tape node wrote a gradient whose bucket may already be in flight: the overlap contract (one
                # writer node per parameter, true for every layer of this model) is broken -- refuse rather than corrupt
                raise RuntimeError('parameter %s received gradient contributions from two tape nodes; run with '
                                   'BTS_DP_NO_OVERLAP=1' % p_.name)
            self.done.add(pid)
            for bi in self.by_param[pid]:
                self.pending[bi] -= 1
                if self.pending[bi] == 0:
                    self._launch(bi)

    def _launch(self, bi):
        from . import ops
        # the bucket's weight gradients were enqueued on the weight-gradient stream: order this bucket behind what that stream
        # holds NOW (an event), not behind the stream itself -- joining it would also make the main stream's data-gradient
        # chain wait for every weight gradient queued so far, which is the overlap the side stream exists for
        ops.wait_side_stream_event('wgrad')
        off, ln, _ = self.buckets[bi]
        m = self.model
        if self.l2_grad is not None:
            g = self.l2_grad
        else:
            fresh = getattr(m, '_l2_val', None) is not None and getattr(m, '_l2_val_gen', None) == self.tape.gen
            g = m._l2_val.grad if fresh else None   # (no regulariser term was added to this step's loss otherwise)
        if g is not None and m._l2_ranges:
            k = l2_grad_scale()
            rg = []
            for o, l, c in m._l2_ranges:        # regulariser ranges clipped to this bucket
                a, b = max(o, off), min(o + l, off + ln)
                if a < b:
                    rg.append((a, b - a, c * k))
            if rg:
                ops.l2_reg_bwd(m.flat_params, m.flat_grads, rg, g)
        self.launched[bi] = True
        self.launch_log.append((bi, getattr(self.tape, 'nodes_replayed', 0), ln * 4))
        if active():
            self.handles.append(_sum_over_ranks(m.flat_grads[off:off + ln], async_op=True))

    def finish(self):
        from . import ops
        gen = self.tape.gen
        for pid, p_ in self.params.items():
            if pid not in self.done:
                if not self.prefilled and p_._gen != gen:   # never written this step: its gradient is exactly the regulariser's
                    ops.fill(p_._gview, 0.0)
                    p_._gen = gen
                self.done.discard(pid)
                self.params_written([p_])
        for bi in range(len(self.buckets)):    # (pad-only tail buckets have no members)
            if not self.launched[bi]:
                self._launch(bi)
        if self.timing and active() and self.model.flat_grads.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for h in self.handles:
                h.wait()
            b.record()
            self._timing_events = getattr(self, '_timing_events', []) + [(a, b)]
        else:
            for h in self.handles:
                h.wait()
        self.handles = []

    def exposed_wait_ms(self):
        """per step since timing was switched on: milliseconds the compute stream waited in finish() for buckets still in flight
        (call after a device synchronize)"""
        ev = getattr(self, '_timing_events', [])
        self._timing_events = []
        return [a.elapsed_time(b) for a, b in ev]


def grad_sync(model):
    """a GradSync for `model` when a process group exists (and BTS_DP_NO_OVERLAP is unset), else None"""
    if not active() or os.environ.get('BTS_DP_NO_OVERLAP') or model.flat_grads is None:
        return None
    gs = getattr(model, '_grad_sync', None)
    if gs is None or gs.model.flat_grads.data_ptr() != model.flat_grads.data_ptr() or (gs.default_sized and gs.bucket_bytes != BUCKET_BYTES):
        gs = GradSync(model)
        model._grad_sync = gs
    return gs
