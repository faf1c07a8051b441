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

BUCKET_BYTES = 64 << 20


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


def all_reduce_sum(t):
    """in-place sum over ranks (C3); no-op on one rank"""
    if active():
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return t


def bucket_ranges(n, elem_bytes=4, bucket_bytes=BUCKET_BYTES):
    per = max(1, bucket_bytes // elem_bytes)
    return [(o, min(per, n - o)) for o in range(0, n, per)]


def all_reduce_flat(flat, bucket_bytes=BUCKET_BYTES):
    """bucketed in-place sum of a flat buffer (C1)"""
    if not active():
        return flat
    handles = []
    for off, ln in bucket_ranges(flat.numel(), flat.element_size(), bucket_bytes):
        handles.append(torch.distributed.all_reduce(flat[off:off + ln], op=torch.distributed.ReduceOp.SUM, async_op=True))
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
        torch.distributed.broadcast(model.flat_params, src=src)
        from .tape import bump_weights_epoch
        bump_weights_epoch()
