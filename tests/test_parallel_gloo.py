"""CPU, world_size 2 over gloo: the data-parallel recipe of SURVEY 8(e) -- shard the batch by sample, all-reduce the raw
Dice/MSE/KL sums in the forward (C3), sum-all-reduce the flat gradient in buckets (C1), pre-divide the rank-identical L2
gradient -- must reproduce the single-process global-batch loss and gradient.  Compute here is the oracle (torch-CPU
autograd); what is under test is bts_amd.parallel and the exchange arithmetic."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global_reference(cfg, P, x, y, mask, eps):
    from oracle import torch_ref as R
    leaves = {k: t.clone().requires_grad_(True) for k, t in P.items()}
    PP = R.ParamSet(); PP.update(leaves); PP.l2 = P.l2
    out = R.model(x, PP, cfg, training=True, inference=False, mask=mask, eps=eps)
    loss = R.dice_vae_loss(x, y, *out) + R.l2_regularisation(PP)
    g = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), torch.cat([t.reshape(-1) for t in g])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bts_amd  # noqa: F401
    from bts_amd import parallel
    from oracle import torch_ref as R
    torch.set_num_threads(2)
    parallel.init_from_env('gloo')
    assert parallel.world() == world and parallel.rank() == rank
    cfg = R.default_config(base_filters=4, groups=2, reduction=2, depth=2)
    crop = (8, 8, 8)
    x, y, mask, eps = R.synthetic_batch(world, crop, latent=4, seed=11, dtype=torch.float64)
    P = R.build_params(cfg, crop, seed=5)
    g = torch.Generator().manual_seed(6)
    for k in P:
        if k.endswith('gn2_g'):
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64)
    sl = slice(rank, rank + 1)                                  # one sample per rank
    leaves = {k: t.clone().requires_grad_(True) for k, t in P.items()}
    PP = R.ParamSet(); PP.update(leaves); PP.l2 = P.l2
    y_pred, y_vae, zm, zl = R.model(x[sl], PP, cfg, training=True, inference=False, mask=mask[sl], eps=eps[sl])
    # local raw sums (the engine: bts_loss_sums), then C3
    ax = (0, 1, 2, 3)
    loc = torch.cat([(y_pred * y[sl]).sum(ax), (y_pred ** 2).sum(ax), (y[sl] ** 2).sum(ax),
                     ((x[sl] - y_vae) ** 2).sum().reshape(1), (zm ** 2 + torch.exp(zl) - zl - 1).sum().reshape(1),
                     torch.tensor([float(x[sl].numel()), float(zm.numel())], dtype=torch.float64)])
    tot = parallel.all_reduce_sum(loc.detach().clone())
    glob = loc + (tot - loc.detach())                           # value = global sums, gradient = this rank's share
    C = 3
    dice = (1.0 - (2 * glob[:C] + 1) / (glob[C:2 * C] + glob[2 * C:3 * C] + 1)).mean()
    loss = dice + 0.1 * glob[3 * C] / tot[3 * C + 2] + 0.1 * glob[3 * C + 1] / tot[3 * C + 3]
    loss = loss + R.l2_regularisation(PP) * parallel.l2_grad_scale()   # rank-identical term, pre-divided
    grads = torch.autograd.grad(loss, list(leaves.values()))
    flat = torch.cat([t.reshape(-1) for t in grads]).contiguous()
    parallel.all_reduce_flat(flat, bucket_bytes=4096)           # C1, many small buckets to exercise the bucketing
    loss_val = loss.detach() + R.l2_regularisation(P) * (1 - parallel.l2_grad_scale())
    if rank == 0:
        ref_loss, ref_flat = _global_reference(cfg, P, x, y, mask, eps)
        q.put((float((loss_val - ref_loss).abs()), float((flat - ref_flat).abs().max()), float(ref_flat.abs().max())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_equals_global_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    dloss, dgrad, gscale = res
    assert dloss < 1e-12, res
    assert dgrad < 1e-10 * max(gscale, 1.0), res
