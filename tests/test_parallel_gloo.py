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


def _ckpt_worker(rank, world, port, folder, q):
    """two ranks run one 'epoch' of random draws, save (all ranks call, rank 0 writes), then FRESH objects load: every rank must get
    back its own augmentation generator and its own dropout / reparameterisation counters (all ranks used to
    resume with rank 0's)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bts_amd  # noqa: F401
    from bts_amd import data as D, parallel, train as T
    from bts_amd.model import Model
    torch.set_num_threads(1)
    parallel.init_from_env('gloo')
    m = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m.build((1, 8, 8, 8, 2))
    parallel.decorrelate_rng(m)                        # (broadcast_parameters does this after the C2 broadcast)
    m.encoder._seed += 3                               # three steps' worth of draws on every rank
    m.vae._seed += 3
    ds = D._Dataset(['a', 'b', 'c', 'd'], 1, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=3, device='cpu', rank=rank, world=world)
    torch.randperm(4, generator=ds.order_gen)
    torch.rand(5 + rank, generator=ds.gen)             # ranks are at different positions of different streams
    want = (int(m.encoder._seed), int(m.vae._seed), torch.rand(4, generator=torch.Generator().set_state(ds.gen.get_state())).tolist(),
            torch.randperm(4, generator=torch.Generator().set_state(ds.order_gen.get_state())).tolist())
    T.save_checkpoint(folder, m, None, completed=True, datasets={'train': ds}, write=(rank == 0))
    torch.distributed.barrier()
    m2 = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m2.build((1, 8, 8, 8, 2))
    T.load_checkpoint(folder, m2)
    ds2 = D._Dataset(['a', 'b', 'c', 'd'], 1, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=77, device='cpu', rank=rank, world=world)
    ds2.load_state_dict({k[len('train/'):]: v for k, v in m2._resume['data'].items() if k.startswith('train/')})
    got = (int(m2.encoder._seed), int(m2.vae._seed), torch.rand(4, generator=ds2.gen).tolist(), torch.randperm(4, generator=ds2.order_gen).tolist())
    q.put((rank, want == got, want[0]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_checkpoint_keeps_per_rank_random_state(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ckpt_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], res
    assert res[0][2] != res[1][2]                      # the ranks' counters differ, and each got its own back


def _fit_worker(rank, world, port, folder, q):
    """fit() with a metric that is NOT all-reduced (rank 1 sees a worse validation Dice) and a save_folder on rank 0 only: the
    round-3 advisor's hang.  Both ranks must take rank 0's save / stop decisions and finish"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bts_amd  # noqa: F401
    from bts_amd import parallel, train as T
    from bts_amd.model import Model
    from bts_amd.util import ScheduledOptim
    parallel.init_from_env('gloo')
    m = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m.build((1, 8, 8, 8, 2))
    opt = ScheduledOptim(1e-4)
    dice = {0: [0.5, 0.4, 0.6, 0.3, 0.2, 0.1], 1: [0.1, 0.7, 0.2, 0.8, 0.9, 0.95]}[rank]     # rank 1 disagrees at every epoch
    ep = [0]

    def estep(x, y):
        return 1.0, dice[ep[0]], dice[ep[0]]

    def tstep(x, y):
        torch.distributed.all_reduce(torch.zeros(1))       # stands for the gradient exchange: a rank that left would hang the other
        return 1.0, 0.5, 0.5
    msgs = []

    def log(s):
        msgs.append(s)
        if s.startswith('Validation.'):
            ep[0] += 1
    hist = T.fit(m, opt, None, None, [(None, None)], [(None, None)], 6, patience=2, save_folder=folder if rank == 0 else None,
                 train_step_fn=tstep, eval_step_fn=estep, log=log)
    q.put((rank, len(hist), sum(1 for s in msgs if s.startswith('Saved')), any('Stopped' in s for s in msgs)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_fit_takes_rank_identical_save_and_stop_decisions(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # rank 0's curve 0.5 0.4 0.6 0.3 0.2 0.1 (patience 2): saves at epochs 0 and 2, stops at epoch 5 -- on BOTH ranks
    assert res[0][1:] == res[1][1:], res
    assert res[0][2] == 2 and res[0][3] and res[0][1] == 6, res
    assert os.path.exists(os.path.join(str(tmp_path), 'checkpoint.safetensors')) or len(os.listdir(str(tmp_path))) > 0
