"""Rank process of tests/test_dp_gpu.py (not a test module): runs the PRODUCT train_step (bts_amd.util.train_step: HIP forward,
C3 loss-sum exchange, backward with the bucketed gradient exchange C1, TF-form Adam) on its shard of a fixed global batch
and dumps loss / summed flat gradient / post-Adam parameters.

usage: dp_worker.py RANK WORLD PORT OUT.pt [steps]
  WORLD == 0  -> single process, no process group, the WHOLE global batch (the reference every DP run must reproduce:
                 the reference itself is single-device, train.py:138)
  WORLD >= 1  -> one sample per rank; all ranks on cuda:0 over gloo (RCCL refuses two ranks on one device; with gloo the
                 exchanges go through parallel._sum_over_ranks' host bounce, everything else is the production code), or -- with
                 BTS_DP_BACKEND=nccl on a box with >= WORLD GPUs -- rank r on cuda:r over RCCL on device buffers
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KW = dict(base_filters=8, groups=2, reduction=2, depth=3)
CROP = (16, 16, 16)
GLOBAL_BATCH = 2
# BTS_DP_TRAINER=bfloat16 | float16: the 16-bit storage step (bts_amd.lowp_train, BASELINE configs[3] = configs[2] per GPU) instead of the fp32 one
TRAINER = os.environ.get('BTS_DP_TRAINER')
if TRAINER:
    KW = dict(base_filters=16, groups=8, reduction=2, depth=3)
    CROP = (32, 32, 32)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    import torch
    import bts_amd  # noqa: F401
    from bts_amd import parallel
    from bts_amd.data import synthetic_batch
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    # BTS_DP_BACKEND=nccl: one DEVICE per rank and RCCL on device buffers (the production path; needs >= WORLD GPUs) instead of ranks
    # sharing cuda:0 over gloo's host bounce
    backend = os.environ.get('BTS_DP_BACKEND', 'gloo')
    didx = rank if (backend == 'nccl' and world >= 1) else 0
    torch.cuda.set_device(didx)
    dev = torch.device('cuda', didx)
    if world >= 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(didx), BTS_FORCE_PG='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
        parallel.init_from_env(backend)
        assert parallel.world() == world and parallel.rank() == rank
        assert torch.distributed.get_backend() == backend
        # bring-up ends HERE: one collective on a device buffer forces the backend to connect the ranks (RCCL connects lazily).  The marker
        # tells the test which side of that line a failure fell on: before it the box could not connect its devices (skip), after it any
        # failure is the product's (fail)
        probe = torch.ones(8, device=dev)
        torch.distributed.all_reduce(probe)
        torch.cuda.synchronize()
        assert float(probe[0]) == world
        open(out + '.pg_ready', 'w').write('ok')
    latent = KW['base_filters'] * 2 ** (KW['depth'] - 2)
    x, y, mask, eps = synthetic_batch(GLOBAL_BATCH, CROP, latent=latent, seed=77)
    sl = slice(0, GLOBAL_BATCH) if world == 0 else slice(rank, GLOBAL_BATCH, world)
    x, y, mask, eps = x[sl].contiguous(), y[sl].contiguous(), mask[sl].contiguous(), eps[sl].contiguous()
    model = Model(**KW)
    model.build((x.shape[0],) + CROP + (2,))
    g = torch.Generator().manual_seed(11 + 1000 * rank)      # ranks start DIFFERENT on purpose: C2 must make them equal
    for p in model.trainable_variables:
        p.t.copy_((torch.randn(p.t.shape, generator=g) * (0.05 if p.t.dim() > 1 else 0.3)).to(dev))
    bump_weights_epoch()
    parallel.broadcast_parameters(model)                     # C2 (rank 0's weights; no-op without a group)
    start = model.flat_params.detach().cpu().clone()
    opt = ScheduledOptim(1e-3)
    opt(epoch=0)
    lf, df = DiceVAELoss(), DiceCoefficient()
    step = train_step
    if TRAINER:
        from bts_amd.lowp_train import LowPrecisionTrainer
        tr = LowPrecisionTrainer(model, TRAINER)
        step = lambda m_, o_, lf_, df_, x_, y_: tr.step(o_, df_, x_, y_)   # noqa: E731
    losses, macros, grads1 = [], [], None
    for s in range(steps):
        model.encoder.set_dropout_mask(mask)                 # one-shot injections: the same draws every step, so the
        model.vae.set_eps(eps)                               # single-process run sees exactly the ranks' samples
        loss, macro, micro = step(model, opt, lf, df, x.to(dev), y.to(dev))
        torch.cuda.synchronize()
        losses.append(float(loss))
        macros.append(float(macro))
        if s == 0:
            grads1 = model.flat_grads.detach().cpu().clone()   # after C1: the global-batch gradient on every rank
    torch.save({'start': start, 'loss': losses, 'macro': macros, 'grads': grads1,
                'params': model.flat_params.detach().cpu().clone(),
                'overlap': parallel.grad_sync(model) is not None}, out)
    if world >= 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
