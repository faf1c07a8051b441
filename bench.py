#!/usr/bin/env python3
"""bench.py -- training volumes/s (2ch x 128^3) of the MI355X-native 3D U-Net+VAE engine.

One "step" = one pass of the hot path over one resident batch: forward (training=True), Dice+0.1*MSE+0.1*KL+L2 loss,
Dice metric, full backward, TF-form Adam (train.py:140-152 of the reference), CLI-default model (base_filters=32,
reduction=8, depth=4, groups=8; args.py:121-143), fp32.  N=1: BASELINE.json configs[1] (batch 1 per GPU).
N>1 (torchrun): one process per GPU, one sample per rank (weak scaling), RCCL all-reduce of the 13 loss sums and of the
flat 168.7 MB gradient buffer.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     dominant kernel (by summed time in the timed region), algorithmic FLOPs / measured HIP-event time
  "cpu_baseline": the oracle's identical step on the host cores, bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)


def pmc_traffic_bytes(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC passes (scripts/pmc_traffic.sh ->
    profiles/rNN_pmc_traffic.json): (2*FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE reports half of the bytes of
    wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None, None
    tab = json.load(open(files[-1]))
    norm = lambda n: n.replace('void ', '').replace(' ', '')
    key = norm(symbol)
    best = None
    for k, v in tab.items():
        nk = norm(k)
        if (nk == key or (('<' not in key) and nk.split('<')[0] == key)) and 'FETCH_SIZE_KiB_mean' in v \
                and 'WRITE_SIZE_KiB_mean' in v:
            if best is None or v['launches'] > best['launches']:
                best = v
    if best is None:
        return None, None
    return (2.0 * best['FETCH_SIZE_KiB_mean'] + best['WRITE_SIZE_KiB_mean']) * 1024.0, os.path.basename(files[-1])


CPU_BASELINE_THREADS = 16   # torch-CPU conv3d stops scaling (and thrashes badly) far below the GPU box's 256 hardware threads


def _cpu_baseline_worker():
    """child process: oracle (torch-CPU restatement) train step on a bounded sample; prints one JSON line"""
    import torch
    from oracle import torch_ref as R
    threads = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    cfg = R.default_config(base_filters=32, reduction=8)

    def one(crop):
        x, y, mask, eps = R.synthetic_batch(1, crop, latent=128, seed=1234)
        P = R.build_params(cfg, crop, seed=0)
        for k in P:
            P[k] = P[k].float()
        t0 = time.time()
        R.train_step(P, cfg, x, y, mask, eps, {}, 1e-4, 1)
        return time.time() - t0

    one((16, 16, 16))                      # warm-up: thread pool, oneDNN primitive caches
    t = one((64, 64, 64))
    print(json.dumps({'value': (64 ** 3 / float(128 ** 3)) / t, 'unit': 'volumes/s', 'cores': threads, 'kind': 'port',
                      'sample': '1 fwd+bwd+Adam step, fp32, CLI-default model, one 2ch x 64^3 crop (= 1/8 of a 128^3 volume) '
                                'after a 16^3 warm-up step; value scaled to 128^3 volumes', 'seconds': round(t, 3)}))


def cpu_baseline(timeout_s=150):
    """oracle train step timed on the host cores in a child process; kind 'port' (the reference needs TensorFlow, which
    cannot be installed here, so the restatement is what can be timed)"""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker'], capture_output=True,
                             text=True, timeout=timeout_s, cwd=ROOT)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {'value': None, 'unit': 'volumes/s', 'cores': CPU_BASELINE_THREADS, 'kind': 'port',
                'sample': 'worker failed: ' + out.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'volumes/s', 'cores': CPU_BASELINE_THREADS, 'kind': 'port',
                'sample': 'bounded sample (64^3 crop step) did not finish within %d s' % timeout_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--crop', type=int, default=128)
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker()
        return

    import torch
    import bts_amd  # noqa: F401
    from bts_amd import ops, parallel
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    from oracle import torch_ref as R   # synthetic input generator only

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU execution path exists for the product)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get('BTS_FORCE_PG'):   # BTS_FORCE_PG=1: 1-rank RCCL group, smoke-tests the N>1 code path
        parallel.init_from_env('nccl')
    dev = torch.device('cuda', local)

    crop = (args.crop,) * 3
    nb = args.batch
    kw = dict(base_filters=32, reduction=8, depth=4, groups=8)
    model = Model(**kw)
    model.build((nb,) + crop + (2,))
    parallel.broadcast_parameters(model)
    x, y, _, _ = R.synthetic_batch(nb, crop, latent=128, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    loss_fn, dice_fn = DiceVAELoss(), DiceCoefficient()

    for _ in range(args.warmup):
        loss, macro, micro = train_step(model, opt, loss_fn, dice_fn, x, y)
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    do_prof = (not args.no_profile)
    if do_prof:
        ops.profile_enable(True)   # HIP events on the launch stream around every igemm_kernel / wgrad_kernel launch
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, macro, micro = train_step(model, opt, loss_fn, dice_fn, x, y)
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if parallel.active():
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    prof = None
    if do_prof:
        ops.profile_enable(False)
        prof = ops.profile_records()

    if rank != 0:
        if parallel.active():
            torch.distributed.destroy_process_group()
        return
    volumes = world * nb * args.steps
    out = {
        'metric': 'training volumes/sec (2ch x %d^3)' % args.crop, 'value': volumes / dt, 'unit': 'volumes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: 2ch x %d^3, batch %d per GPU, fp32, full fwd+bwd with '
                               'Dice+KL+L2 VAE loss + Dice metric + TF-form Adam; CLI-default model '
                               '(base_filters=32, depth=4, groups=8, reduction=8; 42,174,773 params)' % (args.crop, nb),
                   'parallelism': 'dp%d' % world, 'global_batch': world * nb},
        'loss': float(loss), 'macro_dice': float(macro),
    }
    if do_prof and prof:
        agg = {}
        for sym, flops, ms in prof:
            a = agg.setdefault(sym, [0.0, 0.0, 0])
            a[0] += ms * 1e-3
            a[1] += flops
            a[2] += 1
        dom = max(agg.items(), key=lambda kv: kv[1][0])
        sym, (tsec, fl, nl) = dom
        ach = fl / tsec / 1e12
        traffic, tsrc = pmc_traffic_bytes(sym)
        out['roofline'] = {
            'kernel': sym, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': tsrc,
            'launches_per_step': nl / args.steps, 'avg_launch_ms': 1e3 * tsec / nl,
            'algorithmic_gflop_per_launch': fl / nl / 1e9,
            'time_share_of_step': tsec / dt,
        }
        if sym in ('wino_kernel', 'wgw_kernel'):
            # the Winograd form issues 12 matrix instructions where the direct form needs 27: `achieved` follows the
            # contract (ALGORITHMIC direct-conv FLOPs / time) and can exceed the pipe's peak; the rate the matrix pipe
            # really executes, and its fraction of the peak, are reported next to it
            out['roofline']['executed'] = ach * 12.0 / 27.0
            out['roofline']['executed_frac'] = ach * 12.0 / 27.0 / PEAK_F32_MFMA_TFLOPS
            out['roofline']['note'] = ('Winograd F(2x2,3x3) x direct: 12/27 of the algorithmic MACs are executed; '
                                       'frac > 1 means faster than any direct-form kernel could be')
        out['kernel_breakdown'] = {k: {'ms_per_step': 1e3 * v[0] / args.steps, 'tflops': v[1] / v[0] / 1e12,
                                       'launches_per_step': v[2] / args.steps} for k, v in sorted(agg.items())}
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if parallel.active():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
