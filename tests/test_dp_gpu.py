"""-m gpu: the PRODUCT data-parallel path with world_size 2.

The reference is single-device (train.py:138), so the correctness oracle of the sharded run is the single-process run on
the same global batch: two rank processes (fresh interpreters, both on cuda:0, gloo -- RCCL refuses two ranks on one
device) each run bts_amd.util.train_step on ONE sample of a 2-sample batch; C2 (parameter broadcast), C3 (13 loss sums +
Dice table) and C1 (bucketed flat-gradient exchange issued from inside the backward pass, L2 term pre-divided per bucket)
must reproduce the loss, the flat gradient and the post-Adam parameters of one process stepping on both samples.

Stated tolerances: loss <= 1e-6 relative; gradient <= 2e-6 of the gradient's max-abs (the per-sample weight-gradient
partials are summed in a different association: fp64 fixed-order inside one launch vs fp32 add of two ranks' results);
parameters after 2 Adam steps <= 2e-6 absolute at lr 1e-3 (Adam's first steps ~ lr*g/(|g|+3e-6) amplify a gradient
difference d by lr*d/3e-6 where g ~ 0).  Overlapped vs post-backward exchange (BTS_DP_NO_OVERLAP=1): bitwise equal.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dp_worker.py')


def _run(world, out_dir, tag, extra_env=None, steps=2):
    """start the rank processes (or the single global-batch process for world 0), return their dumps"""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.pop('BTS_DP_NO_OVERLAP', None)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    env.update(extra_env or {})
    outs, procs = [], []
    for r in range(max(world, 1)):
        o = os.path.join(out_dir, '%s_r%d.pt' % (tag, r))
        outs.append(o)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), o, str(steps)], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, 'rank process failed:\n' + log[-3000:]
    return [torch.load(o) for o in outs]


def test_two_ranks_equal_single_process_global_batch(tmp_path):
    d = str(tmp_path)
    single = _run(0, d, 'single')[0]
    dp = _run(2, d, 'dp')
    assert dp[0]['overlap'] and dp[1]['overlap'], 'the overlapped GradSync path was not taken'
    # C2: both ranks started from rank 0's weights, which are the single run's
    assert torch.equal(dp[0]['start'], dp[1]['start']) and torch.equal(dp[0]['start'], single['start'])
    # every rank holds the same global loss, gradient and parameters
    assert dp[0]['loss'] == dp[1]['loss'] and dp[0]['macro'] == dp[1]['macro']
    assert torch.equal(dp[0]['grads'], dp[1]['grads']) and torch.equal(dp[0]['params'], dp[1]['params'])
    for a, b in zip(dp[0]['loss'], single['loss']):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (dp[0]['loss'], single['loss'])
    for a, b in zip(dp[0]['macro'], single['macro']):
        assert abs(a - b) <= 1e-6, (dp[0]['macro'], single['macro'])
    gs = float(single['grads'].abs().max())
    ge = float((dp[0]['grads'] - single['grads']).abs().max())
    pe = float((dp[0]['params'] - single['params']).abs().max())
    moved = float((single['params'] - single['start']).abs().max())
    print('grad max-abs %.3e, DP-vs-single |d| %.3e (%.2e rel); params moved %.3e, |d| %.3e' % (gs, ge, ge / gs, moved, pe))
    assert gs > 0 and moved > 1e-4
    assert ge <= 2e-6 * gs
    assert pe <= 2e-6


def test_two_ranks_of_the_16bit_step_equal_the_single_process_global_batch(tmp_path):
    """BASELINE configs[3] runs configs[2]'s step on every GPU: the bf16-storage step (bts_amd.lowp_train) sharded over two ranks against
    one process stepping on both samples.  Unlike the fp32 case the per-sample values are not bit-identical: a batch-1 launch may pick another
    tiling / split-K than the batch-2 launch (different fp32 summation order), and where a sum lands next to a bf16 rounding boundary the
    STORED activation differs by one unit in the last place -- measured: gradient 1.8e-4 of its max-abs apart, loss 1e-6; stated bounds:
    1e-3 and 1e-5; parameters after two Adam steps at lr 1e-3 within 2 lr (a sign flip of a near-zero gradient)."""
    d = str(tmp_path)
    env = {'BTS_DP_TRAINER': 'bfloat16'}
    single = _run(0, d, 'single16', env)[0]
    dp = _run(2, d, 'dp16', env)
    assert torch.equal(dp[0]['start'], dp[1]['start']) and torch.equal(dp[0]['start'], single['start'])
    assert dp[0]['loss'] == dp[1]['loss'] and dp[0]['macro'] == dp[1]['macro']
    assert torch.equal(dp[0]['grads'], dp[1]['grads']) and torch.equal(dp[0]['params'], dp[1]['params'])
    for a, b in zip(dp[0]['loss'], single['loss']):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (dp[0]['loss'], single['loss'])
    gs = float(single['grads'].abs().max())
    ge = float((dp[0]['grads'] - single['grads']).abs().max())
    pe = float((dp[0]['params'] - single['params']).abs().max())
    moved = float((single['params'] - single['start']).abs().max())
    print('16-bit step: grad max-abs %.3e, DP-vs-single |d| %.3e (%.2e rel); params moved %.3e, |d| %.3e' % (gs, ge, ge / gs, moved, pe))
    assert gs > 0 and moved > 1e-4
    assert ge <= 1e-3 * gs
    assert pe <= 2.1e-3


def test_overlapped_exchange_is_bitwise_the_plain_one(tmp_path):
    d = str(tmp_path)
    a = _run(2, d, 'ovl')
    b = _run(2, d, 'plain', {'BTS_DP_NO_OVERLAP': '1'})
    assert a[0]['overlap'] and not b[0]['overlap']
    assert a[0]['loss'] == b[0]['loss']
    assert torch.equal(a[0]['grads'], b[0]['grads'])
    assert torch.equal(a[0]['params'], b[0]['params']) and torch.equal(a[1]['params'], b[1]['params'])


def test_overlapped_exchange_of_the_16bit_step_is_bitwise_the_plain_one(tmp_path):
    """lowp_train.LowPrecisionTrainer.step launches finished gradient buckets from inside its explicit backward (GradSync, as the fp32
    tape does); BTS_DP_NO_OVERLAP=1 restores join -> regulariser -> one blocking exchange.  Same kernels on the same values: bitwise"""
    d = str(tmp_path)
    env = {'BTS_DP_TRAINER': 'bfloat16'}
    a = _run(2, d, 'ovl16', env)
    b = _run(2, d, 'plain16', dict(env, BTS_DP_NO_OVERLAP='1'))
    assert a[0]['overlap'] and not b[0]['overlap']
    assert a[0]['loss'] == b[0]['loss']
    assert torch.equal(a[0]['grads'], b[0]['grads'])
    assert torch.equal(a[0]['params'], b[0]['params']) and torch.equal(a[1]['params'], b[1]['params'])


def test_bench_launcher_starts_the_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must START two ranks (here sharing the box's single GPU) and say so"""
    import json
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--share-gpu', '--crop', '32',
                        '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['ranks_seen'] == 2 and out['config']['global_batch'] == 2 and out['config']['parallelism'] == 'dp2'
    assert out['value'] > 0
    # the first scaling run must carry BASELINE configs[3] itself (bf16 storage, batch 8 per GPU, buckets exchanged from inside the
    # backward) next to the fp32 weak-scaling headline, and explain itself
    c3 = out['also']['configs[3]']
    assert c3['value'] > 0 and c3['dtype'] == 'bf16' and c3['config']['global_batch'] == 16 and c3['config']['parallelism'] == 'dp2', c3
    assert 'configs[3]' in c3['config']['workload'] and c3['exchange']['buckets'], c3
    assert c3['ranks_seen'] == 2
    assert out['summary']['configs[3]'][0] == c3['value']


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ)
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stdout + r.stderr)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank: fewer than 2 GPUs visible')
@pytest.mark.parametrize('trainer', [None, 'bfloat16'])
def test_two_ranks_over_rccl_on_device_buffers(tmp_path, trainer):
    """The production exchange path (torch.distributed 'nccl' = RCCL, device buffers, one GPU per rank, async bucket all-reduces issued from
    inside the backward) -- every other world-2 test here goes through gloo's host bounce because the GPU boxes of the pool have ONE device.
    Runs by itself wherever two devices are visible; the single-process global-batch run is the reference as above."""
    d = str(tmp_path)
    extra = {'BTS_DP_TRAINER': trainer} if trainer else {}
    single = _run(0, d, 'single', extra)[0]
    try:
        dp = _run(2, d, 'rccl', dict(extra, BTS_DP_BACKEND='nccl'))
    except AssertionError as e:
        # Only a failure of the BRING-UP is the box's business (IPC mode, topology): some rank never got through its first collective
        # (tests/dp_worker.py writes `<out>.pg_ready` after it) AND the log names a system-level RCCL error.  ncclInvalidUsage /
        # ncclInvalidArgument (mismatched buckets, a collective issued on one rank only), a hang that ends in an abort, or anything at
        # all after the marker is a product failure and fails the test.
        msg = str(e)
        ready = [os.path.exists(os.path.join(d, 'rccl_r%d.pt.pg_ready' % r)) for r in range(2)]
        if not all(ready) and any(k in msg for k in ('ncclSystemError', 'ncclUnhandledCudaError', 'hipIpc')):
            pytest.skip('RCCL could not connect the two devices on this box: ' + msg[-300:])
        raise
    assert dp[0]['overlap'] and dp[1]['overlap']
    assert torch.equal(dp[0]['start'], dp[1]['start']) and torch.equal(dp[0]['start'], single['start'])
    assert dp[0]['loss'] == dp[1]['loss'] and torch.equal(dp[0]['grads'], dp[1]['grads']) and torch.equal(dp[0]['params'], dp[1]['params'])
    for a, b in zip(dp[0]['loss'], single['loss']):
        assert abs(a - b) <= (1e-6 if trainer is None else 1e-5) * max(1.0, abs(b))
    gs = float(single['grads'].abs().max())
    ge = float((dp[0]['grads'] - single['grads']).abs().max())
    print('RCCL world 2 (%s): gradient |d| %.3e of max-abs %.3e' % (trainer or 'fp32', ge, gs))
    if trainer is None:
        assert ge <= 2e-6 * gs and float((dp[0]['params'] - single['params']).abs().max()) <= 2e-6
    else:      # 16-bit storage: the bounds (and the reason) of test_two_ranks_of_the_16bit_step_equal_the_single_process_global_batch
        assert ge <= 1e-3 * gs and float((dp[0]['params'] - single['params']).abs().max()) <= 2.1e-3
    # and the exchange is the same arithmetic as the post-backward one
    plain = _run(2, d, 'rccl_plain', dict(extra, BTS_DP_BACKEND='nccl', BTS_DP_NO_OVERLAP='1'))
    assert torch.equal(plain[0]['grads'], dp[0]['grads']) and torch.equal(plain[0]['params'], dp[0]['params'])
