"""-m gpu: the training-loop shell on the real engine: `fit` runs epochs of train_step / eval_step, logs the
reference's CSV, checkpoints on validation improvement -- and a run resumed from the container continues BIT-EXACTLY
(weights, Adam moments, step count, LR schedule position and the dropout / reparameterisation RNG counters all persist)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

KW = dict(base_filters=8, groups=2, reduction=2, depth=3)
CROP = (16, 16, 16)


def _setup(seed=0):
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim
    torch.manual_seed(seed)
    m = Model(**KW)
    m.build((1,) + CROP + (2,))
    g = torch.Generator().manual_seed(11)
    for p in m.trainable_variables:   # same deterministic start for every model of this test module
        if p.t.dim() > 1:
            p.t.copy_((torch.randn(p.t.shape, generator=g) * 0.05).to(p.t.device))
    from bts_amd.tape import bump_weights_epoch
    bump_weights_epoch()
    return m, ScheduledOptim(1e-3, n_epochs=4), DiceVAELoss(), DiceCoefficient()


def _data(n, seed):
    dev = torch.device('cuda', 0)
    latent = KW['base_filters'] * 2 ** (KW['depth'] - 2)
    out = []
    for i in range(n):
        x, y, _, _ = R.synthetic_batch(1, CROP, latent=latent, seed=seed + i)
        out.append((x.to(dev), y.to(dev)))
    return out


def test_fit_logs_and_checkpoints(tmp_path):
    from bts_amd import train as T
    m, opt, lf, df = _setup()
    hist = T.fit(m, opt, lf, df, _data(3, 100), _data(2, 200), n_epochs=2, patience=5, save_folder=str(tmp_path),
                 log=lambda s: None)
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(float(h['train_loss']))) for h in hist)
    assert float(hist[1]['train_loss']) < float(hist[0]['train_loss'])      # it learns something on 3 fixed volumes
    lines = open(os.path.join(str(tmp_path), 'train.log')).read().strip().split('\n')
    assert lines[0] == T.LOG_HEADER and len(lines) == 3 and lines[1].startswith('0,0.001,')
    assert os.path.exists(os.path.join(str(tmp_path), T.CHECKPOINT_NAME))
    assert opt.iterations == 6


def test_resume_is_bit_exact(tmp_path):
    from bts_amd import train as T
    train, val = _data(3, 100), _data(1, 200)
    # run A: epochs 0 and 1 back to back, snapshot after epoch 0
    mA, oA, lf, df = _setup()
    T.fit(mA, oA, lf, df, train, val, n_epochs=1, patience=5, log=lambda s: None)
    mA.epoch.assign(1)
    T.save_checkpoint(str(tmp_path), mA, oA)
    T.fit(mA, oA, lf, df, train, val, n_epochs=2, patience=5, log=lambda s: None)
    # run B: fresh process state, restored from the snapshot, then epoch 1
    mB, oB, lf2, df2 = _setup(seed=123)
    for p in mB.trainable_variables:
        p.t.zero_()
    meta = T.load_checkpoint(str(tmp_path), mB, oB)
    assert meta['epoch'] == 1 and oB.iterations == 3
    histB = T.fit(mB, oB, lf2, df2, train, val, n_epochs=2, patience=5, log=lambda s: None)
    assert [h['epoch'] for h in histB] == [1]
    torch.cuda.synchronize()
    assert torch.equal(mA.flat_params, mB.flat_params), 'resumed run diverged: max |d| %.3e' % float(
        (mA.flat_params - mB.flat_params).abs().max())
    sA, sB = oA._state[id(mA.flat_params)], oB._state[id(mB.flat_params)]
    assert torch.equal(sA[0], sB[0]) and torch.equal(sA[1], sB[1]) and oA.iterations == oB.iterations == 6


KW16 = dict(base_filters=16, groups=4, reduction=4, depth=3)


def _setup16(seed=0):
    import bts_amd  # noqa: F401
    from bts_amd.layers import _base
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim
    _base.set_seed(7)
    m = Model(**KW16)
    m.build((1,) + CROP + (2,))
    return m, ScheduledOptim(1e-3, n_epochs=4), DiceVAELoss(), DiceCoefficient()


def _data16(n, seed):
    dev = torch.device('cuda', 0)
    latent = KW16['base_filters'] * 2 ** (KW16['depth'] - 2)
    return [tuple(t.to(dev) for t in R.synthetic_batch(1, CROP, latent=latent, seed=seed + i)[:2]) for i in range(n)]


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
def test_fit_with_the_16bit_step_learns_and_resumes_bit_exactly(tmp_path, dtype):
    """fit(compute_dtype=...) runs the training iterations through lowp_train.LowPrecisionTrainer (validation stays fp32); the
    container holds everything that step needs too (fp32 master weights, Adam moments, RNG counters), so a resumed run is bit-exact."""
    from bts_amd import train as T
    train, val = _data16(3, 100), _data16(1, 200)
    mA, oA, lf, df = _setup16()
    hA = T.fit(mA, oA, lf, df, train, val, n_epochs=1, patience=5, log=lambda s: None, compute_dtype=dtype)
    mA.epoch.assign(1)
    if dtype == 'float16':                   # as if two overflows had halved it: the resumed run must continue at this scale
        mA._trainer16.loss_scale, mA._trainer16._clean_steps = 2.0 ** 14, 1
    T.save_checkpoint(str(tmp_path), mA, oA)
    hA += T.fit(mA, oA, lf, df, train, val, n_epochs=3, patience=5, log=lambda s: None, compute_dtype=dtype)
    assert [h['epoch'] for h in hA] == [0, 1, 2] and oA.iterations == 9
    assert float(hA[2]['train_loss']) < float(hA[0]['train_loss'])
    mB, oB, lf2, df2 = _setup16()
    for p in mB.trainable_variables:
        p.t.zero_()
    T.load_checkpoint(str(tmp_path), mB, oB)
    T.fit(mB, oB, lf2, df2, train, val, n_epochs=3, patience=5, log=lambda s: None, compute_dtype=dtype)
    torch.cuda.synchronize()
    assert torch.equal(mA.flat_params, mB.flat_params), 'max |d| %.3e' % float((mA.flat_params - mB.flat_params).abs().max())
    assert (mB._trainer16.loss_scale, mB._trainer16._clean_steps) == (mA._trainer16.loss_scale, mA._trainer16._clean_steps)
    assert mA._trainer16.loss_scale == (2.0 ** 14 if dtype == 'float16' else 1.0) and mA._trainer16.skipped_steps == 0


def test_fused_block_backward_gives_the_gradients_of_the_separate_routes(monkeypatch):
    """a whole fp32 step with the gate + GroupNorm-2 backward of every block fused (bts_block_bwd, the default) against the same step
    with BTS_FUSE_BLOCK_BWD=0: the same arithmetic with other partial-sum groupings -- gradients equal to fp32 rounding of the sums"""
    import bts_amd  # noqa: F401
    from bts_amd import tape as T
    from bts_amd.util import reduce_sum
    x, y = _data16(1, 300)[0]

    def grads(switch):
        monkeypatch.setenv('BTS_FUSE_BLOCK_BWD', switch)
        m, opt, lf, df = _setup16()
        for blk_norm in [p for p in m.trainable_variables if (p.name or '').endswith('gn2/gamma')]:
            blk_norm.t.fill_(0.7)                       # (the reference initialises GN2's gamma to zero: give the conv branch a gradient)
        T.bump_weights_epoch()
        m.encoder._seed = m.vae._seed = 5
        with T.GradientTape() as tape:
            yp, yv, zm, zl = m(x, training=True, inference=False)
            loss = lf(x, y, yp, yv, zm, zl) + reduce_sum(m.losses)
        tape.gradient(loss, m.trainable_variables)
        torch.cuda.synchronize()
        return m.flat_grads.clone(), m

    g1, m = grads('1')
    g0, _ = grads('0')
    assert float(g0.abs().max()) > 0
    worst = 0.0
    off = 0
    for p in m.trainable_variables:              # (flat_grads holds the parameters in trainable_variables order)
        k = p.t.numel()
        a, b = g1[off:off + k], g0[off:off + k]
        off += k
        worst = max(worst, float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12))
    assert g1.numel() - 8 < off <= g1.numel() and worst <= 2e-5, worst        # (the flat buffer is padded to a vector width)


def test_overlapped_gradient_sync_is_bit_identical(monkeypatch):
    """parallel.GradSync (bucketed all-reduce issued from inside the backward pass, L2 term applied per bucket) against
    the plain path (backward, then all_reduce_gradients) on a 1-rank RCCL group: identical parameters after 3 steps."""
    import bts_amd  # noqa: F401
    from bts_amd import parallel
    from bts_amd.util import train_step
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29617', rank=0, world_size=1)
    try:
        train = _data(3, 300)
        results = []
        for overlap in (True, False):
            if overlap:
                monkeypatch.delenv('BTS_DP_NO_OVERLAP', raising=False)
            else:
                monkeypatch.setenv('BTS_DP_NO_OVERLAP', '1')
            m, opt, lf, df = _setup()
            opt(epoch=0)
            if overlap:
                gs = parallel.grad_sync(m)
                assert gs is not None and len(gs.buckets) >= 1
                parallel_buckets = parallel.GradSync(m, bucket_bytes=1 << 16)   # many small buckets: exercises the bookkeeping
                m._grad_sync = parallel_buckets
                assert len(parallel_buckets.buckets) > 4
            else:
                assert parallel.grad_sync(m) is None
            for x, y in train:
                loss, _, _ = train_step(m, opt, lf, df, x, y)
            torch.cuda.synchronize()
            results.append((m.flat_params.clone(), float(loss)))
        assert results[0][1] == results[1][1]
        assert torch.equal(results[0][0], results[1][0]), 'max |d| %.3e' % float((results[0][0] - results[1][0]).abs().max())
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_follow_the_backward_pass():
    """SURVEY 8e (C1, "reverse-layer order"): the flat gradient buffer is laid out in the order the backward pass finishes the
    layers (vae, decoder, encoder level 3 .. 0), buckets are runs of whole layer groups, and GradSync launches a bucket's
    all-reduce from inside the backward pass as soon as its last gradient is written.  On a 1-rank RCCL group with the CLI-default
    model (42,174,773 parameters; 32^3 crop -- the layout does not depend on it): at least 80 % of the gradient bytes must be in
    flight before the last 10 % of the tape's nodes are replayed, the last bucket is the small one, and every byte is exchanged
    exactly once."""
    import bts_amd  # noqa: F401
    from bts_amd import parallel
    from bts_amd.data import synthetic_batch
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1)
    try:
        os.environ.pop('BTS_DP_NO_OVERLAP', None)
        m = Model(base_filters=32, reduction=8, depth=4, groups=8)
        crop = (32, 32, 32)
        m.build((1,) + crop + (2,))
        # layout: the groups tile the buffer; regulariser ranges sit inside it and do not overlap
        spans = m._group_spans
        assert spans[0][0] == 0 and spans[-1][1] == m.n_params and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        rg = sorted(m._l2_ranges)
        assert len(rg) <= 128 and all(a[0] + a[1] <= b[0] for a, b in zip(rg, rg[1:]))
        x, y, _, _ = synthetic_batch(1, crop, latent=128, seed=5)
        opt = ScheduledOptim(1e-4)
        opt(epoch=0)
        gs = parallel.grad_sync(m)
        assert gs is not None
        sizes = [ln * 4 for _, ln, _ in gs.buckets]
        assert sum(sizes) == m.flat_grads.numel() * 4 and sizes[-1] <= (4 << 20) + 64 and len(sizes) >= 3, sizes
        train_step(m, opt, DiceVAELoss(), DiceCoefficient(), x.cuda(), y.cuda())
        torch.cuda.synchronize()
        log, total_nodes = gs.launch_log, gs.nodes_total
        assert sorted(b for b, _, _ in log) == list(range(len(sizes)))            # every bucket launched exactly once
        early = sum(nbytes for _, at, nbytes in log if at <= 0.9 * total_nodes)
        print('buckets (MB):', ['%.1f' % (v / 1e6) for v in sizes], '; launched after node',
              ['%d/%d' % (at, total_nodes) for _, at, _ in sorted(log)], '; %.1f %% of the bytes in flight before the last 10 %% of nodes'
              % (100.0 * early / sum(sizes)))
        assert early >= 0.8 * sum(sizes)
        assert [b for b, _, _ in log] == sorted(b for b, _, _ in log)             # front to back: the order the backward pass finishes them
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_follow_the_backward_pass_of_the_16bit_step():
    """The same contract for lowp_train.LowPrecisionTrainer.step (BASELINE configs[3] runs configs[2]'s step on every GPU): its explicit
    backward walks vae -> decoder -> encoder and reports each stage's parameters; on a 1-rank RCCL group at least 80 % of the gradient
    bytes are in flight before the last 10 % of the stages, every bucket is exchanged once, front to back, and the gradient equals the
    post-backward exchange's bit for bit."""
    import bts_amd  # noqa: F401
    from bts_amd import parallel
    from bts_amd.data import synthetic_batch
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, ScheduledOptim
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29643', rank=0, world_size=1)
    try:
        os.environ.pop('BTS_DP_NO_OVERLAP', None)
        m = Model(base_filters=32, reduction=8, depth=4, groups=8)
        crop = (32, 32, 32)
        m.build((2,) + crop + (2,))
        x, y, mask, eps = synthetic_batch(2, crop, latent=128, seed=5)
        start = m.flat_params.clone()
        tr = LowPrecisionTrainer(m, 'bfloat16')
        grads = []
        for plain in (False, True):
            if plain:
                os.environ['BTS_DP_NO_OVERLAP'] = '1'
            m.flat_params.copy_(start)
            from bts_amd.tape import bump_weights_epoch
            bump_weights_epoch()
            m.encoder.set_dropout_mask(mask)
            m.vae.set_eps(eps)
            opt = ScheduledOptim(1e-4)
            opt(epoch=0)
            tr.step(opt, DiceCoefficient(), x.cuda(), y.cuda())
            torch.cuda.synchronize()
            grads.append(m.flat_grads.clone())
            if not plain:
                gs = parallel.grad_sync(m)
                assert gs is not None
                sizes = [ln * 4 for _, ln, _ in gs.buckets]
                log, total = list(gs.launch_log), gs.nodes_total
                assert gs.tape.nodes_replayed == total, (gs.tape.nodes_replayed, total)      # every stage reported
                assert sorted(b for b, _, _ in log) == list(range(len(sizes)))
                early = sum(nbytes for _, at, nbytes in log if at <= 0.9 * total)
                print('16-bit step: buckets (MB):', ['%.1f' % (v / 1e6) for v in sizes], '; launched after stage',
                      ['%d/%d' % (at, total) for _, at, _ in sorted(log)], '; %.1f %% of the bytes in flight before the last 10 %%'
                      % (100.0 * early / sum(sizes)))
                assert early >= 0.8 * sum(sizes)
                assert [b for b, _, _ in log] == sorted(b for b, _, _ in log)
        assert torch.equal(grads[0], grads[1])
    finally:
        os.environ.pop('BTS_DP_NO_OVERLAP', None)
        dist.destroy_process_group()


def test_step_fence_bounds_the_run_ahead(monkeypatch):
    """ops.step_fence: however many steps the host enqueues without reading a value back, at most two are in flight (every step holds
    its activations; unbounded run-ahead made the caching allocator grow inside timed regions) -- and the fence only
    waits: the parameters after 6 fenced steps equal those after 6 unfenced ones bit for bit."""
    import bts_amd  # noqa: F401
    from bts_amd import ops
    from bts_amd.util import train_step
    data = _data(6, 500)
    finals = []
    for fenced in (True, False):
        if fenced:
            monkeypatch.delenv('BTS_STEP_FENCE', raising=False)
        else:
            monkeypatch.setenv('BTS_STEP_FENCE', '0')
        ops._fence.clear()
        m, opt, lf, df = _setup()
        opt(epoch=0)
        for x, y in data:
            train_step(m, opt, lf, df, x, y)          # no synchronize, no .item() in between
            q = ops._fence.get((torch.cuda.current_device(), 'train'), [])
            assert len(q) <= 2
            assert (len(q) > 0) == fenced
            for ev in q[:-2]:
                assert ev.query()
        torch.cuda.synchronize()
        finals.append(m.flat_params.clone())
    assert torch.equal(finals[0], finals[1])
