"""CPU: the training-loop shell of SURVEY 8 f-1 (reference train.py:116-216, args.py:193-194): log format, Mean metric,
patience rule, checkpoint container, and the loop itself driven by stub step functions (no kernels run here)."""
import os

import numpy as np
import pytest
import torch

import bts_amd  # noqa: F401
from bts_amd import train as T
from bts_amd.model import Model
from bts_amd.util import ScheduledOptim


def test_log_header_is_the_reference_header():
    # train.py:119-126 joins exactly these eight names with ','
    assert T.LOG_HEADER == 'epoch,lr,train_loss,train_macro_dice,train_micro_dice,val_loss,val_macro_dice,val_micro_dice'


def test_log_row_formats_like_float32_numpy_scalars():
    # train.py:186-193: str(<tf float32>.numpy()) -> numpy's shortest float32 repr
    row = T.log_row(3, 1e-4, 0.75, 0.5, 0.25, 1.5, 0.125, 1.0 / 3.0)
    assert row == '3,1e-04,0.75,0.5,0.25,1.5,0.125,0.33333334'


def test_mean_is_float32_total_over_count():
    m = T.Mean('x')
    assert m.result() == np.float32(0.0)
    vals = [0.1, 0.2, 0.7, 1e-8]
    for v in vals[:2]:
        m.update_state(torch.tensor([v], dtype=torch.float32))
    for v in vals[2:]:
        m.update_state(v)
    tot = np.float32(0.0)
    for v in (vals[2], vals[3]):
        tot = np.float32(tot + np.float32(v))
    dev = np.float32(np.float32(vals[0]) + np.float32(vals[1]))
    assert m.result() == np.float32(np.float32(tot + dev) / np.float32(4))
    m.reset_states()
    assert m.count == 0 and m.result() == np.float32(0.0)


def _reference_patience(vals, limit):
    """train.py:195-208 transcribed as a generator of actions"""
    best, patience, out = 0.0, 0, []
    for v in vals:
        if v > best:
            best, patience = v, 0
            out.append('save')
        elif patience == limit:
            out.append('stop')
            return out
        else:
            patience += 1
            out.append('wait')
    return out


@pytest.mark.parametrize('limit', [0, 1, 3])
def test_patience_rule(limit):
    rng = np.random.RandomState(limit)
    for _ in range(20):
        vals = list(np.round(rng.rand(30), 2))
        tr, got = T.PatienceTracker(limit), []
        for v in vals:
            got.append(tr.update(v))
            if got[-1] == 'stop':
                break
        assert got == _reference_patience(vals, limit)


class _FakeModel(object):
    def __init__(self):
        from bts_amd.model import _EpochVariable
        self.epoch = _EpochVariable()


def test_fit_with_stub_steps_writes_reference_log_and_stops(tmp_path):
    model, opt = _FakeModel(), ScheduledOptim(1e-4, n_epochs=10)
    val_dice = iter([0.2, 0.3, 0.3, 0.1, 0.25, 0.9, 0.9])   # save, save, wait, wait, stop (patience 2)
    saved = []

    def tstep(x, y):
        return torch.tensor([float(x)]), torch.tensor([0.5]), torch.tensor([0.25])

    state = {}

    def estep(x, y):
        if 'v' not in state:
            state['v'] = next(val_dice)
        return torch.tensor([2.0]), torch.tensor([state['v']]), torch.tensor([0.0])

    class Data(list):
        def __iter__(self):
            state.pop('v', None)
            return super().__iter__()

    orig = T.save_checkpoint
    T.save_checkpoint = lambda folder, m, o=None, **kw: saved.append(int(m.epoch.value().numpy()))
    try:
        hist = T.fit(model, opt, None, None, [(1.0, 0), (3.0, 0)], Data([(0, 0), (0, 0)]), n_epochs=10, patience=2,
                     save_folder=str(tmp_path), train_step_fn=tstep, eval_step_fn=estep, log=lambda s: None)
    finally:
        T.save_checkpoint = orig
    assert [h['epoch'] for h in hist] == [0, 1, 2, 3, 4]
    assert saved == [0, 1]
    lines = open(os.path.join(str(tmp_path), 'train.log')).read().strip().split('\n')
    assert lines[0] == T.LOG_HEADER and len(lines) == 6
    e, lr, tl, tm, tmi, vl, vm, vmi = lines[2].split(',')
    assert e == '1' and tl == '2.0' and tm == '0.5' and tmi == '0.25' and vl == '2.0' and vm == '0.3' and vmi == '0.0'
    assert lr == str(np.float32(1e-4 * (1 - 1 / 10.0) ** 0.9))          # util.py:76-80 schedule applied per epoch
    assert int(model.epoch.value().numpy()) == 4                       # train.py:135


def test_fit_resumes_at_model_epoch():
    model, opt = _FakeModel(), ScheduledOptim(1e-4, n_epochs=5)
    model.epoch.assign(3)
    st = lambda x, y: (torch.tensor([1.0]), torch.tensor([0.5]), torch.tensor([0.5]))
    hist = T.fit(model, opt, None, None, [(0, 0)], [(0, 0)], n_epochs=5, patience=10, train_step_fn=st, eval_step_fn=st,
                 log=lambda s: None)
    assert [h['epoch'] for h in hist] == [3, 4]                        # train.py:133 range(model.epoch, n_epochs)


def test_checkpoint_container_round_trip(tmp_path):
    m = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m.build((1, 8, 8, 8, 2))
    g = torch.Generator().manual_seed(5)
    for p in m.trainable_variables:
        p.t.copy_(torch.randn(p.t.shape, generator=g))
    m.epoch.assign(7)
    m.encoder._seed, m.vae._seed = 1234, 99
    opt = ScheduledOptim(1e-4)
    opt.iterations = 41
    opt(epoch=7)
    mom = (torch.randn(m.flat_params.shape, generator=g), torch.rand(m.flat_params.shape, generator=g))
    opt._state[id(m.flat_params)] = mom
    meta = T.save_checkpoint(str(tmp_path), m, opt)
    assert meta['epoch'] == 7 and meta['n_params'] == m.n_params
    assert os.path.exists(os.path.join(str(tmp_path), T.CHECKPOINT_NAME))

    m2 = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m2.build((1, 8, 8, 8, 2))
    opt2 = ScheduledOptim(1e-4)
    T.load_checkpoint(str(tmp_path), m2, opt2)
    assert torch.equal(m2.flat_params, m.flat_params)
    assert int(m2.epoch.value().numpy()) == 7 and m2.encoder._seed == 1234 and m2.vae._seed == 99
    assert opt2.iterations == 41 and opt2.learning_rate == opt.learning_rate
    s2 = opt2._state[id(m2.flat_params)]
    assert torch.equal(s2[0], mom[0]) and torch.equal(s2[1], mom[1])

    m3 = Model(base_filters=8, reduction=2, depth=3, groups=2)   # different architecture: loud failure
    m3.build((1, 8, 8, 8, 2))
    with pytest.raises((KeyError, ValueError)):
        T.load_checkpoint(str(tmp_path), m3)


def test_checkpoint_of_a_finished_epoch_resumes_at_the_next_one(tmp_path):
    """fit() saves after the epoch's last step: the container says next_epoch = epoch + 1, carries the patience tracker
    and the datasets' generator states, and the resumed fit() starts there with all three restored"""
    from bts_amd import data as D
    m = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m.build((1, 8, 8, 8, 2))
    m.epoch.assign(4)
    tr = T.PatienceTracker(5)
    tr.best, tr.patience = 0.625, 3
    ds = D._Dataset(['a', 'b', 'c'], 1, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=3, device='cpu')
    torch.randperm(3, generator=ds.order_gen)                    # advance both generators past their seeds
    torch.rand(5, generator=ds.gen)
    want = (torch.randperm(3, generator=torch.Generator().set_state(ds.order_gen.get_state())).tolist(),
            torch.rand(4, generator=torch.Generator().set_state(ds.gen.get_state())).tolist())
    meta = T.save_checkpoint(str(tmp_path), m, None, completed=True, tracker=tr, datasets={'train': ds, 'val': [1, 2]})
    assert meta['epoch'] == 4 and meta['next_epoch'] == 5 and meta['tracker'] == {'best': 0.625, 'patience': 3}

    m2 = Model(base_filters=8, reduction=2, depth=2, groups=2)
    m2.build((1, 8, 8, 8, 2))
    T.load_checkpoint(str(tmp_path), m2)
    assert int(m2.epoch.value().numpy()) == 5
    ds2 = D._Dataset(['a', 'b', 'c'], 1, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=999, device='cpu')
    seen = {}

    def estep(x, y):
        return torch.tensor([1.0]), torch.tensor([0.5]), torch.tensor([0.5])    # 0.5 < restored best 0.625 -> no save

    class Probe(list):
        def load_state_dict(self, st):
            seen['loaded'] = True
            ds2.load_state_dict(st)

    orig = T.save_checkpoint
    saves = []
    T.save_checkpoint = lambda *a, **k: saves.append(1)
    try:
        hist = T.fit(m2, ScheduledOptim(1e-4, n_epochs=8), None, None, Probe([(0, 0)]), [(0, 0)], n_epochs=6, patience=5,
                     save_folder=str(tmp_path / 'out'), train_step_fn=estep, eval_step_fn=estep, log=lambda s: None)
    finally:
        T.save_checkpoint = orig
    assert [h['epoch'] for h in hist] == [5] and saves == [] and seen.get('loaded')
    assert torch.randperm(3, generator=ds2.order_gen).tolist() == want[0]
    assert torch.rand(4, generator=ds2.gen).tolist() == want[1]


def test_dataset_shards_by_rank_with_equal_lengths():
    from bts_amd import data as D
    files = ['f%d' % i for i in range(7)]
    shards = []
    for r in range(2):
        ds = D._Dataset(files, 2, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=5, device='cpu', rank=r, world=2)
        assert len(ds) == 2                                     # 7 // 2 = 3 examples per rank -> 2 batches of <= 2
        order = torch.randperm(7, generator=ds.order_gen).tolist()
        shards.append(order[r::2][:3])
    assert len(shards[0]) == len(shards[1]) == 3 and not set(shards[0]) & set(shards[1])
    g0 = D._Dataset(files, 2, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=5, device='cpu', rank=0, world=2).gen
    g1 = D._Dataset(files, 2, (8, 8, 8, 2), (8, 8, 8), 3, True, seed=5, device='cpu', rank=1, world=2).gen
    assert torch.rand(4, generator=g0).tolist() != torch.rand(4, generator=g1).tolist()   # per-rank augmentation draws


def test_train_args_round_trip(tmp_path):
    args = {'model_args': {'base_filters': 32, 'reduction': 8}, 'crop_size': [128, 128, 128], 'lr': 1e-4}
    T.save_train_args(str(tmp_path), args)
    assert os.path.basename(T.ARGS_NAME) == 'train_args.pkl'            # args.py:193
    assert T.load_train_args(str(tmp_path)) == args
