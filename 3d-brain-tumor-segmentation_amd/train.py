"""Training-loop shell and checkpoint container (reference train.py:93-216, args.py:176-196).

`fit()` reproduces the reference's epoch loop: optimizer(epoch) schedule call, six tf.keras.metrics.Mean accumulators,
the `train.log` CSV (header train.py:119-126, one row per epoch :186-193), best-validation checkpointing and the
patience rule exactly as written there (:196-208: the epoch that *reaches* `patience` stops, improvements reset it).

Differences, each deliberate:
  * weights: the reference writes Keras HDF5 (`chkpt.hdf5`, train.py:100,201); h5py is not available, so the
    container is safetensors (`chkpt.safetensors`) keyed by the engine's variable names;
  * the reference never saves optimizer state and (SURVEY f-1 caveat) probably not `Model.epoch` either; this
    container persists epoch, Adam m / v / step count and the layers' RNG counters, so a resumed run continues
    bit-exactly (tests/test_train_gpu.py);
  * metric accumulation stays on the device (one scalar add per step) and is read once per epoch -- the reference
    synchronises every step through `.update_state`.
"""
import json
import os
import pickle

import numpy as np
import torch

from . import parallel
from .tape import Tensor, bump_weights_epoch
from .util import reduce_sum, train_step

LOG_HEADER = ','.join(['epoch', 'lr', 'train_loss', 'train_macro_dice', 'train_micro_dice', 'val_loss',
                       'val_macro_dice', 'val_micro_dice'])                       # train.py:119-126
CHECKPOINT_NAME = 'chkpt.safetensors'                                             # reference: chkpt.hdf5 (train.py:201)
ARGS_NAME = 'train_args.pkl'                                                      # args.py:193-194
FORMAT_VERSION = 2      # 2: Adam moments in the backward-completion order of the flat buffers (model._flatten_parameters, round 3)


class Mean(object):
    """tf.keras.metrics.Mean: float32 running total / count (train.py:109-114)."""

    def __init__(self, name='mean'):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self.total = None
        self.count = 0
        self._host = np.float32(0.0)

    def update_state(self, value):
        if isinstance(value, Tensor):
            value = value.t
        if torch.is_tensor(value):
            v = value.detach().reshape(()).to(torch.float32)
            self.total = v.clone() if self.total is None else self.total + v
        else:
            self._host = np.float32(self._host + np.float32(value))
        self.count += 1

    def result(self):
        """np.float32, like `metric.result().numpy()`; 0.0 before the first update (Keras divides no-NaN)"""
        if self.count == 0:
            return np.float32(0.0)
        tot = self._host
        if self.total is not None:
            tot = np.float32(tot + np.float32(self.total.item()))
        return np.float32(tot / np.float32(self.count))


def log_row(epoch, lr, train_loss, train_macro, train_micro, val_loss, val_macro, val_micro):
    """one `train.log` line, formatted like the reference's str(<float32>.numpy()) fields (train.py:186-193)"""
    f = lambda v: str(np.float32(v))
    return ','.join([str(int(epoch)), f(lr), f(train_loss), f(train_macro), f(train_micro), f(val_loss), f(val_macro),
                     f(val_micro)])


class PatienceTracker(object):
    """train.py:195-208 verbatim: improvement -> save + reset; otherwise stop once `patience` epochs have already
    passed without one (the check precedes the increment, so training runs patience + 1 non-improving epochs)."""

    def __init__(self, patience):
        self.limit = patience
        self.best = 0.0
        self.patience = 0

    def update(self, val_macro_dice):
        """-> 'save' | 'stop' | 'wait'"""
        if val_macro_dice > self.best:
            self.best = val_macro_dice
            self.patience = 0
            return 'save'
        if self.patience == self.limit:
            return 'stop'
        self.patience += 1
        return 'wait'


# ---- checkpoint container ------------------------------------------------------------------------------------------
def write_container(path, tensors, meta):
    """tensors: {name: torch tensor}; meta: JSON-able dict (stored in the safetensors header)"""
    from safetensors.torch import save_file
    save_file({k: v.detach().to('cpu').contiguous() for k, v in tensors.items()}, path,
              metadata={'bts': json.dumps(meta)})


def read_container(path):
    from safetensors import safe_open
    tensors = {}
    with safe_open(path, framework='pt', device='cpu') as f:
        meta = json.loads((f.metadata() or {}).get('bts', '{}'))
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, meta


def _rng_layers(model):
    out = {}
    for name in ('encoder', 'vae'):
        lay = getattr(model, name, None)
        if lay is not None and hasattr(lay, '_seed'):
            out[name] = lay
    return out


_RANK_STRIDE = 0x9E3779B97F4A7C15      # parallel.decorrelate_rng: rank r's counters run at base + r * stride (mod 2^63)


def save_checkpoint(folder, model, optimizer=None, completed=False, tracker=None, datasets=None, write=True):
    """weights by variable name + `epoch`; with an optimizer also Adam m / v (flat order) and its step count.

    Data-parallel runs: EVERY rank calls this (it gathers the per-rank random state: each rank's augmentation generator;
    the dropout / reparameterisation counters are stored rank-independent, as their base value), rank 0 alone passes
    write=True.  A resumed rank r gets back ITS generator state and base + r * stride counters, so all ranks continue the
    uninterrupted trajectory instead of replaying rank 0's draws.

    completed=True (what fit() passes: it saves AFTER the epoch's last step): the container also records
    `next_epoch = epoch + 1`, and a resumed fit() starts there -- the reference restarts at the saved epoch
    (train.py:133-135) but carries no optimizer state; with Adam moments, step count and RNG counters persisted, repeating
    that epoch would apply it twice.  tracker: the PatienceTracker (best / patience persist, so the first resumed epoch cannot
    overwrite a better checkpoint).  datasets: {'train': ds, 'val': ds}; objects with state_dict() (data._Dataset: shuffle
    and augmentation generators) are persisted."""
    tensors = {'var/' + p.name: p.t for p in model.trainable_variables}
    rk = parallel.rank() if getattr(model, '_rng_rank', None) is not None else 0      # offset decorrelate_rng applied, if it ran
    meta = {'format': FORMAT_VERSION, 'epoch': int(model.epoch.value().numpy()), 'n_params': int(model.n_params),
            'rng': {k: (int(v._seed) - rk * _RANK_STRIDE) & 0x7FFFFFFFFFFFFFFF for k, v in _rng_layers(model).items()},
            'rng_is_base': True}
    if completed:
        meta['next_epoch'] = meta['epoch'] + 1
    if tracker is not None:
        meta['tracker'] = {'best': float(tracker.best), 'patience': int(tracker.patience)}
    tr16 = getattr(model, '_trainer16', None)          # fit(compute_dtype=...): the dynamic loss scale is training state too
    if tr16 is not None:
        tr16.settle()           # the last step's overflow decision belongs to the state that is saved
        meta['loss_scale'] = {'dtype': tr16.dtype_name, 'scale': float(tr16.loss_scale), 'clean_steps': int(tr16._clean_steps),
                              'skipped_steps': int(tr16.skipped_steps)}
    local = {}
    for key, ds in (datasets or {}).items():
        if hasattr(ds, 'state_dict'):
            for k, v in ds.state_dict().items():
                local['data/%s/%s' % (key, k)] = v
    per_rank = parallel.gather_objects(local)          # a collective when a process group exists: all ranks are here
    for r, st in enumerate(per_rank):
        for k, v in st.items():
            if k.endswith('/gen'):
                tensors['%s@%d' % (k, r)] = v          # augmentation draws: one stream per rank
            elif r == 0:
                tensors[k] = v                         # the shuffle generator is common to all ranks
    if not write:
        return meta
    if optimizer is not None:
        meta['optimizer'] = {'iterations': int(optimizer.iterations), 'learning_rate': float(optimizer.learning_rate),
                             'init_lr': float(optimizer.init_lr), 'n_epochs': float(optimizer.n_epochs)}
        st = optimizer._state.get(id(model.flat_params))
        if st is not None:
            tensors['adam/m'], tensors['adam/v'] = st[0], st[1]
    os.makedirs(folder, exist_ok=True)
    tmp = os.path.join(folder, CHECKPOINT_NAME + '.tmp')
    write_container(tmp, tensors, meta)
    os.replace(tmp, os.path.join(folder, CHECKPOINT_NAME))
    return meta


def load_checkpoint(folder, model, optimizer=None):
    """inverse of save_checkpoint; the model must be built (same architecture).  Returns the stored meta dict."""
    tensors, meta = read_container(os.path.join(folder, CHECKPOINT_NAME))
    fmt = meta.get('format')
    if fmt not in (1, FORMAT_VERSION):
        raise ValueError('unknown checkpoint format %r' % (fmt,))
    missing = [p.name for p in model.trainable_variables if 'var/' + p.name not in tensors]
    if missing:
        raise KeyError('checkpoint lacks %d variables, e.g. %s' % (len(missing), missing[:3]))
    for p in model.trainable_variables:
        src = tensors['var/' + p.name]
        if tuple(src.shape) != tuple(p.t.shape):
            raise ValueError('shape mismatch for %s: %s vs %s' % (p.name, tuple(src.shape), tuple(p.t.shape)))
        p.t.copy_(src.to(p.t.device))
    model.epoch.assign(int(meta.get('next_epoch', meta.get('epoch', 0))))
    # consumed by the next fit(): PatienceTracker state and the datasets' generator states
    data = {}
    mine = '@%d' % parallel.rank()
    for k, v in tensors.items():
        if not k.startswith('data/'):
            continue
        if '@' in k:                                  # per-rank entries: keep this rank's, under the plain name
            if k.endswith(mine):
                data[k[len('data/'):-len(mine)]] = v
        else:
            data.setdefault(k[len('data/'):], v)
    model._resume = {'tracker': meta.get('tracker'), 'data': data, 'loss_scale': meta.get('loss_scale')}
    for k, lay in _rng_layers(model).items():
        if k in meta.get('rng', {}):
            lay._seed = int(meta['rng'][k])
    if meta.get('rng_is_base'):                       # counters were stored without the rank offset: re-apply this rank's
        model._rng_rank = None
        if parallel.active():
            parallel.decorrelate_rng(model)
    if optimizer is not None and 'optimizer' in meta:
        o = meta['optimizer']
        optimizer.iterations = int(o['iterations'])
        optimizer.learning_rate = float(o['learning_rate'])
        if 'adam/m' in tensors and fmt != FORMAT_VERSION:
            # format 1 stored the Adam moments in the flat order of that build (reference variable order); variables are keyed by name and
            # load fine, the moments cannot be mapped without that order: they restart at zero (and the bias correction with them)
            import warnings
            warnings.warn('checkpoint format %r: weights, epoch and random state restored; Adam moments dropped (flat order changed in '
                          'format %d)' % (fmt, FORMAT_VERSION))
            optimizer.iterations = 0
        elif 'adam/m' in tensors:
            dev = model.flat_params.device
            optimizer._state[id(model.flat_params)] = (tensors['adam/m'].to(dev).contiguous(),
                                                       tensors['adam/v'].to(dev).contiguous())
    bump_weights_epoch()
    return meta


def save_train_args(folder, args_dict):
    """args.py:193-194: the run's arguments as a pickled plain dict (`train_args.pkl`), read back by a resumed run or by
    the inference script to rebuild the model (`model_args`, `crop_size`)"""
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, ARGS_NAME), 'wb') as f:
        pickle.dump(dict(args_dict), f)


def load_train_args(folder):
    with open(os.path.join(folder, ARGS_NAME), 'rb') as f:
        return pickle.load(f)


# ---- the loop -------------------------------------------------------------------------------------------------------
def eval_step(model, loss_fn, dice_fn, x, y):
    """validation iteration, train.py:166-173: forward with training=False, loss incl. regularisers, Dice"""
    y_pred, y_vae, z_mean, z_logvar = model(x, training=False, inference=False)
    loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)
    loss = loss + reduce_sum(model.losses)
    macro, micro = dice_fn(y, y_pred)
    return loss, macro, micro


def fit(model, optimizer, loss_fn, dice_fn, train_data, val_data, n_epochs, patience=10, save_folder=None,
        train_step_fn=None, eval_step_fn=None, log=print, compute_dtype=None):
    """The reference's `train(args)` from the logging set-up on (train.py:116-216).

    compute_dtype: None / 'float32' = the fp32 engine (util.train_step); 'bfloat16' / 'float16' = the 16-bit-storage engine
    (lowp_train.LowPrecisionTrainer.step: fp32 master weights, fp32 statistics, fp16 with dynamic loss scaling) for the training
    iterations (its loss is util.DiceVAELoss, like train.py:143-147; loss_fn is then used by validation only) -- validation
    stays on the fp32 engine.

    train_data / val_data: re-iterable collections of (x, y) batches (NDHWC tensors on the device).
    Resumes at `model.epoch` (train.py:133).  Returns the list of per-epoch rows (dicts).  On data-parallel runs every
    rank must call fit() (with the same n_epochs / patience; equally long shards) and iterates its own shard; rank 0 alone writes
    files, and rank 0's validation Dice drives the save / stop decision of all ranks."""
    if train_step_fn is None and compute_dtype not in (None, 'float32'):
        from .lowp_train import LowPrecisionTrainer
        trainer = getattr(model, '_trainer16', None)
        if trainer is None or trainer.dtype_name != compute_dtype:
            trainer = model._trainer16 = LowPrecisionTrainer(model, compute_dtype)
        saved = (getattr(model, '_resume', None) or {}).get('loss_scale')
        if saved and saved.get('dtype') == compute_dtype:
            trainer.loss_scale, trainer._clean_steps = float(saved['scale']), int(saved['clean_steps'])
            trainer.skipped_steps = int(saved.get('skipped_steps', 0))
        train_step_fn = lambda x, y: trainer.step(optimizer, dice_fn, x, y)     # noqa: E731
    tstep = train_step_fn or (lambda x, y: train_step(model, optimizer, loss_fn, dice_fn, x, y))
    estep = eval_step_fn or (lambda x, y: eval_step(model, loss_fn, dice_fn, x, y))
    writer = save_folder is not None and parallel.rank() == 0
    names = ('train_loss', 'train_macro_dice', 'train_micro_dice', 'val_loss', 'val_macro_dice', 'val_micro_dice')
    m = {k: Mean(k) for k in names}
    if writer:
        os.makedirs(save_folder, exist_ok=True)
        with open(os.path.join(save_folder, 'train.log'), 'w') as f:
            f.write(LOG_HEADER + '\n')
    tracker = PatienceTracker(patience)
    resume = getattr(model, '_resume', None)       # left by load_checkpoint
    if resume is not None:
        model._resume = None
        if resume.get('tracker'):
            tracker.best, tracker.patience = float(resume['tracker']['best']), int(resume['tracker']['patience'])
        for key, ds in (('train', train_data), ('val', val_data)):
            st = {k[len(key) + 1:]: v for k, v in resume.get('data', {}).items() if k.startswith(key + '/')}
            if st and hasattr(ds, 'load_state_dict'):
                ds.load_state_dict(st)
    history = []
    for epoch in range(int(model.epoch.value().numpy()), int(n_epochs)):
        log('Epoch {}.'.format(epoch))
        model.epoch.assign(epoch)
        optimizer(epoch=epoch)
        for x, y in train_data:
            loss, macro, micro = tstep(x, y)
            m['train_loss'].update_state(loss)
            m['train_macro_dice'].update_state(macro)
            m['train_micro_dice'].update_state(micro)
        log('Training. Loss: {l: .4f}, Macro Dice: {d1: 1.4f}, Micro Dice: {d2: 1.4f}.'.format(
            l=m['train_loss'].result(), d1=m['train_macro_dice'].result(), d2=m['train_micro_dice'].result()))
        for x, y in val_data:
            loss, macro, micro = estep(x, y)
            m['val_loss'].update_state(loss)
            m['val_macro_dice'].update_state(macro)
            m['val_micro_dice'].update_state(micro)
        log('Validation. Loss: {l: .4f}, Macro Dice: {d1: 1.4f}, Micro Dice: {d2: 1.4f}.'.format(
            l=m['val_loss'].result(), d1=m['val_macro_dice'].result(), d2=m['val_micro_dice'].result()))
        row = {'epoch': epoch, 'lr': np.float32(optimizer.learning_rate)}
        row.update({k: m[k].result() for k in names})
        history.append(row)
        if writer:
            with open(os.path.join(save_folder, 'train.log'), 'a') as f:
                f.write(log_row(epoch, row['lr'], *[row[k] for k in names]) + '\n')
        # The save / stop decision must be the same on every rank: save_checkpoint is a collective (it gathers the per-rank random
        # state) and a rank that stops alone leaves the others in the next gradient exchange.  The built-in DiceCoefficient all-reduces
        # its sums, a custom eval_step_fn / metric need not -- so rank 0's validation Dice decides for all, and whether files are
        # wanted is agreed on as well (a save_folder on rank 0 only is fine)
        val_dice, want_files = float(row['val_macro_dice']), save_folder is not None
        if parallel.active():
            got = parallel.gather_objects((val_dice, want_files))
            val_dice, want_files = got[0][0], any(w for _, w in got)
        action = tracker.update(val_dice)
        if action == 'save':
            if want_files:                       # every rank: the per-rank random state is gathered; rank 0 writes
                save_checkpoint(save_folder, model, optimizer, completed=True, tracker=tracker,
                                datasets={'train': train_data, 'val': val_data}, write=writer)
            log('Saved model weights.')
        elif action == 'stop':
            log('Validation dice has not improved in {} epochs. Stopped training.'.format(patience))
            return history
        for k in names:
            m[k].reset_states()
    return history
