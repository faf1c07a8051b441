"""Input pipeline with the augmentation on the device (reference train.py:12-64).

`prepare_dataset` keeps the reference's signature and per-example transformation (`parse_example`, train.py:14-49):
intensity shift/scale by per-channel sigma, random crop of the concatenated (x, y), independent flips of the three
axes with p = 0.5, one-hot labels minus the background channel -- note the reference applies it to the validation set
too (the same map function serves both, train.py:74-89).  Differences:
  * storage: TFRecord protos (train.py:51-58, preprocess.py:88-96) need TensorFlow; examples are read from `.npz` files
    holding the same two arrays (`x`: (h,w,d,c) float32, `y`: (h,w,d,1) float32);
  * the transformation runs on the GPU (bts_channel_moments + bts_augment_crop: one pass over the crop), the host only
    draws the 2c + 6 random numbers -- at ~70 ms per step a tf.data-style host pipeline would otherwise be the bottleneck;
  * the draws come from a seeded torch generator in a documented order (shift[c], scale[c], 3 crop offsets, 3 flips), so
    an epoch is reproducible; TF's stream cannot be.
"""
import os

import numpy as np
import torch

from . import ops


def synthetic_batch(n, crop, in_ch=2, out_ch=3, seed=1234, dropout_rate=0.2, latent=128):
    """Synthetic BraTS-like batch of SURVEY 8(d), host tensors (x, y, dropout keep-mask, eps):
    x ~ N(0,1) zeroed outside a centred ellipsoid of semi-axes (56,60,52)/128 of the crop (skull-stripped, unit-variance
    channels: preprocess.py:116-124), labels = three nested spheres at a jittered centre -> one-hot minus background
    (train.py:41-43, preprocess.py:36), mask ~ Bernoulli(keep 1-rate) (encoder.py:39,71), eps ~ N(0,1) (vae.py:12).
    All draws from numpy.random.Generator(PCG64(seed)) in that order, so bench.py, the tests and the CPU baseline see the
    same volumes (the oracle carries an identical generator; tests/test_host_logic.py checks they agree)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, H, W = crop
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing='ij')
    semi = np.array([56.0, 60.0, 52.0]) * np.array(crop) / 128.0
    inside = (((zz - D / 2) / semi[0]) ** 2 + ((yy - H / 2) / semi[1]) ** 2 + ((xx - W / 2) / semi[2]) ** 2) <= 1.0
    x = rng.standard_normal((n, D, H, W, in_ch)).astype(np.float32) * inside[None, ..., None]
    y = np.zeros((n, D, H, W, out_ch), np.float32)
    radii = np.array([36.0, 24.0, 12.0]) * min(crop) / 128.0
    for b in range(n):
        centre = np.array([D / 2, H / 2, W / 2]) + rng.uniform(-0.15, 0.15, 3) * np.array(crop)
        r2 = (zz - centre[0]) ** 2 + (yy - centre[1]) ** 2 + (xx - centre[2]) ** 2
        lab = np.zeros((D, H, W), np.int64)
        for k, r in enumerate(radii[:out_ch]):
            lab[r2 <= r * r] = k + 1
        for k in range(out_ch):
            y[b, ..., k] = (lab == k + 1)
    mask = (rng.random((n, D, H, W, in_ch)) >= dropout_rate).astype(np.float32)
    eps = rng.standard_normal((n, latent)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(mask), torch.from_numpy(eps)


class Draws(object):
    """the random numbers of one example (train.py:19-20,26,30-33)"""
    __slots__ = ('shift', 'scale', 'offsets', 'flips')

    def __init__(self, shift, scale, offsets, flips):
        self.shift, self.scale, self.offsets, self.flips = list(shift), list(scale), list(offsets), list(flips)

    @property
    def flip_mask(self):
        return (4 if self.flips[0] else 0) | (2 if self.flips[1] else 0) | (1 if self.flips[2] else 0)


def draw(gen, c, vol_size, crop_size):
    """shift ~ U(-0.1, 0.1)^c, scale ~ U(0.9, 1.1)^c, crop origin uniform over the valid range, flip_k = (u_k > 0.5)"""
    u = torch.rand(2 * c + 6, generator=gen, dtype=torch.float64).tolist()
    shift = [-0.1 + 0.2 * v for v in u[:c]]
    scale = [0.9 + 0.2 * v for v in u[c:2 * c]]
    offs = [min(int(u[2 * c + k] * (vol_size[k] - crop_size[k] + 1)), vol_size[k] - crop_size[k]) for k in range(3)]
    flips = [u[2 * c + 3 + k] > 0.5 for k in range(3)]
    return Draws(shift, scale, offs, flips)


def augment_example(x, y, crop_size, out_ch, draws):
    """device tensors x (h,w,d,c), y (h,w,d,1) -> (x_aug (crop,c), y_onehot (crop,out_ch)) per train.py:17-41"""
    _, var = ops.channel_moments(x)
    return ops.augment_crop(x, y, var, crop_size, draws.offsets, draws.flip_mask, draws.shift, draws.scale, out_ch)


class _Dataset(object):
    """re-iterable epoch of (x, y) device batches.  Data parallel: every rank draws the SAME permutation from
    the shared shuffle generator and keeps positions rank, rank+world, ... of it, truncated to len // world examples so
    that all ranks run the same number of steps (the per-step exchanges would deadlock otherwise); the augmentation draws
    come from a second, per-rank generator."""

    def __init__(self, files, batch_size, prepro_size, crop_size, out_ch, shuffle, seed, device, rank=0, world=1):
        self.files, self.batch_size, self.prepro_size = files, int(batch_size), tuple(prepro_size)
        self.crop_size, self.out_ch, self.shuffle = tuple(crop_size), int(out_ch), shuffle
        self.rank, self.world = int(rank), max(1, int(world))
        self.order_gen = torch.Generator().manual_seed(seed)
        self.gen = torch.Generator().manual_seed(seed + 7919 * (self.rank + 1))
        self.device = device

    def _per_rank(self):
        return len(self.files) // self.world if self.world > 1 else len(self.files)

    def __len__(self):
        return (self._per_rank() + self.batch_size - 1) // self.batch_size

    def state_dict(self):
        """generator states (uint8 tensors): persisted by train.save_checkpoint so a resumed run draws what the
        uninterrupted one would have"""
        return {'order_gen': self.order_gen.get_state().clone(), 'gen': self.gen.get_state().clone()}

    def load_state_dict(self, st):
        self.order_gen.set_state(st['order_gen'].to(torch.uint8).cpu())
        self.gen.set_state(st['gen'].to(torch.uint8).cpu())

    def __iter__(self):
        order = list(range(len(self.files)))
        if self.shuffle:                                                   # train.py:60-61 (buffer = whole file list)
            order = torch.randperm(len(order), generator=self.order_gen).tolist()
        if self.world > 1:
            order = order[self.rank::self.world][:self._per_rank()]
        h, w, d, c = self.prepro_size
        xs, ys = [], []
        for i in order:
            z = np.load(self.files[i])
            x = torch.from_numpy(np.ascontiguousarray(z['x'], dtype=np.float32).reshape(h, w, d, c)).to(self.device)
            y = torch.from_numpy(np.ascontiguousarray(z['y'], dtype=np.float32).reshape(h, w, d, 1)).to(self.device)
            xa, ya = augment_example(x, y, self.crop_size, self.out_ch, draw(self.gen, c, (h, w, d), self.crop_size))
            xs.append(xa)
            ys.append(ya)
            if len(xs) == self.batch_size:
                yield self._emit(xs, ys)
                xs, ys = [], []
        if xs:
            yield self._emit(xs, ys)

    def _emit(self, xs, ys):
        return torch.stack(xs), torch.stack(ys)


class _ChannelsFirstDataset(_Dataset):
    """public NCDHW batches (train.py:45-47 transposes each example); the engine's layers re-lay them out on entry"""

    def _emit(self, xs, ys):
        return torch.stack(xs).permute(0, 4, 1, 2, 3).contiguous(), torch.stack(ys).permute(0, 4, 1, 2, 3).contiguous()


def prepare_dataset(loc, batch_size, prepro_size, crop_size, out_ch, shuffle=True, data_format='channels_last', seed=0,
                    device=None, rank=None, world=None):
    """-> (re-iterable dataset of (x, y) device batches, number of examples)   [train.py:12-64]
    rank / world default to the process group's (one shard of the examples per rank, see _Dataset)."""
    if data_format not in ('channels_last', 'channels_first'):
        raise ValueError('unknown data_format %r' % (data_format,))
    from . import parallel
    rank = parallel.rank() if rank is None else rank
    world = parallel.world() if world is None else world
    files = sorted(os.path.join(loc, f) for f in os.listdir(loc) if f.endswith('.npz'))
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    cls = _Dataset if data_format == 'channels_last' else _ChannelsFirstDataset
    return cls(files, batch_size, prepro_size, crop_size, out_ch, shuffle, seed, dev, rank, world), len(files)
