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
    def __init__(self, files, batch_size, prepro_size, crop_size, out_ch, shuffle, seed, device):
        self.files, self.batch_size, self.prepro_size = files, int(batch_size), tuple(prepro_size)
        self.crop_size, self.out_ch, self.shuffle = tuple(crop_size), int(out_ch), shuffle
        self.gen = torch.Generator().manual_seed(seed)
        self.device = device

    def __len__(self):
        return (len(self.files) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(range(len(self.files)))
        if self.shuffle:                                                   # train.py:60-61 (buffer = whole file list)
            order = torch.randperm(len(order), generator=self.gen).tolist()
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
                yield torch.stack(xs), torch.stack(ys)
                xs, ys = [], []
        if xs:
            yield torch.stack(xs), torch.stack(ys)


def prepare_dataset(loc, batch_size, prepro_size, crop_size, out_ch, shuffle=True, data_format='channels_last', seed=0,
                    device=None):
    """-> (re-iterable dataset of (x, y) device batches, number of examples)   [train.py:12-64]"""
    if data_format != 'channels_last':
        raise NotImplementedError('channels_first public layout is SURVEY 8 f-4 (not built)')
    files = sorted(os.path.join(loc, f) for f in os.listdir(loc) if f.endswith('.npz'))
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    return _Dataset(files, batch_size, prepro_size, crop_size, out_ch, shuffle, seed, dev), len(files)
