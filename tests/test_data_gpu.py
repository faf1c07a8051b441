"""-m gpu: device-side augmentation against the oracle restatement of train.py:14-49 with identical draws;
the .npz-backed prepare_dataset yields reference-shaped, reproducible batches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402


def dev():
    return torch.device('cuda', 0)


@pytest.mark.parametrize('vol,crop,c,out_ch', [((9, 10, 11), (8, 8, 8), 2, 3), ((16, 12, 20), (16, 8, 16), 4, 1), ((5, 5, 5), (5, 5, 5), 1, 3)])
def test_augment_matches_oracle(vol, crop, c, out_ch):
    import bts_amd  # noqa: F401
    from bts_amd import data, ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(vol + (c,), generator=g) * 30 + 50
    y = torch.randint(0, out_ch + 1, vol + (1,), generator=g).float()
    mean, var = ops.channel_moments(x.to(dev()))
    assert torch.allclose(mean.cpu().double(), x.double().mean(dim=(0, 1, 2)), rtol=1e-6)
    assert torch.allclose(var.cpu().double(), x.double().var(dim=(0, 1, 2), unbiased=False), rtol=1e-6)
    gen = torch.Generator().manual_seed(2)
    for _ in range(6):
        d = data.draw(gen, c, vol, crop)
        xr, yr = R.augment_example(x.double(), y.double(), crop, out_ch, d.shift, d.scale, d.offsets, d.flips)
        xa, ya = data.augment_example(x.to(dev()), y.to(dev()), crop, out_ch, d)
        assert tuple(xa.shape) == crop + (c,) and tuple(ya.shape) == crop + (out_ch,)
        assert torch.equal(ya.cpu().double(), yr), 'one-hot labels must be exact'
        err = (xa.cpu().double() - xr).abs().max()
        assert float(err) <= 1e-5 * float(xr.abs().max()), 'augmented intensities: %.3e' % float(err)


def test_prepare_dataset_shapes_and_reproducibility(tmp_path):
    import bts_amd  # noqa: F401
    from bts_amd import data
    rs = np.random.RandomState(0)
    size, crop = (10, 12, 9, 2), (8, 8, 8)
    for i in range(5):
        np.savez(os.path.join(str(tmp_path), 'ex%d.npz' % i), x=rs.randn(*size).astype(np.float32),
                 y=rs.randint(0, 4, size[:3] + (1,)).astype(np.float32))
    ds, n = data.prepare_dataset(str(tmp_path), 2, size, list(crop), 3, shuffle=True, seed=7, device=dev())
    assert n == 5 and len(ds) == 3
    batches = list(ds)
    assert [tuple(b[0].shape) for b in batches] == [(2,) + crop + (2,), (2,) + crop + (2,), (1,) + crop + (2,)]
    assert all(tuple(b[1].shape)[-1] == 3 and float(b[1].max()) <= 1.0 for b in batches)
    ds2, _ = data.prepare_dataset(str(tmp_path), 2, size, list(crop), 3, shuffle=True, seed=7, device=dev())
    for (xa, ya), (xb, yb) in zip(batches, list(ds2)):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
