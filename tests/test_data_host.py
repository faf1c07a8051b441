"""CPU: oracle restatement of the reference's per-example augmentation (train.py:14-49) on a hand example,
and the host-side draw logic of the engine's pipeline."""
import torch

import bts_amd  # noqa: F401
from bts_amd import data
from oracle import torch_ref as R


def test_augment_hand_example():
    x = torch.arange(2 * 2 * 3 * 1, dtype=torch.float64).reshape(2, 2, 3, 1)          # values 0..11, variance 143/12
    y = torch.tensor([[[0., 1., 2.], [3., 1., 0.]], [[2., 2., 1.], [0., 3., 3.]]]).reshape(2, 2, 3, 1)
    sig = (143.0 / 12.0) ** 0.5
    xa, ya = R.augment_example(x, y, (1, 2, 2), 3, shift=[0.1], scale=[1.1], offsets=[1, 0, 1], flips=[False, True, False])
    # window rows z=1, y=0..1, x=1..2 -> values [[7,8],[10,11]], flipped along axis 1 -> [[10,11],[7,8]]
    exp = (torch.tensor([[[10., 11.], [7., 8.]]], dtype=torch.float64) + 0.1 * sig) * 1.1
    assert torch.allclose(xa[..., 0], exp, atol=1e-12)
    lab = torch.tensor([[[3, 3], [2, 1]]])
    assert torch.equal(ya.argmax(-1) + 1, lab) and torch.equal(ya.sum(-1), torch.ones(1, 2, 2, dtype=torch.float64))
    # background (label 0) becomes the all-zero vector: one_hot(out_ch + 1) minus channel 0 (train.py:39-40)
    _, y0 = R.augment_example(x, y, (2, 2, 3), 3, [0.0], [1.0], [0, 0, 0], [False] * 3)
    assert float(y0[0, 0, 0].sum()) == 0.0 and float(y0[0, 1, 0, 2]) == 1.0


def test_draw_ranges_and_reproducibility():
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    seen_flip, seen_off = set(), set()
    for _ in range(200):
        d = data.draw(g1, 2, (155, 190, 147), (128, 128, 128))
        e = data.draw(g2, 2, (155, 190, 147), (128, 128, 128))
        assert d.shift == e.shift and d.scale == e.scale and d.offsets == e.offsets and d.flips == e.flips
        assert all(-0.1 <= s <= 0.1 for s in d.shift) and all(0.9 <= s <= 1.1 for s in d.scale)   # train.py:19-20
        assert all(0 <= o <= v - c for o, v, c in zip(d.offsets, (155, 190, 147), (128, 128, 128)))
        seen_flip.add(d.flip_mask)
        seen_off.add(d.offsets[2])
    assert seen_flip == set(range(8)) and min(seen_off) == 0 and max(seen_off) == 19
    d = data.draw(g1, 2, (8, 8, 8), (8, 8, 8))
    assert d.offsets == [0, 0, 0]
