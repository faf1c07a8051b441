"""CPU: oracle restatement of the full-volume inference wrapper (reference test.py) against hand-worked
answers, and the host-side helpers of the engine's wrapper that need no kernels."""
import torch

import bts_amd  # noqa: F401
from bts_amd import infer
from oracle import torch_ref as R


def test_padding_appends_res_minus_remainder_even_when_divisible():
    x = torch.arange(5 * 8 * 3 * 2, dtype=torch.float32).reshape(5, 8, 3, 2)
    m = torch.ones(5, 8, 3, 1)
    xp, mp, orig = R.pad_to_spatial_res(8, x, m)            # test.py:166-172: 8 - (s % 8) -> 3, 8, 5
    assert tuple(xp.shape) == (8, 16, 8, 2) and tuple(mp.shape) == (8, 16, 8, 1) and orig == [5, 8, 3]
    assert torch.equal(xp[:5, :8, :3], x) and float(xp[5:].abs().sum() + xp[:, 8:].abs().sum() + xp[:, :, 3:].abs().sum()) == 0
    xe, me, oe = infer.pad_to_spatial_res(8, x, m)          # the engine's host helper is the same arithmetic
    assert torch.equal(xe, xp) and torch.equal(me, mp) and oe == orig


def test_flip_list_is_the_reference_order_and_covers_all_subsets():
    ref = R.tta_augment_axes(True)
    assert ref == [[1, 2, 3], [], [1], [2, 3], [2], [1, 3], [3], [1, 2]]      # test.py:95-103
    bit = {1: 4, 2: 2, 3: 1}
    assert infer.augment_axes(True) == [sum(bit[a] for a in f) for f in ref]
    assert sorted(infer.augment_axes(True)) == list(range(8))
    assert infer.augment_axes(False) == [0] and R.tta_augment_axes(False) == [[]]


def test_label_remap_hand_example():
    y = torch.tensor([[[[0.7, 0.2, 0.1], [0.1, 0.8, 0.3], [0.2, 0.3, 0.9], [0.4, 0.4, 0.1], [0.9, 0.1, 0.1]]]])
    bm = torch.tensor([[[[1.0], [1.0], [1.0], [1.0], [0.0]]]])
    lab = R.tta_labels(y * bm, bm, 0.5)
    # classes 0,1,2 -> 1,2,4 (test.py:259-261); 0.4 < threshold -> 0; masked voxel -> 0
    assert lab.reshape(-1).tolist() == [1, 2, 4, 0, 0]


def test_tta_of_a_flip_equivariant_map_is_the_plain_prediction():
    """with a model that commutes with flips the 8-way mean must equal one forward pass (sanity of flip / un-flip)"""
    orig = R.model
    try:
        R.model = lambda x, P, cfg, training=None, inference=None: (torch.sigmoid(x.sum(-1, keepdim=True) * torch.ones(3)),)
        x = torch.randn(4, 6, 5, 2, dtype=torch.float64)
        bm = (torch.rand(4, 6, 5, 1) > 0.2).double()
        mean, std = torch.tensor([0.1, -0.2], dtype=torch.float64), torch.tensor([1.5, 0.7], dtype=torch.float64)
        y = R.tta_predict(x, bm, None, None, mean, std)
        exp = torch.sigmoid(((x - mean) / std).sum(-1, keepdim=True) * torch.ones(3, dtype=torch.float64)) * bm
        assert torch.allclose(y, exp, atol=1e-12)
    finally:
        R.model = orig
