"""-m gpu: the Winograd form of the stride-1 3x3x3 weight gradient (wgw_kernel in csrc/conv_wgrad.hip) against the fp64 oracle,
through bts_conv3d_bwd_weight.  The launcher uses it for whole 16x4x2 sub-tiles and 32-channel groups; every test asserts
through the profiler that `wgw_kernel` really ran, and repeats the call on the direct kernel (BTS_WGW=0).

Tolerance: the contraction bound of SURVEY 8c, |err| <= 8 * eps32 * sum|a_i b_i| + 1e-7 -- the contraction runs over all
voxels, so the bound grows with K while the transforms' extra rounding does not; no widening is needed."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

EPS32 = 2.0 ** -24


def dev():
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def check(got, ref, bound, what):
    err = (got.double().cpu() - ref).abs()
    tol = 8 * EPS32 * bound + 1e-7
    bad = err > tol
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e (tol there %.3e), max|ref| %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(tol.flatten()[err.argmax()]), float(ref.abs().max()))


def kernels_of(fn):
    from bts_amd import ops
    ops.profile_enable(True)
    fn()
    torch.cuda.synchronize()
    names = [r[0] for r in ops.profile_records()]
    ops.profile_enable(False)
    return names


# N, (D,H,W), Cin, Cout: one / several channel groups on either side, N = 2, image borders in every tile, long x rows
CASES = [(1, (4, 8, 32), 32, 32), (2, (2, 4, 16), 64, 32), (1, (6, 12, 48), 32, 64), (1, (8, 8, 16), 96, 64), (1, (16, 16, 64), 32, 32)]


@pytest.mark.parametrize('n,dims,cin,cout', CASES)
def test_wgw_weight_and_bias_gradient(n, dims, cin, cout):
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 1)
    dy = rnd((n, d, h, w, cout), 2)
    wd = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    bd = torch.zeros((cout,), dtype=torch.float64, requires_grad=True)
    (R.conv3d(x.double(), wd, bd) * dy.double()).sum().backward()
    wa = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    (R.conv3d(x.double().abs(), wa, None) * dy.double().abs()).sum().backward()
    bb = dy.double().abs().sum(dim=(0, 1, 2, 3))
    xg, dyg = x.to(dev()), dy.to(dev())
    dw = torch.empty((3, 3, 3, cin, cout), device=dev())
    db = torch.empty((cout,), device=dev())
    names = kernels_of(lambda: ops.conv_bwd_weight(ops.K3S1, xg, dyg, dw, db))
    assert 'wgw_kernel' in names, names
    check(dw, wd.grad, wa.grad, 'wgw dW')
    check(db, bd.grad, bb, 'wgw db')
    # accumulate on top, then the same call on the direct kernel
    ops.conv_bwd_weight(ops.K3S1, xg, dyg, dw, db, accumulate=True)
    check(dw, 2 * wd.grad, 2 * wa.grad, 'wgw dW accumulate')
    os.environ['BTS_WGW'] = '0'
    try:
        dw0 = torch.empty_like(dw)
        names0 = kernels_of(lambda: ops.conv_bwd_weight(ops.K3S1, xg, dyg, dw0, None))
    finally:
        del os.environ['BTS_WGW']
    assert 'wgw_kernel' not in names0
    check(dw0, wd.grad, wa.grad, 'direct dW')


def test_wgw_slab_views_and_folded_duplicate_slice():
    """x and dy as channel slices of wider slabs (ld > C) and the encoder's duplicated input slice (encoder.py:83-87): the
    gradient of the folded weights is scattered back to both copies by the shared finalize kernel"""
    from bts_amd import ops
    n, d, h, w, f, j = 1, 4, 8, 16, 32, 2
    cin_slab, cin_ref, cout = j * f, (j + 1) * f, 32
    slab = rnd((n, d, h, w, cin_slab + 32), 8)
    dslab = rnd((n, d, h, w, cout + 16), 9)
    xs = slab[..., :cin_slab].double()
    dyd = dslab[..., 16:].double()
    wd = torch.zeros((3, 3, 3, cin_ref, cout), dtype=torch.float64, requires_grad=True)
    xcat = torch.cat([xs[..., (j - 1) * f:], xs], dim=-1)
    (R.conv3d(xcat, wd, None) * dyd).sum().backward()
    wa = torch.zeros((3, 3, 3, cin_ref, cout), dtype=torch.float64, requires_grad=True)
    (R.conv3d(xcat.abs(), wa, None) * dyd.abs()).sum().backward()
    sg, dg = slab.to(dev()), dslab.to(dev())
    dw = torch.empty((3, 3, 3, cin_ref, cout), device=dev())
    names = kernels_of(lambda: ops.conv_bwd_weight(ops.K3S1, sg[..., :cin_slab], dg[..., 16:], dw, None, (j - 1) * f, f))
    assert 'wgw_kernel' in names, names
    check(dw, wd.grad, wa.grad, 'wgw folded, strided dW')


def test_wgw_is_deterministic_and_declines_ragged_shapes():
    from bts_amd import ops
    x, dy = rnd((1, 8, 8, 32, 32), 11).to(dev()), rnd((1, 8, 8, 32, 32), 12).to(dev())
    a, b = torch.empty((3, 3, 3, 32, 32), device=dev()), torch.empty((3, 3, 3, 32, 32), device=dev())
    ops.conv_bwd_weight(ops.K3S1, x, dy, a, None)
    ops.conv_bwd_weight(ops.K3S1, x, dy, b, None)
    assert torch.equal(a, b)
    # W not a multiple of 16 / channels not a multiple of 32: the general kernels take the call
    for shp, cin, cout in (((1, 4, 8, 24), 32, 32), ((1, 4, 8, 32), 24, 32)):
        xs, ds = rnd(shp + (cin,), 13).to(dev()), rnd(shp + (cout,), 14).to(dev())
        dw = torch.empty((3, 3, 3, cin, cout), device=dev())
        assert 'wgw_kernel' not in kernels_of(lambda: ops.conv_bwd_weight(ops.K3S1, xs, ds, dw, None))


def test_wgw_full_size_agrees_with_direct_form():
    """BASELINE size (32 -> 32 channels on 128^3): too big for the fp64 oracle, so the Winograd weight gradient is held against
    the engine's own direct kernel (oracle-checked at small sizes) within the sum of both bounds; the bound sum|a_i b_i| is
    the direct kernel's result on |x|, |dy|.  Plus linearity in dy."""
    from bts_amd import ops
    x = rnd((1, 128, 128, 128, 32), 31).to(dev())
    dy = rnd((1, 128, 128, 128, 32), 32).to(dev())
    dy2 = rnd((1, 128, 128, 128, 32), 33).to(dev())
    shape = (3, 3, 3, 32, 32)
    dw, dwb, dws = (torch.empty(shape, device=dev()) for _ in range(3))
    db = torch.empty((32,), device=dev())
    names = kernels_of(lambda: (ops.conv_bwd_weight(ops.K3S1, x, dy, dw, db), ops.conv_bwd_weight(ops.K3S1, x, dy2, dwb, None),
                                ops.conv_bwd_weight(ops.K3S1, x, dy + dy2, dws, None)))
    assert names.count('wgw_kernel') == 3, names
    os.environ['BTS_WGW'] = '0'
    try:
        dw0, bound, bound2 = (torch.empty(shape, device=dev()) for _ in range(3))
        ops.conv_bwd_weight(ops.K3S1, x, dy, dw0, None)
        ops.conv_bwd_weight(ops.K3S1, x.abs(), dy.abs(), bound, None)
        ops.conv_bwd_weight(ops.K3S1, x.abs(), dy2.abs(), bound2, None)
    finally:
        del os.environ['BTS_WGW']
    err = (dw - dw0).abs()
    assert not bool((err > 16 * EPS32 * bound + 1e-6).any()), 'max err %.3e' % float(err.max())
    lin = (dws - (dw + dwb)).abs()
    assert not bool((lin > 32 * EPS32 * (bound + bound2) + 1e-5).any()), 'linearity: max err %.3e' % float(lin.max())
    check(db, dy.double().cpu().sum(dim=(0, 1, 2, 3)), dy.double().cpu().abs().sum(dim=(0, 1, 2, 3)), 'full-size bias gradient')


@pytest.mark.parametrize('n,dims,cin,cout', [(1, (32, 32, 32), 32, 32), (2, (16, 32, 32), 48, 80), (1, (33, 31, 35), 64, 24)])
def test_k1w_streaming_pointwise_weight_gradient(n, dims, cin, cout):
    """k1w_kernel: weight and bias gradient of the 1x1x1 convs on big grids, operands streamed from global memory (no LDS
    staging); channel counts that are not multiples of 32, odd voxel counts, channel slices of wider slabs"""
    from bts_amd import ops
    d, h, w = dims
    xs = rnd((n, d, h, w, cin + 8), 41)
    ds = rnd((n, d, h, w, cout + 4), 42)
    x, dy = xs[..., 8:], ds[..., :cout]
    wd = torch.zeros((1, 1, 1, cin, cout), dtype=torch.float64, requires_grad=True)
    bd = torch.zeros((cout,), dtype=torch.float64, requires_grad=True)
    (R.conv3d(x.double(), wd, bd) * dy.double()).sum().backward()
    bound = torch.einsum('ndhwc,ndhwk->ck', x.double().abs(), dy.double().abs()).reshape(1, 1, 1, cin, cout)
    xg, dg = xs.to(dev())[..., 8:], ds.to(dev())[..., :cout]
    dw = torch.empty((1, 1, 1, cin, cout), device=dev())
    db = torch.empty((cout,), device=dev())
    names = kernels_of(lambda: ops.conv_bwd_weight(ops.K1, xg, dg, dw, db))
    assert 'k1w_kernel' in names, names
    check(dw, wd.grad, bound, 'k1w dW')
    check(db, bd.grad, dy.double().abs().sum(dim=(0, 1, 2, 3)), 'k1w db')
    os.environ['BTS_K1W'] = '0'
    try:
        dw0 = torch.empty_like(dw)
        assert 'k1w_kernel' not in kernels_of(lambda: ops.conv_bwd_weight(ops.K1, xg, dg, dw0, None))
    finally:
        del os.environ['BTS_K1W']
    check(dw0, wd.grad, bound, 'staged dW')
