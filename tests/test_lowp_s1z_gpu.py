"""-m gpu: the z-marching stride-1 3x3x3 convolution of the 16-bit storage path for few channels (csrc/lowp_s1z.hip: Cin 16 | 32,
Cout <= 32, W % 32 == 0, H % 16 == 0 -- the 128^3 level's Conv3D of resnet.py:80-87 / vae.py:92-99 and, on role-swapped images, the
data gradients under train.py:142-151).

Through the C ABI (bts_lp_conv3d_fwd / _bwd_data / _fwd_gn) against the oracle's op on the same 16-bit-rounded operands in fp64 under
|err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref| (+ u * |old| when accumulating), with the library's launch records asserting that
`lp_s1z_kernel` produced the result.  Covered: one and two k-steps, partly filled cout block, several columns and z chunks with a ragged
last chunk, two items per workgroup, slab views on both sides, accumulation, the fused GroupNorm partial sums, BTS_LP_S1Z=0."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


def _ran(fn):
    from bts_amd import ops
    ops.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    ops.profile_enable(False)
    return out, [s for s, _, _ in ops.profile_records()]


CASES = [
    # n, (D,H,W), Cin, Cout, slab_in, slab_out
    (2, (16, 32, 64), 32, 32, False, False),     # two k-steps, 8 columns
    (1, (40, 32, 64), 16, 32, True, True),       # one k-step, slab views, z chunks (40 planes -> 2 x 20)
    (1, (37, 16, 96), 32, 16, False, True),      # half-filled cout block, ragged last z chunk
    (4, (8, 48, 32), 32, 24, True, False),       # cout block three quarters full, 12 columns of 8 planes
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_forward(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    n, (d, h, w), cin, cout, slab_in, slab_out = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((d, h, w, cin, cout)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = R.conv3d(xr, wr, b.double())
    bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), wr.abs(), None) + U[dtype] * ref.abs() + 1e-30
    ldx = cin + 16 if slab_in else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_in else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)
    out = None
    if slab_out:
        buf = torch.full((n, d, h, w, cout + 24), 7.0, dtype=tdt, device=DEV)
        out = buf[..., 8:8 + cout]
    y, syms = _ran(lambda: lowp.conv(ops.K3S1, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout, out=out))
    assert syms == ['lp_s1z_kernel'], syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    if slab_out:
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())
    # the tiled kernel gives the same result up to the summation order
    os.environ['BTS_LP_S1Z'] = '0'
    try:
        y2, syms2 = _ran(lambda: lowp.conv(ops.K3S1, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout))
    finally:
        del os.environ['BTS_LP_S1Z']
    assert 'lp_s1z_kernel' not in syms2
    assert float(((y2.double() - y.double()).cpu().abs() / bound).max()) <= 2.0


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_data_gradient(dtype, accumulate):
    """dx (+)= conv^T(dy) on the flipped-tap image, into a slab-gradient view"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(23)
    n, d, h, w, cin, cout = 2, 16, 32, 64, 32, 32
    dy = torch.randn((n, d, h, w, cout), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    old = torch.randn((n, d, h, w, cin), generator=g)
    dyr, wr, oldr = _round(dy, tdt), _round(wt, tdt), _round(old, tdt)
    xs = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d(xs, wr, None) * dyr).sum().backward()
    ref = xs.grad + (oldr if accumulate else 0)
    xa = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d(xa, wr.abs(), None) * dyr.abs()).sum().backward()
    bound = 8 * 2.0 ** -24 * xa.grad + u * ref.abs() + (u * oldr.abs() if accumulate else 0) + 1e-30
    slab = torch.full((n, d, h, w, cin + 32), 3.0, dtype=tdt, device=DEV)
    dx = slab[..., 16:16 + cin]
    dx.copy_(old.to(tdt).to(DEV))
    wpb = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout, role=ops.ROLE_BWD)
    _, syms = _ran(lambda: lowp.conv_bwd_data(ops.K3S1, code, dy.to(tdt).to(DEV), wpb, dx, accumulate))
    assert syms == ['lp_s1z_kernel'], syms
    err = (dx.double().cpu() - ref.detach()).abs()
    assert float((err / bound).max()) <= 1.0
    assert bool((slab[..., :16] == 3.0).all()) and bool((slab[..., 16 + cin:] == 3.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_fused_groupnorm_statistics(dtype):
    """bts_lp_conv3d_fwd_gn on a shape the streaming kernel takes: y as the plain conv's, mean / rstd as bts_lp_gn_stats of that y"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(5)
    n, d, h, w, cin, cout, groups = 2, 16, 32, 64, 32, 32, 8
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.3).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    norm = GroupNormalization(groups=groups, axis=-1)
    norm.build((None, None, None, None, cout))
    (y, mean, rstd), syms = _ran(lambda: lowp.conv_gn(code, tdt, x, wp, b, cout, norm))
    assert 'lp_s1z_kernel' in syms, syms
    y2 = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
    assert torch.equal(y, y2)
    m2, r2 = lowp.gn_stats(code, y2, groups, norm._mode, norm.epsilon)
    # (the fused sums see the unrounded outputs, the stand-alone pass the stored ones: the difference is the storage rounding's mean)
    assert float((mean - m2).abs().max()) <= 4 * U[dtype] * float(y2.float().abs().mean()) + 1e-6
    assert float((rstd / r2 - 1).abs().max()) <= 4 * U[dtype]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_sixty_four_input_channels_as_two_passes(dtype, accumulate, monkeypatch):
    """Cin = 64 -> Cout <= 32 (the decoder's top block, decoder.py:55-63: 64 -> 32 at 128^3): two marches of the 32-channel kernel over the
    channel halves of the input slab, the second accumulating (BTS_LP_S1Z_PAIR=1 forces the form below its 8 M-voxel threshold).  The
    first pass's result passes through the storage type once: the bound carries one more rounding of the partial result."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    monkeypatch.setenv('BTS_LP_S1Z_PAIR', '1')
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(41)
    n, d, h, w, cin, cout = 2, 12, 32, 64, 64, 24
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    old = torch.randn((n, d, h, w, cout), generator=g)
    xr, wr, oldr = _round(x, tdt), _round(wt, tdt), _round(old, tdt)
    start = oldr if accumulate else b.double()           # what the first pass adds its half of the contraction to
    first = R.conv3d(xr[..., :32], wr[:, :, :, :32], None) + start          # ... and stores, rounded to the storage type
    ref = R.conv3d(xr, wr, None) + start
    # fp32 sums of the two passes + the two storage roundings (of the first pass's result and of the final one)
    bound = 8 * 2.0 ** -24 * (R.conv3d(xr.abs(), wr.abs(), None) + start.abs()) + 1.01 * u * (ref.abs() + first.abs()) + 1e-30
    slab = torch.zeros((n, d, h, w, cin + 16), dtype=tdt, device=DEV)
    slab[..., 8:8 + cin] = x.to(tdt).to(DEV)
    xin = slab[..., 8:8 + cin]
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)
    if accumulate:      # through the data-gradient entry point (it is the one with an accumulate flag): the same image, forward role
        y = old.to(tdt).to(DEV).clone()
        wt_b = wt.permute(0, 1, 2, 4, 3).flip(0, 1, 2).contiguous()      # conv^T with this kernel == the forward conv with wt
        wpb = lowp.pack(ops.K3S1, code, wt_b.to(DEV), cout, cin, role=ops.ROLE_BWD)
        _, syms = _ran(lambda: lowp.conv_bwd_data(ops.K3S1, code, xin, wpb, y, True))
    else:
        y, syms = _ran(lambda: lowp.conv(ops.K3S1, code, tdt, xin, wp, b.to(DEV), cout))
    assert syms == ['lp_s1z_kernel', 'lp_s1z_kernel'], syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    # and the tiled kernel agrees up to the roundings
    monkeypatch.setenv('BTS_LP_S1Z_PAIR', '0')
    if not accumulate:
        y2, syms2 = _ran(lambda: lowp.conv(ops.K3S1, code, tdt, xin, wp, b.to(DEV), cout))
        assert 'lp_s1z_kernel' not in syms2
        assert float(((y2.double() - y.double()).cpu().abs() / bound).max()) <= 2.0


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_two_pass_form_with_fused_groupnorm_statistics(dtype, monkeypatch):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    monkeypatch.setenv('BTS_LP_S1Z_PAIR', '1')
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(6)
    n, d, h, w, cin, cout, groups = 2, 16, 32, 64, 64, 32, 8      # (8 columns x 16 planes: above the streaming kernel's 96-plane floor)
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5).to(DEV)
    b = (torch.randn(cout, generator=g) * 0.3).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    norm = GroupNormalization(groups=groups, axis=-1)
    norm.build((None, None, None, None, cout))
    (y, mean, rstd), syms = _ran(lambda: lowp.conv_gn(code, tdt, x, wp, b, cout, norm))
    assert syms.count('lp_s1z_kernel') == 2, syms
    m2, r2 = lowp.gn_stats(code, y, groups, norm._mode, norm.epsilon)
    assert float((mean - m2).abs().max()) <= 4 * U[dtype] * float(y.float().abs().mean()) + 1e-6
    assert float((rstd / r2 - 1).abs().max()) <= 4 * U[dtype]
    ref = R.conv3d(x.double().cpu(), wt.to(tdt).double().cpu(), b.double().cpu())
    assert float((y.double().cpu() - ref).abs().max()) <= 4 * U[dtype] * float(ref.abs().max())
