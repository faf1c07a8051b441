"""-m gpu: the streaming 1x1x1 convolution of the 16-bit storage path (csrc/lowp_k1.hip: the shortcut conv of resnet.py:96-103,118-121,
the decoder / VAE projections, and their data gradients) at shapes it takes (>= 4096 positions; the small cases of test_lowp_gpu.py keep
exercising the gather kernel).  Through the C ABI, against the oracle's op on the same 16-bit-rounded operands in fp64 under
|err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref| (+ u * |old| when accumulating); the launch records must show `lp_k1_kernel`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


def _records(fn):
    from bts_amd import ops
    ops.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    ops.profile_enable(False)
    return out, [s for s, _, _ in ops.profile_records()]


CASES = [
    # n, (D,H,W), Cin, Cout, slab_in, slab_out
    (2, (16, 16, 24), 48, 32, True, True),       # three k-steps (the operand ring wraps), slab views
    (1, (16, 16, 16), 16, 96, False, False),     # one k-step, three cout blocks (two launch columns, the second half empty)
    (2, (9, 13, 20), 64, 64, False, True),       # 4680 positions: the last workgroup ragged
    (1, (16, 24, 32), 192, 64, True, False),     # twelve k-steps
    (1, (17, 15, 17), 64, 64, False, True),      # 4335 positions: the last quad of voxels ragged (whole-row loads / stores by quads)
    (1, (16, 16, 17), 128, 96, True, True),      # two groups of four k-steps, the second cout group half empty
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
    wt = torch.randn((1, 1, 1, cin, cout), generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = R.conv3d(xr, wr, b.double())
    bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), wr.abs(), None) + U[dtype] * ref.abs() + 1e-30
    ldx = cin + 16 if slab_in else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_in else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K1, code, wt.to(DEV), cin, cout)
    out = None
    if slab_out:
        buf = torch.full((n, d, h, w, cout + 24), 7.0, dtype=tdt, device=DEV)
        out = buf[..., 8:8 + cout]
    y, syms = _records(lambda: lowp.conv(ops.K1, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout, out=out))
    assert syms == ['lp_k1_kernel'], syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    if slab_out:
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_data_gradient(dtype, accumulate):
    """dx (+)= dres . W^T into a slab-gradient view (lowp_train._block_bwd accumulates the shortcut's data gradient there)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(17)
    n, d, h, w, cin, cout = 2, 12, 16, 24, 64, 32
    dy = torch.randn((n, d, h, w, cout), generator=g)
    wt = torch.randn((1, 1, 1, cin, cout), generator=g) * (2.0 / cin) ** 0.5
    old = torch.randn((n, d, h, w, cin), generator=g)
    dyr, wr, oldr = _round(dy, tdt), _round(wt, tdt), _round(old, tdt)
    w2 = wr.reshape(cin, cout)
    ref = dyr @ w2.t() + (oldr if accumulate else 0)
    bound = 8 * 2.0 ** -24 * (dyr.abs() @ w2.abs().t()) + u * ref.abs() + (u * oldr.abs() if accumulate else 0) + 1e-30
    slab = torch.full((n, d, h, w, cin + 32), 3.0, dtype=tdt, device=DEV)
    dx = slab[..., 16:16 + cin]
    dx.copy_(old.to(tdt).to(DEV))
    wpb = lowp.pack(ops.K1, code, wt.to(DEV), cin, cout, role=ops.ROLE_BWD)
    _, syms = _records(lambda: lowp.conv_bwd_data(ops.K1, code, dy.to(tdt).to(DEV), wpb, dx, accumulate))
    assert syms == ['lp_k1_kernel'], syms
    err = (dx.double().cpu() - ref).abs()
    assert float((err / bound).max()) <= 1.0
    assert bool((slab[..., :16] == 3.0).all()) and bool((slab[..., 16 + cin:] == 3.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 16, 16, 16, 32, 64), (1, 20, 24, 20, 128, 64), (1, 17, 15, 17, 64, 32)],
                         ids=['n2-16x16x16', 'n1-20x24x20-ragged-rows', 'n1-17x15x17-ragged-quad'])
def test_fused_global_average_pool(dtype, shape):
    """bts_lp_conv1_gap: the shortcut conv and the squeeze of its output (resnet.py:118-121) in one pass; a single sample's volume
    need not be whole 256-position blocks (the 20x24x20 level of the full inference volume)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(4)
    n, d, h, w, cin, cout = shape
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((1, 1, 1, cin, cout), generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    ref = R.conv3d(_round(x, tdt), _round(wt, tdt), b.double())
    wp = lowp.pack(ops.K1, code, wt.to(DEV), cin, cout)
    (res, gap), syms = _records(lambda: lowp.conv1_gap(code, x.to(tdt).to(DEV), wp, b.to(DEV), cout, tdt))
    assert 'lp_k1_kernel' in syms and 'lp_conv_gather_kernel' not in syms, syms
    assert float((gap.double().cpu() - ref.mean(dim=(1, 2, 3))).abs().max()) <= 2e-6
    bound = 8 * 2.0 ** -24 * R.conv3d(_round(x, tdt).abs(), _round(wt, tdt).abs(), None) + U[dtype] * ref.abs() + 1e-30
    assert float(((res.double().cpu() - ref).abs() / bound).max()) <= 1.0
