"""-m gpu: the transposed-form kernel of the 16-bit storage path (csrc/lowp_up.hip): ConvUpsample's Conv3DTranspose (upsample.py:28-33)
and the data gradient of ConvDownsample's stride-2 Conv3D (downsample.py:28-35) with all eight output-parity classes in one pass, at
shapes it takes (coarse W >= 12, >= 2048 coarse positions; the small cases of test_lowp_gpu.py keep exercising the per-class gather).
Through the C ABI, against the oracle's op on the same 16-bit-rounded operands in fp64 under
|err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref| (+ u * |old| when accumulating); the launch records must show `lp_up_kernel` only."""
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
    # n, coarse (D,H,W), Cin, Cout, slab_in, slab_out
    (2, (8, 12, 32), 64, 32, False, True),       # 32-cout items (8 rows per tile), four k-steps, slab output
    (1, (6, 9, 40), 32, 64, True, False),        # 64-cout items, ragged on every axis, slab input
    (2, (8, 8, 16), 128, 64, False, False),      # 16-wide coarse tiles (two z planes per fragment)
    (1, (10, 12, 20), 16, 96, False, False),     # one k-step, three cout blocks, 16-wide ragged
    (1, (24, 32, 32), 16, 32, False, False),     # 384 items on 256 workgroups: items chained, request ring crosses items every k-step
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_transposed_conv_forward(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    n, (d, h, w), cin, cout, slab_in, slab_out = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((d, h, w, cin, cout)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cout, cin), generator=g) * (2.0 / (27 * cin / 8)) ** 0.5       # Keras Conv3DTranspose layout (kd,kh,kw,Cout,Cin)
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = R.conv3d_transpose(xr, wr, b.double())
    bound = 8 * 2.0 ** -24 * R.conv3d_transpose(xr.abs(), wr.abs(), None) + U[dtype] * ref.abs() + 1e-30
    ldx = cin + 16 if slab_in else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_in else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S2T, code, wt.to(DEV), cin, cout)
    out = None
    if slab_out:
        buf = torch.full((n, 2 * d, 2 * h, 2 * w, cout + 24), 7.0, dtype=tdt, device=DEV)
        out = buf[..., 8:8 + cout]
    y, syms = _records(lambda: lowp.conv(ops.K3S2T, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout, out=out))
    assert syms == ['lp_up_kernel'], syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    if slab_out:
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_stride2_conv_data_gradient(dtype, accumulate):
    """dx (+)= conv_s2^T(dy) into a slab-gradient view (lowp_train._sampler_bwd accumulates the encoder down-samplers' there)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(31)
    n, d, h, w, cin, cout = 1, 16, 24, 48, 32, 64        # forward input dims; dy lives on the half grid
    dy = torch.randn((n, d // 2, h // 2, w // 2, cout), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    old = torch.randn((n, d, h, w, cin), generator=g)
    dyr, wr, oldr = _round(dy, tdt), _round(wt, tdt), _round(old, tdt)
    xs = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d(xs, wr, None, stride=2) * dyr).sum().backward()
    ref = xs.grad + (oldr if accumulate else 0)
    xa = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d(xa, wr.abs(), None, stride=2) * dyr.abs()).sum().backward()
    bound = 8 * 2.0 ** -24 * xa.grad + u * ref.abs() + (u * oldr.abs() if accumulate else 0) + 1e-30
    slab = torch.full((n, d, h, w, cin + 32), 3.0, dtype=tdt, device=DEV)
    dx = slab[..., 16:16 + cin]
    dx.copy_(old.to(tdt).to(DEV))
    wpb = lowp.pack(ops.K3S2, code, wt.to(DEV), cin, cout, role=ops.ROLE_BWD)
    _, syms = _records(lambda: lowp.conv_bwd_data(ops.K3S2, code, dy.to(tdt).to(DEV), wpb, dx, accumulate))
    assert syms == ['lp_up_kernel'], syms
    err = (dx.double().cpu() - ref.detach()).abs()
    assert float((err / bound).max()) <= 1.0
    assert bool((slab[..., :16] == 3.0).all()) and bool((slab[..., 16 + cin:] == 3.0).all())
