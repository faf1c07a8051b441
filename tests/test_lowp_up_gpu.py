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


WG_CASES = [
    # kind, n, forward-input (D,H,W), Cin, Cout, slab_x
    ('K3S2', 2, (8, 12, 36), 32, 64, True),      # stride-2 conv: P = x (fine, slab view), Q = dy (coarse); ragged coarse tiles
    ('K3S2', 1, (16, 16, 32), 64, 128, False),   # two cp blocks, two cq groups of 64
    ('K3S2T', 2, (6, 8, 20), 64, 32, False),     # transposed conv: P = dy (fine), Q = x (coarse): dW (kd,kh,kw,Cout,Cin)
    ('K3S2T', 1, (8, 8, 16), 128, 64, True),
    ('K3S2', 1, (4, 6, 34), 16, 32, False),      # half-empty P channel block (16 of 32), odd coarse width
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
@pytest.mark.parametrize('case', WG_CASES, ids=lambda c: '%s-n%d-%dx%dx%d-%d-%d' % (c[0], c[1], *c[2], c[3], c[4]))
def test_strided_weight_gradients(case, accumulate, dtype):
    """dW, db of the stride-2 conv and the transposed conv from 16-bit operands (csrc/lowp_wgs.hip: transposing LDS reads) against torch
    autograd of the oracle's op on the same rounded operands in fp64; bound 8 * 2^-24 * sum|a_i b_i| (fp32 sums of exact products)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    name, n, (d, h, w), cin, cout, slab_x = case
    kind = getattr(ops, name)
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((name, d, h, w, cin, cout)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    oshape = (n, d // 2, h // 2, w // 2, cout) if kind == ops.K3S2 else (n, 2 * d, 2 * h, 2 * w, cout)
    dy = torch.randn(oshape, generator=g)
    wshape = (3, 3, 3, cin, cout) if kind == ops.K3S2 else (3, 3, 3, cout, cin)
    xr, dyr = _round(x, tdt), _round(dy, tdt)
    fwd = (lambda xx, ww, bb: R.conv3d(xx, ww, bb, stride=2)) if kind == ops.K3S2 else R.conv3d_transpose
    w0 = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    b0 = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (fwd(xr, w0, b0) * dyr).sum().backward()
    wa = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    (fwd(xr.abs(), wa, None) * dyr.abs()).sum().backward()
    old_w = torch.randn(wshape, generator=g)
    old_b = torch.randn(cout, generator=g)
    ref_w = w0.grad + (old_w.double() if accumulate else 0)
    ref_b = b0.grad + (old_b.double() if accumulate else 0)
    ldx = cin + 16 if slab_x else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_x else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    dw, db = old_w.clone().to(DEV), old_b.clone().to(DEV)
    ok, syms = _records(lambda: lowp.conv_bwd_weight(kind, code, xin[..., c0:c0 + cin], dy.to(tdt).to(DEV), dw, db, accumulate=accumulate))
    assert ok and 'lp_wgs_kernel' in syms, syms
    bound_w = 8 * 2.0 ** -24 * wa.grad + 2.0 ** -22 * ref_w.abs() + 1e-30        # (+ the final fp32 rounding of dw (+ old))
    assert float(((dw.double().cpu() - ref_w).abs() / bound_w).max()) <= 1.0
    assert float((db.double().cpu() - ref_b).abs().max()) <= 1e-4 * max(1.0, float(ref_b.abs().max()))
