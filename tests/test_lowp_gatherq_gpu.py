"""-m gpu: the whole-row stride-2 gather kernel of the 16-bit storage path (lp_conv_gatherq_kernel in csrc/lowp.hip: ConvDownsample's
Conv3D, downsample.py:28-35, below the top level, and -- on role-swapped images -- ConvUpsample's data gradient, upsample.py:28-33 under
train.py:151) in its three register shapes: GK = 4 (four k-steps per load group), GK = 2, and VB = 4 (four position groups per wave
sharing every weight fragment: round 5, measured slower and off by default, kept correct).  Small grids, the size thresholds lowered through BTS_LP_GATHERQ_MIN / BTS_LP_GATHERQ_VB4 so that
each form takes them; against the oracle's op on the same 16-bit-rounded operands in fp64 under |err| <= 8 * 2^-24 * sum|a_i b_i| + u |ref|."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


FORMS = {'gk4': dict(BTS_LP_GATHERQ_MIN='1', BTS_LP_GATHERQ_VB4='0'), 'vb4': dict(BTS_LP_GATHERQ_MIN='1', BTS_LP_GATHERQ_VB4='2'),
         'plain': dict(BTS_LP_GATHERQ='0')}
CASES = [
    # n, (D,H,W) of the input, Cin, Cout
    (1, (32, 32, 64), 64, 64),      # KS = 4, Wg = 32: GK = 4 (or VB = 4 with GK = 2)
    (2, (16, 24, 36), 96, 128),     # KS = 6, Wg = 18: GK = 2; ragged position blocks
    (1, (20, 16, 48), 128, 48),     # partly filled second cout block
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('form', list(FORMS))
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_stride2_forward_in_every_register_shape(case, form, dtype, monkeypatch):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv('BTS_LP_S2T', '0')
    n, (d, h, w), cin, cout = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((d, h, w, cin, cout)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = R.conv3d(xr, wr, b.double(), stride=2)
    bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), wr.abs(), None, stride=2) + U[dtype] * ref.abs() + 1e-30
    slab = torch.full((n, d, h, w, cin + 32), 5.0, dtype=tdt, device=DEV)
    slab[..., 16:16 + cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S2, code, wt.to(DEV), cin, cout)
    ops.profile_enable(True)
    y = lowp.conv(ops.K3S2, code, tdt, slab[..., 16:16 + cin], wp, b.to(DEV), cout)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    assert [s for s, _, _ in ops.profile_records()] == ['lp_conv_gather_kernel']
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s %s: error %.3e is %.2fx the stated bound' % (form, dtype, float(err.max()), worst)


@pytest.mark.parametrize('form', ['gk4', 'vb4'])
def test_transposed_conv_data_gradient_accumulates(form, monkeypatch):
    """dx += conv^T(dy) of ConvUpsample (the stride-2 gather over the FINE gradient) into a slab-gradient view"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv('BTS_LP_S2T', '0')
    dtype = 'bfloat16'
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(3)
    n, d, h, w, cin, cout = 1, 16, 16, 32, 64, 64          # coarse grid; dy lives on the doubled grid
    wt = torch.randn((3, 3, 3, cout, cin), generator=g) * (2.0 / (27 * cin)) ** 0.5      # Conv3DTranspose layout (kd,kh,kw,Cout,Cin)
    dy = torch.randn((n, 2 * d, 2 * h, 2 * w, cout), generator=g)
    old = torch.randn((n, d, h, w, cin), generator=g)
    dyr, wr, oldr = _round(dy, tdt), _round(wt, tdt), _round(old, tdt)
    xs = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d_transpose(xs, wr, None) * dyr).sum().backward()
    ref = xs.grad + oldr
    xa = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    (R.conv3d_transpose(xa, wr.abs(), None) * dyr.abs()).sum().backward()
    bound = 8 * 2.0 ** -24 * xa.grad + u * ref.abs() + u * oldr.abs() + 1e-30
    slab = torch.full((n, d, h, w, cin + 16), 3.0, dtype=tdt, device=DEV)
    dx = slab[..., 8:8 + cin]
    dx.copy_(old.to(tdt).to(DEV))
    wpb = lowp.pack(ops.K3S2T, code, wt.to(DEV), cin, cout, role=ops.ROLE_BWD)
    lowp.conv_bwd_data(ops.K3S2T, code, dy.to(tdt).to(DEV), wpb, dx, True)
    torch.cuda.synchronize()
    err = (dx.double().cpu() - ref.detach()).abs()
    assert float((err / bound).max()) <= 1.0
    assert bool((slab[..., :8] == 3.0).all()) and bool((slab[..., 8 + cin:] == 3.0).all())
