"""-m gpu: the two Winograd forms of the 3x3x3 stride-1 convolution -- F(2x2x2,3x3x3) (csrc/conv_wino3.hip, the default) and
F(2x2,3x3) x direct (csrc/conv_wino.hip, BTS_W3=0) -- against the fp64 oracle, through the same C-ABI entry points as the
implicit GEMM (the launcher picks the form; BTS_WINO_MIN_WGS=1 makes it take the small grids used here, and every test asserts
through the profiler that the form under test really ran).  Every test runs once per form.

Tolerance: the input transform adds fp32 values before the multiply (||B^T||_1 = 2 per transformed axis), so the contraction
bound of SURVEY 8c is widened:  |err| <= 32 * eps32 * sum|a_i b_i| + 1e-7  (direct form: 8) -- for BOTH forms: the third
transformed axis measured 1.35x the two-axis error, well inside the same bound."""
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


def check_wino(got, ref, bound, what):
    err = (got.double().cpu() - ref).abs()
    tol = 32 * EPS32 * bound + 1e-7
    bad = err > tol
    assert not bad.any(), '%s: %d/%d elements out of tolerance, max err %.3e (tol there %.3e), max|ref| %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(tol.flatten()[err.argmax()]), float(ref.abs().max()))


def check_close(got, ref, what, rtol, atol):
    err = (got.double().cpu() - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e' % (what, int(bad.sum()), bad.numel(), float(err.max()))


FORM = {'kernel': 'w3_kernel'}   # set per test by the _form fixture


class ran_wino(object):
    """context: asserts that at least `n` launches of the Winograd kernel under test happened inside"""

    def __init__(self, n=1):
        self.n = n

    def __enter__(self):
        from bts_amd import ops
        ops.profile_enable(True)
        return self

    def __exit__(self, *exc):
        from bts_amd import ops
        torch.cuda.synchronize()
        names = [r[0] for r in ops.profile_records()]
        ops.profile_enable(False)
        if exc[0] is None:
            assert names.count(FORM['kernel']) >= self.n, 'expected %s, profiler saw %r' % (FORM['kernel'], names)
        return False


@pytest.fixture(autouse=True, params=['w3', 'wino'])
def _form(request, monkeypatch):
    monkeypatch.setenv('BTS_WINO_MIN_WGS', '1')
    monkeypatch.delenv('BTS_WINO', raising=False)
    if request.param == 'w3':
        monkeypatch.delenv('BTS_W3', raising=False)
        FORM['kernel'] = 'w3_kernel'
    else:
        monkeypatch.setenv('BTS_W3', '0')
        FORM['kernel'] = 'wino_kernel'


CASES = [
    # N, (D,H,W), Cin, Cout  -- aligned, ragged in every axis (D=6, H=10, W=40), several cout blocks, cout < 32, deep cin
    (1, (8, 8, 32), 16, 32), (2, (6, 10, 40), 8, 48), (1, (16, 16, 64), 64, 64), (1, (4, 4, 32), 24, 16),
    (1, (5, 7, 33), 32, 20), (1, (8, 4, 96), 128, 32),
    # 16-wide tiles (two y patches per wave), ragged; long contractions on small grids (split along the input channels)
    (1, (8, 8, 16), 32, 32), (1, (5, 9, 13), 64, 32), (1, (16, 16, 16), 256, 64), (1, (8, 16, 48), 128, 48), (2, (8, 8, 16), 128, 32),
]


@pytest.mark.parametrize('n,dims,cin,cout', CASES)
def test_wino_fwd_and_data_gradient(n, dims, cin, cout):
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 1)
    wt = rnd((3, 3, 3, cin, cout), 2, 0.2)
    b = rnd((cout,), 3)
    xd = x.double().requires_grad_(True)
    ref = R.conv3d(xd, wt.double(), b.double())
    bound = R.conv3d(x.double().abs(), wt.double().abs(), b.double().abs())
    xg, wg, bg = x.to(dev()), wt.to(dev()), b.to(dev())
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg, cin, cout)
    with ran_wino():
        y = ops.conv_fwd(ops.K3S1, xg, wp, bg, cout)
    check_wino(y, ref.detach(), bound, 'wino fwd')
    # the same call on the implicit GEMM: both forms agree within the sum of their bounds
    import os
    os.environ['BTS_WINO'] = '0'
    try:
        y0 = ops.conv_fwd(ops.K3S1, xg, wp, bg, cout)
    finally:
        del os.environ['BTS_WINO']
    check_wino(y, y0.double().cpu(), 1.25 * bound, 'wino vs implicit GEMM')
    dy = rnd(tuple(ref.shape), 4)
    ref.backward(dy.double())
    wpb = ops.conv_pack(ops.K3S1, ops.ROLE_BWD, wg, cin, cout)
    dx = torch.empty_like(xg)
    xa = x.double().abs().requires_grad_(True)
    dxb = torch.autograd.grad(R.conv3d(xa, wt.double().abs(), None), xa, dy.double().abs())[0]
    if cout % 8 == 0 and cin % 4 == 0 and cin >= 16:   # data gradient: contraction over cout, columns = cin
        with ran_wino(2):
            ops.conv_bwd_data(ops.K3S1, dy.to(dev()), wpb, dx, accumulate=False)
            check_wino(dx, xd.grad, dxb, 'wino bwd_data')
            ops.conv_bwd_data(ops.K3S1, dy.to(dev()), wpb, dx, accumulate=True)
        check_wino(dx, 2 * xd.grad, 2 * dxb, 'wino bwd_data accumulate')
    else:  # ineligible channel counts fall back to the implicit GEMM, same results
        ops.conv_bwd_data(ops.K3S1, dy.to(dev()), wpb, dx, accumulate=False)
        check_wino(dx, xd.grad, dxb, 'bwd_data (declined)')


def test_wino_declines_what_it_cannot_do(monkeypatch):
    """narrow grids, odd channel counts and the sigmoid epilogue stay on the implicit GEMM; BTS_WINO=0 turns the form off"""
    from bts_amd import ops
    for (dims, cin, cout, sig, env) in [((8, 8, 8), 16, 32, False, None), ((8, 8, 32), 12, 32, False, None),
                                        ((8, 8, 32), 16, 32, True, None), ((8, 8, 32), 16, 32, False, '0')]:
        if env is not None:
            monkeypatch.setenv('BTS_WINO', env)
        d, h, w = dims
        x = rnd((1, d, h, w, cin), 5)
        wt = rnd((3, 3, 3, cin, cout), 6, 0.2)
        wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.to(dev()), cin, cout)
        ops.profile_enable(True)
        y = ops.conv_fwd(ops.K3S1, x.to(dev()), wp, None, cout, sigmoid=sig)
        torch.cuda.synchronize()
        names = [r[0] for r in ops.profile_records()]
        ops.profile_enable(False)
        assert 'wino_kernel' not in names and 'w3_kernel' not in names
        ref = R.conv3d(x.double(), wt.double(), None)
        if sig:
            ref = torch.sigmoid(ref)
        check_close(y, ref, 'declined case', rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('n,dims,cin,cout,g', [(1, (16, 16, 32), 32, 32, 4), (2, (32, 8, 64), 16, 64, 8), (1, (6, 10, 40), 8, 16, 2),
                                               (1, (16, 16, 16), 32, 32, 4), (1, (16, 16, 16), 128, 32, 2)])
def test_wino_fused_groupnorm_statistics(n, dims, cin, cout, g):
    """conv + slab-mode GroupNorm statistics from the Winograd epilogue (groups of whole 4-plane tiles) or, for the ragged
    case, from the library's own statistics pass over the Winograd output"""
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 61)
    wt = rnd((3, 3, 3, cin, cout), 62, 0.2)
    b = rnd((cout,), 63)
    ref = R.conv3d(x.double(), wt.double(), b.double())
    flat = ref.reshape(n, g, -1)
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.to(dev()), cin, cout)
    with ran_wino():
        y, mean, rstd = ops.conv_fwd_gn(ops.K3S1, x.to(dev()), wp, b.to(dev()), cout, g, 1e-5)
    check_wino(y, ref, R.conv3d(x.double().abs(), wt.double().abs(), b.double().abs()), 'wino conv_fwd_gn y')
    check_close(mean, flat.mean(dim=2).reshape(-1), 'GN mean', rtol=2e-5, atol=2e-5)
    check_close(rstd, (1.0 / torch.sqrt(flat.var(dim=2, unbiased=False) + 1e-5)).reshape(-1), 'GN rstd', rtol=2e-5, atol=2e-5)


def test_wino_shortcut_pair_and_gradient_pair():
    """the fused entry points (conv3x3x3 + conv1x1x1 of the same input; data gradient of both) with the 3x3x3 part in
    Winograd form and the 1x1x1 part as its own launch"""
    from bts_amd import ops
    n, (d, h, w), cin, cout, g = 1, (8, 8, 32), 32, 32, 4
    x = rnd((n, d, h, w, cin), 71)
    w3, b3 = rnd((3, 3, 3, cin, cout), 72, 0.2), rnd((cout,), 73)
    w1, b1 = rnd((1, 1, 1, cin, cout), 74, 0.2), rnd((cout,), 75)
    wp3 = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, w3.to(dev()), cin, cout)
    wp1 = ops.conv_pack(ops.K1, ops.ROLE_FWD, w1.to(dev()), cin, cout)
    r3 = R.conv3d(x.double(), w3.double(), b3.double())
    r1 = R.conv3d(x.double(), w1.double(), b1.double())
    b3d = R.conv3d(x.double().abs(), w3.double().abs(), b3.double().abs())
    b1d = R.conv3d(x.double().abs(), w1.double().abs(), b1.double().abs())
    with ran_wino(2):
        c1, res, mean, rstd = ops.conv_fwd_fused2_gn(x.to(dev()), wp3, b3.to(dev()), wp1, b1.to(dev()), cout, g, 1e-5)
        out = ops.conv_fwd_fused2(x.to(dev()), wp3, b3.to(dev()), wp1, b1.to(dev()), cout)
    check_wino(c1, r3, b3d, 'fused2_gn conv3')
    check_wino(res, r1, b1d, 'fused2_gn conv1')
    flat = r3.reshape(n, g, -1)
    check_close(mean, flat.mean(dim=2).reshape(-1), 'fused2 GN mean', rtol=2e-5, atol=2e-5)
    if out is not None:
        check_wino(out[0], r3, b3d, 'fused2 conv3')
        check_wino(out[1], r1, b1d, 'fused2 conv1')
    # gradient pair into a slab slice, then accumulated
    xq = x.double().requires_grad_(True)
    dy, dy2 = rnd((n, d, h, w, cout), 84), rnd((n, d, h, w, cout), 85)
    ((R.conv3d(xq, w3.double(), None) * dy.double()).sum() + (R.conv3d(xq, w1.double(), None) * dy2.double()).sum()).backward()
    xa = x.double().abs().requires_grad_(True)
    ((R.conv3d(xa, w3.double().abs(), None) * dy.double().abs()).sum() + (R.conv3d(xa, w1.double().abs(), None) * dy2.double().abs()).sum()).backward()
    wpb3 = ops.conv_pack(ops.K3S1, ops.ROLE_BWD, w3.to(dev()), cin, cout)
    wpb1 = ops.conv_pack(ops.K1, ops.ROLE_BWD, w1.to(dev()), cin, cout)
    pad = 16
    slab = torch.full((n, d, h, w, cin + pad), 3.0, device=dev())
    dx = slab[..., pad:]
    with ran_wino(2):
        ops.conv_bwd_data_pair(dy.to(dev()), wpb3, dy2.to(dev()), wpb1, dx, False)
        check_wino(dx, xq.grad, xa.grad, 'gradient pair')
        assert float((slab[..., :pad] - 3.0).abs().max()) == 0.0
        ops.conv_bwd_data_pair(dy.to(dev()), wpb3, dy2.to(dev()), wpb1, dx, True)
    check_wino(dx, 2 * xq.grad, 2 * xa.grad, 'gradient pair accumulate')


def test_wino_slab_views_and_folded_duplicate_slice():
    """channel slices of a level slab as input and output (virtual Concatenate) and the encoder's duplicated input slice
    folded into the Winograd-domain weights (encoder.py:83-87)"""
    from bts_amd import ops
    n, d, h, w, f, j = 1, 8, 8, 32, 16, 2
    cin_slab, cin_ref, cout = j * f, (j + 1) * f, 16
    slab = rnd((n, d, h, w, cin_slab + cout), 8)
    wt = rnd((3, 3, 3, cin_ref, cout), 9, 0.2)
    xs = slab[..., :cin_slab].double()
    xcat = torch.cat([xs[..., (j - 1) * f:], xs], dim=-1)
    ref = R.conv3d(xcat, wt.double(), None)
    bound = R.conv3d(xcat.abs(), wt.double().abs(), None)
    sg = slab.to(dev())
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.to(dev()), cin_ref, cout, cin_slab, (j - 1) * f, f)
    with ran_wino():
        ops.conv_fwd(ops.K3S1, sg[..., :cin_slab], wp, None, cout, out=sg[..., cin_slab:])
    check_wino(sg[..., cin_slab:], ref, 2 * bound, 'folded, strided fwd')
    check_close(sg[..., :cin_slab], slab[..., :cin_slab].double(), 'input slice untouched', rtol=0, atol=0)
    dy = rnd(tuple(ref.shape), 10)
    xq = xs.clone().requires_grad_(True)
    xc = torch.cat([xq[..., (j - 1) * f:], xq], dim=-1)
    (R.conv3d(xc, wt.double(), None) * dy.double()).sum().backward()
    wpb = ops.conv_pack(ops.K3S1, ops.ROLE_BWD, wt.to(dev()), cin_ref, cout, cin_slab, (j - 1) * f, f)
    dx = torch.empty((n, d, h, w, cin_slab), device=dev())
    with ran_wino():
        ops.conv_bwd_data(ops.K3S1, dy.to(dev()), wpb, dx, False)
    check_close(dx, xq.grad, 'folded bwd_data', rtol=1e-4, atol=1e-4)


def test_wino_is_deterministic():
    from bts_amd import ops
    x = rnd((1, 8, 8, 64, 32), 11).to(dev())
    wt = rnd((3, 3, 3, 32, 32), 12, 0.2).to(dev())
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt, 32, 32)
    with ran_wino(2):
        a = ops.conv_fwd(ops.K3S1, x, wp, None, 32).clone()
        b = ops.conv_fwd(ops.K3S1, x, wp, None, 32)
    assert torch.equal(a, b)


def test_wino_full_size_agrees_with_direct_form():
    """BASELINE size (32 -> 32 channels on 128^3, the layer the metric is dominated by): too big for the fp64 oracle, so the
    Winograd form is held against the engine's own direct-form kernel (itself oracle-checked at small sizes) with the sum of
    both bounds, the bound sum|a_i b_i| being evaluated by the direct kernel on |x|, |w|; plus linearity in x."""
    import os
    from bts_amd import ops
    x = rnd((1, 128, 128, 128, 32), 21).to(dev())
    x2 = rnd((1, 128, 128, 128, 32), 22).to(dev())
    wt = rnd((3, 3, 3, 32, 32), 23, 0.1).to(dev())
    b = rnd((32,), 24).to(dev())
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt, 32, 32)
    wpa = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.abs(), 32, 32)
    with ran_wino(3):
        y = ops.conv_fwd(ops.K3S1, x, wp, b, 32)
        ya = ops.conv_fwd(ops.K3S1, x2, wp, None, 32)
        ys = ops.conv_fwd(ops.K3S1, x + x2, wp, b, 32)
    os.environ['BTS_WINO'] = '0'
    try:
        y0 = ops.conv_fwd(ops.K3S1, x, wp, b, 32)
        bound = ops.conv_fwd(ops.K3S1, x.abs(), wpa, b.abs(), 32)
        bound2 = ops.conv_fwd(ops.K3S1, x2.abs(), wpa, None, 32)
    finally:
        del os.environ['BTS_WINO']
    tol = 40 * EPS32 * bound + 1e-7
    err = (y - y0).abs()
    assert not bool((err > tol).any()), 'max err %.3e' % float(err.max())
    lin = (ys - (y + ya)).abs()
    tol2 = 3 * 32 * EPS32 * (bound + bound2) + 1e-6      # three Winograd evaluations + the fp32 sums x + x2, y + ya
    assert not bool((lin > tol2).any()), 'linearity: max err %.3e' % float(lin.max())


@pytest.mark.parametrize('t', [1, 2, 3, 5, 8])
def test_w3_chained_items_any_length(t, monkeypatch):
    """conv_wino3.hip walks T consecutive (tile, cout block) items per workgroup as one stage stream (the successor's halo tiles and
    weight fragments ride in the predecessor's last stages); the launcher picks T per launch, BTS_W3_T forces it: every chain
    length -- including ones that do not divide an XCD's item count, so that the last workgroup's chain is cut short -- gives the
    bits of T = 1, and those agree with the oracle.  (Runs for the F(2x2x2,3x3x3) form only.)"""
    from bts_amd import ops
    if FORM['kernel'] != 'w3_kernel':
        pytest.skip('item chaining of this kind exists in w3_kernel only')
    n, (d, h, w), cin, cout = 2, (8, 12, 48), 24, 80     # 2*2*3*3 = 36 tiles over 8 XCDs (ragged eighths), 3 cout blocks, 3 stages
    x = rnd((n, d, h, w, cin), 31)
    wt = rnd((3, 3, 3, cin, cout), 32, 0.2)
    b = rnd((cout,), 33)
    ref = R.conv3d(x.double(), wt.double(), b.double())
    bound = R.conv3d(x.double().abs(), wt.double().abs(), b.double().abs())
    xg = x.to(dev())
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.to(dev()), cin, cout)
    monkeypatch.setenv('BTS_W3_T', '1')
    with ran_wino():
        y1 = ops.conv_fwd(ops.K3S1, xg, wp, b.to(dev()), cout)
    monkeypatch.setenv('BTS_W3_T', str(t))
    with ran_wino():
        yt = ops.conv_fwd(ops.K3S1, xg, wp, b.to(dev()), cout)
    torch.cuda.synchronize()
    assert torch.equal(y1, yt)
    check_wino(yt, ref, bound, 'w3 chain T=%d' % t)
