"""-m gpu: every C-ABI kernel against the fp64 oracle (oracle/torch_ref.py) on the same seeded inputs.

Tolerances: contractions  |err| <= 8*eps32*sum|a_i b_i| (+1e-7);  element-wise / normalisation
kernels  |err| <= 1e-6 + 1e-5*|ref|  (looser 2e-5 relative where a long fp32 reduction feeds the value).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

EPS32 = 2.0 ** -24


def dev():
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale)


def check_contraction(got, ref, bound, what):
    err = (got.double().cpu() - ref).abs()
    tol = 8 * EPS32 * bound + 1e-7
    bad = (err > tol)
    assert not bad.any(), '%s: %d/%d elements out of tolerance, max err %.3e (tol there %.3e), max|ref| %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(tol.flatten()[err.argmax()]), float(ref.abs().max()))


def check_close(got, ref, what, rtol=1e-5, atol=1e-6):
    err = (got.double().cpu() - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e, max|ref| %.3e' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()))


def ref_conv(kind, x, w, b):
    from bts_amd import ops
    if kind == ops.K3S2T:
        return R.conv3d_transpose(x, w, b)
    return R.conv3d(x, w, b, stride=2 if kind == ops.K3S2 else 1)


def wshape(kind, cin, cout):
    from bts_amd import ops
    k = 1 if kind == ops.K1 else 3
    return (k, k, k, cout, cin) if kind == ops.K3S2T else (k, k, k, cin, cout)


CONV_CASES = [
    # kind, N, (D,H,W), Cin, Cout
    (1, 1, (8, 8, 8), 8, 32), (1, 2, (16, 16, 32), 32, 32), (1, 1, (8, 12, 20), 16, 64), (1, 1, (16, 16, 16), 2, 32),
    (1, 1, (8, 8, 8), 32, 2), (1, 1, (4, 4, 4), 40, 128), (1, 1, (32, 32, 64), 32, 32), (1, 1, (6, 10, 14), 12, 20),
    (1, 1, (16, 16, 48), 64, 64),
    (0, 1, (8, 8, 8), 32, 32), (0, 2, (16, 16, 16), 64, 32), (0, 1, (8, 8, 16), 32, 3), (0, 1, (4, 4, 4), 256, 128),
    (0, 1, (8, 8, 8), 2, 32), (0, 1, (32, 32, 32), 64, 32),
    (2, 1, (8, 8, 8), 32, 32), (2, 2, (16, 16, 16), 32, 64), (2, 1, (4, 4, 4), 128, 16), (2, 1, (8, 16, 32), 16, 32),
    (3, 1, (4, 4, 4), 32, 16), (3, 2, (8, 8, 8), 64, 32), (3, 1, (2, 2, 2), 1, 64), (3, 1, (4, 8, 16), 16, 32),
]


@pytest.mark.parametrize('kind,n,dims,cin,cout', CONV_CASES)
def test_conv_fwd_bwd(kind, n, dims, cin, cout):
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 1)
    wt = rnd(wshape(kind, cin, cout), 2, 0.2)
    b = rnd((cout,), 3)
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    ref = ref_conv(kind, xd, wd, bd)
    bound = ref_conv(kind, x.double().abs(), wt.double().abs(), b.double().abs())
    xg, wg, bg = x.to(dev()), wt.to(dev()), b.to(dev())
    wp = ops.conv_pack(kind, ops.ROLE_FWD, wg, cin, cout)
    y = ops.conv_fwd(kind, xg, wp, bg, cout)
    torch.cuda.synchronize()
    check_contraction(y, ref.detach(), bound, 'conv fwd kind %d' % kind)
    # backward
    dy = rnd(tuple(ref.shape), 4)
    ref.backward(dy.double())
    dyg = dy.to(dev())
    wpb = ops.conv_pack(kind, ops.ROLE_BWD, wg, cin, cout)
    dx = torch.empty_like(xg)
    ops.conv_bwd_data(kind, dyg, wpb, dx, accumulate=False)
    dxb = torch.autograd.grad(ref_conv(kind, xd, wd.abs(), None), xd, dy.double().abs())[0]
    # bound for dx: sum |dy||w| == gradient of the abs problem
    check_contraction(dx, xd.grad, dxb, 'conv bwd_data kind %d' % kind)
    ops.conv_bwd_data(kind, dyg, wpb, dx, accumulate=True)
    check_contraction(dx, 2 * xd.grad, 2 * dxb, 'conv bwd_data accumulate kind %d' % kind)
    dw = torch.empty_like(wg)
    db = torch.empty_like(bg)
    ops.conv_bwd_weight(kind, xg, dyg, dw, None if kind == ops.K3S2T else db)
    dwb = torch.autograd.grad(ref_conv(kind, xd.abs(), wd, None), wd, dy.double().abs())[0]
    check_contraction(dw, wd.grad, dwb, 'conv bwd_weight kind %d' % kind)
    if kind != ops.K3S2T:
        check_contraction(db, bd.grad, dy.double().abs().sum(dim=(0, 1, 2, 3)), 'conv bias grad kind %d' % kind)
    else:
        db = ops.colsum(dyg, sum_over_n=True)
        check_contraction(db, bd.grad, dy.double().abs().sum(dim=(0, 1, 2, 3)), 'colsum bias grad')


@pytest.mark.parametrize('kind,n,dims,cin,cout', [(3, 1, (8, 8, 16), 32, 32), (3, 2, (4, 8, 8), 16, 64), (3, 1, (6, 10, 12), 8, 20),
                                                  (3, 1, (2, 4, 40), 24, 32), (2, 1, (16, 16, 32), 32, 32), (2, 2, (8, 12, 20), 16, 40)])
def test_conv_up_merged_classes(monkeypatch, kind, n, dims, cin, cout):
    """Conv3DTranspose forward / stride-2 data gradient through upm_kernel (all 8 output-parity classes per workgroup):
    forced on for grids the heuristic would leave to the per-class path, incl. ragged tiles and accumulate."""
    monkeypatch.setenv('BTS_IGEMM_UPM_MIN', '1')
    test_conv_fwd_bwd(kind, n, dims, cin, cout)


@pytest.mark.parametrize('n,dims,cin,cout', [(1, (8, 8, 8), 32, 32), (2, (4, 6, 10), 64, 3), (1, (8, 8, 16), 24, 72), (1, (3, 5, 7), 256, 128),
                                             (1, (16, 16, 16), 8, 32)])
def test_conv_k1_streaming(monkeypatch, n, dims, cin, cout):
    """1x1x1 conv through k1s_kernel (operands straight from global memory), forced on for small grids: ragged voxel
    counts, cout not a multiple of 4, accumulate; plus sigmoid + strided slab views."""
    from bts_amd import ops
    monkeypatch.setenv('BTS_IGEMM_K1S_MIN', '1')
    test_conv_fwd_bwd(0, n, dims, cin, cout)
    d, h, w = dims
    slab = rnd((n, d, h, w, cin + 8), 31)
    wt = rnd((1, 1, 1, cin, cout), 32, 0.2)
    b = rnd((cout,), 33)
    ref = torch.sigmoid(R.conv3d(slab[..., 8:].double(), wt.double(), b.double()))
    sg = slab.to(dev())
    wp = ops.conv_pack(0, ops.ROLE_FWD, wt.to(dev()), cin, cout)
    out = torch.zeros((n, d, h, w, cout + 4), device=dev())
    ops.conv_fwd(0, sg[..., 8:], wp, b.to(dev()), cout, out=out[..., 4:], sigmoid=True)
    check_close(out[..., 4:], ref, 'k1s strided sigmoid', rtol=1e-5, atol=1e-6)
    assert float(out[..., :4].abs().max()) == 0.0


@pytest.mark.parametrize('n,dims,cin,cout', [(1, (8, 8, 8), 32, 2), (2, (6, 10, 40), 16, 3), (1, (4, 4, 4), 8, 1), (1, (16, 16, 32), 32, 4)])
def test_conv_small_cout_direct(monkeypatch, n, dims, cin, cout):
    """3x3x3 conv with <= 4 output channels through dsc_kernel (vector-ALU direct form), forced on for small grids:
    ragged tiles, all cout counts, bias; plus sigmoid into a strided slab view."""
    from bts_amd import ops
    monkeypatch.setenv('BTS_IGEMM_DSC_MIN', '1')
    test_conv_fwd_bwd(1, n, dims, cin, cout)
    d, h, w = dims
    slab = rnd((n, d, h, w, cin + 8), 51)
    wt = rnd((3, 3, 3, cin, cout), 52, 0.2)
    b = rnd((cout,), 53)
    ref = torch.sigmoid(R.conv3d(slab[..., 8:].double(), wt.double(), b.double()))
    wp = ops.conv_pack(1, ops.ROLE_FWD, wt.to(dev()), cin, cout)
    out = torch.zeros((n, d, h, w, cout + 4), device=dev())
    ops.conv_fwd(1, slab.to(dev())[..., 8:], wp, b.to(dev()), cout, out=out[..., 4:], sigmoid=True)
    check_close(out[..., 4:], ref, 'dsc strided sigmoid', rtol=1e-5, atol=2e-6)
    assert float(out[..., :4].abs().max()) == 0.0


@pytest.mark.parametrize('kind,n,dims,cin,cout', [(3, 2, (8, 8, 8), 64, 32), (3, 1, (4, 8, 16), 128, 40), (2, 1, (16, 16, 16), 32, 64)])
def test_conv_up_merged_classes_split_k(monkeypatch, kind, n, dims, cin, cout):
    """upm_kernel with the contraction split over workgroups (grids too small to fill the chip): raw partials + the
    fixed-order igemm_reduce_kernel, incl. bias and accumulate; BTS_IGEMM_UPM_MIN=64 makes these small grids take it."""
    monkeypatch.setenv('BTS_IGEMM_UPM_MIN', '64')
    test_conv_fwd_bwd(kind, n, dims, cin, cout)


@pytest.mark.parametrize('kind,n,dims,cin,cout,g', [(1, 1, (16, 16, 32), 32, 32, 8), (1, 2, (32, 16, 32), 16, 64, 8), (1, 1, (8, 8, 8), 8, 16, 8),
                                                    (2, 1, (32, 32, 32), 32, 32, 8), (1, 1, (6, 10, 14), 12, 20, 2), (3, 1, (4, 4, 8), 16, 32, 4)])
def test_conv_with_fused_groupnorm_statistics(monkeypatch, kind, n, dims, cin, cout, g):
    """bts_conv3d_fwd_gn: the conv output and its slab-mode GroupNorm statistics from one call -- epilogue partials where
    the z-slab groups hold whole tiles, the library's own bts_gn_stats fallback elsewhere (and with the fusion disabled);
    both against the fp64 reference (group = contiguous 1/G chunk of each sample's flattened (D,H,W,C) memory)."""
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 61)
    wt = rnd(wshape(kind, cin, cout), 62, 0.2)
    b = rnd((cout,), 63)
    ref = ref_conv(kind, x.double(), wt.double(), b.double())
    flat = ref.reshape(n, g, -1)
    mean_r = flat.mean(dim=2).reshape(-1)
    rstd_r = 1.0 / torch.sqrt(flat.var(dim=2, unbiased=False) + 1e-5).reshape(-1)
    wp = ops.conv_pack(kind, ops.ROLE_FWD, wt.to(dev()), cin, cout)
    for fuse in (True, False):
        if not fuse:
            monkeypatch.setenv('BTS_IGEMM_NOGNFUSE', '1')
        y, mean, rstd = ops.conv_fwd_gn(kind, x.to(dev()), wp, b.to(dev()), cout, g, 1e-5)
        check_contraction(y, ref, ref_conv(kind, x.double().abs(), wt.double().abs(), b.double().abs()), 'conv_fwd_gn y')
        check_close(mean, mean_r, 'fused GN mean', rtol=2e-5, atol=2e-5)
        check_close(rstd, rstd_r, 'fused GN rstd', rtol=2e-5, atol=2e-5)


def test_conv_fused_pair_with_groupnorm_statistics():
    from bts_amd import ops
    n, (d, h, w), cin, cout, g = 1, (16, 16, 32), 32, 32, 8
    x = rnd((n, d, h, w, cin), 71)
    w3, b3 = rnd((3, 3, 3, cin, cout), 72, 0.2), rnd((cout,), 73)
    w1, b1 = rnd((1, 1, 1, cin, cout), 74, 0.2), rnd((cout,), 75)
    wp3 = ops.conv_pack(1, ops.ROLE_FWD, w3.to(dev()), cin, cout)
    wp1 = ops.conv_pack(0, ops.ROLE_FWD, w1.to(dev()), cin, cout)
    out = ops.conv_fwd_fused2_gn(x.to(dev()), wp3, b3.to(dev()), wp1, b1.to(dev()), cout, g, 1e-5)
    assert out is not None
    c1, res, mean, rstd = out
    r3 = R.conv3d(x.double(), w3.double(), b3.double())
    r1 = R.conv3d(x.double(), w1.double(), b1.double())
    check_contraction(c1, r3, R.conv3d(x.double().abs(), w3.double().abs(), b3.double().abs()), 'fused_gn conv3')
    check_contraction(res, r1, R.conv3d(x.double().abs(), w1.double().abs(), b1.double().abs()), 'fused_gn conv1')
    flat = r3.reshape(n, g, -1)
    check_close(mean, flat.mean(dim=2).reshape(-1), 'fused2 GN mean', rtol=2e-5, atol=2e-5)
    check_close(rstd, (1.0 / torch.sqrt(flat.var(dim=2, unbiased=False) + 1e-5)).reshape(-1), 'fused2 GN rstd', rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('n,dims,cin,cout,pad', [(1, (16, 16, 32), 32, 32, 0), (2, (8, 8, 8), 24, 16, 8), (1, (4, 4, 4), 64, 128, 0),
                                                 (1, (32, 32, 32), 48, 64, 16), (1, (6, 10, 14), 12, 8, 0)])
def test_conv_bwd_data_pair(monkeypatch, n, dims, cin, cout, pad):
    """bts_conv3d_bwd_data_pair: dx (+)= bwd(3x3x3)(dy) + bwd(1x1x1)(dy2) in one pass (centre-tap fusion), into a slab slice,
    with accumulation, and with the fusion disabled (two launches) -- all against the fp64 reference."""
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 81).double().requires_grad_(True)
    w3 = rnd((3, 3, 3, cin, cout), 82, 0.2)
    w1 = rnd((1, 1, 1, cin, cout), 83, 0.3)
    dy, dy2 = rnd((n, d, h, w, cout), 84), rnd((n, d, h, w, cout), 85)
    y = R.conv3d(x, w3.double(), None)
    y2 = R.conv3d(x, w1.double(), None)
    (y * dy.double()).sum().backward(retain_graph=True)
    (y2 * dy2.double()).sum().backward()
    ref = x.grad
    xa = x.detach().abs().requires_grad_(True)
    ((R.conv3d(xa, w3.double().abs(), None) * dy.double().abs()).sum() + (R.conv3d(xa, w1.double().abs(), None) * dy2.double().abs()).sum()).backward()
    bound = xa.grad
    wpb3 = ops.conv_pack(1, ops.ROLE_BWD, w3.to(dev()), cin, cout)
    wpb1 = ops.conv_pack(0, ops.ROLE_BWD, w1.to(dev()), cin, cout)
    for fuse in (True, False):
        if not fuse:
            monkeypatch.setenv('BTS_IGEMM_NOPAIR', '1')
        slab = torch.full((n, d, h, w, cin + pad), 3.0, device=dev())
        dx = slab[..., pad:]
        ops.conv_bwd_data_pair(dy.to(dev()), wpb3, dy2.to(dev()), wpb1, dx, False)
        check_contraction(dx, ref, bound, 'bwd_data_pair fuse=%s' % fuse)
        assert pad == 0 or float((slab[..., :pad] - 3.0).abs().max()) == 0.0
        ops.conv_bwd_data_pair(dy.to(dev()), wpb3, dy2.to(dev()), wpb1, dx, True)
        check_contraction(dx, 2 * ref, 2 * bound, 'bwd_data_pair accumulate fuse=%s' % fuse)


def test_conv_strided_views_and_sigmoid():
    """channel slices of a slab as conv input and output (virtual Concatenate), fused sigmoid"""
    from bts_amd import ops
    n, d, h, w = 1, 8, 8, 16
    slab = rnd((n, d, h, w, 48), 5)
    wt = rnd((3, 3, 3, 32, 16), 6, 0.2)
    b = rnd((16,), 7)
    sg = slab.to(dev())
    xin = sg[..., :32]
    out = sg[..., 32:48]
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wt.to(dev()), 32, 16)
    ops.conv_fwd(ops.K3S1, xin, wp, b.to(dev()), 16, out=out, sigmoid=True)
    ref = torch.sigmoid(R.conv3d(slab[..., :32].double(), wt.double(), b.double()))
    check_close(sg[..., 32:48], ref, 'strided sigmoid conv', rtol=2e-5, atol=2e-6)
    check_close(sg[..., :32], slab[..., :32].double(), 'input slice untouched', rtol=0, atol=0)


@pytest.mark.parametrize('n,dims,cin,cout', [(1, (16, 16, 32), 32, 32), (1, (8, 8, 8), 24, 128), (2, (4, 4, 8), 2, 32),
                                             (1, (16, 16, 16), 64, 64)])
def test_conv_fused_shortcut_pair(n, dims, cin, cout):
    """resnet.py:118 + :133-134 in one pass over x: bts_conv3d_fwd_fused2 == the two separate convolutions"""
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, cin), 50)
    w3, b3 = rnd((3, 3, 3, cin, cout), 51, 0.2), rnd((cout,), 52)
    w1, b1 = rnd((1, 1, 1, cin, cout), 53, 0.3), rnd((cout,), 54)
    xg = x.to(dev())
    wp3 = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, w3.to(dev()), cin, cout)
    wp1 = ops.conv_pack(ops.K1, ops.ROLE_FWD, w1.to(dev()), cin, cout)
    out = ops.conv_fwd_fused2(xg, wp3, b3.to(dev()), wp1, b1.to(dev()), cout)
    if out is None:   # tiling without room for the second accumulator set: the engine falls back to two launches
        pytest.skip('not fusable at this shape')
    c1, res = out
    r3 = R.conv3d(x.double(), w3.double(), b3.double())
    r1 = R.conv3d(x.double(), w1.double(), b1.double())
    check_contraction(c1, r3, R.conv3d(x.double().abs(), w3.double().abs(), b3.double().abs()), 'fused conv3')
    check_contraction(res, r1, R.conv3d(x.double().abs(), w1.double().abs(), b1.double().abs()), 'fused conv1')


@pytest.mark.parametrize('kind', [0, 1])
def test_conv_folded_duplicate_slice(kind):
    """encoder.py:83-87: block j sees [o_{j-1}, o_0..o_{j-1}]; the engine reads the slab [o_0..o_{j-1}] once with
    folded weights.  Forward, data-gradient and weight-gradient must equal the duplicated formulation."""
    from bts_amd import ops
    n, d, h, w, f, j = 1, 8, 8, 8, 16, 2
    cin_slab, cin_ref, cout = j * f, (j + 1) * f, 32
    slab = rnd((n, d, h, w, cin_slab), 8)
    k = 1 if kind == 0 else 3
    wt = rnd((k, k, k, cin_ref, cout), 9, 0.2)
    xd = slab.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    xcat = torch.cat([xd[..., (j - 1) * f:], xd], dim=-1)
    ref = R.conv3d(xcat, wd, None)
    bound = R.conv3d(xcat.detach().abs(), wt.double().abs(), None)
    sg, wg = slab.to(dev()), wt.to(dev())
    wp = ops.conv_pack(kind, ops.ROLE_FWD, wg, cin_ref, cout, cin_slab, (j - 1) * f, f)
    y = ops.conv_fwd(kind, sg, wp, None, cout)
    check_contraction(y, ref.detach(), 2 * bound, 'folded fwd')
    dy = rnd(tuple(ref.shape), 10)
    ref.backward(dy.double())
    wpb = ops.conv_pack(kind, ops.ROLE_BWD, wg, cin_ref, cout, cin_slab, (j - 1) * f, f)
    dx = torch.empty_like(sg)
    ops.conv_bwd_data(kind, dy.to(dev()), wpb, dx, False)
    check_close(dx, xd.grad, 'folded bwd_data', rtol=1e-4, atol=1e-4)
    dw = torch.empty_like(wg)
    ops.conv_bwd_weight(kind, sg, dy.to(dev()), dw, None, (j - 1) * f, f)
    check_close(dw, wd.grad, 'folded bwd_weight', rtol=1e-4, atol=1e-4)


def test_conv_pack_batch_matches_single():
    """bts_conv_pack_batch (one launch for every weight image) must produce bit-identical images to bts_conv_pack."""
    from bts_amd import ops
    specs = [(1, 0, 32, 32, 32, 0, 0), (1, 1, 32, 32, 32, 0, 0), (0, 0, 48, 16, 32, 16, 16), (0, 1, 48, 16, 32, 16, 16),
             (2, 0, 16, 24, 16, 0, 0), (3, 1, 8, 40, 8, 0, 0), (1, 0, 2, 32, 2, 0, 0), (1, 1, 32, 3, 32, 0, 0)]
    entries, singles = [], []
    for i, (kind, role, cin_ref, cout, cin_slab, dstart, dshift) in enumerate(specs):
        k = 1 if kind == 0 else 3
        shape = (k, k, k, cout, cin_ref) if kind == 3 else (k, k, k, cin_ref, cout)
        w = rnd(shape, 40 + i, 0.3).to(dev())
        singles.append(ops.conv_pack(kind, role, w, cin_ref, cout, cin_slab, dstart, dshift))
        wp = ops.conv_packed_empty(kind, role, cin_slab, cout, w.device)
        wp.fill_(float('nan'))
        entries.append((kind, role, w, wp, cin_ref, cout, cin_slab, dstart, dshift))
    table = ops.PackTable()
    table.run(entries)
    table.run(entries)  # cached table path
    torch.cuda.synchronize()
    for e, ref in zip(entries, singles):
        assert e[3].shape == ref.shape
        assert torch.equal(e[3], ref), 'batched image differs for %r' % (e[:2] + e[4:],)


GN_CASES = [(1, (8, 8, 8), 32, 8), (2, (8, 8, 16), 16, 8), (1, (16, 16, 16), 64, 8), (2, (4, 4, 4), 256, 8),
            (1, (8, 8, 8), 16, 2), (1, (2, 2, 2), 16, 8), (1, (32, 32, 32), 32, 8),
            (2, (4, 4, 4), 2, 2), (1, (1, 1, 1), 16, 8), (1, (2, 4, 2), 12, 3)]


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('relu', [0, 1])
@pytest.mark.parametrize('n,dims,c,g', GN_CASES)
def test_groupnorm_fwd_bwd(mode, relu, n, dims, c, g):
    from bts_amd import ops
    d, h, w = dims
    x = rnd((n, d, h, w, c), 11) * 2 + 0.5
    gamma = rnd((c,), 12)
    beta = rnd((c,), 13)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    if mode == 0:
        ref = R.group_norm(xd, gd, bd, g, -1)
    else:  # channels_first semantics evaluated on NCDHW, compared in NDHWC
        ref = R.group_norm(xd.permute(0, 4, 1, 2, 3), gd, bd, g, 1).permute(0, 2, 3, 4, 1)
    if relu:
        ref = torch.relu(ref)
    xg = x.to(dev())
    mean, rstd = ops.gn_stats(xg, g, mode)
    y = ops.gn_apply(xg, gamma.to(dev()), beta.to(dev()), mean, rstd, g, mode, relu)
    check_close(y, ref.detach(), 'gn fwd', rtol=2e-5, atol=2e-5)
    dy = rnd((n, d, h, w, c), 14)
    ref.backward(dy.double())
    dgam = torch.empty(c, device=dev())
    dbet = torch.empty(c, device=dev())
    dx = ops.gn_bwd(xg, dy.to(dev()), gamma.to(dev()), beta.to(dev()), mean, rstd, dgam, dbet, g, mode, relu)
    sc = float(xd.grad.abs().max())
    check_close(dx, xd.grad, 'gn dx', rtol=1e-4, atol=1e-4 * sc + 1e-6)
    gs = float(gd.grad.abs().max()) + 1.0
    check_close(dgam, gd.grad, 'gn dgamma', rtol=1e-4, atol=1e-5 * gs)
    check_close(dbet, bd.grad, 'gn dbeta', rtol=1e-4, atol=1e-5 * gs)


def test_groupnorm_strided_output():
    from bts_amd import ops
    x = rnd((1, 8, 8, 8, 16), 15)
    gamma, beta = rnd((16,), 16), rnd((16,), 17)
    slab = torch.zeros((1, 8, 8, 8, 48), device=dev())
    xg = x.to(dev())
    mean, rstd = ops.gn_stats(xg, 8, 0)
    ops.gn_apply(xg, gamma.to(dev()), beta.to(dev()), mean, rstd, 8, 0, 1, out=slab[..., 32:48])
    ref = torch.relu(R.group_norm(x.double(), gamma.double(), beta.double(), 8, -1))
    check_close(slab[..., 32:48], ref, 'gn strided out', rtol=2e-5, atol=2e-5)
    assert float(slab[..., :32].abs().max()) == 0.0


@pytest.mark.parametrize('n,dims,f,r,g', [(1, (8, 8, 8), 32, 4, 8), (2, (8, 8, 8), 16, 8, 8), (1, (4, 4, 8), 256, 32, 8),
                                          (2, (4, 4, 4), 64, 8, 8)])
def test_se_gate_epilogue_fwd_bwd(n, dims, f, r, g):
    """out = res*(sigmoid(res.w_sp)+ch) + relu(GN2(c2)), resnet.py:121-137"""
    from bts_amd import ops
    d, h, w = dims
    v = d * h * w
    res = rnd((n, d, h, w, f), 20)
    c2 = rnd((n, d, h, w, f), 21)
    w1, w2, wsp = rnd((f, r), 22, 0.3), rnd((r, f), 23, 0.3), rnd((f,), 24, 0.3)
    gamma, beta = rnd((f,), 25), rnd((f,), 26)
    rd = res.double().requires_grad_(True)
    w1d, w2d, wspd = [t.double().requires_grad_(True) for t in (w1, w2, wsp)]
    gap = rd.mean(dim=(1, 2, 3))
    hh = torch.relu(gap @ w1d)
    ch = torch.sigmoid(hh @ w2d)
    sp = torch.sigmoid(rd @ wspd)
    conv = torch.relu(R.group_norm(c2.double(), gamma.double(), beta.double(), g, -1))
    ref = rd * (sp.unsqueeze(-1) + ch.reshape(n, 1, 1, 1, f)) + conv
    D = dev()
    rg, c2g = res.to(D), c2.to(D)
    gapg = ops.colsum(rg, scale=1.0 / v)
    check_close(gapg, gap.detach(), 'gap', rtol=1e-5, atol=1e-6)
    hg, chg = ops.se_mlp_fwd(gapg, w1.to(D), w2.to(D))
    check_close(chg, ch.detach(), 'chse', rtol=1e-5, atol=1e-6)
    mean, rstd = ops.gn_stats(c2g, g, 0)
    out = torch.empty_like(rg)
    spg = ops.block_epilogue_fwd(rg, c2g, out, wsp.to(D), chg, gamma.to(D), beta.to(D), mean, rstd, g, 0)
    check_close(out, ref.detach(), 'epilogue', rtol=2e-5, atol=2e-5)
    dout = rnd((n, d, h, w, f), 27)
    ref.backward(dout.double())
    dw1, dw2, dwsp = torch.empty_like(w1, device=D), torch.empty_like(w2, device=D), torch.empty_like(wsp, device=D)
    dres = ops.se_bwd(dout.to(D), rg, spg, gapg, hg, chg, w1.to(D), w2.to(D), wsp.to(D), dw1, dw2, dwsp)
    check_close(dres, rd.grad, 'gate dres', rtol=1e-4, atol=1e-5)
    for got, want, nm in ((dw1, w1d.grad, 'dw1'), (dw2, w2d.grad, 'dw2'), (dwsp, wspd.grad, 'dwsp')):
        check_close(got, want, nm, rtol=1e-4, atol=1e-5 * (float(want.abs().max()) + 1))


@pytest.mark.parametrize('accumulate', [False, True])
@pytest.mark.parametrize('shape', [(2, 8, 16, 16, 32), (1, 16, 16, 32, 64), (2, 4, 8, 8, 256), (1, 8, 8, 16, 8)])
def test_fused_gate_and_groupnorm2_backward(shape, accumulate):
    """bts_block_bwd (one reduce + one apply pass over dout; resnet.py:121-137 under autodiff) against bts_se_bwd followed by bts_gn_bwd
    on the same tensors: same arithmetic element for element, the fp64 partial sums grouped differently -- dc2 / dres and the parameter
    gradients to a few fp32 units; dout arrives as a channel slice of a wider slab; first-write and accumulate into non-zero buffers"""
    from bts_amd import ops
    D = dev()
    n, d, h, w, f = shape
    groups, r = (8 if f >= 8 else f), max(f // 8, 1)
    g = torch.Generator().manual_seed(f + d)
    res = torch.randn(shape, generator=g).to(D)
    c2 = torch.randn(shape, generator=g).to(D)
    slab = torch.randn((n, d, h, w, f + 16), generator=g).to(D)
    dout = slab[..., 8:8 + f]
    sp = torch.sigmoid(torch.randn((n, d, h, w, 1), generator=g)).to(D).contiguous()
    gap = torch.randn((n, f), generator=g).to(D)
    w1 = (0.3 * torch.randn((f, r), generator=g)).to(D)
    w2 = (0.3 * torch.randn((r, f), generator=g)).to(D)
    wsp = (0.3 * torch.randn(f, generator=g)).to(D)
    gamma = (1.0 + 0.3 * torch.randn(f, generator=g)).to(D)
    beta = (0.2 * torch.randn(f, generator=g)).to(D)
    hbuf, ch = ops.se_mlp_fwd(gap, w1, w2)
    mean, rstd = ops.gn_stats(c2, groups, ops.GN_SLAB)
    init = [torch.randn(t.shape, generator=g).to(D) for t in (w1, w2, wsp, gamma, beta)]      # dw1 dw2 dwsp dgamma dbeta
    ref = [t.clone() for t in init]
    dres_r = ops.se_bwd(dout, res, sp, gap, hbuf, ch, w1, w2, wsp, ref[0], ref[1], ref[2], accumulate_params=accumulate)
    dc2_r = ops.gn_bwd(c2, dout, gamma, beta, mean, rstd, ref[3], ref[4], groups, ops.GN_SLAB, True, accumulate_params=accumulate)
    got = [t.clone() for t in init]
    assert ops.block_bwd_takes(res, r, groups, dout, c2)
    dres, dc2 = ops.block_bwd(dout, res, c2, sp, gap, hbuf, ch, w1, w2, wsp, gamma, beta, mean, rstd, groups, got[0], got[1], got[2], got[3], got[4],
                              accumulate_gate_params=accumulate, accumulate_norm_params=accumulate)
    torch.cuda.synchronize()
    for a, b, nm in ((dc2, dc2_r, 'dc2'), (dres, dres_r, 'dres')):
        assert float((a - b).abs().max()) <= 4e-7 * float(b.abs().max()) + 1e-7, (nm, float((a - b).abs().max()), float(b.abs().max()))
    for a, b, nm in zip(got, ref, ('dw1', 'dw2', 'dwsp', 'dgamma', 'dbeta')):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-6, (nm, float((a - b).abs().max()), float(b.abs().max()))


def test_fused_block_backward_declines_what_it_cannot_tile():
    """units that are not whole 1024-element chunks / channel counts that do not divide a chunk: the query says so and the layer keeps
    the two separate calls (tests/test_model_gpu.py runs such models)"""
    from bts_amd import ops
    D = dev()
    for shape, groups in (((1, 4, 4, 4, 8), 4), ((1, 6, 10, 10, 16), 8), ((1, 8, 8, 8, 24), 8)):
        res = torch.zeros(shape, device=D)
        assert not ops.block_bwd_takes(res, 2, groups, res, res), shape


def test_dropout_sample_dense():
    from bts_amd import ops
    D = dev()
    x = rnd((1, 8, 8, 8, 2), 30)
    m = ops.dropout_mask(x.shape, 0.2, 1234, D)
    keep = float(m.float().mean())
    assert 0.7 < keep < 0.9
    y = ops.dropout_apply(x.to(D), m, 0.2)
    check_close(y, R.dropout(x.double(), m.cpu().double(), 0.2), 'dropout')
    e = ops.normal((4096,), 7, D).cpu()
    assert abs(float(e.mean())) < 0.08 and abs(float(e.std()) - 1) < 0.08
    proj, eps = rnd((3, 16), 31), rnd((3, 8), 32)
    pd = proj.double().requires_grad_(True)
    z = R.sample(pd[:, :8], pd[:, 8:], eps.double())
    zg = ops.vae_sample_fwd(proj.to(D), eps.to(D))
    check_close(zg, z.detach(), 'sample')
    dz = rnd((3, 8), 33)
    z.backward(dz.double())
    dproj = torch.zeros((3, 16), device=D)
    ops.vae_sample_bwd(proj.to(D), eps.to(D), dz.to(D), dproj)
    check_close(dproj, pd.grad, 'sample bwd')
    for (n, fin, fout, relu) in [(1, 8192, 256, 0), (3, 128, 512, 1), (2, 64, 40, 1)]:
        xx, ww, bb = rnd((n, fin), 34), rnd((fin, fout), 35, 0.1), rnd((fout,), 36)
        xd, wd, bd = [t.double().requires_grad_(True) for t in (xx, ww, bb)]
        ref = xd @ wd + bd
        if relu:
            ref = torch.relu(ref)
        yg = ops.dense_fwd(xx.to(D), ww.to(D), bb.to(D), relu)
        bound = xx.double().abs() @ ww.double().abs() + bb.double().abs()
        check_contraction(yg, ref.detach(), bound, 'dense fwd')
        dy = rnd((n, fout), 37)
        ref.backward(dy.double())
        g = ops.relu_bwd(yg, dy.to(D)) if relu else dy.to(D)
        dx = torch.empty((n, fin), device=D)
        dw = torch.empty((fin, fout), device=D)
        db = torch.empty((fout,), device=D)
        ops.dense_bwd(xx.to(D), ww.to(D), g, dx, dw, db)
        check_close(dx, xd.grad, 'dense dx', rtol=1e-4, atol=1e-5)
        check_close(dw, wd.grad, 'dense dw', rtol=1e-4, atol=1e-5)
        check_close(db, bd.grad, 'dense db', rtol=1e-4, atol=1e-5)


def test_loss_metric_l2_adam():
    from bts_amd import ops
    D = dev()
    n, d, h, w = 2, 8, 8, 16
    x, y, _, _ = R.synthetic_batch(n, (d, h, w), latent=8)
    logits = rnd((n, d, h, w, 3), 40)
    yv = rnd((n, d, h, w, 2), 41)
    proj = rnd((n, 16), 42, 0.5)
    ld_, yvd, pd = [t.double().requires_grad_(True) for t in (logits, yv, proj)]
    yp = torch.sigmoid(ld_)
    loss = R.dice_vae_loss(x.double(), y.double(), yp, yvd, pd[:, :8], pd[:, 8:])
    ypg = torch.sigmoid(logits.double()).float().to(D)
    xg, yg, yvg, pg = x.to(D), y.to(D), yv.to(D), proj.to(D)
    sums = ops.loss_sums(ypg, yg, xg, yvg, pg)
    lv, parts = ops.loss_value(sums, 3)
    check_close(lv, loss.detach().reshape(1), 'loss value', rtol=1e-5, atol=1e-6)
    loss.backward()
    dyp = torch.empty_like(ypg)
    dyv = torch.empty_like(yvg)
    dpr = torch.empty_like(pg)
    ops.loss_bwd(ypg, yg, xg, yvg, pg, sums, None, dyp, dyv, dpr, through_sigmoid=True)
    check_close(dyp, ld_.grad, 'dlogits', rtol=1e-4, atol=1e-9)
    check_close(dyv, yvd.grad, 'dyvae', rtol=1e-4, atol=1e-10)
    check_close(dpr, pd.grad, 'dproj', rtol=1e-4, atol=1e-9)
    # metric (F8 axes) + bit-exact labels
    macro, micro, labels = R.dice_coefficient(y.double(), ypg.cpu().double())
    table, lab = ops.dice_metric_sums(yg, ypg, True)
    mv = ops.dice_metric_value(table, w, 3, True).cpu()
    assert torch.equal(lab.cpu().long(), labels.long()), 'argmax label map must be bit-exact'
    assert abs(float(mv[0]) - float(macro)) < 1e-6 and abs(float(mv[1]) - float(micro)) < 1e-6
    # L2 + Adam on a flat buffer
    p = rnd((10007,), 43).to(D)
    g = rnd((10007,), 44).to(D)
    ranges = [(0, 4000, 1e-5), (4000, 3001, 3e-5)]
    l2 = ops.l2_reg_fwd(p, ranges)
    pc = p.cpu().double()
    ref = 1e-5 * (pc[:4000] ** 2).sum() + 3e-5 * (pc[4000:7001] ** 2).sum()
    check_close(l2, ref.reshape(1), 'l2 value', rtol=1e-5, atol=1e-9)
    g2 = g.clone()
    ops.l2_reg_bwd(p, g2, ranges)
    gr = g.cpu().double().clone()
    gr[:4000] += 2e-5 * pc[:4000]
    gr[4000:7001] += 6e-5 * pc[4000:7001]
    check_close(g2, gr, 'l2 grad', rtol=1e-5, atol=1e-7)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    pr, mr, vr = p.cpu().double(), torch.zeros(10007, dtype=torch.float64), torch.zeros(10007, dtype=torch.float64)
    import math
    for t in (1, 2, 3):
        lr_t = 1e-4 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_tf_step(p, g, m, v, lr_t, 0.9, 0.999, 1e-7)
        pr, mr, vr = R.adam_tf_step(pr, g.cpu().double(), mr, vr, t, 1e-4)
    check_close(p, pr, 'adam p', rtol=1e-6, atol=1e-7)
    check_close(m, mr, 'adam m', rtol=1e-5, atol=1e-8)
    check_close(v, vr, 'adam v', rtol=5e-5, atol=1e-9)  # (1-beta_2) is inexact in fp32


@pytest.mark.parametrize('n,dims,c,pad', [(1, (4, 6, 8), 5, 0), (2, (2, 2, 2), 16, 3), (1, (8, 4, 10), 32, 8)])
def test_maxpool_and_nearest_upsample(n, dims, c, pad):
    """MaxPooling3D 2/2 (downsample.py:51-70) and UpSampling3D 2 (upsample.py:70-72) forward / backward against torch,
    on dense tensors and on channel slices of a wider slab (ld > C), with and without accumulation."""
    from bts_amd import ops
    d, h, w = dims
    g = torch.Generator().manual_seed(13)
    slab = torch.randn((n, d, h, w, c + pad), generator=g)
    xg = slab.to(dev())[..., pad:]
    xd = slab[..., pad:].double().permute(0, 4, 1, 2, 3).requires_grad_(True)
    # max pool
    y, idx = ops.maxpool2_fwd(xg)
    ref = torch.nn.functional.max_pool3d(xd, 2, 2)
    assert torch.equal(y.cpu().double(), ref.detach().permute(0, 2, 3, 4, 1))
    dy = torch.randn(tuple(y.shape), generator=g)
    ref.backward(dy.double().permute(0, 4, 1, 2, 3))
    gslab = torch.full((n, d, h, w, c + pad), 7.0, device=dev())
    ops.maxpool2_bwd(dy.to(dev()), idx, gslab[..., pad:], False)
    assert torch.equal(gslab[..., pad:].cpu().double(), xd.grad.permute(0, 2, 3, 4, 1))
    assert pad == 0 or float((gslab[..., :pad] - 7.0).abs().max()) == 0.0
    ops.maxpool2_bwd(dy.to(dev()), idx, gslab[..., pad:], True)
    assert torch.allclose(gslab[..., pad:].cpu().double(), 2 * xd.grad.permute(0, 2, 3, 4, 1))
    # nearest upsample
    up = ops.upsample2_fwd(xg)
    xr = slab[..., pad:].double().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    uref = xr.repeat_interleave(2, 2).repeat_interleave(2, 3).repeat_interleave(2, 4)
    assert torch.equal(up.cpu().double(), uref.detach().permute(0, 2, 3, 4, 1))
    out_slab = torch.zeros((n, 2 * d, 2 * h, 2 * w, c + pad), device=dev())
    ops.upsample2_fwd(xg, out=out_slab[..., pad:])
    assert torch.equal(out_slab[..., pad:], up) and (pad == 0 or float(out_slab[..., :pad].abs().max()) == 0.0)
    du = torch.randn(tuple(up.shape), generator=g)
    uref.backward(du.double().permute(0, 4, 1, 2, 3))
    dx = ops.upsample2_bwd(du.to(dev()))
    assert torch.allclose(dx.cpu().double(), xr.grad.permute(0, 2, 3, 4, 1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('n,dims,cout', [(1, (8, 8, 32), 32), (2, (5, 6, 40), 16), (1, (16, 8, 64), 32)])
def test_conv_two_channel_input(monkeypatch, n, dims, cout):
    """c2_kernel: the network's first convolutions (2 input channels) with the channel pair as the MFMA's K -- plain, with the
    fused 1x1x1 shortcut output, and with GroupNorm statistics from the epilogue; all through the usual entry points"""
    from bts_amd import ops
    monkeypatch.setenv('BTS_IGEMM_C2_MIN', '1')
    d, h, w = dims
    x = rnd((n, d, h, w, 2), 91)
    w3, b3 = rnd((3, 3, 3, 2, cout), 92, 0.3), rnd((cout,), 93)
    w1, b1 = rnd((1, 1, 1, 2, cout), 94, 0.3), rnd((cout,), 95)
    r3 = R.conv3d(x.double(), w3.double(), b3.double())
    r1 = R.conv3d(x.double(), w1.double(), b1.double())
    b3d = R.conv3d(x.double().abs(), w3.double().abs(), b3.double().abs())
    b1d = R.conv3d(x.double().abs(), w1.double().abs(), b1.double().abs())
    xg = x.to(dev())
    wp3 = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, w3.to(dev()), 2, cout)
    wp1 = ops.conv_pack(ops.K1, ops.ROLE_FWD, w1.to(dev()), 2, cout)
    ops.profile_enable(True)
    y = ops.conv_fwd(ops.K3S1, xg, wp3, b3.to(dev()), cout)
    pair = ops.conv_fwd_fused2(xg, wp3, b3.to(dev()), wp1, b1.to(dev()), cout)
    g = 4 if d % 16 == 0 else 2 if d % 8 == 0 else 1
    yg, mean, rstd = ops.conv_fwd_gn(ops.K3S1, xg, wp3, b3.to(dev()), cout, g, 1e-5)
    torch.cuda.synchronize()
    names = [r[0] for r in ops.profile_records()]
    ops.profile_enable(False)
    assert names.count('c2_kernel') >= 2, names
    check_contraction(y, r3, b3d, 'c2 conv')
    check_contraction(yg, r3, b3d, 'c2 conv + gn')
    if pair is not None:
        check_contraction(pair[0], r3, b3d, 'c2 fused conv3')
        check_contraction(pair[1], r1, b1d, 'c2 fused conv1')
    flat = r3.reshape(n, g, -1)
    check_close(mean, flat.mean(dim=2).reshape(-1), 'c2 GN mean', rtol=2e-5, atol=2e-5)
    check_close(rstd, (1.0 / torch.sqrt(flat.var(dim=2, unbiased=False) + 1e-5)).reshape(-1), 'c2 GN rstd', rtol=2e-5, atol=2e-5)
