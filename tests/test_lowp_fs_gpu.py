"""-m gpu: conv1 and the shortcut of a ResnetBlock from ONE pass over the block input (bts_lp_conv3d_fwd_gn_shortcut, round 6).

resnet.py:118 (`res = conv3d_ptwise(inputs)`) and resnet.py:134 (`x = conv3d_1(inputs)`) read the same tensor: in the z-marching kernel
(`lp_s1z_kernel<.., FS>`) the 1x1x1 shortcut is a second set of output columns at the centre tap of the input planes already in LDS; the
gate's squeeze (resnet.py:121: mean over the voxels of res) leaves as column sums of the unrounded shortcut output, GroupNorm-1's
statistics (group_norm.py:100-107, slab semantics) as partial sums of conv1's -- all against the ORACLE's ops in fp64 on the same
16-bit-rounded operands: |err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref| for the two stored tensors, 2e-5 (relative for rstd) for the
statistics, 1e-5 of the largest |gap| for the squeeze."""
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
    # n, (D,H,W), Cin, F, slab_in
    (2, (16, 32, 64), 32, 32, False),      # two k-steps
    (1, (40, 32, 64), 16, 32, True),       # one k-step, x a channel slice of a wider tensor, z chunks
    (1, (40, 16, 96), 32, 16, False),      # half-filled column block (both outputs)
    (4, (8, 48, 32), 16, 24, True),        # 24 columns, 12 columns of 8 planes
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_conv1_and_shortcut_from_one_pass(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    n, (d, h, w), cin, f, slab_in = case
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    G = 8
    g = torch.Generator().manual_seed(hash((d, h, w, cin, f)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    w3 = torch.randn((3, 3, 3, cin, f), generator=g) * (2.0 / (27 * cin)) ** 0.5
    w1 = torch.randn((1, 1, 1, cin, f), generator=g) * (2.0 / cin) ** 0.5
    b3 = torch.randn(f, generator=g) * 0.3
    b1 = torch.randn(f, generator=g) * 0.3
    xr, w3r, w1r = _round(x, tdt), _round(w3, tdt), _round(w1, tdt)
    y_ref = R.conv3d(xr, w3r, b3.double())
    r_ref = R.conv3d(xr, w1r, b1.double())
    y_bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), w3r.abs(), None) + u * y_ref.abs() + 1e-30
    r_bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), w1r.abs(), None) + u * r_ref.abs() + 1e-30
    ldx = cin + 16 if slab_in else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_in else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    norm = GroupNormalization(groups=G, axis=-1)
    norm.build((n, d, h, w, f))
    wp3 = lowp.pack(ops.K3S1, code, w3.to(DEV), cin, f)
    wp1 = lowp.pack(ops.K1, code, w1.to(DEV), cin, f)
    out, syms = _ran(lambda: lowp.conv_gn_shortcut(code, tdt, xin[..., c0:c0 + cin], wp3, b3.to(DEV), f, norm, wp1, b1.to(DEV)))
    assert out is not None, 'the fused form declined a shape it is built for'
    y, mean, rstd, res, gap = out
    assert 'lp_s1z_kernel' in syms and not any(s.startswith('lp_k1') or s.startswith('lp_conv_gather') for s in syms), syms
    assert float(((y.double().cpu() - y_ref).abs() / y_bound).max()) <= 1.0
    assert float(((res.double().cpu() - r_ref).abs() / r_bound).max()) <= 1.0
    chunks = y_ref.reshape(n, G, -1)          # slab semantics
    m_ref = chunks.mean(dim=2).reshape(-1)
    s_ref = (chunks.var(dim=2, unbiased=False) + norm.epsilon).rsqrt().reshape(-1)
    assert float((mean.double().cpu() - m_ref).abs().max()) <= 2e-5
    assert float(((rstd.double().cpu() - s_ref).abs() / s_ref).max()) <= 2e-5
    gap_ref = r_ref.reshape(n, -1, f).mean(dim=1)
    assert float((gap.double().cpu() - gap_ref).abs().max()) <= 1e-5 * max(1.0, float(gap_ref.abs().max()))
    # and the same numbers as the two-launch route it replaces (stored tensors bit-equal: the same fp32 accumulation order per output)
    res2, gap2 = lowp.conv1_gap(code, xin[..., c0:c0 + cin], wp1, b1.to(DEV), f, tdt)
    y2, mean2, rstd2 = lowp.conv_gn(code, tdt, xin[..., c0:c0 + cin], wp3, b3.to(DEV), f, norm)
    torch.cuda.synchronize()
    assert torch.equal(y2, y)
    assert float((res2.float() - res.float()).abs().max()) <= 2 * u * float(res.float().abs().max())
    assert float((gap2 - gap).abs().max()) <= 1e-5 * max(1.0, float(gap.abs().max()))


@pytest.mark.parametrize('split', [False, True], ids=['one-tensor', 'two-operands'])
@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_sixty_four_input_channels_as_two_passes(dtype, split, monkeypatch):
    """Cin = 64 (the decoder's top block over [skip | up-sampled], decoder.py:75): conv1 runs as two marches over the channel halves
    (lowp_s1z.hip) and the shortcut rides on both -- the first pass writes res = x[0:32] . W[0:32] + b, the second adds x[32:64] . W[32:64] and
    leaves the squeeze of the result.  BTS_LP_S1Z_PAIR=1 forces the form below its 8 M-voxel threshold.  Both partial results pass through the
    storage type once: the bounds carry that rounding."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    monkeypatch.setenv('BTS_LP_S1Z_PAIR', '1')
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    n, d, h, w, cin, f, G = 2, 16, 32, 64, 64, 32, 8
    g = torch.Generator().manual_seed(77)
    x = torch.randn((n, d, h, w, cin), generator=g)
    w3 = torch.randn((3, 3, 3, cin, f), generator=g) * (2.0 / (27 * cin)) ** 0.5
    w1 = torch.randn((1, 1, 1, cin, f), generator=g) * (2.0 / cin) ** 0.5
    b3 = torch.randn(f, generator=g) * 0.3
    b1 = torch.randn(f, generator=g) * 0.3
    xr, w3r, w1r = _round(x, tdt), _round(w3, tdt), _round(w1, tdt)
    y_ref = R.conv3d(xr, w3r, b3.double())
    r_ref = R.conv3d(xr, w1r, b1.double())
    y_first = R.conv3d(xr[..., :32], w3r[:, :, :, :32], b3.double())        # what the first pass stores, rounded
    r_first = R.conv3d(xr[..., :32], w1r[:, :, :, :32], b1.double())
    y_bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), w3r.abs(), None) + u * (y_ref.abs() + y_first.abs()) + 1e-30
    r_bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), w1r.abs(), None) + u * (r_ref.abs() + r_first.abs()) + 1e-30
    norm = GroupNormalization(groups=G, axis=-1)
    norm.build((n, d, h, w, f))
    xd = x.to(tdt).to(DEV)
    if split:      # the two operands of the concat as dense tensors of their own (decoder.py:75), a guard block behind them
        buf = torch.full((3, n, d, h, w, 32), 7.0, dtype=tdt, device=DEV)
        buf[0].copy_(xd[..., :32])
        buf[1].copy_(xd[..., 32:])
        xin = buf[:2]
    else:
        xin = xd
    wp3 = lowp.pack(ops.K3S1, code, w3.to(DEV), cin, f)
    wp1 = lowp.pack(ops.K1, code, w1.to(DEV), cin, f)
    out, syms = _ran(lambda: lowp.conv_gn_shortcut(code, tdt, xin, wp3, b3.to(DEV), f, norm, wp1, b1.to(DEV)))
    assert out is not None
    y, mean, rstd, res, gap = out
    assert syms.count('lp_s1z_kernel') == 2 and not any(s.startswith('lp_k1') or s.startswith('lp_conv_gather') for s in syms), syms
    assert float(((y.double().cpu() - y_ref).abs() / y_bound).max()) <= 1.0
    assert float(((res.double().cpu() - r_ref).abs() / r_bound).max()) <= 1.0
    # statistics and squeeze of what was STORED plus at most the rounding of the last addition
    yd = y.double().cpu()
    chunks = yd.reshape(n, G, -1)
    assert float((mean.double().cpu() - chunks.mean(dim=2).reshape(-1)).abs().max()) <= 4 * u * float(yd.abs().mean()) + 1e-6
    s_ref = (chunks.var(dim=2, unbiased=False) + norm.epsilon).rsqrt().reshape(-1)
    assert float(((rstd.double().cpu() - s_ref).abs() / s_ref).max()) <= 4 * u
    gap_ref = r_ref.reshape(n, -1, f).mean(dim=1)
    assert float((gap.double().cpu() - gap_ref).abs().max()) <= 2 * u * float(r_ref.abs().mean()) + 1e-5
    # conv1 is the two-pass form either way: bit-equal to it
    y2, mean2, rstd2 = lowp.conv_gn(code, tdt, xd, wp3, b3.to(DEV), f, norm)
    torch.cuda.synchronize()
    assert torch.equal(y2, y)


def test_shapes_outside_the_streaming_kernel_decline():
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    code, tdt = lowp.DTYPES['bfloat16']
    g = torch.Generator().manual_seed(1)
    for shape, cin, f in (((1, 16, 20, 40), 32, 32), ((1, 16, 32, 64), 64, 32), ((1, 16, 32, 64), 32, 64)):      # (64 -> 32 below 8 M voxels: no two-pass form)
        n, d, h, w = shape
        x = torch.randn(shape + (cin,), generator=g).to(tdt).to(DEV)
        w3 = torch.randn((3, 3, 3, cin, f), generator=g).to(DEV)
        w1 = torch.randn((1, 1, 1, cin, f), generator=g).to(DEV)
        norm = GroupNormalization(groups=8, axis=-1)
        norm.build((n, d, h, w, f))
        b = torch.zeros(f, device=DEV)
        assert lowp.conv_gn_shortcut(code, tdt, x, lowp.pack(ops.K3S1, code, w3, cin, f), b, f, norm, lowp.pack(ops.K1, code, w1, cin, f), b) is None
