"""-m gpu: the LDS-DMA staged stride-1 3x3x3 convolution of the 16-bit storage path (csrc/lowp_s1d.hip; the Conv3D of
resnet.py:80-87 and, on role-swapped images, its data gradient under train.py:142-151) at shapes it TAKES -- the small cases of
test_lowp_gpu.py fall below its 4096-voxel floor and keep exercising the register-staged kernel.

Every case runs through the C ABI (bts_lp_conv3d_fwd / _bwd_data / _fwd_gn), is checked against the oracle's op on the same
16-bit-rounded operands in fp64 under the stated bound  |err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref|  (u = 2^-11 fp16, 2^-8 bf16;
+ u * |old| again where the result is accumulated into a stored value), and asserts through the library's launch records that
`lp_s1d_kernel` (not its fallback) produced it.  Covered: both work splits (32-cout items on 32x8x4 tiles, 64-cout items on
32x4x4 tiles), 32- and 16-wide tiles, ragged tiles on every axis, slab views on both sides, more items than workgroups (item
chaining), split-K with the fixed-order reduce, accumulation, the flipped-tap data-gradient image, the folded duplicate slice and
the fused GroupNorm partial sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


def _ran_s1d(fn):
    """run fn() with the library's launch records on -> (result, symbols launched)"""
    from bts_amd import ops
    ops.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    ops.profile_enable(False)
    return out, [s for s, _, _ in ops.profile_records()]


CASES = [
    # n, (D,H,W), Cin, Cout, slab_in, slab_out, what it reaches
    (2, (12, 20, 36), 32, 32, False, False),     # 32-cout items, 32-wide tiles, ragged on every axis
    (1, (16, 24, 40), 16, 64, True, True),       # 64-cout items, one k-step, slab views
    (2, (16, 24, 20), 64, 64, False, True),      # 16-wide tiles (two z planes per fragment), few items -> split-K + reduce
    (2, (20, 20, 20), 32, 32, True, False),      # 16-wide tiles, 32-cout items, ragged z / y / x
    (1, (36, 32, 64), 16, 128, False, False),    # 288 items on 256 workgroups: items chained, two cout groups per tile
    (1, (8, 48, 32), 48, 96, False, False),      # three cout blocks: the last group half empty
    (1, (20, 24, 20), 128, 128, False, False),   # the deepest grid of the 160x192x160 volume: 32-wide tiles on a 20-wide grid, split-K
    (2, (16, 16, 16), 64, 64, False, False),     # 16-wide tiles chosen by the plan's cost model (64-cout items)
    (1, (16, 16, 16), 32, 32, True, False),      # ... 32-cout items, 4096 voxels = the floor
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
    y, syms = _ran_s1d(lambda: lowp.conv(ops.K3S1, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout, out=out))
    assert 'lp_s1d_kernel' in syms and 'lp_conv_s1_kernel' not in syms, syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    if slab_out:      # neighbours of the output slice untouched
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_data_gradient(dtype, accumulate):
    """dx (+)= conv^T(dy): the flipped-tap image on the same kernel, written into a slab-gradient view (the 16-bit trainer
    accumulates conv1's data gradient into the level slab: lowp_train._block_bwd)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(21)
    n, d, h, w, cin, cout = 1, 16, 20, 40, 64, 32
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
    _, syms = _ran_s1d(lambda: lowp.conv_bwd_data(ops.K3S1, code, dy.to(tdt).to(DEV), wpb, dx, accumulate))
    assert 'lp_s1d_kernel' in syms and 'lp_conv_s1_kernel' not in syms, syms
    err = (dx.double().cpu() - ref.detach()).abs()
    assert float((err / bound).max()) <= 1.0
    assert bool((slab[..., :16] == 3.0).all()) and bool((slab[..., 16 + cin:] == 3.0).all())


def test_folded_duplicate_slice():
    """encoder.py:83-87: block j reads [o_{j-1}, o_0 .. o_{j-1}]; the slab [o_0 .. o_{j-1}] is read once with the duplicated slice
    folded into the packed weights -- the DMA part of the image folds like the first part"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES['float16']
    g = torch.Generator().manual_seed(3)
    f, j = 16, 2
    slab = torch.randn((1, 16, 24, 32, j * f), generator=g)
    wt = torch.randn((3, 3, 3, (j + 1) * f, 32), generator=g) * 0.05
    sr = _round(slab, tdt)
    full = torch.cat([sr[..., (j - 1) * f:], sr], dim=-1)
    ref = R.conv3d(full, wt.double(), None)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), (j + 1) * f, 32, j * f, (j - 1) * f, f)
    y, syms = _ran_s1d(lambda: lowp.conv(ops.K3S1, code, tdt, slab.to(tdt).to(DEV), wp, None, 32))
    assert 'lp_s1d_kernel' in syms, syms
    bound = (8 * 2.0 ** -24 + 2 * U['float16']) * R.conv3d(full.abs(), wt.double().abs(), None) + U['float16'] * ref.abs()
    assert float(((y.double().cpu() - ref).abs() / bound).max()) <= 1.0


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 16, 24, 32, 32, 32), (1, 16, 24, 36, 16, 64), (1, 16, 24, 36, 32, 24), (1, 16, 24, 32, 16, 40)],
                         ids=['32cout', '64cout-ragged', '24cout-ragged', '40cout'])
def test_fused_groupnorm_statistics(shape, dtype):
    """bts_lp_conv3d_fwd_gn: the slab-mode GroupNorm (sum, sumsq) partials leave the conv's output side (group_norm.py:100-107 on the
    conv of resnet.py:80-93): mean / rstd against the statistics of the unrounded fp64 conv result"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, cin, cout = shape
    G = 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    ref = R.conv3d(_round(x, tdt), _round(wt, tdt), b.double())
    norm = GroupNormalization(groups=G, axis=-1)
    norm.build((n, d, h, w, cout))
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)
    (y, mean, rstd), syms = _ran_s1d(lambda: lowp.conv_gn(code, tdt, x.to(tdt).to(DEV), wp, b.to(DEV), cout, norm))
    assert 'lp_s1d_kernel' in syms, syms
    # slab semantics: group g of a sample = the g-th contiguous 1/G chunk of its flattened (D,H,W,C) memory
    chunks = ref.reshape(n, G, -1)
    m_ref = chunks.mean(dim=2).reshape(-1)
    r_ref = (chunks.var(dim=2, unbiased=False) + norm.epsilon).rsqrt().reshape(-1)
    assert float((mean.double().cpu() - m_ref).abs().max()) <= 2e-5
    assert float(((rstd.double().cpu() - r_ref).abs() / r_ref).max()) <= 2e-5
    assert float(((y.double().cpu() - ref).abs() / (8 * 2.0 ** -24 * R.conv3d(_round(x, tdt).abs(), _round(wt, tdt).abs(), None) +
                                                     U[dtype] * ref.abs() + 1e-30)).max()) <= 1.0


def test_two_runs_are_bitwise_identical():
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES['bfloat16']
    g = torch.Generator().manual_seed(8)
    x = torch.randn((2, 16, 24, 20, 64), generator=g).to(tdt).to(DEV)       # split-K case: the reduce order is fixed too
    wt = (torch.randn((3, 3, 3, 64, 64), generator=g) * 0.03).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt, 64, 64)
    a = lowp.conv(ops.K3S1, code, tdt, x, wp, None, 64).clone()
    b = lowp.conv(ops.K3S1, code, tdt, x, wp, None, 64)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 16, 16, 16, 32, 32), (1, 16, 24, 40, 32, 32), (2, 16, 16, 16, 64, 64), (2, 16, 16, 16, 96, 32)],
                         ids=lambda c: '%dx%dx%dx%d-%d-%d' % c)
def test_repeated_launches_are_bitwise_identical(shape, dtype):
    """every variant, other work in between (round 6: a build whose fp16 32-cout variant stored through a run-time scalar offset returned
    different garbage on every launch while bf16 and the 64-cout variants were fine -- scripts/s1d_stress.py is the long form)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = (torch.randn((3, 3, 3, cin, cout), generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt, cin, cout)
    first, syms = _ran_s1d(lambda: lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout).clone())
    assert 'lp_s1d_kernel' in syms, syms
    for _ in range(6):
        torch.randn((1 << 18,), device=DEV).sin_()
        y = lowp.conv(ops.K3S1, code, tdt, x, wp, b, cout)
        torch.cuda.synchronize()
        assert torch.equal(y, first)
