"""-m gpu: the reduced-precision STORAGE forward (csrc/lowp.hip, bts_amd/lowp.py; BASELINE configs[4] fp16 inference, the forward
half of configs[2] in bf16).  The reference has no such mode; what is checked:

  * every kernel against the oracle's op on the SAME 16-bit-rounded operands, evaluated in fp64: the only differences left are
    the fp32 summation order and the final rounding of the result to the storage type.  Stated tolerance:
        |err| <= 8 * 2^-24 * sum|a_i b_i|  +  u * |ref|  (+ u * |ref| again where a second rounding is part of the op),
    u = 2^-11 (fp16) or 2^-8 (bf16), the storage type's unit round-off;
  * the whole 16-bit forward against the fp32 engine on the same weights and volume: probabilities, label map, Dice -- reported,
    and bounded by the tolerances stated at the test (fp16: |dp| <= 2e-2, <= 0.2 % label changes; bf16 8x the fp16 bounds).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


def _ref_conv(kind, x, w, b):
    from bts_amd import ops
    if kind == ops.K3S2T:
        return R.conv3d_transpose(x, w, b)
    return R.conv3d(x, w, b, stride=2 if kind == ops.K3S2 else 1)


def _abs_conv(kind, x, w):
    """sum |a_i b_i| per output: the same op on absolute values"""
    return _ref_conv(kind, x.abs(), w.abs(), None)


CONV_CASES = [
    # kind, (D,H,W), Cin, Cout, slab_in (ld > Cin), slab_out
    ('K3S1', (8, 8, 32), 32, 32, False, False),
    ('K3S1', (10, 12, 10), 32, 64, True, True),       # level-3 grid of the 160x192x160 volume: ragged tiles on every axis
    ('K3S1', (4, 8, 16), 64, 128, False, True),
    ('K3S1', (4, 4, 36), 16, 32, True, False),        # two x tiles, the second one ragged
    ('K1', (6, 5, 7), 48, 32, True, True),
    ('K1', (8, 8, 8), 16, 96, False, False),
    ('K3S2', (8, 12, 16), 32, 32, True, False),
    ('K3S2', (4, 4, 4), 64, 64, False, True),
    ('K3S2T', (4, 6, 5), 32, 32, False, True),
    ('K3S2T', (3, 3, 3), 64, 64, True, False),
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: '%s-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_conv_kernels(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    name, (d, h, w), cin, cout, slab_in, slab_out = case
    kind = getattr(ops, name)
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((name, d, h, w, cin, cout)) % 10000)
    n = 2
    x = torch.randn((n, d, h, w, cin), generator=g)
    k = 1 if kind == ops.K1 else 3
    wshape = (k, k, k, cout, cin) if kind == ops.K3S2T else (k, k, k, cin, cout)
    wt = torch.randn(wshape, generator=g) * (2.0 / (k ** 3 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = _ref_conv(kind, xr, wr, b.double())
    bound = 8 * 2.0 ** -24 * _abs_conv(kind, xr, wr) + U[dtype] * ref.abs() + 1e-30
    ldx = cin + 16 if slab_in else cin
    xin = torch.zeros((n, d, h, w, ldx), dtype=tdt, device=DEV)
    c0 = 16 if slab_in else 0
    xin[..., c0:c0 + cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(kind, code, wt.to(DEV), cin, cout)
    out = None
    if slab_out:
        buf = torch.full(tuple(ref.shape[:4]) + (cout + 24,), 7.0, dtype=tdt, device=DEV)
        out = buf[..., 8:8 + cout]
    y = lowp.conv(kind, code, tdt, xin[..., c0:c0 + cin], wp, b.to(DEV), cout, out=out)
    torch.cuda.synchronize()
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s %s: error %.3e is %.2fx the stated bound' % (name, dtype, float(err.max()), worst)
    if slab_out:      # neighbours of the output slice untouched
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())


def test_folded_duplicate_slice_pack():
    """encoder.py:83-87: block j reads [o_{j-1}, o_0 .. o_{j-1}]; the engine reads the slab [o_0 .. o_{j-1}] once with the duplicated
    slice folded into the weights"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES['float16']
    g = torch.Generator().manual_seed(3)
    f, j = 16, 2
    slab = torch.randn((1, 4, 4, 8, j * f), generator=g)
    wt = torch.randn((3, 3, 3, (j + 1) * f, 32), generator=g) * 0.05
    sr = _round(slab, tdt)
    full = torch.cat([sr[..., (j - 1) * f:], sr], dim=-1)                 # what the reference's Concatenate would build
    ref = R.conv3d(full, wt.double(), None)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), (j + 1) * f, 32, j * f, (j - 1) * f, f)
    y = lowp.conv(ops.K3S1, code, tdt, slab.to(tdt).to(DEV), wp, None, 32)
    torch.cuda.synchronize()
    # the folded weight W[first copy] + W[second copy] is rounded once more than either copy: 2 u on top of the conv bound
    bound = (8 * 2.0 ** -24 + 2 * U['float16']) * R.conv3d(full.abs(), wt.double().abs(), None) + U['float16'] * ref.abs()
    assert float(((y.double().cpu() - ref).abs() / bound).max()) <= 1.0


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('mode', ['slab', 'channel'])
def test_groupnorm_colsum_epilogue_head(dtype, mode):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    gmode = ops.GN_SLAB if mode == 'slab' else ops.GN_CHANNEL
    g = torch.Generator().manual_seed(11)
    n, d, h, w, c, G = 2, 8, 6, 8, 32, 8
    x = torch.randn((n, d, h, w, c), generator=g) * 1.7 + 0.4
    gamma, beta = 1 + 0.3 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    xr = _round(x, tdt)
    axis = -1 if mode == 'slab' else 1
    xx = xr if mode == 'slab' else xr.permute(0, 4, 1, 2, 3)
    ref = R.group_norm(xx, gamma.double(), beta.double(), G, axis)
    ref = torch.relu(ref if mode == 'slab' else ref.permute(0, 2, 3, 4, 1))
    xd = x.to(tdt).to(DEV)
    mean, rstd = lowp.gn_stats(code, xd, G, gmode, 1e-5)
    y = lowp.gn_apply(code, xd, gamma.to(DEV), beta.to(DEV), mean, rstd, G, gmode, True)
    torch.cuda.synchronize()
    err = (y.double().cpu() - ref).abs()
    assert float((err / (1e-5 * (1 + ref.abs()) + u * ref.abs() + 1e-30)).max()) <= 1.0
    # global average pool
    gap = lowp.colsum(code, xd, 1.0 / (d * h * w))
    torch.cuda.synchronize()
    assert float((gap.double().cpu() - xr.mean(dim=(1, 2, 3))).abs().max()) <= 1e-6
    # block epilogue: out = res * (sigmoid(res . wsp) + ch) + relu(GN2(c2))
    res = torch.randn((n, d, h, w, c), generator=g)
    wsp = torch.randn(c, generator=g) * 0.3
    ch = torch.rand((n, c), generator=g)
    rr = _round(res, tdt)
    sp = torch.sigmoid((rr * wsp.double()).sum(-1, keepdim=True))
    ref2 = rr * (sp + ch.double().reshape(n, 1, 1, 1, c)) + ref
    buf = torch.zeros((n, d, h, w, c + 16), dtype=tdt, device=DEV)
    out = lowp.block_epilogue(code, res.to(tdt).to(DEV), xd, buf[..., 8:8 + c], wsp.to(DEV), ch.to(DEV), gamma.to(DEV), beta.to(DEV),
                              mean, rstd, G, gmode)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref2).abs()
    assert float((err / (2e-5 * (1 + ref2.abs()) + u * ref2.abs() + 1e-30)).max()) <= 1.0
    # head: sigmoid(x . W + b) in fp32
    wk, bk = torch.randn((c, 3), generator=g) * 0.2, torch.randn(3, generator=g) * 0.1
    yh = lowp.head(code, xd, wk.to(DEV), bk.to(DEV), True)
    torch.cuda.synchronize()
    refh = torch.sigmoid(xr @ wk.double() + bk.double())
    assert yh.dtype == torch.float32 and float((yh.double().cpu() - refh).abs().max()) <= 2e-6


def _model(kw, crop, seed):
    from bts_amd.layers import _base
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    _base.set_seed(seed)
    m = Model(**kw)
    m.build((1,) + crop + (2,))
    g = torch.Generator().manual_seed(seed + 1)
    for p in m.trainable_variables:           # gamma_2 = 0 at init would hide the conv branch
        if p.name.endswith('gamma'):
            p.t.copy_((1.0 + 0.3 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
        elif p.name.endswith('beta') or p.t.dim() == 1:
            p.t.copy_((0.1 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
    bump_weights_epoch()
    return m


def _compare(y_lp, y_32, tag):
    d = (y_lp - y_32).abs()
    lab = lambda y: torch.where(y.max(-1).values > 0.5, y.argmax(-1) + 1, torch.zeros_like(y.argmax(-1)))
    l1, l2 = lab(y_lp), lab(y_32)
    mism = float((l1 != l2).float().mean())
    inter = [(float(((l1 == k) & (l2 == k)).sum()), float((l1 == k).sum() + (l2 == k).sum())) for k in (1, 2, 3)]
    dice = [2 * a / b if b > 0 else 1.0 for a, b in inter]
    print('%s: |dp| max %.3e mean %.3e ; label changes %.4f %% ; label-map Dice vs fp32 %s' %
          (tag, float(d.max()), float(d.mean()), 100 * mism, ' '.join('%.4f' % v for v in dice)))
    return float(d.max()), float(d.mean()), mism


@pytest.mark.parametrize('dtype,scale', [('float16', 1.0), ('bfloat16', 8.0)])
def test_forward_against_fp32_engine_and_oracle(dtype, scale):
    """tiny config at 32^3: 16-bit forward vs the fp32 engine AND vs the fp64 oracle (both on the fp32 master weights)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    kw = dict(base_filters=16, groups=8, reduction=2, depth=3)
    crop = (32, 32, 32)
    m = _model(kw, crop, 5)
    x = torch.randn((1,) + crop + (2,), generator=torch.Generator().manual_seed(9))
    y32 = m(x, training=False, inference=True)[0].t
    ylp = lowp.LowPrecisionForward(m, dtype)(x)
    torch.cuda.synchronize()
    assert ylp.dtype == torch.float32 and ylp.shape == y32.shape
    mx, mean, mism = _compare(ylp, y32, dtype + ' vs fp32 engine')
    assert mx <= 2e-2 * scale and mean <= 1e-3 * scale and mism <= 2e-3 * scale
    cfg = R.default_config(**kw)
    P = R.ParamSet()
    for p in m.trainable_variables:
        P[m.oracle_name(p)] = p.t.detach().cpu().double()
    yo = R.model(x.double(), P, cfg, training=False, inference=True)[0]
    mx, mean, mism = _compare(ylp.cpu().double(), yo, dtype + ' vs fp64 oracle')
    assert mx <= 2e-2 * scale and mean <= 1e-3 * scale and mism <= 2e-3 * scale


def test_full_volume_fp16_inference_against_fp32_engine():
    """BASELINE configs[4]: 155x190x147 padded to 160x192x160 (test.py:164-178), CLI-default model, VAE off, fp16 storage"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    m = _model(dict(base_filters=32, reduction=8, depth=4, groups=8), (128, 128, 128), 77)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 160, 192, 160, 2), generator=g)
    x[:, 155:] = 0
    x[:, :, 190:] = 0
    x[:, :, :, 147:] = 0
    x = x.to(DEV)
    y32 = m(x, training=False, inference=True)[0].t
    ylp = lowp.LowPrecisionForward(m, 'float16')(x)
    torch.cuda.synchronize()
    mx, mean, mism = _compare(ylp, y32, 'fp16 160x192x160 vs fp32 engine')
    assert mx <= 5e-2 and mean <= 2e-3 and mism <= 5e-3


def test_segment_volume_in_fp16_against_the_fp32_pipeline():
    """infer.segment_volume(..., compute_dtype='float16'): pad, 8 flip-TTA forwards in 16-bit storage, un-flip + mean + mask + labels"""
    import bts_amd  # noqa: F401
    from bts_amd import infer
    m = _model(dict(base_filters=16, groups=8, reduction=2, depth=3), (16, 16, 24), 3)
    g = torch.Generator().manual_seed(21)
    vol = (13, 9, 20)
    x = torch.randn(vol + (2,), generator=g) * 40.0 + 100.0
    mask = (torch.rand(vol + (1,), generator=g) > 0.15).float()
    x = x * mask
    mean, std = torch.tensor([95.0, 110.0]), torch.tensor([35.0, 45.0])
    y32, l32 = infer.segment_volume(m, x.to(DEV), mask.to(DEV), mean, std, 4)
    y16, l16 = infer.segment_volume(m, x.to(DEV), mask.to(DEV), mean, std, 4, compute_dtype='float16')
    torch.cuda.synchronize()
    assert y16.shape == y32.shape and l16.dtype == torch.uint8
    d = (y16 - y32).abs()
    mism = float((l16 != l32).float().mean())
    print('TTA pipeline fp16 vs fp32: |dp| max %.3e mean %.3e, label changes %.3f %%' % (float(d.max()), float(d.mean()), 100 * mism))
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 1e-3 and mism <= 5e-3
    assert set(l16.cpu().unique().tolist()) <= {0, 1, 2, 4}


BWD_CASES = [
    ('K3S1', (8, 8, 32), 32, 32), ('K3S1', (10, 12, 10), 48, 64), ('K1', (6, 5, 7), 48, 32),
    ('K3S2', (8, 12, 16), 32, 32), ('K3S2', (4, 4, 4), 48, 64), ('K3S2T', (4, 6, 5), 32, 32), ('K3S2T', (3, 3, 3), 64, 48),
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('accumulate', [False, True])
@pytest.mark.parametrize('case', BWD_CASES, ids=lambda c: '%s-%dx%dx%d-%d-%d' % (c[0], *c[1], c[2], c[3]))
def test_conv_data_gradient_kernels(case, accumulate, dtype):
    """dx (+)= conv^T(dy) on role-swapped 16-bit weight images against torch autograd of the oracle's forward op on the same
    rounded dy / w (fp64).  (D,H,W) = forward input dims; the gradient lands in a channel slice of a wider slab."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    name, (d, h, w), cin, cout = case
    kind = getattr(ops, name)
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(hash((name, d, h, w, cin, cout, 'b')) % 10000)
    n = 2
    k = 1 if kind == ops.K1 else 3
    wshape = (k, k, k, cout, cin) if kind == ops.K3S2T else (k, k, k, cin, cout)
    wt = torch.randn(wshape, generator=g) * (2.0 / (k ** 3 * cout)) ** 0.5
    wr = _round(wt, tdt)

    x0 = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    yshape = _ref_conv(kind, x0, wr, None).shape
    dy = torch.randn(tuple(yshape), generator=g)
    dyr = _round(dy, tdt)
    ref = torch.autograd.grad(_ref_conv(kind, x0, wr, None), x0, dyr)[0]
    x1 = torch.zeros((n, d, h, w, cin), dtype=torch.float64, requires_grad=True)
    mag = torch.autograd.grad(_ref_conv(kind, x1, wr.abs(), None), x1, dyr.abs())[0]
    old = torch.randn((n, d, h, w, cin), generator=g)
    oldr = _round(old, tdt)
    want = ref + (oldr if accumulate else 0.0)
    bound = 8 * 2.0 ** -24 * mag + u * want.abs() + 1e-30
    slab = torch.full((n, d, h, w, cin + 16), 5.0, dtype=tdt, device=DEV)
    dx = slab[..., 8:8 + cin]
    dx.copy_(old.to(tdt).to(DEV))
    wpb = lowp.pack(kind, code, wt.to(DEV), cin, cout, role=ops.ROLE_BWD)
    lowp.conv_bwd_data(kind, code, dy.to(tdt).to(DEV), wpb, dx, accumulate)
    torch.cuda.synchronize()
    err = (dx.double().cpu() - want).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s %s: error %.3e is %.2fx the stated bound' % (name, dtype, float(err.max()), worst)
    assert bool((slab[..., :8] == 5.0).all()) and bool((slab[..., 8 + cin:] == 5.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 8, 16, 16, 32), (1, 16, 16, 16, 64), (2, 8, 8, 16, 256)])
def test_groupnorm_backward_kernel(shape, dtype):
    """16-bit GroupNorm(+ReLU) backward against the fp32 engine's kernel (itself pinned to the oracle by tests/test_kernels_gpu.py) on
    the same rounded x / dy: dx in both precisions, dgamma / dbeta; dy arrives as a channel slice of a wider slab"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, c = shape
    G = 8
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(tdt).to(DEV)
    dyslab = torch.randn((n, d, h, w, c + 16), generator=g).to(tdt).to(DEV)
    dy = dyslab[..., 8:8 + c]
    gamma = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(c, generator=g)).to(DEV)
    mean, rstd = lowp.gn_stats(code, x, G, ops.GN_SLAB, 1e-5)
    x32, dy32 = lowp.uncast(code, x), lowp.uncast(code, dy)
    dg_r, db_r = torch.full((c,), 0.5, device=DEV), torch.full((c,), -0.25, device=DEV)
    dx_r = ops.gn_bwd(x32, dy32, gamma, beta, mean, rstd, dg_r, db_r, G, ops.GN_SLAB, True, accumulate_params=True)
    dg, db = torch.full((c,), 0.5, device=DEV), torch.full((c,), -0.25, device=DEV)
    out = lowp.gn_bwd(code, tdt, x, dy, gamma, beta, mean, rstd, dg, db, G, True)
    assert out is not None
    dx16, dx32 = out
    torch.cuda.synchronize()
    scale = float(dx_r.abs().max())
    assert float((dx32 - dx_r).abs().max()) <= 2e-5 * scale + 1e-6
    assert float((dx16.float() - dx_r).abs().max()) <= U[dtype] * scale * 1.01 + 2e-5 * scale
    for a, b in ((dg, dg_r), (db, db_r)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-5


WG_CASES = [
    # kind, (N,D,H,W), Cin, Cout, fold (dup_start, dup_shift) or None, x is a slab view
    ('K3S1', (2, 8, 8, 16), 32, 32, None, False),
    ('K3S1', (1, 6, 10, 20), 64, 64, None, True),          # ragged tiles, two cout blocks per wave
    ('K3S1', (2, 4, 8, 16), 32, 16, (16, 16), True),       # folded duplicate slice: both copies of the weight get the gradient
    ('K1', (2, 8, 8, 16), 48, 32, None, True),
    ('K1', (1, 4, 8, 16), 32, 128, (16, 16), False),
    # the streaming kernel of lowp_wgd.hip (W % 32 == 0, H % 8 == 0): several columns; z chunks with a ragged last one, two
    # cin blocks and a half-empty cout block on a slab view; the folded duplicate slice
    ('K3S1', (2, 8, 32, 64), 32, 32, None, False),
    ('K3S1', (1, 37, 16, 32), 64, 16, None, True),
    ('K3S1', (2, 16, 16, 32), 32, 32, (16, 16), True),
    ('K3S1', (2, 16, 16, 32), 64, 128, None, False),        # two cin blocks x four cout blocks
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', WG_CASES, ids=lambda c: '%s-%s-%d-%d-%s' % (c[0], 'x'.join(map(str, c[1])), c[2], c[3], 'fold' if c[4] else 'plain'))
def test_weight_gradient_kernel(case, dtype):
    """16-bit weight gradient against torch autograd of the oracle's conv on the same rounded x / dy (fp64), accumulated onto a
    non-zero buffer; the folded case checks that W[first copy] and W[second copy] both receive the slab slice's gradient"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    name, (n, d, h, w), cin, cout, fold, slab = case
    kind = getattr(ops, name)
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 7 + cout)
    k = 1 if kind == ops.K1 else 3
    dup_start, dup_shift = fold if fold else (0, 0)
    cin_ref = cin + dup_shift
    x = torch.randn((n, d, h, w, cin), generator=g)
    dy = torch.randn((n, d, h, w, cout), generator=g)
    xr, dyr = _round(x, tdt), _round(dy, tdt)
    # what the reference's block would have seen: [slab[dup_start:dup_start+shift], slab]
    full = torch.cat([xr[..., dup_start:dup_start + dup_shift], xr], dim=-1) if fold else xr
    wz = torch.zeros((k, k, k, cin_ref, cout), dtype=torch.float64, requires_grad=True)
    bz = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    dw_ref, db_ref = torch.autograd.grad(R.conv3d(full, wz, bz), (wz, bz), dyr)
    wz2 = torch.zeros((k, k, k, cin_ref, cout), dtype=torch.float64, requires_grad=True)
    mag = torch.autograd.grad(R.conv3d(full.abs(), wz2, None), wz2, dyr.abs())[0]
    dw0 = torch.randn(dw_ref.shape, generator=g)
    db0 = torch.randn(cout, generator=g)
    buf = torch.zeros((n, d, h, w, cin + 16), dtype=tdt, device=DEV)
    xin = buf[..., 8:8 + cin] if slab else torch.empty((n, d, h, w, cin), dtype=tdt, device=DEV)
    xin.copy_(x.to(tdt).to(DEV))
    dw, db = dw0.to(DEV).contiguous(), db0.to(DEV).contiguous()
    streams = name == 'K3S1' and w % 32 == 0 and h % 8 == 0 and n * (h // 8) * (w // 32) * d >= 64
    ops.profile_enable(True)
    ok = lowp.conv_bwd_weight(kind, code, xin, dy.to(tdt).to(DEV), dw, db, dup_start, dup_shift, accumulate=True)
    assert ok
    torch.cuda.synchronize()
    ops.profile_enable(False)
    ran = [r[0] for r in ops.profile_records()]
    assert ('lp_wgd_kernel' in ran) == streams, ran
    err = (dw.double().cpu() - (dw_ref + dw0.double())).abs()
    bound = 8 * 2.0 ** -24 * mag + 2.0 ** -22 * (dw_ref.abs() + dw0.double().abs()) + 1e-9
    assert float((err / bound).max()) <= 1.0, 'dw: max err %.3e at %.2fx the bound' % (float(err.max()), float((err / bound).max()))
    eb = (db.double().cpu() - (db_ref + db0.double())).abs()
    assert float(eb.max()) <= 1e-5 * float(db_ref.abs().max()) + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 8, 16, 16, 16), (1, 6, 10, 12, 32), (2, 4, 4, 8, 128), (1, 2, 4, 4, 256)])
def test_gate_backward_kernel(shape, dtype):
    """16-bit gate (squeeze-excitation) backward against the fp32 engine's kernel (pinned to the oracle by tests/test_kernels_gpu.py) on
    the same rounded dout / res: dres in the storage type, SE-MLP and spatial-gate parameter gradients accumulated onto non-zero
    buffers; dout arrives as a channel slice of a wider slab; ragged voxel counts"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, f = shape
    r = f // 2
    g = torch.Generator().manual_seed(f + d)
    res = torch.randn(shape, generator=g).to(tdt).to(DEV)
    slab = torch.randn((n, d, h, w, f + 16), generator=g).to(tdt).to(DEV)
    dout = slab[..., 8:8 + f]
    sp = torch.sigmoid(torch.randn((n, d, h, w, 1), generator=g)).to(DEV).contiguous()
    gap = torch.randn((n, f), generator=g).to(DEV)
    w1 = (0.3 * torch.randn((f, r), generator=g)).to(DEV)
    w2 = (0.3 * torch.randn((r, f), generator=g)).to(DEV)
    wsp = (0.3 * torch.randn(f, generator=g)).to(DEV)
    hbuf, ch = ops.se_mlp_fwd(gap, w1, w2)
    res32, dout32 = lowp.uncast(code, res), lowp.uncast(code, dout)
    init = [torch.randn(t.shape, generator=g).to(DEV) for t in (w1, w2, wsp)]
    ref_g = [t.clone() for t in init]
    dres_r = ops.se_bwd(dout32, res32, sp, gap, hbuf, ch, w1, w2, wsp, *ref_g, accumulate_params=True)
    got_g = [t.clone() for t in init]
    dres = lowp.se_bwd(code, tdt, dout, res, sp, gap, hbuf, ch, w1, w2, wsp, *got_g)
    torch.cuda.synchronize()
    assert dres.dtype == tdt and dres.shape == res.shape
    scale = float(dres_r.abs().max())
    assert float((dres.float() - dres_r).abs().max()) <= U[dtype] * scale * 1.01 + 2e-5 * scale
    for a, b in zip(got_g, ref_g):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 16, 16, 32), 48, 32, True), ((2, 8, 16, 16), 32, 64, False), ((1, 6, 10, 12), 16, 32, False)],
                         ids=['slab-48-32', 'batch2-32-64', 'ragged-fallback'])
def test_shortcut_conv_with_fused_squeeze(case, dtype):
    """bts_lp_conv1_gap: the 1x1x1 shortcut conv and the mean over voxels of its output in one pass -- res equal to the plain 16-bit
    conv's (same kernel), gap against the fp64 mean of the oracle's conv on the rounded operands; input as a slab view; a volume that
    is not whole position blocks takes the two-pass route"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, slab = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((1, 1, 1, cin, cout), generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    buf = torch.zeros((n, d, h, w, cin + 16), dtype=tdt, device=DEV)
    xin = buf[..., 8:8 + cin] if slab else torch.empty((n, d, h, w, cin), dtype=tdt, device=DEV)
    xin.copy_(x.to(tdt).to(DEV))
    wp = lowp.pack(ops.K1, code, wt.to(DEV), cin, cout)
    res, gap = lowp.conv1_gap(code, xin, wp, b.to(DEV), cout, tdt)
    ref_res = lowp.conv(ops.K1, code, tdt, xin, wp, b.to(DEV), cout)
    torch.cuda.synchronize()
    assert torch.equal(res, ref_res)
    y64 = R.conv3d(_round(x, tdt), _round(wt, tdt), b.double())
    ref_gap = y64.mean(dim=(1, 2, 3))
    fused = (d * h * w) % 512 == 0 or (d * h * w) % 128 == 0 and n * d * h * w < 128 * 512
    # fused route: fp32 sums of unrounded outputs; two-pass route (ragged volumes): sums of the STORED, rounded res -- the
    # rounding noise of V values averages down by sqrt(V) only
    tol = (3e-6 if fused else U[dtype]) * float(y64.abs().mean()) + 2e-6
    assert float((gap.double().cpu() - ref_gap).abs().max()) <= tol, (fused, float((gap.double().cpu() - ref_gap).abs().max()), tol)


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 32, 16, 32), 32, 32, 8), ((2, 16, 8, 16), 16, 64, 4), ((1, 16, 16, 16), 128, 32, 2), ((1, 12, 8, 16), 16, 32, 2)],
                         ids=['32-32-g8', 'batch2-16-64-g4', 'splitk-fallback', 'slab-not-whole-tiles'])
def test_conv_with_fused_groupnorm_statistics(case, dtype):
    """bts_lp_conv3d_fwd_gn: y bit-equal to the plain 16-bit conv's, (mean, rstd) against the fp64 slab statistics of the oracle's
    conv on the rounded operands (the fused sums see the unrounded outputs, the two-pass route the stored ones: both within the
    storage type's rounding of a value, divided by the square root of the group size)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, G = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g)
    xin = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)

    class _Norm(object):
        groups, epsilon, _mode = G, 1e-5, ops.GN_SLAB
    y, mean, rstd = lowp.conv_gn(code, tdt, xin, wp, b.to(DEV), cout, _Norm)
    y_ref = lowp.conv(ops.K3S1, code, tdt, xin, wp, b.to(DEV), cout)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    y64 = R.conv3d(_round(x, tdt), _round(wt, tdt), b.double())            # (n, d, h, w, cout)
    slabs = y64.reshape(n, G, -1)                                           # slab semantics: G equal chunks of the flattened sample
    m_ref = slabs.mean(dim=2).reshape(-1)
    r_ref = 1.0 / torch.sqrt(slabs.var(dim=2, unbiased=False) + 1e-5).reshape(-1)
    scale = float(y64.abs().mean())
    assert float((mean.double().cpu() - m_ref).abs().max()) <= U[dtype] * scale * 0.05 + 1e-5
    assert float((rstd.double().cpu() / r_ref - 1).abs().max()) <= U[dtype] * 0.05 + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 16, 64, 64), 32, 32, 8, 8), ((2, 8, 64, 64), 16, 32, 4, 8), ((1, 24, 48, 96), 32, 16, 8, 4), ((1, 64, 32, 32), 32, 32, 8, 8),
                                  ((1, 16, 64, 64), 32, 32, 2, 8)],
                         ids=['32-32-g8', 'batch2-16-32-g4', '16-couts-three-columns', 'z-chunks', 'classes-of-16-declined'])
def test_conv_with_groupnorm_applied_to_its_input_planes(case, dtype):
    """bts_lp_conv3d_gnin_fwd_gn (conv2 of a block reading conv1's RAW output in a forward without a backward, resnet.py:133-136): the
    arithmetic of bts_lp_gn_apply on the planes in LDS, zero padding after the normalisation -- y, mean, rstd bit-equal to
    gn_apply + conv_gn; shapes / class counts the kernel does not take in this form are declined (None)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, gin, gout = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 7 + cout + d)
    c1 = (torch.randn((n, d, h, w, cin), generator=g) * 1.5 + 0.3).to(tdt).to(DEV)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)

    class _P(object):
        def __init__(self, t):
            self.t = t

    class _NormIn(object):
        groups, epsilon, _mode = gin, 1e-5, ops.GN_SLAB
        gamma = _P((1.0 + 0.3 * torch.randn(cin, generator=g)).to(DEV))
        beta = _P((0.3 * torch.randn(cin, generator=g)).to(DEV))

    class _NormOut(object):
        groups, epsilon, _mode = gout, 1e-5, ops.GN_SLAB
    m1, r1 = lowp.gn_stats(code, c1, gin, ops.GN_SLAB, 1e-5)
    got = lowp.conv_gn_normed_input(code, tdt, c1, _NormIn, m1, r1, True, wp, b, cout, _NormOut)
    if cin // gin > 8:
        assert got is None
        return
    assert got is not None
    a = lowp.gn_apply(code, c1, _NormIn.gamma.t, _NormIn.beta.t, m1, r1, gin, ops.GN_SLAB, True)
    y_ref, m_ref, r_ref = lowp.conv_gn(code, tdt, a, wp, b, cout, _NormOut)
    torch.cuda.synchronize()
    assert float(a.float().abs().max()) > 0 and float((a == 0).float().mean()) > 0.05       # (the ReLU bites)
    assert torch.equal(got[0], y_ref)
    assert torch.equal(got[1], m_ref) and torch.equal(got[2], r_ref)


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 16, 64, 64), 32, 32, 8), ((2, 8, 64, 64), 16, 32, 4), ((1, 64, 32, 32), 32, 16, 8)],
                         ids=['32-32-g8', 'batch2-16-32-g4', 'z-chunks-16-couts'])
def test_weight_gradient_with_groupnorm_applied_to_its_input_planes(case, dtype):
    """bts_lp_conv3d_gnin_bwd_weight (conv2's weight gradient from conv1's RAW output: the streaming kernel normalises its planes in LDS
    with bts_lp_gn_apply's arithmetic) against the same kernel on the materialised relu(GN(x)): dw and db bit-equal, first write and
    accumulate"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, gin = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 11 + cout + d)
    c1 = (torch.randn((n, d, h, w, cin), generator=g) * 1.5 + 0.3).to(tdt).to(DEV)
    dy = torch.randn((n, d, h, w, cout), generator=g).to(tdt).to(DEV)

    class _P(object):
        def __init__(self, t):
            self.t = t

    class _NormIn(object):
        groups, epsilon, _mode = gin, 1e-5, ops.GN_SLAB
        gamma = _P((1.0 + 0.3 * torch.randn(cin, generator=g)).to(DEV))
        beta = _P((0.3 * torch.randn(cin, generator=g)).to(DEV))

    class _NormOut(object):
        groups, epsilon, _mode = 8, 1e-5, ops.GN_SLAB
    assert lowp.gnin_train_ok(c1, cout, _NormIn, _NormOut)
    m1, r1 = lowp.gn_stats(code, c1, gin, ops.GN_SLAB, 1e-5)
    a = lowp.gn_apply(code, c1, _NormIn.gamma.t, _NormIn.beta.t, m1, r1, gin, ops.GN_SLAB, True)
    for accumulate in (False, True):
        init_w = torch.randn((3, 3, 3, cin, cout), generator=g).to(DEV)
        init_b = torch.randn(cout, generator=g).to(DEV)
        dw_ref, db_ref = init_w.clone(), init_b.clone()
        assert lowp.conv_bwd_weight(ops.K3S1, code, a, dy, dw_ref, db_ref, accumulate=accumulate) is not False
        dw, db = init_w.clone(), init_b.clone()
        lowp.conv_bwd_weight_normed_input(code, c1, _NormIn, m1, r1, dy, dw, db, accumulate=accumulate)
        torch.cuda.synchronize()
        assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
        assert float((dw - init_w).abs().max()) > 0


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 20, 24, 20), 256, 256, 8), ((1, 8, 12, 20), 128, 128, 8), ((2, 8, 8, 8), 256, 64, 4)],
                         ids=['deepest-inference-level', '128-128', 'batch2-256-64-g4'])
def test_split_channel_conv_leaves_the_groupnorm_statistics_of_its_stored_output(case, dtype):
    """Layers whose grid splits the input channels (few voxels, many channels: the 20x24x20 level of the full inference volume) get
    their GroupNorm statistics from the split's finish (lp_splitk_reduce_gn_kernel): the sums are those of the STORED values, as in
    the statistics pass: y bit-equal, mean and rstd equal to conv + bts_lp_gn_stats up to the order of the fp64 partial sums."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, G = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin + cout + d)
    xin = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g).to(DEV)
    wp = lowp.pack(ops.K3S1, code, wt.to(DEV), cin, cout)

    class _Norm(object):
        groups, epsilon, _mode = G, 1e-5, ops.GN_SLAB
    y, mean, rstd = lowp.conv_gn(code, tdt, xin, wp, b, cout, _Norm)
    y_ref = lowp.conv(ops.K3S1, code, tdt, xin, wp, b, cout)
    m_ref, r_ref = lowp.gn_stats(code, y_ref, G, ops.GN_SLAB, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    assert float((mean - m_ref).abs().max()) <= 1e-6 and float((rstd / r_ref - 1).abs().max()) <= 1e-6      # (same values, other partial sums)
    sl = y_ref.double().reshape(n, G, -1)
    assert float((mean.double() - sl.mean(dim=2).reshape(-1)).abs().max()) <= 1e-5
    assert float((rstd.double() * torch.sqrt(sl.var(dim=2, unbiased=False) + 1e-5).reshape(-1) - 1).abs().max()) <= 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', [((1, 16, 16, 32), 32, 32, 8), ((2, 8, 8, 16), 64, 64, 8), ((1, 12, 10, 24), 32, 16, 4), ((1, 6, 8, 16), 16, 32, 8),
                                  ((1, 4, 4, 8), 32, 32, 8)],
                         ids=['32-32-g8', 'batch2-64-64-two-cout-blocks', 'ragged-16-couts-g4', 'slab-not-whole-planes', 'too-small-for-the-merged-kernel'])
def test_transposed_conv_with_fused_groupnorm_statistics(case, dtype):
    """bts_lp_convT3d_fwd_gn (ConvUpsample's conv -> GroupNormalization, upsample.py:28-43): y bit-equal to the plain 16-bit transposed
    conv's, (mean, rstd) against the fp64 slab statistics of the oracle's transposed conv on the rounded operands; cases where the
    merged kernel declines or a group is not whole fine planes take the two-pass route inside the same entry point"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, G = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 5 + cout + d)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cout, cin), generator=g) * (2.0 / (27 * cin / 8)) ** 0.5
    b = torch.randn(cout, generator=g)
    xin = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S2T, code, wt.to(DEV), cin, cout)

    class _Norm(object):
        groups, epsilon, _mode = G, 1e-5, ops.GN_SLAB
    y, mean, rstd = lowp.convT_gn(code, tdt, xin, wp, b.to(DEV), cout, _Norm)
    y_ref = lowp.conv(ops.K3S2T, code, tdt, xin, wp, b.to(DEV), cout)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (n, 2 * d, 2 * h, 2 * w, cout) and torch.equal(y, y_ref)
    y64 = R.conv3d_transpose(_round(x, tdt), _round(wt, tdt), b.double())
    slabs = y64.reshape(n, G, -1)
    m_ref = slabs.mean(dim=2).reshape(-1)
    r_ref = 1.0 / torch.sqrt(slabs.var(dim=2, unbiased=False) + 1e-5).reshape(-1)
    scale = float(y64.abs().mean())
    assert float((mean.double().cpu() - m_ref).abs().max()) <= U[dtype] * scale * 0.05 + 1e-5
    assert float((rstd.double().cpu() / r_ref - 1).abs().max()) <= U[dtype] * 0.05 + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_backward_apply_passes_emit_the_producing_convs_bias_gradient(dtype):
    """bts_lp_gn_bwd / bts_lp_se_bwd with dbias: the column sums of the tensor the pass writes (the dy of the conv in front of the
    GroupNorm, resnet.py:80-93; of the 1x1x1 shortcut, resnet.py:96-103) leave from the same pass; they must equal the sums of the
    fp32 values the pass computed, added to what the slot held"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(23)
    n, d, h, w, c, G = 2, 8, 16, 16, 32, 8
    x = (torch.randn((n, d, h, w, c), generator=g) * 1.5).to(tdt).to(DEV)
    dy = torch.randn((n, d, h, w, c), generator=g).to(tdt).to(DEV)
    gamma, beta = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV), (0.2 * torch.randn(c, generator=g)).to(DEV)
    mean, rstd = lowp.gn_stats(code, x, G, ops.GN_SLAB, 1e-5)
    dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    db = torch.full((c,), 0.25, device=DEV)
    dx16, dx32 = lowp.gn_bwd(code, tdt, x, dy, gamma, beta, mean, rstd, dgam, dbet, G, True, want_f32=True, dbias=db)
    torch.cuda.synchronize()
    want = dx32.double().sum(dim=(0, 1, 2, 3)).cpu() + 0.25
    assert float((db.double().cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    # gate backward
    f, r = 32, 4
    res = torch.randn((n, d, h, w, f), generator=g).to(tdt).to(DEV)
    dout = torch.randn((n, d, h, w, f), generator=g).to(tdt).to(DEV)
    wsp = (torch.randn(f, generator=g) * 0.3).to(DEV)
    w1, w2 = (torch.randn((f, r), generator=g) * 0.3).to(DEV), (torch.randn((r, f), generator=g) * 0.3).to(DEV)
    gap = res.float().mean(dim=(1, 2, 3)).contiguous()
    hb, ch = ops.se_mlp_fwd(gap, w1, w2)
    sp = torch.sigmoid((res.float() * wsp).sum(-1)).reshape(-1).contiguous()
    dw1, dw2, dwsp = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(wsp)
    db2 = torch.full((f,), -0.5, device=DEV)
    dres = lowp.se_bwd(code, tdt, dout, res, sp, gap, hb, ch, w1, w2, wsp, dw1, dw2, dwsp, dbias=db2)
    torch.cuda.synchronize()
    want2 = dres.double().sum(dim=(0, 1, 2, 3)).cpu() - 0.5         # (sums of the ROUNDED values: the kernel adds the unrounded ones)
    u = U[dtype]
    tol = u * float(dres.double().abs().sum(dim=(0, 1, 2, 3)).max()) / (n * d * h * w) ** 0.5 * 8 + 1e-4
    assert float((db2.double().cpu() - want2).abs().max()) <= tol


# ---- the non-default samplers on 16-bit tensors (args.py:136-141) ---------------------------------------------------------------
@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape,view', [((2, 8, 12, 20, 32), None), ((1, 4, 6, 10, 24), (8, 16)), ((1, 16, 16, 16, 8), None)])
def test_maxpool2_and_its_gradient_are_exact(dtype, shape, view):
    """MaxPooling3D(2) selects stored values: bit-exact against torch on the same 16-bit tensor, ties included (values drawn from a
    small set); the gradient lands on the first maximum in scan order like the fp32 engine's kernel"""
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(5)
    n, d, h, w, c = shape
    full = (torch.randint(-6, 7, shape, generator=g).float() * 0.25).to(tdt).cuda()
    x = full if view is None else full[..., view[0]:view[0] + view[1]]
    y, idx = lowp.maxpool2(code, x)
    ref = torch.nn.functional.max_pool3d(x.float().permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)
    assert torch.equal(y.float(), ref)
    y2, none = lowp.maxpool2(code, x, want_idx=False)
    assert none is None and torch.equal(y2, y)
    # against the fp32 engine's kernel (same first-maximum rule)
    y32, idx32 = ops.maxpool2_fwd(x.float().contiguous())
    assert torch.equal(idx, idx32) and torch.equal(y.float(), y32)
    dy = (torch.randint(-8, 9, tuple(y.shape), generator=g).float() * 0.125).to(tdt).cuda()
    dx = torch.full(tuple(x.shape), 1.0, dtype=tdt, device='cuda')
    lowp.maxpool2_bwd(code, dy, idx, dx, True)
    dref = torch.zeros(tuple(x.shape), dtype=torch.float32, device='cuda')
    ops.maxpool2_bwd(dy.float(), idx32, dref, False)
    assert torch.equal(dx.float(), dref + 1.0)
    lowp.maxpool2_bwd(code, dy, idx, dx, False)
    assert torch.equal(dx.float(), dref)


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_upsample2_and_its_gradient(dtype):
    from bts_amd import lowp
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(6)
    x = torch.randn((2, 4, 6, 10, 24), generator=g).to(tdt).cuda()
    slab = torch.zeros((2, 8, 12, 20, 64), dtype=tdt, device='cuda')
    lowp.upsample2(code, x, out=slab[..., 16:40])
    ref = x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    assert torch.equal(slab[..., 16:40], ref) and float(slab[..., :16].abs().max()) == 0 and float(slab[..., 40:].abs().max()) == 0
    dy = torch.randn((2, 8, 12, 20, 64), generator=g).to(tdt).cuda()
    dx = lowp.upsample2_bwd(code, dy[..., 16:40])
    r = dy[..., 16:40].float().reshape(2, 4, 2, 6, 2, 10, 2, 24).sum(dim=(2, 4, 6))
    assert torch.equal(dx, r.to(tdt))           # fp32 sum of 8 stored values, rounded once
    old = torch.randn(tuple(dx.shape), generator=g).to(tdt).cuda()
    acc = old.clone()
    lowp.upsample2_bwd(code, dy[..., 16:40], dx=acc, accumulate=True)
    assert float((acc.float() - (r + old.float())).abs().max()) <= 2.0 ** (-7 if dtype == 'bfloat16' else -10) * float(r.abs().max() + 3)


def test_batched_weight_packing_equals_the_single_calls():
    """bts_lp_pack_batch (one launch for every image of the trainer) writes bit-for-bit what bts_lp_pack writes image by image: all four
    kinds, both roles, a folded duplicate slice, a 2-channel first block (K < one matrix step), partly filled cout blocks"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    g = torch.Generator().manual_seed(77)
    for dtype in ('float16', 'bfloat16'):
        code, tdt = lowp.DTYPES[dtype]
        cases = [(ops.K3S1, ops.ROLE_FWD, 32, 32, None, 0, 0), (ops.K3S1, ops.ROLE_BWD, 64, 24, None, 0, 0), (ops.K3S1, ops.ROLE_FWD, 48, 64, 32, 16, 16),
                 (ops.K3S1, ops.ROLE_BWD, 48, 64, 32, 16, 16), (ops.K3S1, ops.ROLE_FWD, 2, 32, None, 0, 0), (ops.K1, ops.ROLE_FWD, 48, 32, 32, 16, 16),
                 (ops.K1, ops.ROLE_BWD, 32, 8, None, 0, 0), (ops.K3S2, ops.ROLE_FWD, 32, 64, None, 0, 0), (ops.K3S2, ops.ROLE_BWD, 32, 64, None, 0, 0),
                 (ops.K3S2T, ops.ROLE_FWD, 128, 64, None, 0, 0), (ops.K3S2T, ops.ROLE_BWD, 16, 8, None, 0, 0)]
        entries, singles = [], []
        for kind, role, cin_ref, cout, cin_slab, ds, dsh in cases:
            k = 1 if kind == ops.K1 else 3
            shape = (k, k, k, cout, cin_ref) if kind == ops.K3S2T else (k, k, k, cin_ref, cout)
            w = torch.randn(shape, generator=g).to(DEV)
            single = lowp.pack(kind, code, w, cin_ref, cout, cin_slab, ds, dsh, role=role)
            wp = torch.full_like(single, 0x5a5a)
            entries.append((kind, role, w, wp, cin_ref, cout, cin_ref if cin_slab is None else cin_slab, ds, dsh))
            singles.append(single)
        tab = lowp.PackTable()
        tab.run(code, entries)
        torch.cuda.synchronize()
        for e, single in zip(entries, singles):
            assert torch.equal(e[3], single), (dtype, e[0], e[1], e[4], e[5])
        # a second run reuses the device table (same pointers) and overwrites in place
        for e in entries:
            e[3].fill_(0)
        tab.run(code, entries)
        torch.cuda.synchronize()
        assert all(torch.equal(e[3], s) for e, s in zip(entries, singles))


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(2, 8, 16, 16, 32), (1, 16, 16, 32, 64), (2, 4, 8, 8, 256)])
def test_fused_gate_and_groupnorm2_backward(shape, dtype):
    """bts_lp_block_bwd (one reduce + one apply pass over dout) against bts_lp_se_bwd followed by bts_lp_gn_bwd on the same tensors: dc2 and
    dres to one unit in the last place of the storage type; parameter and bias gradients accumulated
    onto non-zero buffers; dout arrives as a channel slice of a wider slab"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, f = shape
    groups, r = 8, max(f // 8, 1)
    g = torch.Generator().manual_seed(f + d)
    res = torch.randn(shape, generator=g).to(tdt).to(DEV)
    c2 = torch.randn(shape, generator=g).to(tdt).to(DEV)
    slab = torch.randn((n, d, h, w, f + 16), generator=g).to(tdt).to(DEV)
    dout = slab[..., 8:8 + f]
    sp = torch.sigmoid(torch.randn((n, d, h, w, 1), generator=g)).to(DEV).contiguous()
    gap = torch.randn((n, f), generator=g).to(DEV)
    w1 = (0.3 * torch.randn((f, r), generator=g)).to(DEV)
    w2 = (0.3 * torch.randn((r, f), generator=g)).to(DEV)
    wsp = (0.3 * torch.randn(f, generator=g)).to(DEV)
    gamma = (1.0 + 0.3 * torch.randn(f, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(f, generator=g)).to(DEV)
    hbuf, ch = ops.se_mlp_fwd(gap, w1, w2)
    mean, rstd = lowp.gn_stats(code, c2, groups, ops.GN_SLAB, 1e-5)
    init = [torch.randn(t.shape, generator=g).to(DEV) for t in (w1, w2, wsp, gamma, beta, wsp, wsp)]      # dw1 dw2 dwsp dgamma dbeta db_pt db_c2
    ref = [t.clone() for t in init]
    dres_r = lowp.se_bwd(code, tdt, dout, res, sp, gap, hbuf, ch, w1, w2, wsp, ref[0], ref[1], ref[2], dbias=ref[5])
    dc2_r, _ = lowp.gn_bwd(code, tdt, c2, dout, gamma, beta, mean, rstd, ref[3], ref[4], groups, True, want_f32=False, dbias=ref[6])
    got = [t.clone() for t in init]
    out = lowp.block_bwd(code, tdt, dout, res, c2, sp, gap, hbuf, ch, w1, w2, wsp, gamma, beta, mean, rstd, groups, got[0], got[1], got[2],
                         got[3], got[4], dbias_pt=got[5], dbias_c2=got[6])
    assert out is not None
    dres, dc2 = out
    torch.cuda.synchronize()
    assert float((dc2.float() - dc2_r.float()).abs().max()) <= U[dtype] * float(dc2_r.float().abs().max()) * 1.01      # (same sums; the compiler contracts the apply formula differently)
    scale = float(dres_r.float().abs().max())
    assert float((dres.float() - dres_r.float()).abs().max()) <= U[dtype] * scale * 1.01
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('c', [1, 2, 3, 4, 5, 16])
def test_cast_into_a_padded_view(c, dtype):
    """bts_lp_cast: fp32 rows of c channels -> the storage type inside rows of 16 (how the 2-channel volume becomes one matrix step);
    round-to-nearest-even like torch's own conversion, the pad columns untouched"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(c)
    x = (torch.randn((2, 5, 7, 9, c), generator=g) * 3).cuda()
    for first in (0, 2, 3):
        if first + c > 24:
            continue
        buf = torch.full((2, 5, 7, 9, 24), 7.0, dtype=tdt, device=x.device)
        out = buf[..., first:first + c]
        lowp.cast(code, tdt, x, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, x.to(tdt))
        assert bool((buf[..., :first] == 7.0).all()) and bool((buf[..., first + c:] == 7.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('c', [1, 2, 3, 4])
def test_cast_with_zero_padding_to_a_matrix_step(c, dtype):
    """bts_lp_cast_pad16: fp32 rows of c <= 4 channels (also as a channel slice of wider rows) -> dense rows of 16 in the storage type,
    the live columns rounded like torch's own conversion, the tail exactly zero, the guard bytes behind the tensor untouched"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(10 + c)
    wide = (torch.randn((2, 5, 7, 9, c + 3), generator=g) * 3).cuda()
    for x in (wide[..., :c].contiguous(), wide[..., 1:1 + c]):
        out = lowp.cast_pad16(code, tdt, x)
        torch.cuda.synchronize()
        assert tuple(out.shape) == (2, 5, 7, 9, 16) and out.dtype == tdt
        assert torch.equal(out[..., :c], x.to(tdt)) and bool((out[..., c:] == 0).all())


GNB_CASES = [
    # (N, D, H, W), GroupNorm channels (= conv2 input channels), dy channels, groups, dy is a slab view, fused form expected
    ((2, 32, 32, 64), 32, 32, 8, False, True),       # z chunks of 16 planes over groups of 4: flushed at every group boundary
    ((2, 128, 32, 64), 16, 32, 4, True, True),       # z chunks of 16 planes inside groups of 32: several runs per (sample, group)
    ((2, 64, 32, 64), 16, 32, 2, True, False),       # 8 classes per group: more than the epilogue form folds (4)
    ((1, 16, 16, 32), 32, 32, 8, False, False),      # too small for the streaming kernel: the two steps back to back
    ((2, 8, 16, 16), 64, 64, 8, False, False),       # 64 channels: the tiled kernel, no epilogue form (yet)
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', GNB_CASES, ids=lambda c: '%s-%d-%d-g%d-%s' % ('x'.join(map(str, c[0])), c[1], c[2], c[3], 'fused' if c[5] else 'plain'))
def test_conv2_data_gradient_with_groupnorm1_backward(case, dtype):
    """bts_lp_conv3d_bwd_data_gn_bwd (conv2's data gradient + GroupNorm-1 (+ReLU) backward, class sums from the conv's epilogue where
    the z-marching kernel runs the layer) against bts_lp_conv3d_bwd_data followed by bts_lp_gn_bwd on the same tensors: da identical
    (same kernel, same stores), dc / dc32 / dgamma / dbeta / dbias equal up to the order of the fp32 class sums; and against the fp32
    engine's GroupNorm backward on the stored da (the oracle-pinned kernel of tests/test_kernels_gpu.py)"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cg, cdy, G, slab, want_fused = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cg * 3 + cdy + d)
    k = torch.randn((3, 3, 3, cg, cdy), generator=g) * (2.0 / (27 * cg)) ** 0.5
    wpb = lowp.pack(ops.K3S1, code, k.to(DEV), cg, cdy, role=ops.ROLE_BWD)
    c = (torch.randn((n, d, h, w, cg), generator=g) * 1.5 + 0.3).to(tdt).to(DEV)
    if slab:
        dys = torch.randn((n, d, h, w, cdy + 16), generator=g).to(tdt).to(DEV)
        dy = dys[..., 8:8 + cdy]
    else:
        dy = torch.randn((n, d, h, w, cdy), generator=g).to(tdt).to(DEV)
    gamma = (1 + 0.3 * torch.randn(cg, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(cg, generator=g)).to(DEV)
    mean, rstd = lowp.gn_stats(code, c, G, ops.GN_SLAB, 1e-5)
    # reference: the two entry points
    da_r = torch.empty_like(c)
    lowp.conv_bwd_data(ops.K3S1, code, dy, wpb, da_r, False)
    dg_r, db_r, dbias_r = torch.full((cg,), 0.5, device=DEV), torch.full((cg,), -0.25, device=DEV), torch.full((cg,), 0.125, device=DEV)
    dc_r, dc32_r = lowp.gn_bwd(code, tdt, c, da_r, gamma, beta, mean, rstd, dg_r, db_r, G, True, want_f32=True, dbias=dbias_r)
    dg, db, dbias = torch.full((cg,), 0.5, device=DEV), torch.full((cg,), -0.25, device=DEV), torch.full((cg,), 0.125, device=DEV)
    out = lowp.conv_bwd_data_gn_bwd(code, tdt, dy, wpb, c, gamma, beta, mean, rstd, dg, db, G, True, want_f32=True, dbias=dbias)
    assert out is not None
    da, dc, dc32, fused = out
    torch.cuda.synchronize()
    assert fused == want_fused, (fused, want_fused)
    assert torch.equal(da, da_r)
    scale = float(dc32_r.abs().max())
    assert float((dc32 - dc32_r).abs().max()) <= 2e-5 * scale + 1e-6
    assert float((dc.float() - dc_r.float()).abs().max()) <= U[dtype] * scale * 1.01
    for a, b in ((dg, dg_r), (db, db_r), (dbias, dbias_r)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-5, (a, b)
    # and the fp32 engine's kernel on the stored da
    dg32, db32 = torch.full((cg,), 0.5, device=DEV), torch.full((cg,), -0.25, device=DEV)
    dx32 = ops.gn_bwd(lowp.uncast(code, c), lowp.uncast(code, da), gamma, beta, mean, rstd, dg32, db32, G, ops.GN_SLAB, True, accumulate_params=True)
    assert float((dc32 - dx32).abs().max()) <= 2e-5 * float(dx32.abs().max()) + 1e-6
    for a, b in ((dg, dg32), (db, db32)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-5


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('mode', ['slab', 'channel'])
def test_last_block_epilogue_with_the_output_head_in_it(dtype, mode):
    """bts_lp_block_epilogue_head (inference tail: decoder.py:55-63 after the top block): y = sigmoid(out . W + b) with `out` never written,
    against the fp64 formula on the stored 16-bit operands (the fused kernel keeps `out` in fp32, so it is held to an fp32 bound, tighter
    than the two-kernel route that rounds `out` to the storage type in between); shapes outside its tiling are declined, not mangled"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    gmode = ops.GN_SLAB if mode == 'slab' else ops.GN_CHANNEL
    g = torch.Generator().manual_seed(17)
    n, d, h, w, c, G = 2, 8, 8, 16, 32, 8
    x = torch.randn((n, d, h, w, c), generator=g) * 1.7 + 0.4
    res = torch.randn((n, d, h, w, c), generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    wsp = torch.randn(c, generator=g) * 0.3
    ch = torch.rand((n, c), generator=g)
    wk, bk = torch.randn((c, 3), generator=g) * 0.2, torch.randn(3, generator=g) * 0.1
    xr, rr = _round(x, tdt), _round(res, tdt)
    axis = -1 if mode == 'slab' else 1
    xx = xr if mode == 'slab' else xr.permute(0, 4, 1, 2, 3)
    gn = R.group_norm(xx, gamma.double(), beta.double(), G, axis)
    gn = torch.relu(gn if mode == 'slab' else gn.permute(0, 2, 3, 4, 1))
    sp = torch.sigmoid((rr * wsp.double()).sum(-1, keepdim=True))
    out_ref = rr * (sp + ch.double().reshape(n, 1, 1, 1, c)) + gn
    ref = torch.sigmoid(out_ref @ wk.double() + bk.double())
    xd, rd = x.to(tdt).to(DEV), res.to(tdt).to(DEV)
    mean, rstd = lowp.gn_stats(code, xd, G, gmode, 1e-5)
    args = (wsp.to(DEV), ch.to(DEV), gamma.to(DEV), beta.to(DEV), mean, rstd, G, gmode)
    y = lowp.block_epilogue_head(code, rd, xd, *args, wk.to(DEV), bk.to(DEV), True)
    torch.cuda.synchronize()
    assert y is not None and y.dtype == torch.float32 and tuple(y.shape) == (n, d, h, w, 3)
    err = float((y.double().cpu() - ref).abs().max())
    assert err <= 2e-5, err                     # (fp32 arithmetic on the stored operands; the statistics come from the 16-bit kernel's fp32 sums)
    # the two-kernel route rounds `out` to the storage type before the head reads it: it agrees within that rounding
    out = lowp.block_epilogue(code, rd, xd, torch.empty((n, d, h, w, c), dtype=tdt, device=DEV), *args)
    y2 = lowp.head(code, out, wk.to(DEV), bk.to(DEV), True)
    torch.cuda.synchronize()
    bound = 0.25 * U[dtype] * (out_ref.abs() @ wk.double().abs()) + 2e-5       # (sigmoid is 1/4-Lipschitz)
    assert bool(((y2.double().cpu() - y.double().cpu()).abs() <= bound).all())
    # outside the tiling (units that are not whole 2048-element chunks): declined
    assert lowp.block_epilogue_head(code, rd[:, :, :3, :4].contiguous(), xd[:, :, :3, :4].contiguous(), *args, wk.to(DEV), bk.to(DEV), True) is None


def test_inference_forward_with_and_without_the_fused_head(monkeypatch):
    """LowPrecisionForward: the top decoder block's epilogue carries the output head by default; BTS_LP_FUSE_HEAD=0 restores the two
    launches.  Same labels, probabilities within the storage rounding of the block output."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    m = _model(dict(base_filters=16, groups=8, reduction=2, depth=3), (32, 32, 32), 5)
    x = torch.randn((1, 32, 32, 32, 2), generator=torch.Generator().manual_seed(9)).to(DEV)
    ops.profile_enable(False)
    y1 = lowp.LowPrecisionForward(m, 'float16')(x)
    monkeypatch.setenv('BTS_LP_FUSE_HEAD', '0')
    y0 = lowp.LowPrecisionForward(m, 'float16')(x)
    torch.cuda.synchronize()
    assert y1.shape == y0.shape == (1, 32, 32, 32, 3)
    d = (y1 - y0).abs()
    assert 0.0 < float(d.max()) <= 2e-3, float(d.max())        # (different roundings: not bitwise equal, and not far apart)
    assert float((y1.argmax(-1) != y0.argmax(-1)).float().mean()) <= 1e-3


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape,f,groups', [((1, 8, 8, 64), 32, 8), ((2, 6, 10, 40), 16, 2), ((1, 12, 7, 33), 24, 4)],
                         ids=['aligned-32', 'ragged-16', 'ragged-24'])
def test_first_block_kernel_on_the_raw_two_channel_volume(shape, f, groups, dtype):
    """bts_lp_first_block_fwd (csrc/lowp_c2.hip): conv1 (3x3x3, 2 -> F) + its GroupNorm statistics, the shortcut (1x1x1, 2 -> F) and the
    squeeze of the first ResnetBlock from ONE pass over the fp32 volume (resnet.py:30-37,80-87,118,121 at encoder level 0), against the
    oracle's ops on the operands rounded to the storage type, |err| <= 8 * 2^-24 * sum|a b| + u |ref|; whole and ragged tiles, F = 16 / 24 / 32"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    from bts_amd.layers.group_norm import GroupNormalization
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    n, d, h, w = shape
    g = torch.Generator().manual_seed(31 + f)
    x = torch.randn((n, d, h, w, 2), generator=g) * 1.3 + 0.2
    w3 = torch.randn((3, 3, 3, 2, f), generator=g) * 0.2
    b3 = torch.randn(f, generator=g) * 0.3
    w1 = torch.randn((1, 1, 1, 2, f), generator=g) * 0.5
    b1 = torch.randn(f, generator=g) * 0.3
    xr, w3r, w1r = _round(x, tdt), _round(w3, tdt), _round(w1, tdt)
    c1_ref = R.conv3d(xr, w3r, b3.double())
    c1_bound = 8 * 2.0 ** -24 * (R.conv3d(xr.abs(), w3r.abs(), None) + b3.double().abs()) + u * c1_ref.abs() + 1e-30
    res_ref = R.conv3d(xr, w1r, b1.double())
    res_bound = 8 * 2.0 ** -24 * (R.conv3d(xr.abs(), w1r.abs(), None) + b1.double().abs()) + u * res_ref.abs() + 1e-30
    norm = GroupNormalization(groups=groups, axis=-1)
    norm.build((None, None, None, None, f))
    out = lowp.first_block(code, tdt, x.to(DEV), w3.to(DEV), b3.to(DEV), w1.to(DEV), b1.to(DEV), f, norm)
    torch.cuda.synchronize()
    assert out is not None
    c1, mean, rstd, res, gap = out
    assert float(((c1.double().cpu() - c1_ref).abs() / c1_bound).max()) <= 1.0
    assert float(((res.double().cpu() - res_ref).abs() / res_bound).max()) <= 1.0
    # GroupNorm-1 statistics (slab semantics: group g = z planes [g D/G, (g+1) D/G), all channels) of the UNROUNDED conv output
    zt = d // groups
    for i in range(n):
        for gq in range(groups):
            blk = c1_ref[i, gq * zt:(gq + 1) * zt]
            m_ref, v_ref = float(blk.mean()), float(blk.var(unbiased=False))
            assert abs(float(mean[i * groups + gq]) - m_ref) <= 1e-5 * (1 + abs(m_ref))
            assert abs(float(rstd[i * groups + gq]) * (v_ref + 1e-5) ** 0.5 - 1.0) <= 1e-4
    gap_ref = res_ref.mean(dim=(1, 2, 3))
    assert float((gap.double().cpu() - gap_ref).abs().max()) <= 1e-5 * (1 + float(gap_ref.abs().max()))


def test_inference_forward_with_and_without_the_first_block_kernel(monkeypatch):
    """LowPrecisionForward takes the raw volume through bts_lp_first_block_fwd by default; BTS_LP_C2=0 restores the zero-padded 16-channel
    copy and the generic kernels.  Same operands, different summation order: probabilities within the 16-bit route's own noise"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    m = _model(dict(base_filters=16, groups=8, reduction=2, depth=3), (32, 32, 32), 5)
    x = torch.randn((1, 32, 32, 32, 2), generator=torch.Generator().manual_seed(9)).to(DEV)
    ops.profile_enable(True)
    y1 = lowp.LowPrecisionForward(m, 'float16')(x)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    assert 'lp_c2_kernel' in [s for s, _, _ in ops.profile_records()]
    monkeypatch.setenv('BTS_LP_C2', '0')
    y0 = lowp.LowPrecisionForward(m, 'float16')(x)
    torch.cuda.synchronize()
    d = (y1 - y0).abs()
    assert float(d.max()) <= 5e-3 and float(d.mean()) <= 2e-4, (float(d.max()), float(d.mean()))
    assert float((y1.argmax(-1) != y0.argmax(-1)).float().mean()) <= 2e-3


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('c', [1, 2, 3])
def test_dropout_and_cast_in_one_pass_equal_the_three_passes(dtype, c):
    """encoder.py:39,71 on the way into the first block: bts_lp_dropout_cast_pad16 draws what bts_dropout_mask draws for the same seed
    (checked against the mask itself: kept elements scaled by 1 / (1 - rate), dropped ones zero, the 16-channel tail zero) and is
    bit-identical to bts_dropout_mask + bts_dropout_apply + bts_lp_cast_pad16"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 9, 10, 11, c), generator=g).to(dev)
    rate, seed = 0.2, 12345
    mask = ops.dropout_mask(x.shape, rate, seed, dev)
    want = lowp.cast_pad16(code, tdt, ops.dropout_apply(x, mask, rate))
    got = lowp.dropout_cast_pad16(code, tdt, x, rate, seed)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    keep = mask.bool()
    assert 0.7 < float(keep.float().mean()) < 0.9
    ref = torch.where(keep, x * (1.0 / (1.0 - rate)), torch.zeros_like(x)).to(tdt)
    assert torch.equal(got[..., :c], ref) and bool((got[..., c:] == 0).all())


def test_inference_forward_with_the_skip_part_of_the_decoder_blocks_on_a_side_stream(monkeypatch):
    """BTS_LP_EARLY_SKIP=1 (measured and closed in round 6: slower; kept as an A/B switch): a decoder block's conv1 and shortcut over
    [skip | up-sampled] (decoder.py:75) as the skip part, started next to the deepest levels on a side stream, plus the up-sampled part added
    to it (bts_lp_conv3d_fwd_gn_acc / bts_lp_conv1_gap_acc).  The same contraction split over its input channels, one more rounding of the
    partial sum: same labels, probabilities within the storage rounding; and the entry points' statistics / squeeze are those of the SUMS
    (a GroupNorm over the partial only would move y_pred by far more than 2e-3)."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    m = _model(dict(base_filters=16, groups=8, reduction=2, depth=3), (32, 32, 32), 5)
    x = torch.randn((1, 32, 32, 32, 2), generator=torch.Generator().manual_seed(9)).to(DEV)
    y0 = lowp.LowPrecisionForward(m, 'float16')(x)
    monkeypatch.setenv('BTS_LP_EARLY_SKIP', '1')
    ops.profile_enable(True)
    y1 = lowp.LowPrecisionForward(m, 'float16')(x)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    assert y1.shape == y0.shape == (1, 32, 32, 32, 3)
    d = (y1 - y0).abs()
    assert 0.0 < float(d.max()) <= 2e-3, float(d.max())
    assert float((y1.argmax(-1) != y0.argmax(-1)).float().mean()) <= 1e-3


@pytest.mark.parametrize('dtype,scale', [('float16', 1.0), ('bfloat16', 8.0)])
def test_inference_forward_with_level_0_as_two_dense_operands(dtype, scale, monkeypatch):
    """decoder.py:75 at the top level as a list of (ptr, C) segments: the encoder's level-0 output and the up-sampled tensor are two
    dense 32-channel tensors, the top decoder block's conv1 + GroupNorm statistics + shortcut + squeeze one fused launch pair over them
    (bts_lp_conv3d_fwd_gn_shortcut, x_split).  Held against the fp64 ORACLE and the fp32 engine like the slab route, next to the slab route
    (BTS_LP_INF_SPLIT=0) itself; the launch records say which form ran."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    kw = dict(base_filters=32, groups=8, reduction=4, depth=3)
    crop = (40, 16, 96)
    m = _model(kw, crop, 11)
    x = torch.randn((1,) + crop + (2,), generator=torch.Generator().manual_seed(3)).to(DEV)
    y32 = m(x, training=False, inference=True)[0].t

    def run():
        ops.profile_enable(True)
        y = lowp.LowPrecisionForward(m, dtype)(x)
        torch.cuda.synchronize()
        names = [r[0] for r in ops.profile_records()]
        ops.profile_enable(False)
        return y, names
    y1, n1 = run()
    monkeypatch.setenv('BTS_LP_INF_SPLIT', '0')
    y0, n0 = run()
    # split: the top decoder block's conv1 and shortcut are two z-marching launches (one per operand) instead of a tiled conv + a 1x1x1 conv
    cnt = lambda names, k: sum(1 for q in names if q == k)
    print('profiled launches: split %s ; slab %s' % (sorted(set((k, cnt(n1, k)) for k in n1)), sorted(set((k, cnt(n0, k)) for k in n0))))
    assert cnt(n1, 'lp_s1z_kernel') > cnt(n0, 'lp_s1z_kernel') and cnt(n1, 'lp_k1_kernel') == cnt(n0, 'lp_k1_kernel') - 1
    cfg = R.default_config(**kw)
    P = R.ParamSet()
    for p in m.trainable_variables:
        P[m.oracle_name(p)] = p.t.detach().cpu().double()
    yo = R.model(x.cpu().double(), P, cfg, training=False, inference=True)[0]
    for tag, y in (('split', y1), ('slab', y0)):
        mx, mean, mism = _compare(y, y32, '%s %s vs fp32 engine' % (dtype, tag))
        assert mx <= 2e-2 * scale and mean <= 1e-3 * scale and mism <= 2e-3 * scale
        mx, mean, mism = _compare(y.cpu().double(), yo, '%s %s vs fp64 oracle' % (dtype, tag))
        assert mx <= 2e-2 * scale and mean <= 1e-3 * scale and mism <= 2e-3 * scale
    d = (y1 - y0).abs()
    assert 0.0 < float(d.max()) <= 5e-3 * scale, float(d.max())
