"""-m gpu: conv1's and the shortcut's data gradients of a ResnetBlock in ONE launch (bts_lp_conv3d_bwd_data_sc, round 6).

resnet.py:118 (`res = conv3d_ptwise(inputs)`) and resnet.py:134 (`x = conv3d_1(inputs)`) read the same tensor, so under train.py:151
their data gradients meet in d(inputs) = conv3x3x3^T(dc1) + conv1x1x1^T(dres).  The fused kernels (`lp_s1d_kernel<.., SC>` on 64-cout
items, `lp_s1z_kernel<.., SC>` on the few-channel 128^3 level) run the shortcut's contraction as an extra K-segment at the centre tap.

Every case goes through the C ABI and is checked against the ORACLE's ops (R.conv3d under autograd, fp64, on the same 16-bit-rounded
operands) under  |err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref| (+ u * |old| when accumulating), with the launch records asserting which
kernel produced it -- and that no 1x1x1 launch happened where the fused form claims to have run.  Shapes the fused kernels decline must
give the same answer through the two-launch route inside the same entry point (one extra rounding: the stored intermediate)."""
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


def _reference(dyr, dy2r, w3r, w1r, shape_x):
    """d(inputs) of conv3x3x3(inputs, w3) . dy + conv1x1x1(inputs, w1) . dy2 and the matching sum of absolute products"""
    xs = torch.zeros(shape_x, dtype=torch.float64, requires_grad=True)
    (R.conv3d(xs, w3r, None) * dyr).sum().backward()
    g3 = xs.grad.clone()
    xs.grad = None
    (R.conv3d(xs, w1r, None) * dy2r).sum().backward()
    g1 = xs.grad.clone()
    xa = torch.zeros(shape_x, dtype=torch.float64, requires_grad=True)
    ((R.conv3d(xa, w3r.abs(), None) * dyr.abs()).sum() + (R.conv3d(xa, w1r.abs(), None) * dy2r.abs()).sum()).backward()
    return g3, g1, xa.grad


CASES = [
    # n, (D,H,W), Cin (columns of dx), F (channels of dy / dy2), slab_dx, slab_dy2, accumulate, kernel expected
    (1, (16, 20, 40), 64, 32, True, False, False, 'lp_s1d_kernel'),      # 64-cout items, 32-wide tiles, ragged y / x, two k-steps
    (1, (16, 20, 40), 64, 32, False, True, True, 'lp_s1d_kernel'),       # ... accumulating, dy2 a slice of a wider tensor
    (2, (16, 24, 20), 64, 64, False, False, False, 'lp_s1d_kernel'),     # 16-wide tiles, few items -> split-K + reduce
    (1, (36, 32, 64), 128, 16, False, False, True, 'lp_s1d_kernel'),     # two cout groups per tile, items chained, one k-step
    (1, (8, 48, 32), 96, 48, True, True, False, 'lp_s1d_kernel'),        # three cout blocks: the last group half empty; three k-steps
    (2, (16, 32, 64), 32, 32, False, False, False, 'lp_s1z_kernel'),     # z-marching kernel, two k-steps
    (1, (40, 32, 64), 32, 16, True, True, True, 'lp_s1z_kernel'),        # ... one k-step, z chunks, accumulating into a slab view
    (1, (37, 16, 96), 16, 32, False, False, False, 'lp_s1z_kernel'),     # half-filled column block, ragged last z chunk
    (1, (16, 20, 40), 32, 32, False, False, False, None),                # 32-cout items of the tiled kernel: declined -> two launches
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d-%d%s' % (c[0], *c[1], c[2], c[3], '-acc' if c[6] else ''))
def test_fused_data_gradients(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    n, (d, h, w), cin, f, slab_dx, slab_dy2, accumulate, expect = case
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(hash((d, h, w, cin, f)) % 10000)
    dy = torch.randn((n, d, h, w, f), generator=g)
    dy2 = torch.randn((n, d, h, w, f), generator=g)
    w3 = torch.randn((3, 3, 3, cin, f), generator=g) * (2.0 / (27 * f)) ** 0.5
    w1 = torch.randn((1, 1, 1, cin, f), generator=g) * (2.0 / f) ** 0.5
    old = torch.randn((n, d, h, w, cin), generator=g)
    dyr, dy2r, w3r, w1r, oldr = _round(dy, tdt), _round(dy2, tdt), _round(w3, tdt), _round(w1, tdt), _round(old, tdt)
    g3, g1, gabs = _reference(dyr, dy2r, w3r, w1r, (n, d, h, w, cin))
    ref = g3 + g1 + (oldr if accumulate else 0)
    bound = 8 * 2.0 ** -24 * gabs + u * ref.abs() + (u * oldr.abs() if accumulate else 0) + 1e-30
    if expect is None:      # two launches: the 3x3x3 result is stored (rounded) before the 1x1x1 one is added to it
        bound = bound + u * (g3 + (oldr if accumulate else 0)).abs()
    if slab_dx:
        slab = torch.full((n, d, h, w, cin + 32), 3.0, dtype=tdt, device=DEV)
        dx = slab[..., 16:16 + cin]
    else:
        slab = None
        dx = torch.empty((n, d, h, w, cin), dtype=tdt, device=DEV)
    dx.copy_(old.to(tdt).to(DEV))
    if slab_dy2:
        wide = torch.full((n, d, h, w, f + 24), 5.0, dtype=tdt, device=DEV)
        dy2d = wide[..., 8:8 + f]
        dy2d.copy_(dy2.to(tdt).to(DEV))
    else:
        dy2d = dy2.to(tdt).to(DEV)
    wpb3 = lowp.pack(ops.K3S1, code, w3.to(DEV), cin, f, role=ops.ROLE_BWD)
    wpb1 = lowp.pack(ops.K1, code, w1.to(DEV), cin, f, role=ops.ROLE_BWD)
    fused, syms = _ran(lambda: lowp.conv_bwd_data_sc(code, dy.to(tdt).to(DEV), wpb3, dy2d, wpb1, dx, accumulate))
    if expect is None:
        assert not fused
    else:
        assert fused and expect in syms, (fused, syms)
        assert not any(s.startswith('lp_k1') or s.startswith('lp_conv_gather') for s in syms), syms       # no 1x1x1 launch
    err = (dx.double().cpu() - ref.detach()).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    # the shortcut term is really there: without it the error would be of the size of g1
    assert float((dx.double().cpu() - (ref - g1).detach()).abs().max()) > 0.1 * float(g1.abs().max())
    if slab is not None:
        assert bool((slab[..., :16] == 3.0).all()) and bool((slab[..., 16 + cin:] == 3.0).all())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('shape', [(1, 16, 20, 40, 64, 32), (2, 8, 16, 36, 128, 16)], ids=['2x32', '4x32-ragged'])
def test_split_output(shape, dtype):
    """the gradient of a concat of 32-channel tensors (decoder.py:75) written as dense 32-channel tensors, one per column block"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    n, d, h, w, cin, f = shape
    code, tdt = lowp.DTYPES[dtype]
    u = U[dtype]
    g = torch.Generator().manual_seed(77)
    dy = torch.randn((n, d, h, w, f), generator=g)
    dy2 = torch.randn((n, d, h, w, f), generator=g)
    w3 = torch.randn((3, 3, 3, cin, f), generator=g) * (2.0 / (27 * f)) ** 0.5
    w1 = torch.randn((1, 1, 1, cin, f), generator=g) * (2.0 / f) ** 0.5
    dyr, dy2r, w3r, w1r = _round(dy, tdt), _round(dy2, tdt), _round(w3, tdt), _round(w1, tdt)
    g3, g1, gabs = _reference(dyr, dy2r, w3r, w1r, (n, d, h, w, cin))
    ref = g3 + g1
    assert lowp.conv_bwd_data_sc_split_ok(n, d, h, w, cin, f)
    parts = torch.full((cin // 32 + 1, n, d, h, w, 32), 9.0, dtype=tdt, device=DEV)      # (one guard block behind)
    wpb3 = lowp.pack(ops.K3S1, code, w3.to(DEV), cin, f, role=ops.ROLE_BWD)
    wpb1 = lowp.pack(ops.K1, code, w1.to(DEV), cin, f, role=ops.ROLE_BWD)
    fused, syms = _ran(lambda: lowp.conv_bwd_data_sc(code, dy.to(tdt).to(DEV), wpb3, dy2.to(tdt).to(DEV), wpb1, parts[:cin // 32], False))
    assert fused and 'lp_s1d_kernel' in syms, (fused, syms)
    got = torch.cat([parts[b].double().cpu() for b in range(cin // 32)], dim=-1)
    bound = 8 * 2.0 ** -24 * gabs + u * ref.abs() + 1e-30
    assert float(((got - ref.detach()).abs() / bound).max()) <= 1.0
    assert bool((parts[cin // 32] == 9.0).all())
    # shapes the fused tiled kernel does not take say so up front
    assert not lowp.conv_bwd_data_sc_split_ok(1, 16, 32, 64, 32, 32)


def test_folded_duplicate_slice():
    """encoder.py:83-87: block j reads [o_{j-1}, o_0 .. o_{j-1}] through the slab [o_0 .. o_{j-1}] with the duplicated slice folded into the
    weights; in the data-gradient role the fold sits on the COLUMNS (dx's channels) of both images -- the fused launch must add the
    gradient of both copies, for the 3x3x3 and the 1x1x1 kernel alike"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES['bfloat16']
    u = U['bfloat16']
    g = torch.Generator().manual_seed(11)
    f, j = 32, 2
    n, d, h, w = 1, 16, 24, 32
    cin_ref, cin_slab = (j + 1) * f, j * f
    dy = torch.randn((n, d, h, w, f), generator=g)
    dy2 = torch.randn((n, d, h, w, f), generator=g)
    w3 = torch.randn((3, 3, 3, cin_ref, f), generator=g) * 0.05
    w1 = torch.randn((1, 1, 1, cin_ref, f), generator=g) * 0.2
    dyr, dy2r, w3r, w1r = _round(dy, tdt), _round(dy2, tdt), w3.double(), w1.double()
    g3, g1, gabs = _reference(dyr, dy2r, w3r, w1r, (n, d, h, w, cin_ref))
    full = g3 + g1                                                 # gradient of [o_{j-1}, o_0 .. o_{j-1}]
    ref = full[..., f:].clone()
    ref[..., (j - 1) * f:] += full[..., :f]                        # the duplicated slice o_{j-1} receives both
    ab = gabs[..., f:].clone()
    ab[..., (j - 1) * f:] += gabs[..., :f]
    dx = torch.empty((n, d, h, w, cin_slab), dtype=tdt, device=DEV)
    wpb3 = lowp.pack(ops.K3S1, code, w3.to(DEV), cin_ref, f, cin_slab, (j - 1) * f, f, role=ops.ROLE_BWD)
    wpb1 = lowp.pack(ops.K1, code, w1.to(DEV), cin_ref, f, cin_slab, (j - 1) * f, f, role=ops.ROLE_BWD)
    fused, syms = _ran(lambda: lowp.conv_bwd_data_sc(code, dy.to(tdt).to(DEV), wpb3, dy2.to(tdt).to(DEV), wpb1, dx, False))
    assert fused and 'lp_s1d_kernel' in syms, (fused, syms)
    bound = (8 * 2.0 ** -24 + 2 * u) * ab + u * ref.abs() + 1e-30   # (folded weights: sum of two roundings of the image)
    assert float(((dx.double().cpu() - ref).abs() / bound).max()) <= 1.0


def test_switch_runs_the_two_launches():
    """BTS_LP_SC is read once per process, so the A/B switch is exercised in a child interpreter: same numbers, lp_k1 launched"""
    import os
    import subprocess
    import sys
    code = r'''
import torch, bts_amd
from bts_amd import lowp, ops
DEV = torch.device('cuda', 0)
c, tdt = lowp.DTYPES['bfloat16']
g = torch.Generator().manual_seed(2)
dy = torch.randn((1, 16, 20, 40, 32), generator=g).to(tdt).to(DEV)
dy2 = torch.randn((1, 16, 20, 40, 32), generator=g).to(tdt).to(DEV)
w3 = (torch.randn((3, 3, 3, 64, 32), generator=g) * 0.05).to(DEV)
w1 = (torch.randn((1, 1, 1, 64, 32), generator=g) * 0.2).to(DEV)
dx = torch.empty((1, 16, 20, 40, 64), dtype=tdt, device=DEV)
ops.profile_enable(True)
fused = lowp.conv_bwd_data_sc(c, dy, lowp.pack(ops.K3S1, c, w3, 64, 32, role=ops.ROLE_BWD), dy2, lowp.pack(ops.K1, c, w1, 64, 32, role=ops.ROLE_BWD), dx, False)
torch.cuda.synchronize()
syms = [s for s, _, _ in ops.profile_records()]
print('RESULT', int(fused), int(any(s.startswith('lp_k1') for s in syms)), float(dx.float().abs().sum()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for sw in ('1', '0'):
        env = dict(os.environ, BTS_LP_SC=sw, PYTHONPATH=root)
        r = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][0].split()[1:])
    assert outs[0][:2] == ['1', '0'] and outs[1][:2] == ['0', '1'], outs
    a, b = float(outs[0][2]), float(outs[1][2])
    assert abs(a - b) <= 2e-3 * abs(a), (a, b)
