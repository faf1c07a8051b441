"""-m gpu: out-of-bounds guard bands (GPU AddressSanitizer is not available on this pool -- this is the substitute).

The kernels mask lanes through buffer-descriptor range tricks (offset 0x80000000 against num_records 0x7fffffff): a stray store that
lands INSIDE the descriptor but past a tensor's end would go unseen by every parity test.  Here every buffer the host side hands to the
C ABI -- outputs, saved activations, gradient buffers, partial-sum scratch and each `*_workspace` buffer, sized EXACTLY by its query --
is carved out of a larger allocation pre-filled with the byte 0xA5, 4 KB of it on either side; after the run every guard byte must
still read 0xA5.  The allocator is swapped in for `torch.empty / empty_like / zeros / zeros_like / full` of every bts_amd module and
for `ops.workspace` (which normally hands out a grow-only buffer of at least 1 MB, i.e. would hide an over-run of the queried size).

What runs under it: complete training steps of small models through every conv-form switch of DESIGN section 8 (3-D Winograd, 2-D
Winograd, implicit GEMM incl. split-K, upm / k1s / dsc / c2 forced on by their thresholds, Winograd / direct / streaming weight
gradients), at ragged and minimum crops and with batch 2; the 16-bit forward and training step; and the 16-bit kernels with their own
tiling floors (LDS-DMA stride-1 conv, streaming 1x1x1, transposed form, strided weight gradients) on slab views with ld > C."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

PAD = 4096
FILL = 0xA5


class Guarded(object):
    """stand-in for the `torch` module inside bts_amd: allocation calls return views into guard-banded buffers"""

    def __init__(self):
        self.live = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _alloc(self, shape, dtype, device, zero=False, value=None):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        dtype = dtype or torch.float32
        numel = 1
        for s in shape:
            numel *= s
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        big = torch.full((nbytes + 2 * PAD,), FILL, dtype=torch.uint8, device=device if device is not None else 'cpu')
        view = big[PAD:PAD + nbytes].view(dtype).view(shape)
        if zero:
            view.zero_()
        if value is not None:
            view.fill_(value)
        if big.is_cuda:
            self.live.append((big, nbytes))
        return view

    def empty(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = shape[0]
        return self._alloc(shape, dtype, device)

    def zeros(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = shape[0]
        return self._alloc(shape, dtype, device, zero=True)

    def full(self, shape, value, dtype=None, device=None, **kw):
        return self._alloc(shape, dtype, device, value=value)

    def empty_like(self, t, dtype=None, **kw):
        return self._alloc(tuple(t.shape), dtype or t.dtype, t.device)

    def zeros_like(self, t, dtype=None, **kw):
        return self._alloc(tuple(t.shape), dtype or t.dtype, t.device, zero=True)

    def check(self):
        torch.cuda.synchronize()
        bad = 0
        for big, nbytes in self.live:
            if not bool((big[:PAD] == FILL).all()) or not bool((big[PAD + nbytes:] == FILL).all()):
                bad += 1
        n = len(self.live)
        self.live = []
        assert bad == 0, '%d of %d guarded buffers were written outside their bounds' % (bad, n)
        return n


@pytest.fixture
def guard(monkeypatch):
    import bts_amd
    from bts_amd import ops
    g = Guarded()
    import sys
    for name in ('model', 'util', 'tape', 'ops', 'lowp', 'lowp_train', 'parallel', 'data', 'infer', 'train', 'layers._base', 'layers.resnet',
                 'layers.group_norm', 'layers.downsample', 'layers.upsample', 'layers.vae', 'layers.encoder', 'layers.decoder'):
        importlib.import_module('bts_amd.' + name)
    mods = [m for n, m in list(sys.modules.items()) if n == 'bts_amd' or n.startswith('bts_amd.')]
    for m in mods:
        if getattr(m, 'torch', None) is torch:
            monkeypatch.setattr(m, 'torch', g)

    def exact_workspace(nbytes, device):       # exactly what the *_workspace query asked for -- no 1 MB floor, no reuse
        return g._alloc((max(int(nbytes), 1),), torch.uint8, device)
    monkeypatch.setattr(ops, 'workspace', exact_workspace)
    from bts_amd import lowp
    if getattr(lowp, 'ops', None) is ops:
        pass
    yield g


def _fresh_model(kw, shape, seed=3):
    from bts_amd.layers import _base
    from bts_amd.model import Model
    from bts_amd.tape import bump_weights_epoch
    _base.set_seed(seed)
    m = Model(**kw)
    m.build(shape + (2,))
    gen = torch.Generator().manual_seed(seed + 1)
    for p in m.trainable_variables:
        if p.name.endswith('gamma'):
            p.t.copy_((1.0 + 0.3 * torch.randn(p.t.shape, generator=gen)).to(p.t.device))
    bump_weights_epoch()
    return m


FORMS = {
    'default': {},
    'wino2d': {'BTS_W3': '0'},
    'direct': {'BTS_WINO': '0', 'BTS_WGW': '0', 'BTS_K1W': '0'},
    'small-forms': {'BTS_IGEMM_UPM_MIN': '1', 'BTS_IGEMM_K1S_MIN': '1', 'BTS_IGEMM_DSC_MIN': '1', 'BTS_IGEMM_C2_MIN': '1',
                    'BTS_WINO_MIN_WGS': '1'},
    'one-stream': {'BTS_WGRAD_STREAM': '0', 'BTS_GATE_STREAM': '0', 'BTS_IGEMM_NOGNFUSE': '1'},
}
SHAPES = [
    (dict(base_filters=8, groups=2, reduction=2, depth=3), (1, 8, 16, 24)),      # ragged tiles on every axis, minimum depth-3 crop in z
    (dict(base_filters=16, groups=8, reduction=2, depth=3), (2, 32, 16, 40)),    # batch 2, Winograd-sized top level, ragged x
    (dict(base_filters=32, groups=8, reduction=8, depth=4), (1, 16, 16, 32)),    # the CLI-default model: 256-channel level, split-K grids
]


@pytest.mark.parametrize('form', list(FORMS))
@pytest.mark.parametrize('kw,shape', SHAPES, ids=['f8-8x16x24', 'f16-n2-32x16x40', 'cli-16x16x32'])
def test_fp32_train_step_stays_inside_its_buffers(guard, monkeypatch, kw, shape, form):
    from bts_amd.data import synthetic_batch
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    m = _fresh_model(kw, shape)
    latent = kw['base_filters'] * 2 ** (kw['depth'] - 2)
    x, y, _, _ = synthetic_batch(shape[0], shape[1:], latent=latent, seed=12)
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    for _ in range(2):        # the second step re-packs the weight images and re-uses the optimiser state
        loss, _, _ = train_step(m, opt, DiceVAELoss(), DiceCoefficient(), x.cuda(), y.cuda())
    assert float(loss) == float(loss)
    n = guard.check()
    assert n > 200, n          # (the step allocated: saved activations, gradients, scratch)


@pytest.mark.parametrize('filters', [16, 8])
@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_16bit_forward_and_train_step_stay_inside_their_buffers(guard, dtype, filters):
    """filters = 8 (round-3 advisor finding: a GroupNorm gradient zero-padded to a 16-channel matrix step is wider than an 8-filter
    weight-gradient slot): the 16-bit engine contracts over whole 16-channel steps and refuses such a model BY NAME at construction;
    the trainer's weight-gradient helper additionally raises on any declined or mis-sized launch instead of leaving scratch unwritten"""
    from bts_amd import lowp
    from bts_amd.data import synthetic_batch
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.util import DiceCoefficient, ScheduledOptim
    kw, shape = dict(base_filters=filters, groups=8 if filters == 16 else 4, reduction=2, depth=3), (2, 32, 32, 48)
    m = _fresh_model(kw, shape)
    if filters % 16:
        with pytest.raises(ValueError, match='multiple of 16'):
            LowPrecisionTrainer(m, dtype)
        with pytest.raises(ValueError, match='multiple of 16'):
            lowp.LowPrecisionForward(m, dtype)
        return
    x, y, _, _ = synthetic_batch(shape[0], shape[1:], latent=2 * filters, seed=7)
    yp = lowp.LowPrecisionForward(m, dtype)(x[:1, :24, :, :40].contiguous())          # ragged forward-only volume
    assert yp.dtype == torch.float32
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    tr = LowPrecisionTrainer(m, dtype)
    for _ in range(2):
        loss, _, _ = tr.step(opt, DiceCoefficient(), x, y)
    assert float(loss) == float(loss)
    g = m.flat_grads / tr.last_grad_scale        # (float16: the buffer keeps the dynamic loss scale, Adam un-scales as it reads)
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) < 1e3, float(g.abs().max())     # (no uninitialised scratch added in)
    assert guard.check() > 200


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_16bit_kernels_with_tiling_floors_stay_inside_slab_views(guard, dtype):
    """the kernels that only take grids above a size floor, each on channel-slice views of wider buffers (ld > C): a store one voxel past
    the view's end, or into a neighbouring slice, lands in guarded bytes or in the 0xA5-filled channels checked below"""
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    dev = torch.device('cuda', 0)
    gen = torch.Generator().manual_seed(2)

    def slab(n, d, h, w, c, extra=24, lead=8):
        buf = guard._alloc((n, d, h, w, c + extra), tdt, dev)
        buf.view(torch.uint8).fill_(FILL)
        v = buf[..., lead:lead + c]
        v.copy_(torch.randn((n, d, h, w, c), generator=gen).to(tdt).to(dev))
        return buf, v, lead, c

    def untouched(buf, lead, c):
        b8 = buf.view(torch.uint8).reshape(-1, buf.shape[-1], buf.element_size())
        return bool((b8[:, :lead] == FILL).all()) and bool((b8[:, lead + c:] == FILL).all())

    # LDS-DMA stride-1 conv: ragged 32-wide and 16-wide tiles, both work splits, accumulation
    for (n, d, h, w, cin, cout) in [(1, 16, 20, 40, 32, 32), (2, 16, 24, 20, 16, 64), (1, 32, 24, 20, 64, 64)]:
        xb, xv, _, _ = slab(n, d, h, w, cin)
        yb, yv, yl, yc = slab(n, d, h, w, cout)
        wt = (torch.randn((3, 3, 3, cin, cout), generator=gen) * 0.05).to(dev)
        b = torch.randn(cout, generator=gen).to(dev)
        lowp.conv(ops.K3S1, code, tdt, xv, lowp.pack(ops.K3S1, code, wt, cin, cout), b, cout, out=yv)
        lowp.conv_bwd_data(ops.K3S1, code, yv, lowp.pack(ops.K3S1, code, wt, cin, cout, role=ops.ROLE_BWD), xv, True)
        torch.cuda.synchronize()
        assert untouched(yb, yl, yc) and untouched(xb, 8, cin)
    # streaming 1x1x1 (ragged last block) and its accumulating data gradient
    xb, xv, _, _ = slab(2, 9, 13, 20, 48)
    yb, yv, yl, yc = slab(2, 9, 13, 20, 32)
    wt = (torch.randn((1, 1, 1, 48, 32), generator=gen) * 0.1).to(dev)
    lowp.conv(ops.K1, code, tdt, xv, lowp.pack(ops.K1, code, wt, 48, 32), None, 32, out=yv)
    lowp.conv_bwd_data(ops.K1, code, yv, lowp.pack(ops.K1, code, wt, 48, 32, role=ops.ROLE_BWD), xv, True)
    torch.cuda.synchronize()
    assert untouched(yb, yl, yc) and untouched(xb, 8, 48)
    # transposed form (ragged coarse grid) and the stride-2 conv's data gradient
    xb, xv, _, _ = slab(1, 8, 11, 24, 32)
    yb, yv, yl, yc = slab(1, 16, 22, 48, 64)
    wt = (torch.randn((3, 3, 3, 64, 32), generator=gen) * 0.05).to(dev)
    lowp.conv(ops.K3S2T, code, tdt, xv, lowp.pack(ops.K3S2T, code, wt, 32, 64), None, 64, out=yv)
    torch.cuda.synchronize()
    assert untouched(yb, yl, yc)
    wt2 = (torch.randn((3, 3, 3, 64, 32), generator=gen) * 0.05).to(dev)                     # stride-2 conv 64 -> 32: dy on (8, 11, 24)
    lowp.conv_bwd_data(ops.K3S2, code, xv, lowp.pack(ops.K3S2, code, wt2, 64, 32, role=ops.ROLE_BWD), yv, True)
    torch.cuda.synchronize()
    assert untouched(yb, yl, yc)
    # strided weight gradients (guarded dw / db / workspace)
    dw = guard._alloc((3, 3, 3, 64, 32), torch.float32, dev, zero=True)
    db = guard._alloc((32,), torch.float32, dev, zero=True)
    dyc = torch.randn((1, 8, 11, 24, 32), generator=gen).to(tdt).to(dev)
    assert lowp.conv_bwd_weight(ops.K3S2, code, yv, dyc, dw, db, accumulate=True)
    dwt = guard._alloc((3, 3, 3, 64, 32), torch.float32, dev, zero=True)
    assert lowp.conv_bwd_weight(ops.K3S2T, code, xv, yv.contiguous(), dwt, None, accumulate=False)
    # z-marching stride-1 conv for few channels (lowp_s1z.hip): several columns, z chunks with a ragged last one, slab views, accumulate
    for (n, d, h, w, cin, cout) in [(1, 37, 16, 96, 32, 16), (2, 16, 32, 64, 16, 32)]:
        xb, xv, _, _ = slab(n, d, h, w, cin)
        yb, yv, yl, yc = slab(n, d, h, w, cout)
        wt = (torch.randn((3, 3, 3, cin, cout), generator=gen) * 0.05).to(dev)
        b = torch.randn(cout, generator=gen).to(dev)
        ops.profile_enable(True)
        lowp.conv(ops.K3S1, code, tdt, xv, lowp.pack(ops.K3S1, code, wt, cin, cout), b, cout, out=yv)
        if cin == cout or cout % 16 == 0:
            lowp.conv_bwd_data(ops.K3S1, code, yv, lowp.pack(ops.K3S1, code, wt, cin, cout, role=ops.ROLE_BWD), xv, True)
        torch.cuda.synchronize()
        ops.profile_enable(False)
        assert 'lp_s1z_kernel' in [r[0] for r in ops.profile_records()]
        assert untouched(yb, yl, yc) and untouched(xb, 8, cin)
    # LDS-tiled stride-2 conv of the 32-channel top level (lowp_s2t.hip): ragged tiles on every axis, slab views on both sides; and the
    # whole-row-load gather (64 -> 64: quads of voxels per load instruction) on a grid whose last quad row ends the tensor
    for (n, d, h, w, cin, cout, sym) in [(2, 60, 44, 72, 32, 24, 'lp_s2t_kernel'), (1, 48, 40, 72, 64, 64, 'lp_conv_gather_kernel')]:
        xb, xv, _, _ = slab(n, d, h, w, cin)
        yb, yv, yl, yc = slab(n, d // 2, h // 2, w // 2, cout)
        wt = (torch.randn((3, 3, 3, cin, cout), generator=gen) * 0.05).to(dev)
        b = torch.randn(cout, generator=gen).to(dev)
        ops.profile_enable(True)
        lowp.conv(ops.K3S2, code, tdt, xv, lowp.pack(ops.K3S2, code, wt, cin, cout), b, cout, out=yv)
        torch.cuda.synchronize()
        ops.profile_enable(False)
        assert sym in [r[0] for r in ops.profile_records()]
        assert untouched(yb, yl, yc) and untouched(xb, 8, cin)
    # streaming stride-1 weight gradient (lowp_wgd.hip): x a slab view, guarded dw and an exact-size workspace
    xb, xv, _, _ = slab(1, 37, 16, 32, 64)
    dyc = torch.randn((1, 37, 16, 32, 16), generator=gen).to(tdt).to(dev)
    dw3 = guard._alloc((3, 3, 3, 64, 16), torch.float32, dev, zero=True)
    ops.profile_enable(True)
    assert lowp.conv_bwd_weight(ops.K3S1, code, xv, dyc, dw3, None, accumulate=True)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    assert 'lp_wgd_kernel' in [r[0] for r in ops.profile_records()]
    assert untouched(xb, 8, 64)
    # all images in one launch (bts_lp_pack_batch) into guarded image buffers of exactly bts_lp_packed_bytes
    entries = []
    for kind, role, cin, cout in ((ops.K3S1, ops.ROLE_FWD, 32, 24), (ops.K3S1, ops.ROLE_BWD, 16, 64), (ops.K1, ops.ROLE_FWD, 48, 8), (ops.K3S2T, ops.ROLE_FWD, 32, 16)):
        k = 1 if kind == ops.K1 else 3
        shape = (k, k, k, cout, cin) if kind == ops.K3S2T else (k, k, k, cin, cout)
        wsrc = torch.randn(shape, generator=gen).to(dev)
        nbytes = lowp.lib().query('bts_lp_packed_bytes', kind, role, cin, cout)
        entries.append((kind, role, wsrc, guard._alloc((nbytes // 2,), torch.int16, dev), cin, cout, cin, 0, 0))
    lowp.PackTable().run(code, entries)
    assert guard.check() > 16


def test_the_guard_itself_notices_a_stray_store(guard):
    """negative control: one byte written just past a guarded buffer must fail the check"""
    t = guard.empty((4, 8), dtype=torch.float32, device=torch.device('cuda', 0))
    base = guard.live[-1][0]
    base[PAD + 4 * 8 * 4] = 0          # first guard byte behind the tensor
    with pytest.raises(AssertionError):
        guard.check()
    assert t.shape == (4, 8)
