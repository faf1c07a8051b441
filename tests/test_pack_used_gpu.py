"""-m gpu: weight images are re-packed in the forms their layers READ (csrc/conv_igemm.hip, "Which parts of a K3S1 image are READ").
A BTS_CONV_K3S1 image carries three forms of the same weights; after an optimiser step (train.py:152) only the forms the dispatcher has
picked so far are rewritten, and a form picked later is packed on the spot.  Checked against the fp64 oracle's convolution on the NEW
weights, so a stale form would show."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

EPS32 = 2.0 ** -24


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32) * scale


def _check(got, ref, bound, what):
    err = (got.double().cpu() - ref).abs()
    tol = 32 * EPS32 * bound + 1e-7
    assert not (err > tol).any(), '%s: max err %.3e' % (what, float(err.max()))


def _kernels_of(fn):
    from bts_amd import ops
    ops.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    ops.profile_enable(False)
    return out, [s for s, _, _ in ops.profile_records()]


def test_repack_writes_the_forms_in_use_and_packs_the_others_on_demand(monkeypatch):
    import bts_amd  # noqa: F401
    from bts_amd import ops
    from bts_amd._lib import lib
    monkeypatch.setenv('BTS_WINO_MIN_WGS', '1')
    monkeypatch.delenv('BTS_WINO', raising=False)
    monkeypatch.delenv('BTS_W3', raising=False)
    monkeypatch.delenv('BTS_PACK_USED', raising=False)
    D = torch.device('cuda:0')
    cin, cout = 32, 32
    x = _rnd((1, 8, 8, 32, cin), 1)
    w = _rnd((3, 3, 3, cin, cout), 2, 0.2)
    b = _rnd((cout,), 3)
    xg, wg, bg = x.to(D), w.to(D), b.to(D)
    ref = lambda wt: (R.conv3d(x.double(), wt.double(), b.double()), R.conv3d(x.double().abs(), wt.double().abs(), b.double().abs()))
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg, cin, cout)          # born whole: all three forms
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'w3_kernel' in names
    _check(y, *ref(w), 'first weights, F(2x2x2,3x3x3)')
    whole = wp.clone()
    # "optimiser step": new weights, re-pack through a descriptor table -- the table describes the form read so far only
    w2 = _rnd((3, 3, 3, cin, cout), 7, 0.2)
    wg.copy_(w2.to(D))
    table = ops.PackTable()
    entry = [(ops.K3S1, ops.ROLE_FWD, wg, wp, cin, cout, cin, 0, 0)]
    table.run(entry)
    torch.cuda.synchronize()
    pairs = (cin // 8) * 2 * 32 * 4
    assert wp.numel() == (27 + 48 + 64) * pairs
    assert torch.equal(wp[:75 * pairs], whole[:75 * pairs]), 'forms nobody read were rewritten'
    assert not torch.equal(wp[75 * pairs:], whole[75 * pairs:])
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'w3_kernel' in names
    _check(y, *ref(w2), 'new weights, the form in use')
    # another form is picked (as another geometry would): it must be packed on the spot from the NEW weights
    gen = lib().query('bts_conv_pack_generation')
    monkeypatch.setenv('BTS_W3', '0')
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'wino_kernel' in names and 'w3_kernel' not in names
    _check(y, *ref(w2), 'new weights, F(2x2,3x3) x direct packed on demand')
    monkeypatch.setenv('BTS_WINO', '0')
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'wino_kernel' not in names and 'w3_kernel' not in names
    _check(y, *ref(w2), 'new weights, implicit GEMM packed on demand')
    assert lib().query('bts_conv_pack_generation') >= gen + 2
    # the next re-pack describes all three forms (the table notices the generation change by itself)
    w3_ = _rnd((3, 3, 3, cin, cout), 9, 0.2)
    wg.copy_(w3_.to(D))
    table.run(entry)
    torch.cuda.synchronize()
    fresh = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg.clone(), cin, cout)
    assert torch.equal(wp, fresh)
    for env, what in ((dict(BTS_WINO='0'), 'implicit GEMM'), (dict(BTS_W3='0'), 'F(2x2,3x3) x direct'), ({}, 'F(2x2x2,3x3x3)')):
        monkeypatch.delenv('BTS_WINO', raising=False)
        monkeypatch.delenv('BTS_W3', raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _check(ops.conv_fwd(ops.K3S1, xg, wp, bg, cout), *ref(w3_), 'third weights, ' + what)


def test_full_pack_switch(monkeypatch):
    """BTS_PACK_USED=0: every re-pack writes all forms (the A/B switch of the change above)"""
    import bts_amd  # noqa: F401
    from bts_amd import ops
    monkeypatch.setenv('BTS_WINO_MIN_WGS', '1')
    monkeypatch.setenv('BTS_PACK_USED', '0')
    D = torch.device('cuda:0')
    cin, cout = 16, 32
    xg = _rnd((1, 8, 8, 32, cin), 1).to(D)
    wg = _rnd((3, 3, 3, cin, cout), 2, 0.2).to(D)
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg, cin, cout)
    ops.conv_fwd(ops.K3S1, xg, wp, None, cout)
    wg.mul_(2.0)
    ops.PackTable().run([(ops.K3S1, ops.ROLE_FWD, wg, wp, cin, cout, cin, 0, 0)])
    torch.cuda.synchronize()
    assert torch.equal(wp, ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg.clone(), cin, cout))


def test_whole_image_pack_between_two_runs_of_a_cached_narrow_table(monkeypatch):
    """Round-5 advisor finding: an image sits in a cached descriptor table that writes ONE form; bts_conv_pack re-writes the whole image
    (marking all forms fresh); the weights change and the cached table runs again -- it writes its one form only, so a launch that then
    selects another form must pack it on the spot.  (The registry used to mark all three forms as table-covered at the bts_conv_pack,
    and the later launch read the OLD weights' form.)  Also: two tables over one image, the narrower one run last; and a run without the
    host copy of the table (nothing may count as fresh then)."""
    import ctypes
    import bts_amd  # noqa: F401
    from bts_amd import ops
    from bts_amd._lib import lib
    monkeypatch.setenv('BTS_WINO_MIN_WGS', '1')
    for k in ('BTS_WINO', 'BTS_W3', 'BTS_PACK_USED'):
        monkeypatch.delenv(k, raising=False)
    D = torch.device('cuda:0')
    cin, cout = 32, 32
    x = _rnd((1, 8, 8, 32, cin), 11)
    b = _rnd((cout,), 13)
    xg, bg = x.to(D), b.to(D)
    ref = lambda wt: (R.conv3d(x.double(), wt.double(), b.double()), R.conv3d(x.double().abs(), wt.double().abs(), b.double().abs()))
    w1 = _rnd((3, 3, 3, cin, cout), 12, 0.2)
    wg = w1.to(D)
    wp = ops.conv_pack(ops.K3S1, ops.ROLE_FWD, wg, cin, cout)
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'w3_kernel' in names                                   # form 4 is the one in use
    entry = [(ops.K3S1, ops.ROLE_FWD, wg, wp, cin, cout, cin, 0, 0)]
    narrow = ops.PackTable()
    narrow.run(entry)                                             # the cached table: describes form 4 only
    key = narrow.key
    # the whole image again through the single-image entry point (same source, same parameters)
    lib().call('bts_conv_pack', ops.K3S1, ops.ROLE_FWD, ctypes.c_void_p(wg.data_ptr()), ctypes.c_void_p(wp.data_ptr()), cin, cout, cin, 0, 0,
               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    w2 = _rnd((3, 3, 3, cin, cout), 14, 0.2)
    wg.copy_(w2.to(D))
    narrow.run(entry)                                             # same key: the CACHED table runs, writing form 4 from the new weights
    assert narrow.key == key
    monkeypatch.setenv('BTS_W3', '0')
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'wino_kernel' in names and 'w3_kernel' not in names
    _check(y, *ref(w2), 'form 2 after bts_conv_pack + cached narrow table: must come from the NEW weights')
    # two tables over the image: the wide one (forms 2 + 4 are in use now) is built and run, then the weights change and the OLD narrow one
    # runs (its device copy still describes form 4 only) -- form 2 is stale again
    wide = ops.PackTable()
    wide.run(entry)
    w3_ = _rnd((3, 3, 3, cin, cout), 15, 0.2)
    wg.copy_(w3_.to(D))
    lib().call('bts_conv_pack_batch', ctypes.c_void_p(narrow.dev.data_ptr()), ctypes.cast(narrow.host, ctypes.c_void_p), narrow.n, narrow.blocks,
               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    y, names = _kernels_of(lambda: ops.conv_fwd(ops.K3S1, xg, wp, bg, cout))
    assert 'wino_kernel' in names
    _check(y, *ref(w3_), 'form 2 after the older, narrower table ran last')
    # a run without the host copy: the library cannot know what was written -- every form is packed on demand
    w4 = _rnd((3, 3, 3, cin, cout), 16, 0.2)
    wg.copy_(w4.to(D))
    lib().call('bts_conv_pack_batch', ctypes.c_void_p(narrow.dev.data_ptr()), None, narrow.n, narrow.blocks,
               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _check(ops.conv_fwd(ops.K3S1, xg, wp, bg, cout), *ref(w4), 'form 2 after a table run without its host copy')
    monkeypatch.setenv('BTS_WINO', '0')
    _check(ops.conv_fwd(ops.K3S1, xg, wp, bg, cout), *ref(w4), 'form 1 after a table run without its host copy')
    # released images leave the registry
    assert lib()._bts_conv_pack_forget(ctypes.c_void_p(wp.data_ptr())) == 1
    assert lib()._bts_conv_pack_forget(ctypes.c_void_p(wp.data_ptr())) == 0
