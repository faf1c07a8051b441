"""CPU: the host-only planning code of the C-ABI library (split-K plans, workspace sizing, dispatch thresholds, descriptor tables) walked
over every layer shape of the three single-GPU BASELINE configurations (configs[1]: 1 x 128^3, configs[2]: 8 x 128^3, configs[4]:
1 x 160x192x160; CLI-default model) and the option matrix of row f-4 (max-pool / linear samplers change which kinds occur,
not the shapes).  No GPU, no compute call: every function here returns before any HIP call.  tests/test_asan_host.py runs this file and
tests/test_abi.py once more against the AddressSanitizer + UBSan build of the host code (`make -C .../csrc asan`).

What is asserted: the queries are total (no crash, no negative size other than the documented -1 "this form declines the shape"),
deterministic (the same answer twice) and monotone where the ABI says so (a workspace for N samples is never smaller than for one)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRIDS = [(1, 128, 128, 128), (8, 128, 128, 128), (1, 160, 192, 160)]
F, DEPTH, G, R = 32, 4, 8, 8


def layer_shapes(n, d, h, w):
    """(kind, N, D, H, W, Cin, Cout) of every conv of the CLI-default model on an (n, d, h, w) volume: encoder.py:43-67,
    decoder.py:38-63, vae.py:53-99"""
    K1, K3S1, K3S2, K3S2T = 0, 1, 2, 3
    out = []

    def block(lvl, cin, f):
        s = (n, d >> lvl, h >> lvl, w >> lvl)
        out.append((K1,) + s + (cin, f))
        out.append((K3S1,) + s + (cin, f))
        out.append((K3S1,) + s + (f, f))
        out.append((K1,) + s + (f, 1))
    for lvl in range(DEPTH):
        f = F << lvl
        cin = 2 if lvl == 0 else f // 2
        for b in range(lvl + 1):
            block(lvl, cin if b == 0 else f * (b + 1) if lvl else f, f)
        if lvl < DEPTH - 1:
            out.append((K3S2, n, d >> lvl, h >> lvl, w >> lvl, f * (lvl + 1), f))
    top = F << (DEPTH - 1)
    for lvl in range(DEPTH - 2, -1, -1):       # decoder
        f = F << lvl
        cin = top * DEPTH if lvl == DEPTH - 2 else f * 2
        out.append((K3S2T, n, d >> (lvl + 1), h >> (lvl + 1), w >> (lvl + 1), cin, f))
        out.append((K1, n, d >> (lvl + 1), h >> (lvl + 1), w >> (lvl + 1), cin, f))      # linear up-sampler (f-4)
        block(lvl, f * (lvl + 1) + f, f)
    out.append((K1, n, d, h, w, F, 3))
    if (d, h, w) == (128, 128, 128):            # the VAE branch is tied to the training crop (vae.py:101-111)
        out.append((K3S2, n, d >> 3, h >> 3, w >> 3, top * DEPTH, 16))
        for lvl in range(DEPTH - 1, -1, -1):
            f = F << lvl
            out.append((K3S2T, n, d >> (lvl + 1), h >> (lvl + 1), w >> (lvl + 1), 1 if lvl == DEPTH - 1 else f * 2, f))
            if lvl < DEPTH - 1:
                block(lvl, f, f)
        out.append((K3S1, n, d, h, w, F, 2))
    return sorted(set(out))


@pytest.fixture(scope='module')
def L():
    import bts_amd  # noqa: F401
    from bts_amd._lib import lib
    lb = lib()
    want = os.environ.get('BTS_EXPECT_LIB')        # the sanitizer run: make sure the instrumented build is the one that answered
    if want:
        assert want in open('/proc/self/maps').read(), '%s is not mapped into this process' % want
    return lb


@pytest.mark.parametrize('grid', GRIDS, ids=lambda g: '%dx%dx%dx%d' % g)
def test_conv_planning_queries_are_total_and_deterministic(L, grid):
    n, d, h, w = grid
    shapes = layer_shapes(n, d, h, w)
    assert len(shapes) >= 40
    q7 = ['bts_conv3d_fwd_workspace', 'bts_conv3d_bwd_data_workspace', 'bts_conv3d_bwd_weight_workspace', 'bts_lp_conv3d_workspace',
          'bts_lp_conv3d_bwd_data_workspace', 'bts_lp_conv3d_bwd_weight_workspace']
    c7 = ['bts_conv3d_fwd_config', 'bts_conv3d_bwd_data_config']
    for (kind, N, D, H, W, ci, co) in shapes:
        for name in q7 + c7:
            f = getattr(L, '_' + name)
            a, b = f(kind, N, D, H, W, ci, co), f(kind, N, D, H, W, ci, co)
            assert a == b and a >= -3, (name, kind, N, D, H, W, ci, co, a, b)     # -1: the form declines; -3: BTS_ERR_UNSUPPORTED from a config query
            if name in q7 and N > 1 and a >= 0:
                one = f(kind, 1, D, H, W, ci, co)
                assert one <= a or one < 0 or a == 0, (name, kind, N, D, H, W, ci, co, one, a)
        if kind == 1:
            for name, args in (('bts_conv3d_fwd_can_fuse', (N, D, H, W, ci, co)), ('bts_conv3d_bwd_data_pair_workspace', (N, D, H, W, ci, co)),
                               ('bts_conv3d_fwd_gn_workspace', (kind, N, D, H, W, ci, co, G)),
                               ('bts_lp_conv3d_fwd_gn_workspace', (N, D, H, W, ci, co, G)),
                               ('bts_lp_conv3d_gnin_fwd_gn_workspace', (N, D, H, W, ci, co, G, G)),
                               ('bts_lp_conv3d_gnin_train_ok', (N, D, H, W, ci, co, G, G)),
                               ('bts_lp_conv3d_bwd_data_gn_bwd_workspace', (N, D, H, W, ci, co, G))):
                f = getattr(L, '_' + name)
                a = f(*args)
                assert a == f(*args) and a >= -4, (name, args, a)
        if kind == 3:
            a = L._bts_lp_convT3d_fwd_gn_workspace(N, D, H, W, ci, co, G)
            assert a == L._bts_lp_convT3d_fwd_gn_workspace(N, D, H, W, ci, co, G) and a >= -4
        for role in (0, 1):
            if co >= 1 and ci >= 1:
                fl = L._bts_conv_packed_floats(kind, role, ci, co)
                by = L._bts_lp_packed_bytes(kind, role, ci, co)
                assert fl > 0 and by != 0, (kind, role, ci, co, fl, by)


@pytest.mark.parametrize('grid', GRIDS, ids=lambda g: '%dx%dx%dx%d' % g)
def test_normalisation_and_gate_workspaces(L, grid):
    n, d, h, w = grid
    for lvl in range(DEPTH):
        V = (d >> lvl) * (h >> lvl) * (w >> lvl)
        f = F << lvl
        for mode in (0, 1):
            assert L._bts_gn_workspace(n, V, f, G, mode) >= 0 and L._bts_gn_bwd_workspace(n, V, f, G, mode) >= 0
        assert L._bts_lp_gn_workspace(n, V, f, G) >= -1 and L._bts_lp_gn_bwd_workspace(n, V, f, G) >= -1
        assert L._bts_colsum_workspace(n, V, f) >= 0 and L._bts_lp_colsum_workspace(n, V, f) >= -1
        assert L._bts_se_bwd_workspace(n, V, f, f // R) >= 0 and L._bts_lp_se_bwd_workspace(n, V, f, f // R) >= -1
        assert L._bts_block_bwd_workspace(n, V, f, f // R, G) >= -1 and L._bts_lp_block_bwd_workspace(n, V, f, f // R, G) >= -1
        assert L._bts_lp_conv1_gap_workspace(n, V, f) >= -1
        assert L._bts_channel_moments_workspace(f) >= 0
    assert L._bts_dense_workspace(n, 8192, 256) >= 0 and L._bts_dense_workspace(n, 128, 512) >= 0
    assert L._bts_loss_workspace() > 0 and L._bts_l2_workspace() > 0 and L._bts_lp_head_bwd_workspace(32, 3) >= 0


def test_pack_descriptor_tables_are_written_inside_their_bounds(L):
    """bts_conv_pack_desc / bts_lp_pack_desc fill entry `index` of a host table of `*_desc_bytes()` entries: a guard band behind the
    table must stay untouched (the ASan run checks the same thing from the allocator's side)"""
    shapes = [s for s in layer_shapes(1, 128, 128, 128) if s[5] % 8 == 0 and s[6] % 8 == 0][:24]
    for desc_bytes, desc, sizes in ((L._bts_conv_pack_desc_bytes, L._bts_conv_pack_desc, lambda k, r, ci, co: L._bts_conv_packed_floats(k, r, ci, co) * 4),
                                    (L._bts_lp_pack_desc_bytes, L._bts_lp_pack_desc, lambda k, r, ci, co: L._bts_lp_packed_bytes(k, r, ci, co))):
        nb = desc_bytes()
        assert nb > 0
        guard = 256
        buf = ctypes.create_string_buffer(b'\xa5' * (nb * len(shapes) + guard), nb * len(shapes) + guard)
        first = 0
        for i, (kind, N, D, H, W, ci, co) in enumerate(shapes):
            for role in (0, 1):
                if sizes(kind, role, ci, co) <= 0:
                    continue
                # (fake, never dereferenced device addresses: the descriptor only records them)
                r = desc(ctypes.cast(buf, ctypes.c_void_p), i, first, kind, role, ctypes.c_void_p(0x1000), ctypes.c_void_p(0x100000), ci, co, ci, 0, 0)
                assert r > 0, (kind, role, ci, co, r)      # = blocks of this entry
            first += r
        assert buf.raw[nb * len(shapes):] == b'\xa5' * guard
