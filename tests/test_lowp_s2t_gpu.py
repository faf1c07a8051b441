"""-m gpu: the LDS-tiled stride-2 3x3x3 convolution of the 16-bit storage path (csrc/lowp_s2t.hip: the Conv3D of downsample.py:30-48 at
the top level, 32 -> <= 32 channels) at shapes it takes (>= 512 tiles of 16x4x2 outputs, even extents; the small cases of
test_lowp_gpu.py keep exercising the gather kernels).  Through the C ABI, against the oracle's op on the same 16-bit-rounded operands in
fp64 under |err| <= 8 * 2^-24 * sum|a_i b_i| + u * |ref|; the launch records must show `lp_s2t_kernel`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)
U = {'float16': 2.0 ** -11, 'bfloat16': 2.0 ** -8}


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


CASES = [
    # n, (D,H,W), Cout, slab_in, slab_out
    (2, (64, 64, 64), 32, True, False),       # 512 tiles, whole; the input a 32-channel view of a 64-channel slab (the model's case)
    (2, (60, 68, 72), 24, False, True),       # ragged tiles on every axis (30x34x36 outputs), 24 couts, output into a slab view
    (1, (96, 64, 128), 8, True, True),        # 8 couts (one 16-byte piece per voxel pair)
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d-%dx%dx%d-%d' % (c[0], *c[1], c[2]))
def test_forward(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    n, (d, h, w), cout, slab_in, slab_out = case
    cin = 32
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(hash((d, h, w, cout)) % 10000)
    x = torch.randn((n, d, h, w, cin), generator=g)
    wt = torch.randn((3, 3, 3, cin, cout), generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr = _round(x, tdt), _round(wt, tdt)
    ref = R.conv3d(xr, wr, b.double(), stride=2)
    bound = 8 * 2.0 ** -24 * R.conv3d(xr.abs(), wr.abs(), None, stride=2) + U[dtype] * ref.abs() + 1e-30
    ldx = 64 if slab_in else cin
    xin = torch.full((n, d, h, w, ldx), 5.0, dtype=tdt, device=DEV)       # (the other half of the slab must never be read as data)
    xin[..., :cin] = x.to(tdt).to(DEV)
    wp = lowp.pack(ops.K3S2, code, wt.to(DEV), cin, cout)
    out = None
    if slab_out:
        buf = torch.full((n, d // 2, h // 2, w // 2, cout + 24), 7.0, dtype=tdt, device=DEV)
        out = buf[..., 8:8 + cout]
    ops.profile_enable(True)
    y = lowp.conv(ops.K3S2, code, tdt, xin[..., :cin], wp, b.to(DEV), cout, out=out)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    syms = [s for s, _, _ in ops.profile_records()]
    assert syms == ['lp_s2t_kernel'], syms
    err = (y.double().cpu() - ref).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, '%s: error %.3e is %.2fx the stated bound' % (dtype, float(err.max()), worst)
    if slab_out:
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + cout:] == 7.0).all())
    # the gather kernel on the same operands: the two forms agree to the rounding of the result
    import os
    os.environ['BTS_LP_S2T'] = '0'
    try:
        y2 = lowp.conv(ops.K3S2, code, tdt, xin[..., :cin], wp, b.to(DEV), cout)
    finally:
        del os.environ['BTS_LP_S2T']
    assert float((y2.float() - y.float()).abs().max()) <= 4 * U[dtype] * float(ref.abs().max())
