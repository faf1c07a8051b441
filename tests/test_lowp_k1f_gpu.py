"""-m gpu: the weight gradients of the TWO convolutions that read a ResnetBlock's input from ONE pass over it (bts_lp_conv3d_bwd_weight_pair,
round 6): resnet.py:134 conv1 (3x3x3) and resnet.py:118 the shortcut (1x1x1) are both applied to `inputs`, so under train.py:151
  dW3[t][c][k] = sum_v x[v + off_t][c] dc1[v][k],   dW1[c][k] = sum_v x[v][c] dres[v][k]
share their P operand.  `lp_wgd_kernel<.., K1F>` forms dW1 as one more accumulator, fed by the centre-tap fragments of x and by planes of
dres that ride in Q-ring slots the 3x3x3 contraction has finished with.

Against torch autograd of the ORACLE's convs on the same rounded x / dy (fp64), accumulated onto non-zero buffers, under the weight-gradient
bound of tests/test_lowp_gpu.py (8 * 2^-24 * sum|a b| + 2^-22 * (|ref| + |old|)); launch records assert ONE weight-gradient kernel ran (no
1x1x1 launch); shapes outside the streaming kernel decline up front."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

DEV = torch.device('cuda', 0)


def _round(t, tdt):
    return t.to(tdt).to(torch.float64)


CASES = [
    # (N,D,H,W), Cin, Cout, fold (dup_start, dup_shift) or None, x a slab view, dres a slab view, bias gradient wanted
    ((2, 8, 32, 64), 32, 32, None, False, False, True),        # several columns
    ((1, 37, 16, 32), 64, 16, None, True, True, False),        # z chunks with a ragged last one, two cin blocks, half-empty cout block, views
    ((2, 16, 16, 32), 32, 32, (16, 16), True, False, True),    # the folded duplicate slice: both copies of BOTH kernels get the gradient
    ((2, 16, 16, 32), 64, 128, None, False, False, False),     # two cin blocks x four cout blocks
    ((1, 24, 24, 32), 16, 32, None, False, True, True),        # a 16-channel input (the padded first block), three columns
]


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: '%s-%d-%d-%s' % ('x'.join(map(str, c[0])), c[1], c[2], 'fold' if c[3] else 'plain'))
def test_pair_of_weight_gradients(case, dtype):
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    (n, d, h, w), cin, cout, fold, slab_x, slab_q, want_db = case
    code, tdt = lowp.DTYPES[dtype]
    g = torch.Generator().manual_seed(cin * 5 + cout)
    dup_start, dup_shift = fold if fold else (0, 0)
    cin_ref = cin + dup_shift
    x = torch.randn((n, d, h, w, cin), generator=g)
    dy3 = torch.randn((n, d, h, w, cout), generator=g)
    dy1 = torch.randn((n, d, h, w, cout), generator=g)
    xr, dy3r, dy1r = _round(x, tdt), _round(dy3, tdt), _round(dy1, tdt)
    full = torch.cat([xr[..., dup_start:dup_start + dup_shift], xr], dim=-1) if fold else xr

    def grads(k, dyr):
        wz = torch.zeros((k, k, k, cin_ref, cout), dtype=torch.float64, requires_grad=True)
        bz = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        dw_ref, db_ref = torch.autograd.grad(R.conv3d(full, wz, bz), (wz, bz), dyr)
        wz2 = torch.zeros((k, k, k, cin_ref, cout), dtype=torch.float64, requires_grad=True)
        mag = torch.autograd.grad(R.conv3d(full.abs(), wz2, None), wz2, dyr.abs())[0]
        return dw_ref, db_ref, mag
    dw3_ref, db3_ref, mag3 = grads(3, dy3r)
    dw1_ref, _, mag1 = grads(1, dy1r)
    dw3_0 = torch.randn(dw3_ref.shape, generator=g)
    dw1_0 = torch.randn(dw1_ref.shape, generator=g)
    db_0 = torch.randn(cout, generator=g)
    bufx = torch.zeros((n, d, h, w, cin + 16), dtype=tdt, device=DEV)
    xin = bufx[..., 8:8 + cin] if slab_x else torch.empty((n, d, h, w, cin), dtype=tdt, device=DEV)
    xin.copy_(x.to(tdt).to(DEV))
    bufq = torch.full((n, d, h, w, cout + 16), 5.0, dtype=tdt, device=DEV)
    q1 = bufq[..., 8:8 + cout] if slab_q else torch.empty((n, d, h, w, cout), dtype=tdt, device=DEV)
    q1.copy_(dy1.to(tdt).to(DEV))
    dw3, dw1, db = dw3_0.to(DEV).contiguous(), dw1_0.to(DEV).contiguous(), db_0.to(DEV).contiguous()
    ops.profile_enable(True)
    ok = lowp.conv_bwd_weight_pair(code, xin, dy3.to(tdt).to(DEV), q1, dw3, dw1, db if want_db else None, dup_start, dup_shift, accumulate=True)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    assert ok, 'the streaming kernel declined a shape it is built for'
    ran = [r[0] for r in ops.profile_records()]
    assert ran.count('lp_wgd_kernel') == 1 and 'lp_wgrad_kernel' not in ran, ran
    for name, got, ref, old, mag in (('dw3', dw3, dw3_ref, dw3_0, mag3), ('dw1', dw1, dw1_ref, dw1_0, mag1)):
        err = (got.double().cpu() - (ref + old.double())).abs()
        bound = 8 * 2.0 ** -24 * mag + 2.0 ** -22 * (ref.abs() + old.double().abs()) + 1e-9
        assert float((err / bound).max()) <= 1.0, '%s: max err %.3e at %.2fx the bound' % (name, float(err.max()), float((err / bound).max()))
    if want_db:
        assert float((db.double().cpu() - (db3_ref + db_0.double())).abs().max()) <= 1e-5 * float(db3_ref.abs().max()) + 1e-5
    # the same numbers as the two launches it replaces (same partial layout, same fixed-order finalize)
    dw3b, dw1b = dw3_0.to(DEV).contiguous(), dw1_0.to(DEV).contiguous()
    assert lowp.conv_bwd_weight(ops.K3S1, code, xin, dy3.to(tdt).to(DEV), dw3b, None, dup_start, dup_shift, accumulate=True)
    assert lowp.conv_bwd_weight(ops.K1, code, xin, q1, dw1b, None, dup_start, dup_shift, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(dw3b, dw3)
    assert float((dw1b - dw1).abs().max()) <= 2.0 ** -20 * float(dw1.abs().max())


@pytest.mark.parametrize('dtype', ['float16', 'bfloat16'])
def test_input_as_a_list_of_dense_operands(dtype):
    """x_split: the block input as TWO dense 32-channel tensors (the operands of the decoder's concat, decoder.py:75) -- a
    workgroup of the streaming kernel owns one 32-channel block of P anyway.  Same numbers as the one-tensor call, bit for bit."""
    import bts_amd  # noqa: F401
    from bts_amd import lowp, ops
    code, tdt = lowp.DTYPES[dtype]
    n, d, h, w, cin, cout = 2, 16, 16, 32, 64, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn((n, d, h, w, cin), generator=g).to(tdt).to(DEV)
    dy3 = torch.randn((n, d, h, w, cout), generator=g).to(tdt).to(DEV)
    dy1 = torch.randn((n, d, h, w, cout), generator=g).to(tdt).to(DEV)
    buf = torch.full((3, n, d, h, w, 32), 7.0, dtype=tdt, device=DEV)
    buf[0].copy_(x[..., :32])
    buf[1].copy_(x[..., 32:])
    outs = []
    for xin in (x, buf[:2]):
        dw3 = torch.zeros((3, 3, 3, cin, cout), device=DEV)
        dw1 = torch.zeros((1, 1, 1, cin, cout), device=DEV)
        assert lowp.conv_bwd_weight_pair(code, xin, dy3, dy1, dw3, dw1, None, 0, 0, accumulate=False)
        outs.append((dw3, dw1))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and against the oracle (the one-tensor form is held to it case by case above; one check here)
    xr, q3 = x.double().cpu(), dy3.double().cpu()
    wz = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(R.conv3d(xr, wz, None), wz, q3)[0]
    wz2 = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    mag = torch.autograd.grad(R.conv3d(xr.abs(), wz2, None), wz2, q3.abs())[0]
    assert float(((outs[1][0].double().cpu() - ref).abs() / (8 * 2.0 ** -24 * mag + 2.0 ** -22 * ref.abs() + 1e-9)).max()) <= 1.0


def test_shapes_outside_the_streaming_kernel_decline():
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    code, tdt = lowp.DTYPES['bfloat16']
    for (n, d, h, w), cin, cout in (((1, 6, 10, 20), 64, 64), ((2, 8, 8, 16), 32, 32)):
        x = torch.zeros((n, d, h, w, cin), dtype=tdt, device=DEV)
        q = torch.zeros((n, d, h, w, cout), dtype=tdt, device=DEV)
        dw3 = torch.zeros((3, 3, 3, cin, cout), device=DEV)
        dw1 = torch.zeros((1, 1, 1, cin, cout), device=DEV)
        assert not lowp.conv_bwd_weight_pair(code, x, q, q, dw3, dw1, None)
