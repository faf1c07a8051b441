"""Thin, checked Python wrappers over the C ABI (include/bts_hip.h): torch tensors in, raw pointers out.

torch is used for device memory and the current HIP stream only.  Every function enqueues on
torch.cuda.current_stream() and raises RuntimeError on a non-zero status -- there is no CPU path.
Activations are [N,D,H,W,C] fp32 tensors whose last-dim stride is 1 and whose voxel stride (`ld`) may exceed C
(channel slices of a slab).
"""
import ctypes
import weakref

import torch

from ._lib import ERRORS, lib

K1, K3S1, K3S2, K3S2T = 0, 1, 2, 3
ROLE_FWD, ROLE_BWD = 0, 1
FLAG_SIGMOID, FLAG_ACCUM = 1, 2
GN_SLAB, GN_CHANNEL = 0, 1

_workspaces = {}

# ---- optional live kernel timing (bench.py roofline): the library records HIP events on the launch stream around its
# igemm_kernel / wgrad_kernel launches (bts_profile_*), i.e. the same per-launch durations rocprofv3 reports ----
_CFG = {0: '2,1,4,1', 1: '2,2,4,1', 2: '1,2,2,2', 3: '1,1,2,2', 4: '1,1,4,1', 5: '4,1,4,1'}


_VARIANTS = {33: {1: '<0,5>', 2: '<0,4>', 3: '<1,5>', 4: '<1,4>'},          # lp_s1d_kernel<MODE, TXL>
             32: {1: '<3x3x3>', 2: '<1x1x1>'}}                                 # lp_wgrad_kernel by kernel size


def kernel_symbol(sym, detail=False):
    """name of a profiled launch; bits 16+ of `sym` carry a kernel-specific variant, shown only with detail=True (tests count launches
    by the plain name, bench.py's kernel_breakdown tells the variants apart)"""
    var, sym = sym >> 16, sym & 0xffff
    if var and detail and sym in _VARIANTS:
        return kernel_symbol(sym) + _VARIANTS[sym].get(var, '<%d>' % var)
    if sym == 20:
        return 'upm_kernel'
    if sym == 21:
        return 'k1s_kernel'
    if sym == 22:
        return 'dsc_kernel'
    if sym == 23:
        return 'wino_kernel'
    if sym == 24:
        return 'wgw_kernel'
    if sym == 25:
        return 'c2_kernel'
    if sym == 26:
        return 'k1w_kernel'
    if sym == 27:
        return 'w3_kernel'
    if sym == 30:
        return 'lp_conv_s1_kernel'
    if sym == 31:
        return 'lp_conv_gather_kernel'
    if sym == 32:
        return 'lp_wgrad_kernel'
    if sym == 33:
        return 'lp_s1d_kernel'
    if sym == 34:
        return 'lp_k1_kernel'
    if sym == 35:
        return 'lp_up_kernel'
    if sym == 36:
        return 'lp_wgs_kernel'
    if sym == 37:
        return 'lp_wgd_kernel'
    if sym == 38:
        return 'lp_s1z_kernel'
    if sym == 39:
        return 'lp_s2t_kernel'
    if sym == 40:
        return 'lp_c2_kernel'
    if sym >= 100:  # 100 + (MODE << 2 | FIXG)
        return 'wgrad_kernel<%d,%d>' % ((sym - 100) >> 2, (sym - 100) & 3)
    return 'igemm_kernel<%s,%d>' % (_CFG[sym & 7], 4 if sym & 8 else 1)


def profile_enable(on):
    lib().call('bts_profile_enable', 1 if on else 0)


def profile_records(detail=False):
    """-> [(symbol, algorithmic_flops, ms)] ; synchronise the device first.  detail: kernel variants in the symbol names"""
    L = lib()
    out = []
    sym, fl, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
    for i in range(L._bts_profile_count()):
        L.call('bts_profile_get', i, ctypes.byref(sym), ctypes.byref(fl), ctypes.byref(ms))
        out.append((kernel_symbol(sym.value, detail), fl.value, ms.value))
    return out


def conv_bwd_data_pair(dy, wp_bwd, dy2, wp2_bwd, dx, accumulate):
    """dx (+)= bwd_data(3x3x3)(dy) + bwd_data(1x1x1)(dy2): both gradient paths into a ResNet block's input in one pass"""
    n, d, h, w, cin = dx.shape
    cout = dy.shape[4]
    nb = lib().query('bts_conv3d_bwd_data_pair_workspace', n, d, h, w, cin, cout)
    ws = workspace(nb, dy.device) if nb > 0 else None
    lib().call('bts_conv3d_bwd_data_pair', _p(dy), _p(wp_bwd), _p(dy2), _p(wp2_bwd), _p(dx), _p(ws), nb, n, d, h, w, cin,
               ld_of(dx), cout, ld_of(dy), ld_of(dy2), FLAG_ACCUM if accumulate else 0, _stream())
    return dx


def conv_flops(kind, n, d, h, w, cin, cout):
    """algorithmic FLOPs of one conv call, SURVEY 8(d) convention (MAC = 2; (d,h,w) = forward INPUT dims)"""
    v = n * d * h * w
    if kind == K1:
        return 2.0 * cin * cout * v
    if kind == K3S2:
        return 2.0 * 27 * cin * cout * (v // 8)
    return 2.0 * 27 * cin * cout * v   # K3S1; K3S2T counts 27*Cin*Cout*V_in MACs


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_side = {}
_side_enabled = True


_fence = {}


def step_fence(kind='train', depth=2):
    """Bound how far the host runs ahead of the GPU: call at the START of a step -- waits (host side) until the step issued `depth`
    calls earlier has finished, then notes this one.  A step enqueues in 2-12 ms what the GPU runs in 7-80 ms; left unbounded, the
    host gets many steps ahead, every one of them holding its activations (blocks freed on a side stream cannot be handed out again
    before that stream's work has run), the caching allocator answers with new hipMalloc segments inside the timed region -- the
    reserved pool was seen to grow from 100 to 140 GB for 42 GB of live tensors, and steps to take 120-400 ms instead of 79.  With
    at most `depth` steps in flight the allocation pattern is the same every step.  BTS_STEP_FENCE=0 disables it (A/B)."""
    import os
    if not torch.cuda.is_available() or os.environ.get('BTS_STEP_FENCE') == '0':
        return
    key = (torch.cuda.current_device(), kind)
    q = _fence.setdefault(key, [])
    while len(q) >= depth:
        q.pop(0).synchronize()
    ev = torch.cuda.Event()
    q.append(ev)
    return ev


def step_fence_done(ev):
    """call at the END of the step with what step_fence returned: the event is recorded behind everything the step enqueued on the
    current stream (side streams are joined before a step returns)"""
    if ev is not None:
        ev.record(torch.cuda.current_stream())


def enable_side_streams(on):
    """switch the extra streams off / on at run time (bench.py measures per-kernel launch durations with one stream: a kernel
    that shares the chip with another stream's kernels has no launch duration of its own)"""
    global _side_enabled
    join_side_stream()
    _side_enabled = bool(on)


def side_stream(which='wgrad'):
    """Extra HIP streams of a training step (one process still drives one GPU):
      'wgrad'  the weight gradients (a third of the step, needed only by the optimiser) run there while the data-gradient
               chain -- with its many short normalisation / gate kernels that leave most CUs idle -- keeps the main stream;
      'gate'   a ResNet block's shortcut / squeeze-excitation branch (HBM-bound 1x1x1 conv, pooling, tiny MLP and their
               gradients: resnet.py:118-130) next to its conv branch (matrix-pipe-bound), joined where the two meet.
    BTS_WGRAD_STREAM=0 / BTS_GATE_STREAM=0 put the respective work back on the main stream (A/B aids)."""
    import os
    if not _side_enabled or not torch.cuda.is_available() or os.environ.get('BTS_%s_STREAM' % which.upper()) == '0':
        return None
    key = (torch.cuda.current_device(), which)
    s = _side.get(key)
    if s is None:
        s = torch.cuda.Stream(device=key[0])
        _side[key] = s
    return s


def join_side_stream():
    """the current stream waits for everything enqueued on the side streams so far (before the regulariser / the gradient
    exchange / the optimiser touch the parameter gradients)"""
    if not torch.cuda.is_available():
        return
    dev = torch.cuda.current_device()
    for (d, _), s in _side.items():
        if d == dev:
            torch.cuda.current_stream().wait_stream(s)


def wait_side_stream_event(which='wgrad'):
    """the current stream waits for what side stream `which` holds at this moment (an event recorded there now); later launches on
    the side stream are not waited for.  A gloo group drains the producing stream on the host instead (parallel._sum_over_ranks)."""
    if not torch.cuda.is_available():
        return
    s = _side.get((torch.cuda.current_device(), which))
    if s is None:
        return
    ev = torch.cuda.Event()
    ev.record(s)
    torch.cuda.current_stream().wait_event(ev)


def workspace(nbytes, device):
    """one grow-only scratch buffer per device AND stream; all users of a buffer are ordered on that stream"""
    key = (device.type, device.index, torch.cuda.current_stream().cuda_stream if device.type == 'cuda' else 0)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _check(t, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError('%s must live on the GPU (no CPU fallback exists for the product path)' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be float32' % name)


def ld_of(t):
    """voxel stride of an NDHWC view (validates that the view is a channel slice of a dense NDHWC buffer)"""
    _check(t)
    if t.dim() != 5 or t.stride(4) != 1 and t.shape[4] != 1:
        raise RuntimeError('expected an [N,D,H,W,C] tensor with unit channel stride, got strides %s' % (t.stride(),))
    n, d, h, w, c = t.shape
    ld = t.stride(3)
    if w == 1:
        ld = t.stride(2) if h > 1 else (t.stride(1) if d > 1 else (t.stride(0) if n > 1 else c))
    exp = (d * h * w * ld, h * w * ld, w * ld, ld)
    for i in range(4):
        if t.shape[i] > 1 and t.stride(i) != exp[i]:
            raise RuntimeError('tensor is not a channel slice of a dense NDHWC buffer: strides %s' % (t.stride(),))
    if ld < c:
        raise RuntimeError('bad voxel stride')
    return ld


def conv_kind_taps(kind):
    return 1 if kind == K1 else 27


def conv_pack(kind, role, w, cin_ref, cout, cin_slab=None, dup_start=0, dup_shift=0):
    """reference-layout kernel -> packed image for the implicit-GEMM kernel"""
    _check(w, 'kernel')
    cin_slab = cin_ref if cin_slab is None else cin_slab
    n = lib().query('bts_conv_packed_floats', kind, role, cin_slab, cout)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    wc = w.contiguous()
    lib().call('bts_conv_pack', kind, role, _p(wc), _p(wp), cin_ref, cout, cin_slab, dup_start, dup_shift,
               _stream())
    if wc is not w:      # packed from a temporary copy: the library must not re-pack a form on demand from that address later
        lib()._bts_conv_pack_forget(_p(wp))
    else:                # the library keys its record of the image by address: drop it with the tensor
        fin = weakref.finalize(wp, _forget_image, wp.data_ptr())
        fin.atexit = False
    return wp


def _forget_image(ptr):
    try:
        lib()._bts_conv_pack_forget(ctypes.c_void_p(ptr))
    except Exception:      # (interpreter shutdown)
        pass


class PackTable(object):
    """Device-resident descriptor table for bts_conv_pack_batch: entries = [(kind, role, w, wp, cin_ref, cout, cin_slab,
    dup_start, dup_shift)] with torch tensors w (reference layout) / wp (packed image).  Rebuilt only when a pointer or
    the entry list changes."""

    def __init__(self):
        self.key = None
        self.dev = None
        self.host = None
        self.n = 0
        self.blocks = 0

    def run(self, entries):
        if not entries:
            return
        # (+ the library's form-usage generation: images are re-packed in the forms their layers read; a table built before a new form
        # was first read is rebuilt here, once)
        key = (lib().query('bts_conv_pack_generation'),) + tuple((e[0], e[1], e[2].data_ptr(), e[3].data_ptr()) + tuple(e[4:]) for e in entries)
        if key != self.key:
            L = lib()
            nb = L._bts_conv_pack_desc_bytes()
            host = (ctypes.c_char * (nb * len(entries)))()
            first = 0
            for i, (kind, role, w, wp, cin_ref, cout, cin_slab, dup_start, dup_shift) in enumerate(entries):
                _check(w, 'kernel')
                _check(wp, 'packed image')
                if not w.is_contiguous():
                    raise ValueError('conv_pack_batch: kernels must be contiguous')
                r = L._bts_conv_pack_desc(ctypes.cast(host, ctypes.c_void_p), i, first, kind, role, _p(w), _p(wp), cin_ref,
                                          cout, cin_slab, dup_start, dup_shift)
                if r <= 0:
                    raise RuntimeError('bts_conv_pack_desc failed: %s' % ERRORS.get(r, r))
                first += r
            dev = entries[0][3].device
            self.dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
            self.host = host      # (kept: the library reads each entry's forms from the host copy of the table it is asked to run)
            self.key, self.n, self.blocks = key, len(entries), first
        lib().call('bts_conv_pack_batch', _p(self.dev), ctypes.cast(self.host, ctypes.c_void_p), self.n, self.blocks, _stream())


def conv_packed_empty(kind, role, cin_slab, cout, device):
    return torch.empty(lib().query('bts_conv_packed_floats', kind, role, cin_slab, cout), dtype=torch.float32, device=device)


def conv_out_shape(kind, x_shape, cout):
    n, d, h, w, _ = x_shape
    if kind == K3S2:
        return (n, d // 2, h // 2, w // 2, cout)
    if kind == K3S2T:
        return (n, 2 * d, 2 * h, 2 * w, cout)
    return (n, d, h, w, cout)


def conv_fwd(kind, x, wp, bias, cout, out=None, sigmoid=False):
    n, d, h, w, cin = x.shape
    if out is None:
        out = torch.empty(conv_out_shape(kind, x.shape, cout), dtype=torch.float32, device=x.device)
    nb = lib().query('bts_conv3d_fwd_workspace', kind, n, d, h, w, cin, cout)
    ws = workspace(nb, x.device) if nb > 0 else None
    if True:
        lib().call('bts_conv3d_fwd', kind, _p(x), _p(wp), _p(bias), _p(out), _p(ws), nb, n, d, h, w, cin, ld_of(x), cout,
                   ld_of(out), FLAG_SIGMOID if sigmoid else 0, _stream())
    return out


def conv_fwd_fused2(x, wp3, bias3, wp1, bias1, cout):
    """(c1, res) = (conv3x3x3(x)+bias3, conv1x1x1(x)+bias1) from one pass over x, or None when the selected tiling cannot
    hold the second accumulator set (the caller then launches the two convolutions separately)"""
    n, d, h, w, cin = x.shape
    if not lib()._bts_conv3d_fwd_can_fuse(n, d, h, w, cin, cout):
        return None
    c1 = torch.empty((n, d, h, w, cout), dtype=torch.float32, device=x.device)
    res = torch.empty_like(c1)
    lib().call('bts_conv3d_fwd_fused2', _p(x), _p(wp3), _p(bias3), _p(c1), _p(wp1), _p(bias1), _p(res), n, d, h, w, cin,
               ld_of(x), cout, cout, cout, _stream())
    return c1, res


def conv_fwd_gn(kind, x, wp, bias, cout, groups, eps):
    """(y, mean, rstd): y = conv(x) + bias (dense) and the slab-mode GroupNorm statistics of y, in one pass where the
    tiled kernel can emit them from its epilogue (the library falls back to bts_gn_stats otherwise)"""
    n, d, h, w, cin = x.shape
    y = torch.empty(conv_out_shape(kind, x.shape, cout), dtype=torch.float32, device=x.device)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    nb = lib().query('bts_conv3d_fwd_gn_workspace', kind, n, d, h, w, cin, cout, groups)
    ws = workspace(nb, x.device)
    lib().call('bts_conv3d_fwd_gn', kind, _p(x), _p(wp), _p(bias), _p(y), _p(ws), nb, n, d, h, w, cin, ld_of(x), cout, groups,
               float(eps), _p(mean), _p(rstd), _stream())
    return y, mean, rstd


def conv_fwd_fused2_gn(x, wp3, bias3, wp1, bias1, cout, groups, eps):
    """conv_fwd_fused2 with the GroupNorm statistics of its 3x3x3 output: (c1, res, mean, rstd) or None"""
    n, d, h, w, cin = x.shape
    if not lib()._bts_conv3d_fwd_can_fuse(n, d, h, w, cin, cout):
        return None
    c1 = torch.empty((n, d, h, w, cout), dtype=torch.float32, device=x.device)
    res = torch.empty_like(c1)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    nb = lib().query('bts_conv3d_fwd_gn_workspace', K3S1, n, d, h, w, cin, cout, groups)
    ws = workspace(nb, x.device)
    lib().call('bts_conv3d_fwd_fused2_gn', _p(x), _p(wp3), _p(bias3), _p(c1), _p(wp1), _p(bias1), _p(res), _p(ws), nb, n, d, h, w,
               cin, ld_of(x), cout, cout, groups, float(eps), _p(mean), _p(rstd), _stream())
    return c1, res, mean, rstd


def conv_bwd_data(kind, dy, wp_bwd, dx, accumulate):
    """dx: [N,D,H,W,Cin] view of the forward input's gradient"""
    n, d, h, w, cin = dx.shape
    cout = dy.shape[4]
    nb = lib().query('bts_conv3d_bwd_data_workspace', kind, n, d, h, w, cin, cout)
    ws = workspace(nb, dy.device) if nb > 0 else None
    if True:
        lib().call('bts_conv3d_bwd_data', kind, _p(dy), _p(wp_bwd), _p(dx), _p(ws), nb, n, d, h, w, cin, ld_of(dx), cout,
                   ld_of(dy), FLAG_ACCUM if accumulate else 0, _stream())
    return dx


def conv_bwd_weight(kind, x, dy, dw, db, dup_start=0, dup_shift=0, accumulate=False):
    n, d, h, w, cin = x.shape
    cout = dy.shape[4]
    nb = lib().query('bts_conv3d_bwd_weight_workspace', kind, n, d, h, w, cin, cout)
    ws = workspace(nb, x.device)
    if True:
        lib().call('bts_conv3d_bwd_weight', kind, _p(x), _p(dy), _p(dw), _p(db), _p(ws), nb, n, d, h, w, cin, ld_of(x),
                   cout, ld_of(dy), dup_start, dup_shift, 1 if accumulate else 0, _stream())


def gn_stats(x, groups, mode, eps=1e-5):
    """x dense [N,D,H,W,C] -> (mean, rstd) each (N*G,)"""
    if not x.is_contiguous():
        raise RuntimeError('GroupNormalization statistics need a dense tensor')
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    nb = lib().query('bts_gn_workspace', n, v, c, groups, mode)
    ws = workspace(nb, x.device)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib().call('bts_gn_stats', _p(x), _p(mean), _p(rstd), _p(ws), nb, n, v, c, groups, mode, eps, _stream())
    return mean, rstd


def gn_apply(x, gamma, beta, mean, rstd, groups, mode, relu, out=None):
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    if out is None:
        out = torch.empty_like(x)
    lib().call('bts_gn_apply', _p(x), _p(out), _p(gamma), _p(beta), _p(mean), _p(rstd), n, v, c, ld_of(out), groups, mode,
               1 if relu else 0, _stream())
    return out


def gn_bwd(x, dy, gamma, beta, mean, rstd, dgamma, dbeta, groups, mode, relu, accumulate_params=False):
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    nb = lib().query('bts_gn_bwd_workspace', n, v, c, groups, mode)
    ws = workspace(nb, x.device)
    dx = torch.empty_like(x)
    lib().call('bts_gn_bwd', _p(x), _p(dy), _p(dx), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dgamma), _p(dbeta), _p(ws),
               nb, n, v, c, ld_of(dy), groups, mode, 1 if relu else 0, 1 if accumulate_params else 0, _stream())
    return dx


def colsum(x, scale=1.0, sum_over_n=False, out=None, accumulate=False):
    """x [N,...,C] (channel slice allowed) -> [N,C] (or [C]) column sums * scale"""
    n, c = x.shape[0], x.shape[-1]
    rows = x.numel() // (n * c)
    ld = ld_of(x) if x.dim() == 5 else x.stride(-2)
    nb = lib().query('bts_colsum_workspace', n, rows, c)
    ws = workspace(nb, x.device)
    if out is None:
        out = torch.empty((c,) if sum_over_n else (n, c), dtype=torch.float32, device=x.device)
    lib().call('bts_colsum', _p(x), _p(out), _p(ws), nb, n, rows, c, ld, float(scale), 1 if sum_over_n else 0,
               1 if accumulate else 0, _stream())
    return out


def se_mlp_fwd(gap, w1, w2):
    n, f = gap.shape
    r = w1.shape[1]
    h = torch.empty((n, r), dtype=torch.float32, device=gap.device)
    ch = torch.empty((n, f), dtype=torch.float32, device=gap.device)
    lib().call('bts_se_mlp_fwd', _p(gap), _p(w1), _p(w2), _p(h), _p(ch), n, f, r, _stream())
    return h, ch


def block_epilogue_fwd(res, c2, out, wsp, ch, gamma, beta, mean, rstd, groups, mode):
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    sp = torch.empty(n * v, dtype=torch.float32, device=res.device)
    lib().call('bts_block_epilogue_fwd', _p(res), _p(c2), _p(out), _p(sp), _p(wsp), _p(ch), _p(gamma), _p(beta), _p(mean),
               _p(rstd), n, v, f, ld_of(out), groups, mode, _stream())
    return sp


def se_bwd(dout, res, sp, gap, h, ch, w1, w2, wsp, dw1, dw2, dwsp, accumulate_params=False):
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    r = w1.shape[1]
    nb = lib().query('bts_se_bwd_workspace', n, v, f, r)
    ws = workspace(nb, res.device)
    dres = torch.empty_like(res)
    ds = torch.empty(n * v, dtype=torch.float32, device=res.device)
    dgap = torch.empty((n, f), dtype=torch.float32, device=res.device)
    lib().call('bts_se_bwd', _p(dout), _p(res), _p(sp), _p(gap), _p(h), _p(ch), _p(w1), _p(w2), _p(wsp), _p(dres), _p(ds),
               _p(dgap), _p(dw1), _p(dw2), _p(dwsp), _p(ws), nb, n, v, f, r, ld_of(dout), 1 if accumulate_params else 0,
               _stream())
    return dres


def block_bwd_takes(res, r, groups, dout, c2):
    """does the fused gate + GroupNorm-2 backward take this block?  Asked BEFORE the grad slots are claimed, so it checks everything
    block_bwd and bts_block_bwd check (tiling, row stride, contiguity, 16-byte alignment of the three streamed tensors): once this
    says yes, block_bwd does not decline."""
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    return (lib().probe('bts_block_bwd_workspace', n, v, f, r, groups) >= 0 and ld_of(dout) % 4 == 0 and res.is_contiguous()
            and c2.is_contiguous() and all(t.data_ptr() % 16 == 0 for t in (dout, res, c2)))


def block_bwd(dout, res, c2, sp, gap, h, ch, w1, w2, wsp, gamma, beta, mean, rstd, groups, dw1, dw2, dwsp, dgamma, dbeta,
              accumulate_gate_params=False, accumulate_norm_params=False):
    """gate backward + GroupNorm-2 backward (slab mode, ReLU) of a ResnetBlock in one pair of passes -> (dres, dc2), or None where
    the fused kernels do not take the shape (the caller runs se_bwd and gn_bwd)"""
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    r = w1.shape[1]
    nb = lib().probe('bts_block_bwd_workspace', n, v, f, r, groups)      # (-1: outside the fused kernels' tiling, not an error)
    if not block_bwd_takes(res, r, groups, dout, c2):
        return None
    ws = workspace(nb, res.device)
    dres = torch.empty_like(res)
    dc2 = torch.empty_like(c2)
    ds = torch.empty(n * v, dtype=torch.float32, device=res.device)
    dgap = torch.empty((n, f), dtype=torch.float32, device=res.device)
    lib().call('bts_block_bwd', _p(dout), ld_of(dout), _p(res), _p(c2), _p(sp), _p(gap), _p(h), _p(ch), _p(w1), _p(w2), _p(wsp), _p(gamma), _p(beta),
               _p(mean), _p(rstd), _p(dres), _p(dc2), _p(ds), _p(dgap), _p(dw1), _p(dw2), _p(dwsp), _p(dgamma), _p(dbeta), _p(ws), nb, n, v, f, r,
               groups, 1 if accumulate_gate_params else 0, 1 if accumulate_norm_params else 0, _stream())
    return dres, dc2


def dropout_mask(shape, rate, seed, device):
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    lib().call('bts_dropout_mask', _p(m), m.numel(), float(rate), int(seed) & (2 ** 64 - 1), _stream())
    return m


def dropout_apply(x, mask, rate):
    y = torch.empty_like(x)
    lib().call('bts_dropout_apply', _p(x), _p(mask), _p(y), x.numel(), float(rate), _stream())
    return y


def normal(shape, seed, device):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    lib().call('bts_normal', _p(out), out.numel(), int(seed) & (2 ** 64 - 1), _stream())
    return out


def vae_sample_fwd(proj, eps):
    n, l2 = proj.shape
    z = torch.empty((n, l2 // 2), dtype=torch.float32, device=proj.device)
    lib().call('bts_vae_sample_fwd', _p(proj), _p(eps), _p(z), n, l2 // 2, _stream())
    return z


def vae_sample_bwd(proj, eps, dz, dproj):
    n, l2 = proj.shape
    lib().call('bts_vae_sample_bwd', _p(proj), _p(eps), _p(dz), _p(dproj), n, l2 // 2, _stream())


def fill(t, v):
    lib().call('bts_fill', _p(t), t.numel(), float(v), _stream())
    return t


def axpy(y, x, a=1.0):
    lib().call('bts_axpy', _p(y), _p(x), y.numel(), float(a), _stream())


def add_strided(dst, src, accumulate):
    """dst[..., :C] (+)= src[..., :C] for NDHWC channel-slice views"""
    c = src.shape[-1]
    rows = src.numel() // c
    ldd = ld_of(dst) if dst.dim() == 5 else dst.stride(-2)
    lds = ld_of(src) if src.dim() == 5 else src.stride(-2)
    lib().call('bts_add_strided', _p(dst), _p(src), rows, c, ldd, lds, 1 if accumulate else 0, _stream())


def scalar_lincomb(a, b, ca=1.0, cb=1.0):
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    lib().call('bts_scalar_lincomb', _p(out), _p(a), _p(b), float(ca), float(cb), _stream())
    return out


def relu_bwd(y, dy):
    dx = torch.empty_like(y)
    lib().call('bts_relu_bwd', _p(y), _p(dy), _p(dx), y.numel(), _stream())
    return dx


def sigmoid_bwd(y, dy):
    c = y.shape[-1]
    rows = y.numel() // c
    dx = torch.empty(y.shape, dtype=torch.float32, device=y.device)
    lib().call('bts_sigmoid_bwd', _p(y), _p(dy), _p(dx), rows, c, ld_of(y), ld_of(dy), _stream())
    return dx


def dense_fwd(x, w, b, relu):
    n, fin = x.shape
    fout = w.shape[1]
    nb = lib().query('bts_dense_workspace', n, fin, fout)
    ws = workspace(nb, x.device)
    y = torch.empty((n, fout), dtype=torch.float32, device=x.device)
    lib().call('bts_dense_fwd', _p(x), _p(w), _p(b), _p(y), _p(ws), nb, n, fin, fout, 1 if relu else 0, _stream())
    return y


def dense_bwd(x, w, g, dx, dw, db, accumulate_dx=False, accumulate_params=False):
    n, fin = x.shape
    fout = w.shape[1]
    lib().call('bts_dense_bwd', _p(x), _p(w), _p(g), _p(dx), _p(dw), _p(db), n, fin, fout, 1 if accumulate_dx else 0,
               1 if accumulate_params else 0, _stream())


def loss_sums(y_pred, y, x, y_vae, proj):
    """-> sums (3C+4,) float64 device tensor (see include/bts_hip.h)"""
    n, c = y_pred.shape[0], y_pred.shape[4]
    v = y_pred.shape[1] * y_pred.shape[2] * y_pred.shape[3]
    sums = torch.empty(3 * c + 4, dtype=torch.float64, device=y_pred.device)
    nb = lib().query('bts_loss_workspace')
    ws = workspace(nb, y_pred.device)
    has_vae = x is not None
    cx = x.shape[4] if has_vae else 0
    lz = proj.shape[1] // 2 if proj is not None else 0
    lib().call('bts_loss_sums', _p(y_pred), _p(y), _p(x), _p(y_vae), _p(proj), _p(sums), _p(ws), nb, n, v, c, ld_of(y_pred),
               ld_of(y), cx, ld_of(x) if has_vae else 0, ld_of(y_vae) if has_vae else 0, lz, _stream())
    return sums


def loss_value(sums, c, has_vae=True):
    loss = torch.empty(1, dtype=torch.float32, device=sums.device)
    parts = torch.empty(3, dtype=torch.float32, device=sums.device)
    lib().call('bts_loss_value', _p(sums), _p(loss), _p(parts), c, 1 if has_vae else 0, _stream())
    return loss, parts


def loss_bwd(y_pred, y, x, y_vae, proj, sums, gscale, dypred, dyvae, dproj, through_sigmoid=False):
    n, c = y_pred.shape[0], y_pred.shape[4]
    v = y_pred.shape[1] * y_pred.shape[2] * y_pred.shape[3]
    has_vae = x is not None
    cx = x.shape[4] if has_vae else 0
    lz = proj.shape[1] // 2 if proj is not None else 0
    lib().call('bts_loss_bwd', _p(y_pred), _p(y), _p(x), _p(y_vae), _p(proj), _p(sums), _p(gscale), _p(dypred), _p(dyvae),
               _p(dproj), n, v, c, ld_of(y_pred), ld_of(y), cx, ld_of(x) if has_vae else 0, ld_of(y_vae) if has_vae else 0,
               lz, 1 if through_sigmoid else 0, _stream())


def dice_metric_sums(y_true, y_pred, channels_last_axes=True, want_labels=True):
    n, d, h, w, c = y_pred.shape
    cells = w if channels_last_axes else 1
    table = torch.empty(cells * c * 3, dtype=torch.float64, device=y_pred.device)
    labels = torch.empty((n, d, h, w), dtype=torch.uint8, device=y_pred.device) if want_labels else None
    lib().call('bts_dice_metric_sums', _p(y_true), _p(y_pred), _p(labels), _p(table), n, d * h * w, w, c, ld_of(y_true),
               ld_of(y_pred), 1 if channels_last_axes else 0, _stream())
    return table, labels


def dice_metric_value(table, w, c, channels_last_axes=True):
    out = torch.empty(2, dtype=torch.float32, device=table.device)
    lib().call('bts_dice_metric_value', _p(table), _p(out), w, c, 1 if channels_last_axes else 0, _stream())
    return out


def _ranges(ranges):
    nr = len(ranges)
    off = (ctypes.c_long * max(nr, 1))(*[r[0] for r in ranges])
    ln = (ctypes.c_long * max(nr, 1))(*[r[1] for r in ranges])
    cf = (ctypes.c_float * max(nr, 1))(*[r[2] for r in ranges])
    return off, ln, cf, nr


def l2_reg_fwd(params_flat, ranges):
    """ranges: [(offset, length, coefficient)] (<= 128) into the flat parameter buffer"""
    off, ln, cf, nr = _ranges(ranges)
    out = torch.empty(1, dtype=torch.float32, device=params_flat.device)
    nb = lib().query('bts_l2_workspace')
    ws = workspace(nb, params_flat.device)
    lib().call('bts_l2_reg_fwd', _p(params_flat), ctypes.cast(off, ctypes.c_void_p), ctypes.cast(ln, ctypes.c_void_p),
               ctypes.cast(cf, ctypes.c_void_p), nr, _p(out), _p(ws), nb, _stream())
    return out


def l2_reg_bwd(params_flat, grads_flat, ranges, gscale=None):
    off, ln, cf, nr = _ranges(ranges)
    lib().call('bts_l2_reg_bwd', _p(params_flat), _p(grads_flat), ctypes.cast(off, ctypes.c_void_p),
               ctypes.cast(ln, ctypes.c_void_p), ctypes.cast(cf, ctypes.c_void_p), nr, _p(gscale), _stream())


def adam_tf_step(p, g, m, v, lr_t, beta1, beta2, eps, gmul=1.0, skip=None):
    """skip: device int32[1]; the whole update is dropped on the device when it is non-zero (grad_nonfinite below)"""
    if skip is not None:
        lib().call('bts_adam_tf_step_guarded', _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr_t), float(beta1), float(beta2),
                   float(eps), float(gmul), _p(skip), _stream())
        return
    lib().call('bts_adam_tf_step', _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr_t), float(beta1), float(beta2),
               float(eps), float(gmul), _stream())


def grad_nonfinite(g, flag):
    """flag[0] (device int32) = 1 iff any element of the flat fp32 gradient is Inf / NaN"""
    lib().call('bts_grad_nonfinite', _p(g), g.numel(), _p(flag), _stream())


# ---- full-volume inference helpers ----
def flip_affine(src, flip_mask=0, mean=None, std=None, scale=1.0, out=None, accumulate=False):
    """out (+)= scale * t(flip(src)); flip_mask bits 4|2|1 = reverse D|H|W; t = (v - mean[c]) / std[c] when given"""
    _check(src, 'src')
    if not src.is_contiguous():
        raise ValueError('flip_affine: src must be a dense NDHWC tensor')
    n, d, h, w, c = src.shape
    if out is None:
        if accumulate:
            raise ValueError('flip_affine: accumulate needs an output tensor')
        out = torch.empty_like(src)
    lib().call('bts_flip_affine', _p(src), _p(out), _p(mean), _p(std), n, d, h, w, c, int(flip_mask), float(scale),
               1 if accumulate else 0, _stream())
    return out


def tta_finish(prob, bmask, threshold=0.5, want_probabilities=True, want_labels=True):
    """-> (prob * bmask, uint8 label map): argmax + 1, >= 3 -> 4, 0 where masked out / below threshold"""
    n, d, h, w, c = prob.shape
    y = torch.empty_like(prob) if want_probabilities else None
    labels = torch.empty((n, d, h, w), dtype=torch.uint8, device=prob.device) if want_labels else None
    lib().call('bts_tta_finish', _p(prob.contiguous()), _p(bmask.contiguous()), _p(y), _p(labels), n * d * h * w, c,
               float(threshold), _stream())
    return y, labels


# ---- training-time augmentation on the device ----
def channel_moments(x):
    """per-channel (mean, population variance) over all voxels of a dense (..., C) tensor, C <= 16 -> two (C,) tensors"""
    _check(x, 'x')
    c = x.shape[-1]
    nvox = x.numel() // c
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    nb = lib().query('bts_channel_moments_workspace', c)
    ws = workspace(nb, x.device)
    lib().call('bts_channel_moments', _p(x.contiguous()), _p(mean), _p(var), _p(ws), nb, nvox, c, c, _stream())
    return mean, var


def augment_crop(x, y, var, crop, offsets, flip_mask, shift, scale, out_ch):
    """x: (S0,S1,S2,C), y: (S0,S1,S2) or (S0,S1,S2,1) float labels, var: (C,) device tensor; shift/scale: C python floats
    -> (x_out (T0,T1,T2,C), y_out (T0,T1,T2,out_ch))"""
    s0, s1, s2, c = x.shape
    t0, t1, t2 = crop
    xo = torch.empty((t0, t1, t2, c), dtype=torch.float32, device=x.device)
    yo = torch.empty((t0, t1, t2, out_ch), dtype=torch.float32, device=x.device)
    sh = (ctypes.c_float * c)(*[float(v) for v in shift])
    sc = (ctypes.c_float * c)(*[float(v) for v in scale])
    lib().call('bts_augment_crop', _p(x.contiguous()), _p(y.contiguous()), _p(var), _p(xo), _p(yo), s0, s1, s2, c, t0, t1, t2,
               int(offsets[0]), int(offsets[1]), int(offsets[2]), int(flip_mask), ctypes.cast(sh, ctypes.c_void_p),
               ctypes.cast(sc, ctypes.c_void_p), int(out_ch), _stream())
    return xo, yo


# ---- non-default samplers ----
def maxpool2_fwd(x):
    n, d, h, w, c = x.shape
    y = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    idx = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=torch.uint8, device=x.device)
    lib().call('bts_maxpool2_fwd', _p(x), _p(y), _p(idx), n, d, h, w, c, ld_of(x), c, _stream())
    return y, idx


def maxpool2_bwd(dy, idx, dx, accumulate):
    """dx: [N,D,H,W,C] view of the input's gradient (may be a slab slice)"""
    n, d, h, w, c = dx.shape
    lib().call('bts_maxpool2_bwd', _p(dy), _p(idx), _p(dx), n, d, h, w, c, ld_of(dy), ld_of(dx), 1 if accumulate else 0, _stream())
    return dx


def upsample2_fwd(x, out=None):
    n, d, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, 2 * d, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    lib().call('bts_upsample2_fwd', _p(x), _p(out), n, d, h, w, c, ld_of(x), ld_of(out), _stream())
    return out


def upsample2_bwd(dy, dx=None, accumulate=False):
    n, d2, h2, w2, c = dy.shape
    if dx is None:
        dx = torch.empty((n, d2 // 2, h2 // 2, w2 // 2, c), dtype=torch.float32, device=dy.device)
    lib().call('bts_upsample2_bwd', _p(dy), _p(dx), n, d2 // 2, h2 // 2, w2 // 2, c, ld_of(dy), ld_of(dx),
               1 if accumulate else 0, _stream())
    return dx
