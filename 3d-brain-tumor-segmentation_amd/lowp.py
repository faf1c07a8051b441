"""Reduced-precision STORAGE forward of the model (BASELINE configs[4]: full-volume fp16 inference with the VAE branch off,
model.py:67-68 / test.py:128-134; the same kernels in bf16 are the forward half of configs[2]).

Activations and packed weights are 16-bit, every sum is fp32: convolutions on v_mfma_f32_32x32x16_{f16,bf16}, GroupNorm
statistics, SE gates and the sigmoid head in fp32 from the stored values (csrc/lowp.hip).  The master weights stay the
model's fp32 variables; 16-bit weight images are re-packed when the weights epoch changes.  The reference has no reduced
precision mode, so the parity reference of this path is the fp32 engine (tests/test_lowp_gpu.py).

The graph walked here is the fp32 layers' own (encoder.py:69-101 incl. the folded duplicated dense connections F4,
resnet.py:116-138, downsample.py:41-45, upsample.py:39-43, decoder.py:65-83) with the same virtual concatenation through
level slabs.  The first block reads the 2-channel input volume: a 16-channel matrix step is the 16-bit instructions' floor,
so the volume is stored zero-padded to 16 channels (the pad multiplies zeros on both sides).
"""
import ctypes
import os

import torch

from . import ops
from ._lib import ERRORS, lib
from .tape import weights_epoch

DTYPES = {'float16': (1, torch.float16), 'bfloat16': (2, torch.bfloat16), 'fp16': (1, torch.float16),
          'bf16': (2, torch.bfloat16), 'f16': (1, torch.float16)}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ld(t):
    """voxel stride in elements of a [N,D,H,W,C] tensor or channel-slice view"""
    if t.stride(-1) != 1:
        raise RuntimeError('channel stride must be 1')
    ld = t.stride(-2)
    n, d, h, w, _ = t.shape
    if (w > 1 and t.stride(3) != ld) or (h > 1 and t.stride(2) != ld * w) or (d > 1 and t.stride(1) != ld * w * h) or \
            (n > 1 and t.stride(0) != ld * w * h * d):
        raise RuntimeError('tensor is not a channel slice of a dense NDHWC buffer')
    return ld


def is_split(t):
    """a concat of 32-channel tensors kept as a contiguous (B, N, D, H, W, 32) buffer: its operands never sit side by side (decoder.py:75;
    SURVEY K13: virtual concat as a list of (ptr, C) segments)"""
    return t.dim() == 6


def xdims(t):
    """(n, d, h, w, channels, voxel stride, elements between the 32-channel operands or 0) of a block input: a 5-d view or a split concat"""
    if is_split(t):
        if not t.is_contiguous() or t.shape[-1] != 32:
            raise RuntimeError('a split concat is a contiguous (B, N, D, H, W, 32) tensor')
        b, n, d, h, w, _ = t.shape
        return n, d, h, w, 32 * b, 32, t.stride(0)
    n, d, h, w, c = t.shape
    return n, d, h, w, c, _ld(t), 0


# ---- C-ABI wrappers ----------------------------------------------------------------------------------------------------
def pack(kind, code, w, cin_ref, cout, cin_slab=None, dup_start=0, dup_shift=0, role=ops.ROLE_FWD):
    cin_slab = cin_ref if cin_slab is None else cin_slab
    nbytes = lib().query('bts_lp_packed_bytes', kind, role, cin_slab, cout)
    wp = torch.empty(nbytes // 2, dtype=torch.int16, device=w.device)
    lib().call('bts_lp_pack', kind, role, code, _p(w), _p(wp), cin_ref, cout, cin_slab, dup_start, dup_shift, _stream())
    return wp


class PackTable(object):
    """Device-resident descriptor table for bts_lp_pack_batch: entries = [(kind, role, w, wp, cin_ref, cout, cin_slab, dup_start,
    dup_shift)] with torch tensors w (reference layout, fp32) / wp (packed 16-bit image).  Rebuilt only when a pointer or the entry
    list changes (ops.PackTable is the fp32 engine's)."""

    def __init__(self):
        self.key = None
        self.dev = None
        self.n = 0
        self.blocks = 0

    def run(self, code, entries):
        if not entries:
            return
        import ctypes
        key = tuple((e[0], e[1], e[2].data_ptr(), e[3].data_ptr()) + tuple(e[4:]) for e in entries)
        if key != self.key:
            L = lib()
            nb = L._bts_lp_pack_desc_bytes()
            host = (ctypes.c_char * (nb * len(entries)))()
            first = 0
            for i, (kind, role, w, wp, cin_ref, cout, cin_slab, dup_start, dup_shift) in enumerate(entries):
                if not w.is_contiguous() or w.dtype != torch.float32:
                    raise ValueError('lp_pack_batch: kernels must be contiguous fp32 tensors')
                r = L._bts_lp_pack_desc(ctypes.cast(host, ctypes.c_void_p), i, first, kind, role, _p(w), _p(wp), cin_ref, cout, cin_slab,
                                        dup_start, dup_shift)
                if r <= 0:
                    raise RuntimeError('bts_lp_pack_desc failed: %s' % ops.ERRORS.get(r, r))
                first += r
            self.dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(entries[0][3].device)
            self.key, self.n, self.blocks = key, len(entries), first
        lib().call('bts_lp_pack_batch', code, _p(self.dev), self.n, self.blocks, _stream())


def conv_bwd_data(kind, code, dy, wp_bwd, dx, accumulate):
    """dx (+)= conv^T(dy); dx: (N,D,H,W,Cin) view of the forward input's gradient, dy: the forward output's"""
    n, d, h, w, cin = dx.shape
    cout = dy.shape[-1]
    nb = lib().query('bts_lp_conv3d_bwd_data_workspace', kind, n, d, h, w, cin, cout)
    ws = ops.workspace(nb, dx.device) if nb > 0 else None
    lib().call('bts_lp_conv3d_bwd_data', kind, code, _p(dy), _p(wp_bwd), _p(dx), _p(ws) if ws is not None else None, nb, n, d, h, w,
               cin, _ld(dx), cout, _ld(dy), 1 if accumulate else 0, _stream())
    return dx


def conv_bwd_data_sc(code, dy, wp_bwd, dy2, wp2_bwd, dx, accumulate):
    """dx (+)= conv3x3x3^T(dy) + conv1x1x1^T(dy2): conv1's and the shortcut's data gradients of a ResnetBlock (resnet.py:80-87,96-103 read
    the same `inputs`) in one launch where the fused kernels take the shape, else as the two launches; -> True if fused.
    dx: an (N,D,H,W,Cin) view, or a contiguous (B,N,D,H,W,32) tensor = the Cin = 32 B columns as B dense tensors (the gradient of a concat
    of 32-channel tensors, decoder.py:75; only where conv_bwd_data_sc_split_ok says so)"""
    split = 0
    if dx.dim() == 6:
        if not dx.is_contiguous() or dx.shape[-1] != 32:
            raise RuntimeError('conv_bwd_data_sc: a split dx is a contiguous (B,N,D,H,W,32) tensor')
        nblk, n, d, h, w, _ = dx.shape
        cin, lddx, split = 32 * nblk, 32, dx.stride(0)
    else:
        n, d, h, w, cin = dx.shape
        lddx = _ld(dx)
    cout = dy.shape[-1]
    if tuple(dy2.shape) != tuple(dy.shape):
        raise RuntimeError('conv_bwd_data_sc: dy %s and dy2 %s differ' % (tuple(dy.shape), tuple(dy2.shape)))
    nb = lib().query('bts_lp_conv3d_bwd_data_sc_workspace', n, d, h, w, cin, cout)
    ws = ops.workspace(nb, dx.device) if nb > 0 else None
    fused = ctypes.c_int(0)
    lib().call('bts_lp_conv3d_bwd_data_sc', code, _p(dy), _p(wp_bwd), _p(dy2), _p(wp2_bwd), _p(dx), split, _p(ws) if ws is not None else None,
               nb, n, d, h, w, cin, lddx, cout, _ld(dy), _ld(dy2), 1 if accumulate else 0, ctypes.byref(fused), _stream())
    return bool(fused.value)


def conv_bwd_data_sc_split_ok(n, d, h, w, cin, cout):
    """does the fused launch write its (cin = 32 B)-column result as B dense 32-channel tensors at this shape?"""
    return cin % 32 == 0 and cin >= 64 and bool(lib().probe('bts_lp_conv3d_bwd_data_sc_split_ok', n, d, h, w, cin, cout))


def conv(kind, code, tdt, x, wp, bias, cout, out=None):
    n, d, h, w, cin = x.shape
    do, ho, wo = (d, h, w) if kind in (ops.K1, ops.K3S1) else ((d + 1) // 2, (h + 1) // 2, (w + 1) // 2) if kind == ops.K3S2 \
        else (2 * d, 2 * h, 2 * w)
    if out is None:
        out = torch.empty((n, do, ho, wo, cout), dtype=tdt, device=x.device)
    elif tuple(out.shape) != (n, do, ho, wo, cout):
        raise RuntimeError('conv output view has shape %s, expected %s' % (tuple(out.shape), (n, do, ho, wo, cout)))
    nb = lib().query('bts_lp_conv3d_workspace', kind, n, d, h, w, cin, cout)
    ws = ops.workspace(nb, x.device) if nb > 0 else None
    lib().call('bts_lp_conv3d_fwd', kind, code, _p(x), _p(wp), _p(bias) if bias is not None else None, _p(out),
               _p(ws) if ws is not None else None, nb, n, d, h, w, cin, _ld(x), cout, _ld(out), _stream())
    return out


def wgrad_supported(kind, cin, cout):
    """shapes the 16-bit weight-gradient kernels take (channel counts in whole 16-byte chunks; the stride-1 3x3x3 / 1x1x1 kernel also
    wants a power-of-two number of them per cout block)"""
    if kind in (ops.K3S2, ops.K3S2T):
        return cin % 8 == 0 and cout % 8 == 0
    return kind in (ops.K3S1, ops.K1) and cin % 8 == 0 and cout % 8 == 0 and cout <= 256 and ((cout // 8) & (cout // 8 - 1)) == 0


def conv_bwd_weight(kind, code, x, dy, dw, db, dup_start=0, dup_shift=0, accumulate=True):
    """16-bit weight gradient of a conv (stride-1 3x3x3, 1x1x1; stride-2 3x3x3 and its transposed sibling through the transposing-read
    kernel of csrc/lowp_wgs.hip: x on the forward-input grid, dy on the half / doubled grid) -> True, or False when the shape is outside
    the kernels' reach (the caller then runs the fp32 kernel on widened copies)"""
    n, d, h, w, cin = x.shape
    cout = dy.shape[-1]
    if not wgrad_supported(kind, cin, cout) or (db is not None and not dy.is_contiguous()):
        return False
    if kind == ops.K3S2 and ((d | h | w) & 1):
        return False
    nb = lib().query('bts_lp_conv3d_bwd_weight_workspace', kind, n, d, h, w, cin, cout)
    ws = ops.workspace(nb, x.device)
    lib().call('bts_lp_conv3d_bwd_weight', kind, code, _p(x), _p(dy), _p(dw), _p(db) if db is not None else None, _p(ws), nb, n, d, h, w, cin,
               _ld(x), cout, _ld(dy), dup_start, dup_shift, 1 if accumulate else 0, _stream())
    return True


def conv_bwd_weight_pair(code, x, dy3, dy1, dw3, dw1, db3, dup_start=0, dup_shift=0, accumulate=True):
    """The weight gradients of conv1 (3x3x3, from dy3) and of the shortcut (1x1x1, from dy1) of a ResnetBlock -- both read the block input x
    (resnet.py:118,134) -- from ONE pass over x (bts_lp_conv3d_bwd_weight_pair) -> True, or False (nothing launched) where the streaming
    kernel does not take the shape: the caller runs conv_bwd_weight twice"""
    n, d, h, w, cin, ldx, xsplit = xdims(x)
    cout = dy3.shape[-1]
    if tuple(dy1.shape) != tuple(dy3.shape) or not wgrad_supported(ops.K3S1, cin, cout) or (db3 is not None and not dy3.is_contiguous()):
        return False
    nb = lib().probe('bts_lp_conv3d_bwd_weight_pair_workspace', n, d, h, w, cin, cout)
    if nb < 0:
        return False
    ws = ops.workspace(nb, x.device)
    r = lib().probe('bts_lp_conv3d_bwd_weight_pair', code, _p(x), xsplit, _p(dy3), _p(dy1), _p(dw3), _p(dw1), _p(db3) if db3 is not None else None,
                    _p(ws), nb, n, d, h, w, cin, ldx, cout, _ld(dy3), _ld(dy1), dup_start, dup_shift, 1 if accumulate else 0, _stream())
    if r == 1:
        return False
    if r != 0:
        raise RuntimeError('bts_lp_conv3d_bwd_weight_pair failed: %s' % ERRORS.get(r, 'hipError %d' % r))
    return True


def cast(code, tdt, src, out=None):
    c = src.shape[-1]
    rows = src.numel() // c
    if out is None:
        out = torch.empty(src.shape, dtype=tdt, device=src.device)
    lib().call('bts_lp_cast', code, _p(src), src.stride(-2), _p(out), out.stride(-2), rows, c, _stream())
    return out


def cast_pad16(code, tdt, src):
    """fp32 (..., C <= 4) -> storage type (..., 16) with a zero tail, one pass (the 2- / 1-channel tensors as whole matrix steps)"""
    c = src.shape[-1]
    rows = src.numel() // c
    out = torch.empty(tuple(src.shape[:-1]) + (16,), dtype=tdt, device=src.device)
    lib().call('bts_lp_cast_pad16', code, _p(src), src.stride(-2), _p(out), rows, c, _stream())
    return out


def dropout_cast_pad16(code, tdt, src, rate, seed):
    """dropout (the draw of ops.dropout_mask for the same seed) + cast_pad16 of a dense fp32 (..., C <= 4) volume in one pass"""
    c = src.shape[-1]
    if not src.is_contiguous():
        raise RuntimeError('dropout_cast_pad16: dense volume expected')
    out = torch.empty(tuple(src.shape[:-1]) + (16,), dtype=tdt, device=src.device)
    lib().call('bts_lp_dropout_cast_pad16', code, _p(src), _p(out), src.numel() // c, c, float(rate), int(seed) & (2 ** 64 - 1), _stream())
    return out


def uncast(code, src):
    c = src.shape[-1]
    rows = src.numel() // c
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    lib().call('bts_lp_uncast', code, _p(src), src.stride(-2), _p(out), out.stride(-2), rows, c, _stream())
    return out


def conv_gn(code, tdt, x, wp, bias, cout, norm, acc_into=None):
    """(y, mean, rstd): y = conv3x3x3(x) + bias (dense, storage type) and GroupNorm `norm`'s statistics of y -- from the conv's
    epilogue where the library can (slab mode: the layout the reference's channels_last GroupNormalization reduces over).
    acc_into: a dense tensor that already holds part of the contraction (other input channels, with the bias): the conv is ADDED to it
    and the statistics are those of the sums (bts_lp_conv3d_fwd_gn_acc)"""
    n, d, h, w, cin = x.shape
    if norm._mode != ops.GN_SLAB:
        if acc_into is not None:
            raise NotImplementedError('conv_gn(acc_into=...) is the slab-mode (channels_last) form only')
        y = conv(ops.K3S1, code, tdt, x, wp, bias, cout)
        return (y,) + tuple(gn_stats(code, y, norm.groups, norm._mode, norm.epsilon))
    y = acc_into if acc_into is not None else torch.empty((n, d, h, w, cout), dtype=tdt, device=x.device)
    if acc_into is not None and (tuple(y.shape) != (n, d, h, w, cout) or not y.is_contiguous()):
        raise RuntimeError('conv_gn: acc_into must be a dense %s tensor' % ((n, d, h, w, cout),))
    mean = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    nb = lib().query('bts_lp_conv3d_fwd_gn_workspace', n, d, h, w, cin, cout, norm.groups)
    ws = ops.workspace(nb, x.device)
    if acc_into is not None:
        lib().call('bts_lp_conv3d_fwd_gn_acc', code, _p(x), _p(wp), _p(bias) if bias is not None else None, _p(y), _p(mean), _p(rstd), _p(ws), nb,
                   n, d, h, w, cin, _ld(x), cout, norm.groups, float(norm.epsilon), 1, _stream())
    else:
        lib().call('bts_lp_conv3d_fwd_gn', code, _p(x), _p(wp), _p(bias), _p(y), _p(mean), _p(rstd), _p(ws), nb, n, d, h, w, cin, _ld(x), cout,
                   norm.groups, float(norm.epsilon), _stream())
    return y, mean, rstd


def conv_gn_shortcut(code, tdt, x, wp, bias, cout, norm, wp_pt, bias_pt):
    """(y, mean, rstd, res, gap) -- conv1 with `norm`'s statistics AND the block's shortcut conv with the gate's squeeze from ONE pass over
    the block input (resnet.py:118,134 read the same `inputs`), or None where the streaming kernel does not take the shape (the caller
    runs conv1_gap + conv_gn)"""
    if norm._mode != ops.GN_SLAB:
        return None
    n, d, h, w, cin, ldx, xsplit = xdims(x)
    if xsplit and cin != 64:
        return None
    nb = lib().probe('bts_lp_conv3d_fwd_gn_shortcut_workspace', n, d, h, w, cin, ldx, cout, norm.groups)
    if nb < 0:
        return None
    y = torch.empty((n, d, h, w, cout), dtype=tdt, device=x.device)
    res = torch.empty((n, d, h, w, cout), dtype=tdt, device=x.device)
    mean = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    gap = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    ws = ops.workspace(nb, x.device)
    r = lib().probe('bts_lp_conv3d_fwd_gn_shortcut', code, _p(x), xsplit, _p(wp), _p(bias), _p(y), _p(mean), _p(rstd), _p(wp_pt), _p(bias_pt), _p(res),
                    _p(gap), _p(ws), nb, n, d, h, w, cin, ldx, cout, norm.groups, float(norm.epsilon), _stream())
    if r == 1:
        return None
    if r != 0:
        raise RuntimeError('bts_lp_conv3d_fwd_gn_shortcut failed: %s' % ERRORS.get(r, 'hipError %d' % r))
    return y, mean, rstd, res, gap


def conv_gn_normed_input(code, tdt, x, norm_in, mean_in, rstd_in, relu_in, wp, bias, cout, norm):
    """(y, mean, rstd) with y = conv3x3x3([relu](norm_in(x))) + bias and `norm`'s statistics of y: the GroupNorm of the conv's INPUT is
    applied to each input plane inside the conv kernel (bts_lp_conv3d_gnin_fwd_gn) -- for forwards whose normalised tensor nobody else
    reads (inference).  None where the library does not take the shape in this form: the caller runs gn_apply + conv_gn."""
    if norm._mode != ops.GN_SLAB or norm_in._mode != ops.GN_SLAB or not x.is_contiguous() or not relu_in:
        return None
    n, d, h, w, cin = x.shape
    nb = lib().probe('bts_lp_conv3d_gnin_fwd_gn_workspace', n, d, h, w, cin, cout, norm_in.groups, norm.groups)
    if nb < 0:
        return None
    y = torch.empty((n, d, h, w, cout), dtype=tdt, device=x.device)
    mean = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    ws = ops.workspace(nb, x.device)
    lib().call('bts_lp_conv3d_gnin_fwd_gn', code, _p(x), _p(norm_in.gamma.t), _p(norm_in.beta.t), _p(mean_in), _p(rstd_in), norm_in.groups,
               1 if relu_in else 0, _p(wp), _p(bias), _p(y), _p(mean), _p(rstd), _p(ws), nb, n, d, h, w, cin, cout, norm.groups,
               float(norm.epsilon), _stream())
    return y, mean, rstd


def gnin_train_ok(x, cout, norm_in, norm):
    """can conv2 of a block run WITHOUT the normalised tensor in training -- forward through conv_gn_normed_input, weight gradient through
    conv_bwd_weight_normed_input -- for raw input x (dense (N,D,H,W,Cin)) of GroupNorm `norm_in`?"""
    if norm._mode != ops.GN_SLAB or norm_in._mode != ops.GN_SLAB or not x.is_contiguous():
        return False
    n, d, h, w, cin = x.shape
    return lib().probe('bts_lp_conv3d_gnin_train_ok', n, d, h, w, cin, cout, norm_in.groups, norm.groups) == 1


def conv_bwd_weight_normed_input(code, x, norm_in, mean_in, rstd_in, dy, dw, db=None, accumulate=True):
    """dw (+)= the 3x3x3 weight gradient of a conv whose input was relu(norm_in(x)), from the RAW x (bts_lp_conv3d_gnin_bwd_weight); db
    (+)= the column sums of dy.  Raises where gnin_train_ok said no."""
    n, d, h, w, cin = x.shape
    cout = dy.shape[-1]
    if dw.shape[-2] != cin or dw.shape[-1] != cout or not x.is_contiguous():
        raise RuntimeError('conv_bwd_weight_normed_input: dw %s does not match x %s / dy %s' % (tuple(dw.shape), tuple(x.shape), tuple(dy.shape)))
    nb = lib().query('bts_lp_conv3d_bwd_weight_workspace', ops.K3S1, n, d, h, w, cin, cout)
    ws = ops.workspace(nb, x.device)
    lib().call('bts_lp_conv3d_gnin_bwd_weight', code, _p(x), _p(norm_in.gamma.t), _p(norm_in.beta.t), _p(mean_in), _p(rstd_in), norm_in.groups,
               _p(dy), _p(dw), _p(db) if db is not None else None, _p(ws), nb, n, d, h, w, cin, cout, _ld(dy), 1 if accumulate else 0, _stream())


def gn_stats(code, x, groups, mode, eps):
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    if not x.is_contiguous():
        raise RuntimeError('GroupNormalization statistics need a dense tensor')
    nb = lib().query('bts_lp_gn_workspace', n, v, c, groups)
    ws = ops.workspace(nb, x.device)
    mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib().call('bts_lp_gn_stats', code, _p(x), _p(mean), _p(rstd), _p(ws), nb, n, v, c, groups, mode, float(eps), _stream())
    return mean, rstd


def gn_apply(code, x, gamma, beta, mean, rstd, groups, mode, relu, out=None):
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    if out is None:
        out = torch.empty_like(x)
    lib().call('bts_lp_gn_apply', code, _p(x), _p(out), _p(gamma), _p(beta), _p(mean), _p(rstd), n, v, c, _ld(out), groups, mode,
               1 if relu else 0, _stream())
    return out


def gn_bwd(code, tdt, x, dy, gamma, beta, mean, rstd, dgamma, dbeta, groups, relu, want_f32=True, dbias=None):
    """slab-mode GroupNorm backward on 16-bit tensors -> (dx in the storage type, dx in fp32 or None); None if the shape is outside the
    kernel's tiling (the caller then runs the fp32 kernel on widened copies).  dbias: fp32 view that receives (+=) the column sums of
    dx -- the bias gradient of the conv that produced x -- from the same pass"""
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    L = v * c // groups
    if (v * c) % groups or L % 2048 or c // groups > 32 or 256 % (c // groups) or c > 256 or c & (c - 1):
        return None
    nb = lib().query('bts_lp_gn_bwd_workspace', n, v, c, groups)
    ws = ops.workspace(nb, x.device)
    dx = torch.empty_like(x)
    dx32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    lib().call('bts_lp_gn_bwd', code, _p(x), _p(dy), _p(dx), _p(dx32) if dx32 is not None else None, _p(gamma), _p(beta), _p(mean), _p(rstd),
               _p(dgamma), _p(dbeta), _p(ws), nb, n, v, c, _ld(dy), groups, 1 if relu else 0, 1, _p(dbias) if dbias is not None else None,
               _stream())
    return dx, dx32


def convT_gn(code, tdt, x, wp, bias, cout, norm):
    """(y, mean, rstd): y = Conv3DTranspose(k3, s2, 'same')(x) + bias (dense fine tensor, storage type) and GroupNorm `norm`'s statistics
    of y -- from the merged transposed-conv kernel's epilogue where the library can (slab mode, whole fine planes per group)"""
    n, d, h, w, cin = x.shape
    if norm._mode != ops.GN_SLAB:
        y = conv(ops.K3S2T, code, tdt, x, wp, bias, cout)
        return (y,) + tuple(gn_stats(code, y, norm.groups, norm._mode, norm.epsilon))
    y = torch.empty((n, 2 * d, 2 * h, 2 * w, cout), dtype=tdt, device=x.device)
    mean = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    nb = lib().query('bts_lp_convT3d_fwd_gn_workspace', n, d, h, w, cin, cout, norm.groups)
    ws = ops.workspace(nb, x.device)
    lib().call('bts_lp_convT3d_fwd_gn', code, _p(x), _p(wp), _p(bias), _p(y), _p(mean), _p(rstd), _p(ws), nb, n, d, h, w, cin, _ld(x), cout,
               norm.groups, float(norm.epsilon), _stream())
    return y, mean, rstd


def conv_bwd_data_gn_bwd(code, tdt, dy, wp_bwd, c, gamma, beta, mean, rstd, dgamma, dbeta, groups, relu, want_f32=True, dbias=None):
    """conv2's data gradient followed by GroupNorm-1's backward (resnet.py:80-93 in reverse) as one library call -> (da, dc, dc32 | None,
    fused): da = conv3x3x3^T(dy) stored in the storage type, dc the gradient of the GroupNorm input c; where the z-marching conv kernel
    takes the layer, GroupNorm's class sums come out of its epilogue (no reduce pass over da and c).  None when the GroupNorm kernels'
    tiling does not fit (gn_bwd's conditions): the caller runs the two steps itself"""
    n, d, h, w, cg = c.shape
    v = d * h * w
    L = v * cg // groups
    if (v * cg) % groups or L % 2048 or cg // groups > 32 or 256 % (cg // groups) or cg > 256 or cg & (cg - 1) or not c.is_contiguous():
        return None
    cdy = dy.shape[-1]
    nb = lib().query('bts_lp_conv3d_bwd_data_gn_bwd_workspace', n, d, h, w, cg, cdy, groups)
    ws = ops.workspace(nb, c.device)
    da = torch.empty_like(c)
    dc = torch.empty_like(c)
    dc32 = torch.empty(c.shape, dtype=torch.float32, device=c.device) if want_f32 else None
    fused = ctypes.c_int(0)
    lib().call('bts_lp_conv3d_bwd_data_gn_bwd', code, _p(dy), _p(wp_bwd), _p(da), _p(c), _p(dc), _p(dc32) if dc32 is not None else None,
               _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dgamma), _p(dbeta), _p(ws), nb, n, d, h, w, cg, cdy, _ld(dy), groups,
               1 if relu else 0, 1, _p(dbias) if dbias is not None else None, ctypes.byref(fused), _stream())
    return da, dc, dc32, bool(fused.value)


def conv1_gap(code, x, wp, bias, cout, tdt, acc_into=None):
    """(res, gap): the block's 1x1x1 shortcut conv and the mean over voxels of its output (the gate's squeeze) in one pass.
    acc_into: a dense tensor holding part of the contraction already (see conv_gn): the conv is added to it, gap = mean of the sums"""
    n, d, h, w, cin = x.shape
    res = acc_into if acc_into is not None else torch.empty((n, d, h, w, cout), dtype=tdt, device=x.device)
    gap = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    nb = lib().query('bts_lp_conv1_gap_workspace', n, d * h * w, cout)
    ws = ops.workspace(nb, x.device)
    if acc_into is not None:
        lib().call('bts_lp_conv1_gap_acc', code, _p(x), _p(wp), _p(bias) if bias is not None else None, _p(res), _p(gap), _p(ws), nb, n, d, h, w,
                   cin, _ld(x), cout, cout, 1, _stream())
    else:
        lib().call('bts_lp_conv1_gap', code, _p(x), _p(wp), _p(bias), _p(res), _p(gap), _p(ws), nb, n, d, h, w, cin, _ld(x), cout, cout,
                   _stream())
    return res, gap


def colsum(code, x, scale):
    n, c = x.shape[0], x.shape[4]
    v = x.shape[1] * x.shape[2] * x.shape[3]
    nb = lib().query('bts_lp_colsum_workspace', n, v, c)
    ws = ops.workspace(nb, x.device)
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    lib().call('bts_lp_colsum', code, _p(x), _p(out), _p(ws), nb, n, v, c, float(scale), _stream())
    return out


def block_epilogue(code, res, c2, out, wsp, ch, gamma, beta, mean, rstd, groups, mode, sp_out=None):
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    lib().call('bts_lp_block_epilogue', code, _p(res), _p(c2), _p(out), _p(sp_out) if sp_out is not None else None, _p(wsp), _p(ch), _p(gamma), _p(beta), _p(mean), _p(rstd), n, v,
               f, _ld(out), groups, mode, _stream())
    return out


def first_block(code, tdt, x, w3, b3, w1, b1, f, norm):
    """the first ResnetBlock's conv1 and shortcut from the RAW fp32 2-channel volume x (N,D,H,W,2) in one pass (csrc/lowp_c2.hip) ->
    (c1, mean1, rstd1, res, gap), or None where the kernel does not take the shape (the caller casts the volume into a zero-padded
    16-channel tensor and runs the generic kernels)"""
    if norm._mode != ops.GN_SLAB or x.dtype != torch.float32 or x.shape[-1] != 2 or not x.is_contiguous():
        return None
    n, d, h, w, _ = x.shape
    nb = lib().probe('bts_lp_first_block_workspace', n, d, h, w, f, norm.groups)
    if nb < 0:
        return None
    ws = ops.workspace(nb, x.device)
    c1 = torch.empty((n, d, h, w, f), dtype=tdt, device=x.device)
    res = torch.empty((n, d, h, w, f), dtype=tdt, device=x.device)
    mean = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n * norm.groups, dtype=torch.float32, device=x.device)
    gap = torch.empty((n, f), dtype=torch.float32, device=x.device)
    lib().call('bts_lp_first_block_fwd', code, _p(x), _p(w3), _p(b3), _p(w1), _p(b1), _p(c1), _p(res), _p(mean), _p(rstd), _p(gap), _p(ws), nb,
               n, d, h, w, f, norm.groups, float(norm.epsilon), _stream())
    return c1, mean, rstd, res, gap


def block_epilogue_head(code, res, c2, wsp, ch, gamma, beta, mean, rstd, groups, mode, head_w, head_b, sigmoid=True):
    """the last block's epilogue and the 1x1x1 output head in one pass (the block output is never written) -> y (N,D,H,W,K) fp32, or None
    where the fused kernel does not take the shape (the caller runs block_epilogue + head)"""
    n, d, h, w, f = res.shape
    k = head_w.shape[-1]
    y = torch.empty((n, d, h, w, k), dtype=torch.float32, device=res.device)
    r = lib().probe('bts_lp_block_epilogue_head', code, _p(res), _p(c2), _p(y), _p(wsp), _p(ch), _p(gamma), _p(beta), _p(mean), _p(rstd),
                    _p(head_w), _p(head_b) if head_b is not None else None, n, d * h * w, f, groups, mode, k, 1 if sigmoid else 0, _stream())
    if r == -3:          # BTS_ERR_UNSUPPORTED: outside the fused kernel's shapes
        return None
    if r != 0:
        raise RuntimeError('bts_lp_block_epilogue_head failed: %s' % ERRORS.get(r, 'hipError %d' % r))
    return y


def se_bwd(code, tdt, dout, res, sp, gap, h, ch, w1, w2, wsp, dw1, dw2, dwsp, dbias=None):
    """gate backward on 16-bit tensors -> dres (storage type, dense); parameter gradients accumulate; dbias: fp32 view that receives
    (+=) the column sums of dres, the shortcut conv's bias gradient"""
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    r = w1.shape[1]
    nb = lib().query('bts_lp_se_bwd_workspace', n, v, f, r)
    ws = ops.workspace(nb, res.device)
    dres = torch.empty_like(res)
    ds = torch.empty(n * v, dtype=torch.float32, device=res.device)
    dgap = torch.empty((n, f), dtype=torch.float32, device=res.device)
    lib().call('bts_lp_se_bwd', code, _p(dout), _p(res), _p(sp), _p(gap), _p(h), _p(ch), _p(w1), _p(w2), _p(wsp), _p(dres), _p(ds), _p(dgap),
               _p(dw1), _p(dw2), _p(dwsp), _p(ws), nb, n, v, f, r, _ld(dout), 1, _p(dbias) if dbias is not None else None, _stream())
    return dres


def block_bwd(code, tdt, dout, res, c2, sp, gap, h, ch, w1, w2, wsp, gamma, beta, mean, rstd, groups, dw1, dw2, dwsp, dgamma, dbeta,
              dbias_pt=None, dbias_c2=None):
    """gate backward + GroupNorm-2 (+ReLU) backward of a ResnetBlock in one pair of passes -> (dres, dc2) in the storage type, or None
    outside the fused kernels' tiling (the caller then runs se_bwd and gn_bwd).  Parameter / bias gradients accumulate."""
    n, f = res.shape[0], res.shape[4]
    v = res.shape[1] * res.shape[2] * res.shape[3]
    L = v * f // groups
    cg = f // groups
    if (v * f) % groups or L % 2048 or cg > 32 or 256 % cg or f > 256 or f & (f - 1) or not (res.is_contiguous() and c2.is_contiguous()):
        return None
    r = w1.shape[1]
    nb = lib().query('bts_lp_block_bwd_workspace', n, v, f, r, groups)
    ws = ops.workspace(nb, res.device)
    dres = torch.empty_like(res)
    dc2 = torch.empty_like(c2)
    ds = torch.empty(n * v, dtype=torch.float32, device=res.device)
    dgap = torch.empty((n, f), dtype=torch.float32, device=res.device)
    lib().call('bts_lp_block_bwd', code, _p(dout), _ld(dout), _p(res), _p(c2), _p(sp), _p(gap), _p(h), _p(ch), _p(w1), _p(w2), _p(wsp), _p(gamma),
               _p(beta), _p(mean), _p(rstd), _p(dres), _p(dc2), _p(ds), _p(dgap), _p(dw1), _p(dw2), _p(dwsp), _p(dgamma), _p(dbeta),
               _p(dbias_pt) if dbias_pt is not None else None, _p(dbias_c2) if dbias_c2 is not None else None, _p(ws), nb, n, v, f, r, groups,
               _stream())
    return dres, dc2


def head_bwd(code, tdt, x, dpre, w, dw, db, accumulate=True):
    """output head backward in one pass over the 16-bit activations -> dx (storage type); dw (C,K) / db (K) fp32 accumulate.
    None when the head is outside the kernel's shapes (C in {16,32,64}, K <= 4): the caller then runs the fp32 kernels"""
    n, d, h, wd, c = x.shape
    k = dpre.shape[-1]
    if c not in (16, 32, 64) or k > 4 or not dpre.is_contiguous():
        return None
    dx = torch.empty((n, d, h, wd, c), dtype=tdt, device=x.device)
    nb = lib().query('bts_lp_head_bwd_workspace', c, k)
    ws = ops.workspace(nb, x.device)
    lib().call('bts_lp_head_bwd', code, _p(x), _p(dpre), _p(w), _p(dx), _p(dw), _p(db) if db is not None else None, _p(ws), nb,
               n * d * h * wd, c, _ld(x), c, k, 1 if accumulate else 0, _stream())
    return dx


def maxpool2(code, x, want_idx=True):
    """MaxPooling3D(2) on a 16-bit view (downsample.py:51-70) -> (y dense, idx uint8 or None)"""
    n, d, h, w, c = x.shape
    if (d | h | w) & 1:
        raise ValueError('MaxDownsample needs even spatial sizes, got %s' % (tuple(x.shape),))
    y = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if want_idx else None
    lib().call('bts_lp_maxpool2_fwd', code, _p(x), _p(y), _p(idx) if want_idx else None, n, d, h, w, c, _ld(x), _ld(y), _stream())
    return y, idx


def maxpool2_bwd(code, dy, idx, dx, accumulate):
    n, d, h, w, c = dx.shape
    lib().call('bts_lp_maxpool2_bwd', code, _p(dy), _p(idx), _p(dx), n, d, h, w, c, _ld(dy), _ld(dx), 1 if accumulate else 0, _stream())
    return dx


def upsample2(code, x, out=None):
    """UpSampling3D(2) (nearest-neighbour repeat, upsample.py:69) of a 16-bit view into `out` (a view is fine)"""
    n, d, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, 2 * d, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
    lib().call('bts_lp_upsample2_fwd', code, _p(x), _p(out), n, d, h, w, c, _ld(x), _ld(out), _stream())
    return out


def upsample2_bwd(code, dy, dx=None, accumulate=False):
    """gradient of the repeat: sums of the 8 fine voxels of each coarse one (fp32 sums) -> dx (dense unless given)"""
    n, d2, h2, w2, c = dy.shape
    if dx is None:
        dx = torch.empty((n, d2 // 2, h2 // 2, w2 // 2, c), dtype=dy.dtype, device=dy.device)
    lib().call('bts_lp_upsample2_bwd', code, _p(dy), _p(dx), n, d2 // 2, h2 // 2, w2 // 2, c, _ld(dy), _ld(dx), 1 if accumulate else 0,
               _stream())
    return dx


def head(code, x, w, bias, sigmoid=True):
    n, d, h, wd, c = x.shape
    k = w.shape[-1]
    y = torch.empty((n, d, h, wd, k), dtype=torch.float32, device=x.device)
    lib().call('bts_lp_head', code, _p(x), _p(w), _p(bias) if bias is not None else None, _p(y), n * d * h * wd, c, _ld(x), k,
               1 if sigmoid else 0, _stream())
    return y


# ---- the forward graph -------------------------------------------------------------------------------------------------
def gate_branch(code, tdt, x, wp_pt, bias_pt, f, se_w1, se_w2, side=True, acc_into=None, after=None):
    """A ResnetBlock's shortcut / squeeze-excitation branch (resnet.py:118-126: 1x1x1 conv with the gate's squeeze from its epilogue,
    then the two small dense layers) -> (res, gap, (h, ch), stream it ran on or None).  HBM-bound and independent of the conv branch
    until the block epilogue, so -- like the fp32 engine -- it goes to the 'gate' side stream next to the matrix-pipe-bound convs
    (BTS_GATE_STREAM=0 or side=False: main stream); the caller makes its stream wait for the returned one before the epilogue.
    Measured (in one job, alternating): fp16 full-volume inference 145.0 -> 148.3 volumes/s; the batch-8 training forward, whose
    convs already fill the chip, is unchanged (and pays for a second allocator pool), so the trainer keeps it on the main stream"""
    gate = ops.side_stream('gate') if side else None
    if gate is None:
        if after is not None:
            torch.cuda.current_stream().wait_event(after)
        res, gap = conv1_gap(code, x, wp_pt, bias_pt, f, tdt, acc_into=acc_into)
        hbuf, ch = ops.se_mlp_fwd(gap, se_w1, se_w2)
        return res, gap, (hbuf, ch), None
    main = torch.cuda.current_stream()
    gate.wait_stream(main)                     # x is complete once the main stream gets here
    if after is not None:                      # (acc_into: the partial result another stream wrote)
        gate.wait_event(after)
    with torch.cuda.stream(gate):
        res, gap = conv1_gap(code, x, wp_pt, bias_pt, f, tdt, acc_into=acc_into)
        hbuf, ch = ops.se_mlp_fwd(gap, se_w1, se_w2)
    x.record_stream(gate)
    if acc_into is not None:
        acc_into.record_stream(gate)
    for t in (res, gap, hbuf, ch):             # allocated on the gate stream, consumed (and later freed) on the main one
        t.record_stream(main)
    return res, gap, (hbuf, ch), gate


class LowPrecisionForward(object):
    """model(x, training=False, inference=True) with 16-bit storage: `LowPrecisionForward(model, 'float16')(x)` -> y_pred fp32
    [N,D,H,W,out_ch] (model.py:63-68), or [N,out_ch,D,H,W] from NCDHW volumes for a model built with
    data_format='channels_first' (args.py:121-123: memory stays NDHWC, GroupNorm takes its channel-group form).  Both sampler
    families of args.py:136-141 (conv | max, conv | linear)."""

    def __init__(self, model, dtype='float16'):
        if dtype not in DTYPES:
            raise ValueError('dtype must be one of %s' % sorted(DTYPES))
        self.model = model
        self.code, self.tdt = DTYPES[dtype]
        self._packs = {}
        from .layers.downsample import ConvDownsample, MaxDownsample
        from .layers.upsample import ConvUpsample, LinearUpsample
        self._max, self._linear = MaxDownsample, LinearUpsample
        for convs, down in model.encoder.levels:
            if down is not None and not isinstance(down, (ConvDownsample, MaxDownsample)):
                raise NotImplementedError('unknown down-sampling layer %r' % type(down).__name__)
        for up, _ in model.decoder.levels:
            if not isinstance(up, (ConvUpsample, LinearUpsample)):
                raise NotImplementedError('unknown up-sampling layer %r' % type(up).__name__)
        self.channels_first = model.data_format == 'channels_first'
        # MEASURED AND CLOSED (round 6, OFF by default; BTS_LP_EARLY_SKIP=1 switches it on for A/B and the parity test).  Decoder blocks read
        # [skip | up-sampled] (decoder.py:75): the skip part of conv1 and of the shortcut (with their biases) needs only the ENCODER level's
        # output, so it can run on a side stream next to the two deepest levels, whose grids leave half the chip idle on a single volume
        # (round 4's batch sweep: 150 -> 185 volumes/s from batch 1 to 8), and the decoder adds the up-sampled part to it
        # (bts_lp_conv3d_fwd_gn_acc / bts_lp_conv1_gap_acc).  The same contraction split over its input channels; the partial sum passes
        # through the storage type once.  Result: 6.26 -> 6.56 ms started right away, 6.26 -> 6.62 ms started where the main stream enters
        # the deep levels (profiles/r06_ab_e6*.txt): the split costs a read-modify-write of c1 and res per decoder block and two launches
        # more, and the side kernels (one 137-KB workgroup per CU) find no CU to run on until a main-stream kernel drains
        self.early_skip = os.environ.get('BTS_LP_EARLY_SKIP', '0') == '1'
        self.fuse_head = os.environ.get('BTS_LP_FUSE_HEAD', '1') != '0'      # (=0: block epilogue and output head as two launches; A/B)
        self.fuse_first = os.environ.get('BTS_LP_C2', '1') != '0'            # (=0: padded 16-channel copy + generic kernels for the first block; A/B)
        # the top decoder block's [skip | up-sampled] input (decoder.py:75) as two DENSE 32-channel operands instead of a 64-channel slab: its
        # conv1, shortcut and squeeze are then two z-marching passes over whole 128-byte lines (bts_lp_conv3d_fwd_gn_shortcut with x_split),
        # and the level's other readers (down-sampler, up-sampler's GroupNorm) see dense tensors too (=0: the slab; A/B)
        self.split_level0 = os.environ.get('BTS_LP_INF_SPLIT', '1') != '0'
        bf = getattr(model.encoder, 'base_filters', 16)
        if bf % 16 != 0:
            # every 16-bit convolution contracts over whole matrix steps of 16 input channels (v_mfma_f32_32x32x16): an 8-filter level
            # cannot feed its own second conv.  Refused here, by name, rather than by an alignment status from the first launch
            raise ValueError('the 16-bit engine needs base_filters to be a multiple of 16 (got %d); use the fp32 engine for such models' % bf)

    def _packed(self, key, kind, param, cin_ref, cout, cin_slab=None, dup_start=0, dup_shift=0):
        ent = self._packs.get(key)
        sig = (kind, cin_ref, cout, cin_slab, dup_start, dup_shift, id(param))
        if ent is None or ent[0] != weights_epoch() or ent[1] != sig:
            ent = (weights_epoch(), sig, pack(kind, self.code, param.t, cin_ref, cout, cin_slab, dup_start, dup_shift))
            self._packs[key] = ent
        return ent[2]

    def _packed_rows(self, key, kind, param, c0, c1, cout):
        """packed image of the input-channel rows [c0, c1) of a kernel (the part of a conv over a concat that reads one operand)"""
        ent = self._packs.get(key)
        sig = (kind, c0, c1, cout, id(param))
        if ent is None or ent[0] != weights_epoch() or ent[1] != sig:
            rows = param.t[..., c0:c1, :].contiguous()
            ent = (weights_epoch(), sig, pack(kind, self.code, rows, c1 - c0, cout))
            self._packs[key] = ent
        return ent[2]

    def _skip_part(self, blk, skip):
        """conv1 and shortcut of decoder block `blk` over its SKIP operand only (with the biases), on the 'skip' side stream ->
        (c1 partial, res partial, event) or None"""
        side = ops.side_stream('skip')
        if side is None or blk.norm1._mode != ops.GN_SLAB:
            return None
        code, tdt = self.code, self.tdt
        f = blk.filters
        cres = skip.shape[-1]
        key = id(blk)
        wp3 = self._packed_rows((key, 'c1s'), ops.K3S1, blk.conv1_k, 0, cres, f)
        wp1 = self._packed_rows((key, 'pts'), ops.K1, blk.ptwise_k, 0, cres, f)
        main = torch.cuda.current_stream()
        side.wait_stream(main)                      # the encoder level is complete once the main stream gets here
        with torch.cuda.stream(side):
            c1p = conv(ops.K3S1, code, tdt, skip, wp3, blk.conv1_b.t, f)
            resp = conv(ops.K1, code, tdt, skip, wp1, blk.ptwise_b.t, f)
            ev = torch.cuda.Event()
            ev.record(side)
        skip.record_stream(side)
        for t in (c1p, resp):                       # allocated on the side stream, finished and freed on others
            t.record_stream(main)
        return c1p, resp, ev

    def _gn(self, norm, c, relu, out=None):
        m, r = gn_stats(self.code, c, norm.groups, norm._mode, norm.epsilon)
        return gn_apply(self.code, c, norm.gamma.t, norm.beta.t, m, r, norm.groups, norm._mode, relu, out=out)

    def _block(self, blk, x, out, fold=None, head=None, first=None, early=None, cres=0):
        """ResnetBlock.call (resnet.py:116-138); x: 16-bit view, out: 16-bit view or None.  head = (W (C,K), b (K)): this is the last block
        and its only reader is the sigmoid output head -- returns ('head', y_pred) where the fused epilogue takes the shape"""
        code, tdt = self.code, self.tdt
        f, g = blk.filters, blk.groups
        n, d, h, w, cin = xdims(x)[:5]      # (x: a 5-d view, or the two operands of the level-0 concat as a (2, N, D, H, W, 32) buffer)
        dup_start, dup_shift = fold if fold else (0, 0)
        v = d * h * w
        key = id(blk)
        if first is None and cin % 16 != 0:
            raise RuntimeError('16-bit convolutions step over 16 input channels; got a %d-channel view' % cin)
        # (the 2-channel input volume arrives zero-padded to 16 channels, see __call__: blk.cin_ref real channels, the rest
        # of the k-step multiplies zeros in both operands)
        cin_slab = min(cin, blk.cin_ref) if fold is None else cin
        if first is not None:      # conv1, its statistics, the shortcut and the squeeze came out of the first-block kernel
            c1, m1, r1, res, gap = first
            _, ch = ops.se_mlp_fwd(gap, blk.se_w1.t, blk.se_w2.t)
            gate = None
        elif early is not None:    # a decoder block whose skip part is (being) computed on the side stream: add the up-sampled part
            c1p, resp, ev = early
            xu = x[..., cres:]
            wp_pt = self._packed_rows((key, 'ptu'), ops.K1, blk.ptwise_k, cres, cin, f)
            wp_c1 = self._packed_rows((key, 'c1u'), ops.K3S1, blk.conv1_k, cres, cin, f)
            torch.cuda.current_stream().wait_event(ev)
            res, gap, (_, ch), gate = gate_branch(code, tdt, xu, wp_pt, None, f, blk.se_w1.t, blk.se_w2.t, acc_into=resp, after=ev)
            c1, m1, r1 = conv_gn(code, tdt, xu, wp_c1, None, f, blk.norm1, acc_into=c1p)
        elif is_split(x):          # conv1 + statistics, shortcut + squeeze: one launch pair over the two dense operands
            wp_pt = self._packed((key, 'pt'), ops.K1, blk.ptwise_k, blk.cin_ref, f, cin_slab, 0, 0)
            wp_c1 = self._packed((key, 'c1'), ops.K3S1, blk.conv1_k, blk.cin_ref, f, cin_slab, 0, 0)
            both = conv_gn_shortcut(code, tdt, x, wp_c1, blk.conv1_b.t, f, blk.norm1, wp_pt, blk.ptwise_b.t)
            if both is None:
                raise RuntimeError('split concat input: the fused two-pass launch declined a shape its query accepted')
            c1, m1, r1, res, gap = both
            _, ch = ops.se_mlp_fwd(gap, blk.se_w1.t, blk.se_w2.t)
            gate = None
        else:
            wp_pt = self._packed((key, 'pt'), ops.K1, blk.ptwise_k, blk.cin_ref, f, cin_slab, dup_start, dup_shift)
            wp_c1 = self._packed((key, 'c1'), ops.K3S1, blk.conv1_k, blk.cin_ref, f, cin_slab, dup_start, dup_shift)
            res, gap, (_, ch), gate = gate_branch(code, tdt, x, wp_pt, blk.ptwise_b.t, f, blk.se_w1.t, blk.se_w2.t)
            c1, m1, r1 = conv_gn(code, tdt, x, wp_c1, blk.conv1_b.t, f, blk.norm1)      # conv + the statistics of its output
        wp_c2 = self._packed((id(blk), 'c2'), ops.K3S1, blk.conv2_k, f, f)
        # conv2 reads relu(GN1(c1)); nobody else does in a forward without a backward: where the library can, GN1 + ReLU are applied to
        # conv2's input planes inside the conv kernel and the apply pass (1 read + 1 write of the tensor) goes away
        fused = conv_gn_normed_input(code, tdt, c1, blk.norm1, m1, r1, True, wp_c2, blk.conv2_b.t, f, blk.norm2)
        if fused is not None:
            c2, m2, r2 = fused
        else:
            a = gn_apply(code, c1, blk.norm1.gamma.t, blk.norm1.beta.t, m1, r1, g, blk.norm1._mode, True)
            c2, m2, r2 = conv_gn(code, tdt, a, wp_c2, blk.conv2_b.t, f, blk.norm2)
            del a
        del c1
        if gate is not None:
            torch.cuda.current_stream().wait_stream(gate)       # the epilogue is where the two branches meet (resnet.py:130,137)
        if head is not None and out is None and self.fuse_head:
            yh = block_epilogue_head(code, res, c2, blk.spatial_k.t.reshape(-1), ch, blk.norm2.gamma.t, blk.norm2.beta.t, m2, r2, g,
                                     blk.norm2._mode, head[0], head[1], True)
            if yh is not None:
                return ('head', yh)
        if out is None:
            out = torch.empty((n, d, h, w, f), dtype=tdt, device=res.device)
        return block_epilogue(code, res, c2, out, blk.spatial_k.t.reshape(-1), ch, blk.norm2.gamma.t, blk.norm2.beta.t, m2, r2, g,
                              blk.norm2._mode)

    def _level0_split_ok(self, n, d, h, w, nb, f):
        """can level 0 live as two dense 32-channel operands?  One encoder block there (its output is the whole skip), a decoder that
        concatenates [skip | up-sampled] of 32 channels each, and the library taking the pair in its fused two-pass form"""
        m = self.model
        if not self.split_level0 or self.early_skip or nb != 1 or f != 32 or len(m.decoder.levels) < 1 or len(m.encoder.levels) < 2:
            return False
        up, blk = m.decoder.levels[-1]
        if up.filters != f or blk.filters != f or blk.norm1._mode != ops.GN_SLAB:
            return False
        return lib().probe('bts_lp_conv3d_fwd_gn_shortcut_workspace', n, d, h, w, 2 * f, f, f, blk.norm1.groups) >= 0

    def _down(self, lay, x):
        if isinstance(lay, self._max):                                                   # downsample.py:51-70
            return maxpool2(self.code, x, want_idx=False)[0]
        wp = self._packed((id(lay), 'f'), ops.K3S2, lay.conv_k, lay.cin, lay.filters)
        c = conv(ops.K3S2, self.code, self.tdt, x, wp, lay.conv_b.t, lay.filters)
        return self._gn(lay.norm, c, True)

    def _up(self, lay, x, out):
        if isinstance(lay, self._linear):                                                # upsample.py:49-79: 1x1x1 conv, then repeat
            wp = self._packed((id(lay), 'f'), ops.K1, lay.ptwise_k, lay.cin, lay.filters)
            c = conv(ops.K1, self.code, self.tdt, x, wp, lay.ptwise_b.t, lay.filters)
            return upsample2(self.code, c, out=out)
        wp = self._packed((id(lay), 'f'), ops.K3S2T, lay.conv_k, lay.cin, lay.filters)
        c, m_, r_ = convT_gn(self.code, self.tdt, x, wp, lay.conv_b.t, lay.filters, lay.norm)      # conv + the statistics of its output
        return gn_apply(self.code, c, lay.norm.gamma.t, lay.norm.beta.t, m_, r_, lay.norm.groups, lay.norm._mode, True, out=out)

    def __call__(self, x):
        m = self.model
        if not m.built:
            raise RuntimeError('build the model first (its weights belong to a training crop)')
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if not x.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError('no MI355X visible: the engine has no CPU execution path')
            x = x.cuda()
        fence = ops.step_fence('infer', depth=3)
        x = x.float()
        if self.channels_first:                       # raw NCDHW in (tape.as_tensor does the same for the fp32 engine)
            x = x.permute(0, 2, 3, 4, 1)
        x = x.contiguous()
        if any(s % (2 ** (m.encoder.depth - 1)) for s in x.shape[1:4]):
            raise ValueError('spatial sizes must be multiples of %d (test.py:164-178 pads to that)' % 2 ** (m.encoder.depth - 1))
        enc0 = m.encoder.levels[0][0][0]
        first = None
        if self.fuse_first and x.shape[-1] == 2 and enc0.cin_ref == 2:
            # the raw 2-channel volume straight into the first block's two convolutions (csrc/lowp_c2.hip): no padded 16-channel copy
            first = first_block(self.code, self.tdt, x, enc0.conv1_k.t, enc0.conv1_b.t, enc0.ptwise_k.t, enc0.ptwise_b.t, enc0.filters, enc0.norm1)
        # the input volume in the storage type, zero-padded to one 16-channel matrix step (in_ch = 2: model.py:18)
        if first is not None:
            pass
        elif x.shape[-1] <= 4:
            x = cast_pad16(self.code, self.tdt, x)
        else:
            cpad = (x.shape[-1] + 15) // 16 * 16
            xin = torch.zeros(tuple(x.shape[:4]) + (cpad,), dtype=self.tdt, device=x.device)
            cast(self.code, self.tdt, x, out=xin[..., :x.shape[-1]])
            x = xin
        enc, dec = m.encoder, m.decoder
        n = x.shape[0]
        residuals = []
        early, pending = {}, []
        deep = max(len(enc.levels) - 2, 1)            # first of the two deepest levels
        cur = x
        for i, (convs, down) in enumerate(enc.levels):
            d, h, w = cur.shape[1:4]
            f = enc.base_filters * 2 ** i
            nb = len(convs)
            spare = f if i < enc.depth - 1 else 0
            if i == 0 and self._level0_split_ok(n, d, h, w, nb, f):
                # level 0 as two dense operands: [0] this level's output (the skip), [1] the up-sampled tensor the decoder writes later
                pair = torch.empty((2, n, d, h, w, f), dtype=self.tdt, device=x.device)
                self._block(convs[0], cur, pair[0], first=first)
                residuals.append((pair, f))
                if down is not None:
                    cur = self._down(down, pair[0])
                continue
            slab = torch.empty((n, d, h, w, nb * f + spare), dtype=self.tdt, device=x.device)
            for j, blk in enumerate(convs):
                out = slab[..., j * f:(j + 1) * f]
                if j == 0:
                    self._block(blk, cur, out, first=first if i == 0 else None)
                else:
                    self._block(blk, slab[..., :j * f], out, fold=((j - 1) * f, f))       # encoder.py:83-87
            residuals.append((slab, nb * f))
            if down is not None:
                if self.early_skip and i < len(dec.levels):
                    # the decoder block of this level reads [this level's output | up-sampled]: its skip part is ready to run from here on
                    pending.append((i, dec.levels[len(dec.levels) - 1 - i][1], slab[..., :nb * f]))
                cur = self._down(down, slab[..., :nb * f])                                # encoder.py:97-98
            if i + 1 >= deep:
                # ... and is STARTED where the main stream enters the two deepest levels: their grids (360 and 120-144 workgroups on the
                # full inference volume) leave CUs idle, the levels above fill the chip (started right away the side work only competed
                # with them: 6.26 -> 6.56 ms, profiles/r06_ab_e6.txt)
                for lv, dblk, view in pending:
                    early[lv] = self._skip_part(dblk, view)
                pending = []
        slab, used = residuals[-1]
        y = slab[..., :used]
        hw = dec.out_k.t.reshape(dec.out_k.t.shape[-2], dec.out_k.t.shape[-1])
        nlev = len(dec.levels)
        for li, ((up, blk), (slab, cres)) in enumerate(zip(dec.levels, residuals[-2::-1])):
            f = up.filters
            lvl = nlev - 1 - li                                                            # the encoder level whose output is the skip
            if is_split(slab):
                self._up(up, y, slab[1])
                y = self._block(blk, slab, None, head=(hw, dec.out_b.t) if li == nlev - 1 else None)
                continue
            self._up(up, y, slab[..., cres:cres + f])                                     # decoder.py:72
            # (the top block's output has one reader, the output head: folded into its epilogue where the library can)
            y = self._block(blk, slab[..., :cres + f], None, head=(hw, dec.out_b.t) if li == nlev - 1 else None,  # decoder.py:75-78
                            early=early.get(lvl), cres=cres)
        if isinstance(y, tuple):
            yp = y[1]
        else:
            yp = head(self.code, y, hw, dec.out_b.t, True)
        ops.step_fence_done(fence)
        return yp.permute(0, 4, 1, 2, 3) if self.channels_first else yp      # the public layout (a view, like Tensor.public())
