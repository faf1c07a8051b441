"""ResnetBlock -- drop-in for layers/resnet.py of the reference (constructor :8-113, call :116-138).

    res  = conv1x1(inputs)                                    :118
    chse = sigmoid(W2^T relu(W1^T GAP(res)))                  :121-124
    spse = sigmoid(conv1x1(res; F->1, no bias))               :127
    res  = res * (spse + chse)                                :130   (gate on the SHORTCUT, sum of the two gates)
    a = relu(GN1(conv3(inputs))) ; c = relu(GN2(conv3(a)))    :133-136
    return res + c                                            :137

One layer call = one tape node: forward launches ~12 HIP kernels, backward ~20 (SURVEY Appendix A').
`inputs` may be a channel-slice view of a level slab; `out` an output slice; `fold=(dup_start, dup_shift)` tells the
block that the reference would have seen [o_{j-1}, slab...] (encoder.py:83-87) so its weights carry dup_shift extra
input channels that are folded at pack time.
"""
import os

import torch

from .. import ops
from ._base import Layer, Tensor, as_tensor, check_data_format, current_tape, gn_mode_of
from .group_norm import GroupNormalization


class ResnetBlock(Layer):
    def __init__(self, filters, data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, name=None):
        super(ResnetBlock, self).__init__(name=name)
        self.data_format = check_data_format(data_format)
        self.config = {'filters': filters, 'data_format': data_format, 'reduction': reduction, 'l2_scale': l2_scale,
                       'groups': groups}
        if filters % reduction != 0:
            raise ValueError('Reduction ratio, {}, must be a factor of number of channels, {}.'.format(reduction, filters))
        self.filters = filters
        self.groups = groups
        self.reduction = reduction
        self.l2_scale = l2_scale
        self.norm1 = self.track(GroupNormalization(groups=groups, axis=-1, beta_initializer='zeros',
                                                   gamma_initializer='ones', beta_regularizer=l2_scale,
                                                   gamma_regularizer=l2_scale, name=self.name + '/gn1',
                                                   semantics=gn_mode_of(data_format)))
        self.norm2 = self.track(GroupNormalization(groups=groups, axis=-1, beta_initializer='zeros',
                                                   gamma_initializer='zeros', beta_regularizer=l2_scale,
                                                   gamma_regularizer=l2_scale, name=self.name + '/gn2',
                                                   semantics=gn_mode_of(data_format)))

    def build(self, input_shape, fold=None):
        cin = input_shape[-1] + (fold[1] if fold else 0)
        f, l2 = self.filters, self.l2_scale
        self.cin_ref = cin
        self.ptwise_k = self.add_weight('ptwise_k', (1, 1, 1, cin, f), 'he_normal', l2)
        self.ptwise_b = self.add_weight('ptwise_b', (f,), 'zeros')
        self.se_w1 = self.add_weight('se_w1', (f, f // self.reduction), 'he_normal', l2)
        self.se_w2 = self.add_weight('se_w2', (f // self.reduction, f), 'he_normal', l2)
        self.spatial_k = self.add_weight('spatial_k', (1, 1, 1, f, 1), 'he_normal', l2)
        self.conv1_k = self.add_weight('conv1_k', (3, 3, 3, cin, f), 'he_normal', l2)
        self.conv1_b = self.add_weight('conv1_b', (f,), 'zeros')
        self.norm1.build((None, None, None, None, f))
        self.conv2_k = self.add_weight('conv2_k', (3, 3, 3, f, f), 'he_normal', l2)
        self.conv2_b = self.add_weight('conv2_b', (f,), 'zeros')
        self.norm2.build((None, None, None, None, f))
        self.built = True

    @property
    def trainable_variables(self):
        # creation order of the reference constructor: ptwise, SE denses, spatial, conv1, GN1, conv2, GN2
        n1, n2 = self.norm1._params, self.norm2._params
        p = self._params
        return p[:7] + n1 + p[7:9] + n2

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.filters,)

    def __call__(self, inputs, training=None, out=None, fold=None):
        inputs = as_tensor(inputs, data_format=self.data_format)   # raw NCDHW -> NDHWC for channels_first
        if not self.built:
            self.build(tuple(inputs.shape), fold)
        return self.call(inputs, training=training, out=out, fold=fold)

    def call(self, inputs, training=None, out=None, fold=None):
        x = as_tensor(inputs, data_format=self.data_format)
        f, g = self.filters, self.groups
        n, d, h, w, cin = x.shape
        dup_start, dup_shift = fold if fold else (0, 0)
        if cin + dup_shift != self.cin_ref:
            raise ValueError('ResnetBlock built for %d input channels, got %d' % (self.cin_ref, cin + dup_shift))
        v = d * h * w
        K1, K3 = ops.K1, ops.K3S1
        pk = lambda key, kind, role, p, ci, co, folded: self.packed(
            key, kind, role, p, ci, co, cin if folded else None, dup_start if folded else 0, dup_shift if folded else 0)
        wp_pt = pk('pt_f', K1, ops.ROLE_FWD, self.ptwise_k, self.cin_ref, f, True)
        wp_c1 = pk('c1_f', K3, ops.ROLE_FWD, self.conv1_k, self.cin_ref, f, True)
        wp_c2 = pk('c2_f', K3, ops.ROLE_FWD, self.conv2_k, f, f, False)
        # shortcut conv + conv1 share the input tile: one fused pass where the tiling has the registers for it
        # channels_last GroupNorm statistics (z-slab groups) come out of the producing conv's epilogue where possible
        slab = self.norm1._mode == ops.GN_SLAB
        m1 = r1 = None
        res = None
        if slab:
            fused = ops.conv_fwd_fused2_gn(x.t, wp_c1, self.conv1_b.t, wp_pt, self.ptwise_b.t, f, g, self.norm1.epsilon)
            if fused is not None:
                c1, res, m1, r1 = fused
        else:
            fused = ops.conv_fwd_fused2(x.t, wp_c1, self.conv1_b.t, wp_pt, self.ptwise_b.t, f)
            if fused is not None:
                c1, res = fused
        # gate branch (resnet.py:118-130): shortcut conv (HBM-bound), global average pool, SE MLP -- on the 'gate' stream next to
        # the conv branch when the shortcut did not come out of the fused first-block kernel
        main = torch.cuda.current_stream()
        gate = ops.side_stream('gate') if res is None else None

        def gate_branch(res_):
            if res_ is None:
                res_ = ops.conv_fwd(K1, x.t, wp_pt, self.ptwise_b.t, f)
            gap_ = ops.colsum(res_, scale=1.0 / v)
            hbuf_, ch_ = ops.se_mlp_fwd(gap_, self.se_w1.t, self.se_w2.t)
            return res_, gap_, hbuf_, ch_

        if gate is not None:
            gate.wait_stream(main)                     # x is complete once the main stream gets here
            with torch.cuda.stream(gate):
                res, gap, hbuf, ch = gate_branch(None)
            x.t.record_stream(gate)
            for t in (res, gap, hbuf, ch):             # allocated on the gate stream, consumed (and later freed) on the main one
                t.record_stream(main)
        # conv branch
        if fused is None:
            if slab:
                c1, m1, r1 = ops.conv_fwd_gn(K3, x.t, wp_c1, self.conv1_b.t, f, g, self.norm1.epsilon)
            else:
                c1 = ops.conv_fwd(K3, x.t, wp_c1, self.conv1_b.t, f)
        if gate is None:
            res, gap, hbuf, ch = gate_branch(res)
        if m1 is None:
            m1, r1 = ops.gn_stats(c1, g, self.norm1._mode, self.norm1.epsilon)
        a = ops.gn_apply(c1, self.norm1.gamma.t, self.norm1.beta.t, m1, r1, g, self.norm1._mode, True)
        if slab:
            c2, m2, r2 = ops.conv_fwd_gn(K3, a, wp_c2, self.conv2_b.t, f, g, self.norm2.epsilon)
        else:
            c2 = ops.conv_fwd(K3, a, wp_c2, self.conv2_b.t, f)
            m2, r2 = ops.gn_stats(c2, g, self.norm2._mode, self.norm2.epsilon)
        if gate is not None:
            main.wait_stream(gate)                     # the epilogue is where the two branches meet (resnet.py:130,137)
        if out is None:
            out = Tensor(torch.empty((n, d, h, w, f), dtype=torch.float32, device=x.t.device))
        wsp = self.spatial_k.t.reshape(-1)
        sp = ops.block_epilogue_fwd(res, c2, out.t, wsp, ch, self.norm2.gamma.t, self.norm2.beta.t, m2, r2, g, self.norm2._mode)
        out.cf = self.data_format == 'channels_first'
        tape = current_tape()
        if tape is not None:
            def backward():
                self._backward(x, out, res, gap, hbuf, ch, sp, c1, m1, r1, a, c2, m2, r2, dup_start, dup_shift)
            tape.record(backward)
        return out

    def _backward(self, x, out, res, gap, hbuf, ch, sp, c1, m1, r1, a, c2, m2, r2, dup_start, dup_shift):
        dout = out.grad
        if dout is None:
            return
        f, g = self.filters, self.groups
        cin = x.shape[-1]
        K1, K3 = ops.K1, ops.K3S1
        n1, n2 = self.norm1, self.norm2
        from .group_norm import group_norm_backward
        # ---- gate branch (independent of the conv branch until both gradients meet in dx): on the 'gate' stream
        gw1, a1 = self.se_w1.grad_slot()
        gw2, a2 = self.se_w2.grad_slot()
        gws, a3 = self.spatial_k.grad_slot()
        main = torch.cuda.current_stream()
        gate = ops.side_stream('gate')
        # gate + GroupNorm-2 backward in one pair of passes where the library takes the shape (slab mode: the layout the reference's
        # channels_last GroupNormalization reduces over); BTS_FUSE_BLOCK_BWD=0: the two separate routes (A/B)
        fused = None
        if n2._mode == ops.GN_SLAB and n2.gamma is not None and n2.beta is not None and os.environ.get('BTS_FUSE_BLOCK_BWD', '1') != '0':
            fused = self._fused_gate_gn2_backward(dout, res, c2, sp, gap, hbuf, ch, m2, r2, (gw1, a1), (gw2, a2), (gws, a3))

        def gate_backward():
            acc_ = a1
            if not (a1 == a2 == a3):
                for buf, ac in ((gw1, a1), (gw2, a2), (gws, a3)):
                    if not ac:
                        ops.fill(buf, 0.0)
                acc_ = True
            return ops.se_bwd(dout, res, sp, gap, hbuf, ch, self.se_w1.t, self.se_w2.t, self.spatial_k.t.reshape(-1), gw1, gw2,
                              gws.reshape(-1), accumulate_params=acc_)

        if fused is not None:
            dres, dc2 = fused
            gate = None
        elif gate is not None:
            gate.wait_stream(main)                     # dout is complete once the main stream gets here
            with torch.cuda.stream(gate):
                dres = gate_backward()
            for t in (dout, res, sp, gap, hbuf, ch):
                t.record_stream(gate)
            dres.record_stream(main)
        # ---- conv branch: GN2 -> conv2 -> GN1 -> conv1
        if fused is None:
            dc2 = group_norm_backward(n2, c2, dout, n2.gamma.t, n2.beta.t, m2, r2, True)
        wpb2 = self.packed('c2_b', K3, ops.ROLE_BWD, self.conv2_k, f, f)
        da = torch.empty_like(a)
        ops.conv_bwd_data(K3, dc2, wpb2, da, False)
        _wgrad(K3, a, dc2, self.conv2_k, self.conv2_b)
        del dc2
        dc1 = group_norm_backward(n1, c1, da, n1.gamma.t, n1.beta.t, m1, r1, True)
        del da
        need_dx = x.requires_grad
        if fused is not None:
            pass
        elif gate is None:
            dres = gate_backward()
        else:
            main.wait_stream(gate)                     # both gradients into x leave in one pass below
        if need_dx:   # dx (+)= conv1^T dc1 + shortcut^T dres: the 1x1x1 term rides on the centre tap of the 3x3x3 sweep
            dx, acc = x.grad_slot()
            wpb1 = self.packed('c1_b', K3, ops.ROLE_BWD, self.conv1_k, self.cin_ref, f, cin, dup_start, dup_shift)
            wpbp = self.packed('pt_b', K1, ops.ROLE_BWD, self.ptwise_k, self.cin_ref, f, cin, dup_start, dup_shift)
            ops.conv_bwd_data_pair(dc1, wpb1, dres, wpbp, dx, acc)
        _wgrad(K3, x.t, dc1, self.conv1_k, self.conv1_b, dup_start, dup_shift)
        del dc1
        _wgrad(K1, x.t, dres, self.ptwise_k, self.ptwise_b, dup_start, dup_shift)

    def _fused_gate_gn2_backward(self, dout, res, c2, sp, gap, hbuf, ch, m2, r2, s1, s2, s3):
        """(dres, dc2) through ops.block_bwd, or None where it declines; parameter gradients go into the grad slots"""
        n2 = self.norm2
        (gw1, a1), (gw2, a2), (gws, a3) = s1, s2, s3
        if not (torch.is_tensor(dout) and dout.shape[-1] == self.filters):
            return None
        if not ops.block_bwd_takes(res, self.se_w1.t.shape[1], n2.groups, dout, c2):
            return None
        dg, ag = n2.gamma.grad_slot()
        db, ab = n2.beta.grad_slot()
        if not (a1 == a2 == a3):
            for buf, ac in ((gw1, a1), (gw2, a2), (gws, a3)):
                if not ac:
                    ops.fill(buf, 0.0)
            a1 = True
        if ag != ab:
            if not ag:
                ops.fill(dg, 0.0)
            if not ab:
                ops.fill(db, 0.0)
            ag = True
        out = ops.block_bwd(dout, res, c2, sp, gap, hbuf, ch, self.se_w1.t, self.se_w2.t, self.spatial_k.t.reshape(-1), n2.gamma.t, n2.beta.t,
                            m2, r2, n2.groups, gw1, gw2, gws.reshape(-1), dg, db, accumulate_gate_params=a1, accumulate_norm_params=ag)
        if out is None:     # the grad slots are claimed: a fallback would accumulate into memory nothing wrote
            raise RuntimeError('ops.block_bwd declined a block ops.block_bwd_takes accepted')
        return out

    def get_config(self):
        return self.config


def _wgrad(kind, x, dy, kparam, bparam, dup_start=0, dup_shift=0):
    """weight + bias gradient into the parameters' grad slots -- enqueued on the side stream (ops.side_stream): nothing in
    the backward pass reads a parameter gradient, so these launches only have to finish before the regulariser / gradient
    exchange / optimiser (ops.join_side_stream there)"""
    side = ops.side_stream()
    if side is None:
        return _wgrad_here(kind, x, dy, kparam, bparam, dup_start, dup_shift)
    side.wait_stream(torch.cuda.current_stream())       # x and dy are complete once the main stream gets here
    with torch.cuda.stream(side):
        _wgrad_here(kind, x, dy, kparam, bparam, dup_start, dup_shift)
    for t in (x, dy):                                   # temporaries the main stream frees: the allocator must not hand
        t.record_stream(side)                           # their memory out before the side stream has read them


def _wgrad_here(kind, x, dy, kparam, bparam, dup_start=0, dup_shift=0):
    dw, aw = kparam.grad_slot()
    db, ab = (None, aw) if bparam is None else bparam.grad_slot()
    if bparam is not None and aw != ab:
        if not aw:
            ops.fill(dw, 0.0)
        if not ab:
            ops.fill(db, 0.0)
        aw = True
    if kind == ops.K3S2T:
        ops.conv_bwd_weight(kind, x, dy, dw, None, accumulate=aw)
        if bparam is not None:
            ops.colsum(dy, sum_over_n=True, out=db, accumulate=aw)
    else:
        ops.conv_bwd_weight(kind, x, dy, dw, db, dup_start, dup_shift, accumulate=aw)
