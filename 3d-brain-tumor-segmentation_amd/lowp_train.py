"""One training step with 16-bit STORAGE (BASELINE configs[2]: bf16, batch 8; the step itself is train.py:140-152).

Activations and their gradients live in HBM as bf16 (or fp16), packed weight images too; every sum is fp32 and the master
weights, their gradients and the Adam moments stay the model's flat fp32 buffers.  Where each piece of the step runs:

  forward convolutions, GroupNorm, pooling, block epilogue            16-bit kernels of csrc/lowp.hip (as bts_amd.lowp)
  data gradients of every convolution                                  the same kernels on role-swapped weight images
                                                                       (bts_lp_conv3d_bwd_data), accumulating into slab gradients
  weight gradients of the stride-1 3x3x3 / 1x1x1 convolutions,         16-bit kernels too (bts_lp_conv3d_bwd_weight: the contraction
  GroupNorm gradients                                                  over voxels gathers its operand fragments from LDS;
                                                                       bts_lp_gn_bwd)
  weight gradients of the strided / transposed convs and of the       the fp32 engine's kernels on operands widened by a
  2-channel first block; squeeze-excitation gradients; the dense       staging pass (bts_lp_uncast) -- exact on the stored values
  VAE head; loss, metric, regulariser, Adam

The graph is the fp32 layers' own (same virtual concatenation through level slabs, same folded duplicate slices), written
out explicitly -- forward saves what the backward needs, the backward walks it in reverse -- so the fp32 engine (the parity
reference of this path: tests/test_lowp_train_gpu.py) is not touched.  Parameter gradients are accumulated into a zeroed
flat buffer; slab gradients are zero-initialised and accumulated into (a 16-bit accumulation: every partial sum is rounded
to the storage type, which is part of what "16-bit storage" means and is bounded by the parity test).
"""
import math
import os

import torch

from . import lowp, ops, parallel
from .lowp import DTYPES, block_epilogue, cast, colsum, conv, conv1_gap, conv_bwd_data, conv_gn, gn_apply, gn_stats, head, uncast
from .tape import Tensor, bump_weights_epoch, weights_epoch


def _wgrad16(kind, code, x, dy, dw, *a, **kw):
    """lowp.conv_bwd_weight where the caller has already checked the shape: a declined launch would leave `dw` unwritten (scratch
    tensors are uninitialised memory), so it is an error here, never a silent skip"""
    cout = dw.shape[-2] if kind == ops.K3S2T else dw.shape[-1]           # (Conv3DTranspose kernels are (kd, kh, kw, Cout, Cin))
    if dy.shape[-1] != cout or not lowp.conv_bwd_weight(kind, code, x, dy, dw, *a, **kw):
        raise RuntimeError('16-bit weight gradient declined: kind %d, x %s, dy %s, dw %s' % (kind, tuple(x.shape), tuple(dy.shape), tuple(dw.shape)))


class _Holder(object):
    """what _pk needs of a parameter: .t"""

    def __init__(self, t):
        self.t = t


class _Progress(object):
    """what parallel.GradSync asks of a tape: a generation, the number of backward stages and how many of them have run"""

    def __init__(self, sync, stages):
        from .tape import _generation_of_last_tape
        self.sync = sync
        self.gen = _generation_of_last_tape()
        self.nodes = [None] * stages
        self.nodes_replayed = 0


class LowPrecisionTrainer(object):
    """dtype 'bfloat16' (BASELINE configs[2]) or 'float16'.

    float16 stores activation GRADIENTS in a type whose normal range ends at 6e-5: the gradients of a mean-reduced loss over N x 128^3
    voxels (~1e-7) would be subnormal and lose most of their bits (tests/test_lowp_fullsize_gpu.py measured it in round 3).  The float16
    trainer therefore always runs with dynamic loss scaling: the backward is seeded with `loss_scale` (a power of two: exact) instead
    of 1 and the summed flat fp32 gradient is checked for overflow ON THE DEVICE (ops.grad_nonfinite: one read pass into a flag word);
    Adam un-scales the gradient as it reads it and drops the whole update when the flag is set (ops.adam_tf_step(skip=flag)) -- no host
    round trip between the gradient exchange and the optimiser on any rank.  The host reads the flag ONE STEP LATER (settle(): the copy
    finished long before): a skipped step is then counted, the optimiser's `iterations` taken back by one and the scale halved; after
    `growth_interval` clean steps it doubles.  So the scale of step k+1 still is step k's: an overflow costs two skipped steps, not one.
    State readers (checkpoints, tests) call settle() first.  bfloat16 has fp32's exponent range and runs unscaled (loss_scale = 1, no
    check) unless a scale is asked for."""

    def __init__(self, model, dtype='bfloat16', loss_scale=None, growth_interval=200):
        self.fwd = lowp.LowPrecisionForward(model, dtype)      # checks samplers / layout, owns the forward weight images
        self.model = model
        self.code, self.tdt = DTYPES[dtype]
        self.dtype_name = dtype
        if loss_scale is None:
            loss_scale = 2.0 ** 16 if self.tdt == torch.float16 else 1.0
        if loss_scale <= 0 or 2.0 ** round(math.log2(loss_scale)) != loss_scale:
            raise ValueError('loss_scale must be a power of two, got %r' % (loss_scale,))
        self.loss_scale = float(loss_scale)
        self.dynamic_scale = self.tdt == torch.float16 or self.loss_scale != 1.0
        self.growth_interval = int(growth_interval)
        self._clean_steps = 0
        self.skipped_steps = 0
        self._consecutive_skips = 0
        self.last_grad_scale = 1.0  # what model.flat_grads still carries after step(): the loss scale (Adam un-scales as it reads)
        self._pending = None        # (event, pinned host flag, optimiser, scale of that step): the overflow decision not yet read
        self._flags = None          # two device flag words + pinned host mirrors, used alternately
        self._packs = {}
        self._pads = {}
        self._pack_table = lowp.PackTable()
        # gate + GroupNorm-2 backward of a block in one pair of passes (bts_lp_block_bwd): 89.4 -> 87.5 ms per batch-8 step measured by
        # interleaved rounds inside one process (scripts/lp_fuse_ab.py; A/B between processes drowns in the pool's run-to-run spread).
        # BTS_LP_FUSE_BLOCK_BWD=0: the two separate routes
        self.fuse_block_bwd = os.environ.get('BTS_LP_FUSE_BLOCK_BWD', '1') != '0'
        # conv2's data gradient + GroupNorm-1's backward through bts_lp_conv3d_bwd_data_gn_bwd (class sums from the conv's epilogue where
        # the streaming kernel runs the layer); BTS_LP_FUSE_GN1_BWD=0: the two separate calls (A/B)
        self.fuse_gn1_bwd = os.environ.get('BTS_LP_FUSE_GN1_BWD', '1') != '0'
        # GroupNorm-1 + ReLU applied inside conv2's forward and weight-gradient kernels where both can (the normalised tensor is never
        # written); BTS_LP_FUSE_GN1_APPLY=0: the separate apply pass everywhere (A/B)
        self.fuse_gn1_apply = os.environ.get('BTS_LP_FUSE_GN1_APPLY', '1') != '0'
        # conv1 + shortcut + squeeze from one pass over the block input (bts_lp_conv3d_fwd_gn_shortcut); BTS_LP_FS=0 in the library's
        # environment is the A/B switch
        self.fuse_shortcut_fwd = True
        # level 0's [skip | up-sampled] as two dense operands (see step()); BTS_LP_FWD_SPLIT=0: the 64-wide slab (A/B)
        self.split_level0 = os.environ.get('BTS_LP_FWD_SPLIT', '1') != '0'
        self.last_labels = None
        self._clock = None

    # ---- weight images ----
    def _pk(self, key, kind, param, cin_ref, cout, cin_slab=None, dup_start=0, dup_shift=0, role=ops.ROLE_FWD):
        """packed 16-bit image of `param` for (kind, role).  Images live in persistent buffers; the first request after the parameters
        changed (optimiser step, assign, load) re-packs EVERY image this trainer has handed out in one launch (bts_lp_pack_batch) on the
        current stream -- 194 launches of ~6 us per step before"""
        ent = self._packs.get(key)
        sig = (kind, role, cin_ref, cout, cin_slab, dup_start, dup_shift, id(param))
        if ent is None or ent[1] != sig:
            wp = lowp.pack(kind, self.code, param.t, cin_ref, cout, cin_slab, dup_start, dup_shift, role=role)
            ent = [weights_epoch(), sig, wp, param]
            self._packs[key] = ent
        elif ent[0] != weights_epoch():
            self._repack_all()
        return ent[2]

    def _padded(self, key, param, width):
        """persistent fp32 copy of `param` zero-padded along its LAST axis to `width` (a kernel's output channels or a bias), as an
        object _pk accepts; the live columns are refreshed by _repack_all, i.e. before any image is packed from it (round 3 rebuilt
        the padded kernels with zeros + slice copy + a pack launch of their own on every step, in both directions)"""
        ent = self._pads.get(key)
        if ent is None or ent[1] is not param:
            buf = torch.zeros(tuple(param.t.shape[:-1]) + (width,), dtype=torch.float32, device=param.t.device)
            buf[..., :param.t.shape[-1]].copy_(param.t)
            ent = (_Holder(buf), param, [weights_epoch()])
            self._pads[key] = ent
        elif ent[2][0] != weights_epoch():
            self._refresh_pads()
        return ent[0]

    def _refresh_pads(self):
        ep = weights_epoch()
        for holder, param, stamp in self._pads.values():
            if stamp[0] != ep:
                holder.t[..., :param.t.shape[-1]].copy_(param.t)
                stamp[0] = ep

    def _repack_all(self):
        ep = weights_epoch()
        self._refresh_pads()
        batch = []
        for ent in self._packs.values():
            if ent[0] == ep:
                continue
            kind, role, cin_ref, cout, cin_slab, dup_start, dup_shift, _ = ent[1]
            w = ent[3].t
            if w.is_contiguous():
                batch.append((kind, role, w, ent[2], cin_ref, cout, cin_ref if cin_slab is None else cin_slab, dup_start, dup_shift))
            else:
                ent[2] = lowp.pack(kind, self.code, w, cin_ref, cout, cin_slab, dup_start, dup_shift, role=role)
            ent[0] = ep
        self._pack_table.run(self.code, batch)

    def _f32(self, t):
        return uncast(self.code, t)

    def _b16(self, t, out=None):
        return cast(self.code, self.tdt, t, out=out)

    def _b16_k(self, t):
        """fp32 gradient -> storage type as a matrix-instruction operand: channels zero-padded to a multiple of 16 (the
        contraction steps over 16 channels; only the VAE's 8- / 16-filter down-sampling conv of small models needs the pad)"""
        c = t.shape[-1]
        if c % 16 == 0:
            return self._b16(t)
        buf = torch.zeros(tuple(t.shape[:-1]) + ((c + 15) // 16 * 16,), dtype=self.tdt, device=t.device)
        cast(self.code, self.tdt, t, out=buf[..., :c])
        return buf

    def _gn_bwd(self, norm, c, dy, mean, rstd, want_f32=True, dbias=None):
        """GroupNorm (+ReLU) backward -> (dc in the storage type with channels padded to a matrix step, dc in fp32 or None);
        parameter gradients accumulate.  16-bit kernel where its tiling fits, else the fp32 kernel on widened copies.  dbias: the
        producing conv's bias-gradient slot; self._db_done says whether the pass filled it (else the weight gradient must)"""
        r = None
        self._db_done = False
        if norm._mode == ops.GN_SLAB:
            pad = c.shape[-1] % 16 != 0
            r = lowp.gn_bwd(self.code, self.tdt, c, dy, norm.gamma.t, norm.beta.t, mean, rstd, self._gslot(norm.gamma), self._gslot(norm.beta),
                            norm.groups, True, want_f32=want_f32 or pad, dbias=dbias)
            self._db_done = r is not None and dbias is not None
        if r is not None and r[0].shape[-1] % 16 == 0:
            return r
        if r is not None:
            return self._b16_k(r[1]), r[1]
        dc = ops.gn_bwd(self._f32(c), self._f32(dy), norm.gamma.t, norm.beta.t, mean, rstd, self._gslot(norm.gamma), self._gslot(norm.beta),
                        norm.groups, norm._mode, True, accumulate_params=True)
        return self._b16_k(dc), dc

    @staticmethod
    def _wg(tensors, fn):
        """weight-gradient launches go to the side stream (ops.side_stream('wgrad'), as in the fp32 step): nothing in the backward
        pass reads a parameter gradient; `tensors` are the temporaries the launches read (the main stream frees them)"""
        side = ops.side_stream('wgrad')
        if side is None:
            return fn()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        for t in tensors:
            t.record_stream(side)

    def _written(self, params):
        """one backward stage finished writing these parameters' gradients (the weight-gradient launches may still sit on the side
        stream: GradSync orders a bucket behind that stream's events, not behind the stream).  Data parallel: a bucket of the flat
        gradient buffer whose last member this was gets the regulariser term and is all-reduced while the backward goes on"""
        clock = self._clock
        if clock is None:
            return
        clock.nodes_replayed += 1
        clock.sync.params_written(params)

    def _level0_split_ok(self, n, d, h, w, nb, f, spare, dec):
        """level 0 as two dense 32-channel operands: only where EVERY reader of the pair takes the list (there is no single-tensor fallback)"""
        if not self.split_level0 or nb != 1 or f != 32 or spare != 32 or not dec.levels or not self.fuse_shortcut_fwd:
            return False
        up, blk = dec.levels[-1]
        from ._lib import lib
        if up.filters != 32 or blk.norm1._mode != ops.GN_SLAB or not lowp.wgrad_supported(ops.K3S1, 64, blk.filters):
            return False
        L = lib()
        return (lowp.conv_bwd_data_sc_split_ok(n, d, h, w, 64, blk.filters)
                and L.probe('bts_lp_conv3d_fwd_gn_shortcut_workspace', n, d, h, w, 64, 32, blk.filters, blk.norm1.groups) >= 0
                and L.probe('bts_lp_conv3d_bwd_weight_pair_workspace', n, d, h, w, 64, blk.filters) >= 0)

    @staticmethod
    def _backward_stages(n_vae, n_dec, blocks_per_level):
        """stages of the explicit backward that report parameters: vae.out, (block, up) per VAE level, vae.upsample, the dense pair,
        vae.downsample; decoder.out, (block, up) per decoder level; every encoder block and down-sampler"""
        return 1 + 2 * n_vae + 3 + 1 + 2 * n_dec + sum(blocks_per_level) + len(blocks_per_level) - 1

    @staticmethod
    def _gslot(p):
        """fp32 gradient view of a parameter inside the model's flat gradient buffer (zeroed at the start of the step)"""
        return p._gview

    # ================================================================================================================
    # forward pieces (each returns what its backward needs)
    # ================================================================================================================
    def _block_fwd(self, blk, x, out, fold=None):
        code, tdt = self.code, self.tdt
        f, g = blk.filters, blk.groups
        n, d, h, w, cin = lowp.xdims(x)[:5]      # (x: a 5-d view, or the two operands of the level-0 concat as a (2, N, D, H, W, 32) buffer)
        dup_start, dup_shift = fold if fold else (0, 0)
        v = d * h * w
        cin_slab = min(cin, blk.cin_ref) if fold is None else cin       # (the zero-padded 2-channel input: cin 16, cin_ref 2)
        key = id(blk)
        wp_pt = self._pk((key, 'pt'), ops.K1, blk.ptwise_k, blk.cin_ref, f, cin_slab, dup_start, dup_shift)
        wp_c1 = self._pk((key, 'c1'), ops.K3S1, blk.conv1_k, blk.cin_ref, f, cin_slab, dup_start, dup_shift)
        # conv1 (+ the statistics of its output) and the shortcut conv (+ the gate's squeeze) from ONE pass over x where the z-marching
        # kernel takes the layer (round 6: the shortcut is a second set of output columns at the centre tap); else the shortcut conv +
        # squeeze in one pass of their own.  Then the SE-MLP (main stream: see lowp.gate_branch)
        both = lowp.conv_gn_shortcut(code, tdt, x, wp_c1, blk.conv1_b.t, f, blk.norm1, wp_pt, blk.ptwise_b.t) if self.fuse_shortcut_fwd else None
        if both is not None:
            c1, m1, r1, res, gap = both
            hbuf, ch = ops.se_mlp_fwd(gap, blk.se_w1.t, blk.se_w2.t)
            gate = None
        elif lowp.is_split(x):
            raise RuntimeError('split concat input: the fused two-pass launch declined a shape its query accepted')
        else:
            res, gap, (hbuf, ch), gate = lowp.gate_branch(code, tdt, x, wp_pt, blk.ptwise_b.t, f, blk.se_w1.t, blk.se_w2.t, side=False)
            c1, m1, r1 = conv_gn(code, tdt, x, wp_c1, blk.conv1_b.t, f, blk.norm1)      # conv + the statistics of its output
        wp_c2 = self._pk((key, 'c2'), ops.K3S1, blk.conv2_k, f, f)
        # relu(GN1(c1)) has two readers, conv2's forward and conv2's weight gradient: where both kernels can normalise their input planes
        # themselves (the streaming kernels of the 128^3 level), the tensor is never written (a = None: the backward knows)
        a = None
        if self.fuse_gn1_apply and lowp.wgrad_supported(ops.K3S1, f, f) and lowp.gnin_train_ok(c1, f, blk.norm1, blk.norm2):
            c2, m2, r2 = lowp.conv_gn_normed_input(code, tdt, c1, blk.norm1, m1, r1, True, wp_c2, blk.conv2_b.t, f, blk.norm2)
        else:
            a = gn_apply(code, c1, blk.norm1.gamma.t, blk.norm1.beta.t, m1, r1, g, blk.norm1._mode, True)
            c2, m2, r2 = conv_gn(code, tdt, a, wp_c2, blk.conv2_b.t, f, blk.norm2)
        if out is None:
            out = torch.empty((n, d, h, w, f), dtype=tdt, device=x.device)
        sp = torch.empty(n * v, dtype=torch.float32, device=x.device)
        if gate is not None:
            torch.cuda.current_stream().wait_stream(gate)       # the epilogue is where the two branches meet (resnet.py:130,137)
        block_epilogue(code, res, c2, out, blk.spatial_k.t.reshape(-1), ch, blk.norm2.gamma.t, blk.norm2.beta.t, m2, r2, g,
                       blk.norm2._mode, sp_out=sp)
        return out, dict(blk=blk, x=x, res=res, c1=c1, m1=m1, r1=r1, a=a, c2=c2, m2=m2, r2=r2, gap=gap, hbuf=hbuf, ch=ch, sp=sp,
                         fold=(dup_start, dup_shift), cin_slab=cin_slab)

    def _block_bwd(self, s, dout, dx, first=False):
        """dout: 16-bit gradient of the block output (dense or a slab-gradient view); dx: 16-bit gradient view of the block input
        to ACCUMULATE into, or None (the input volume).  first: this call is the first writer of dx (uninitialised memory): conv1's
        data gradient writes it, the shortcut's accumulates -- no zero fill, no read of the old values"""
        code = self.code
        blk = s['blk']
        f, g = blk.filters, blk.groups
        n1, n2 = blk.norm1, blk.norm2
        dup_start, dup_shift = s['fold']
        key = id(blk)
        x = s['x']
        cin_slab = s['cin_slab']
        # conv2's weight gradient on the 16-bit kernel?  (f is a multiple of 16 here: LowPrecisionForward refuses other models at construction)
        lp2 = lowp.wgrad_supported(ops.K3S1, f, f)
        # conv1 / shortcut weight gradients on the 16-bit kernel; the first block reads the 2-channel volume zero-padded to one matrix
        # step: its gradients are taken over all 16 stored channels into a scratch tensor and the live rows added to the real slots
        xc = lowp.xdims(x)[4]
        pad_in = cin_slab < xc
        lp1 = lowp.wgrad_supported(ops.K3S1, xc, f)
        # conv branch: GN2 (+ReLU) -> conv2 -> GN1 (+ReLU) -> conv1
        # gate backward and GroupNorm-2 backward both read dout: one pair of passes where the fused kernels' tiling fits
        fused = None
        if lp2 and lp1 and n2._mode == ops.GN_SLAB and self.fuse_block_bwd:
            fused = lowp.block_bwd(code, self.tdt, dout, s['res'], s['c2'], s['sp'], s['gap'], s['hbuf'], s['ch'], blk.se_w1.t, blk.se_w2.t,
                                   blk.spatial_k.t.reshape(-1), n2.gamma.t, n2.beta.t, s['m2'], s['r2'], n2.groups, self._gslot(blk.se_w1),
                                   self._gslot(blk.se_w2), self._gslot(blk.spatial_k).reshape(-1), self._gslot(n2.gamma), self._gslot(n2.beta),
                                   dbias_pt=self._gslot(blk.ptwise_b), dbias_c2=self._gslot(blk.conv2_b))
        if fused is not None:
            dres_fused, dc2_16 = fused
            dc2 = None
            self._db_done = True
        else:
            dc2_16, dc2 = self._gn_bwd(n2, s['c2'], dout, s['m2'], s['r2'], want_f32=not lp2, dbias=self._gslot(blk.conv2_b) if lp2 else None)
        if lp2 and s['a'] is None:      # the forward never wrote relu(GN1(c1)): the weight-gradient kernel forms it from c1 on the way in
            db2 = None if self._db_done else self._gslot(blk.conv2_b)
            c1_, m1_, r1_ = s['c1'], s['m1'], s['r1']
            self._wg((c1_, dc2_16), lambda: lowp.conv_bwd_weight_normed_input(code, c1_, n1, m1_, r1_, dc2_16, self._gslot(blk.conv2_k), db2,
                                                                              accumulate=True))
        elif lp2:
            a16 = s['a']
            db2 = None if self._db_done else self._gslot(blk.conv2_b)
            self._wg((a16, dc2_16), lambda: _wgrad16(ops.K3S1, code, a16, dc2_16, self._gslot(blk.conv2_k), db2, accumulate=True))
        else:
            a32 = self._f32(s['a'])
            self._wg((a32, dc2), lambda: ops.conv_bwd_weight(ops.K3S1, a32, dc2, self._gslot(blk.conv2_k), self._gslot(blk.conv2_b),
                                                             accumulate=True))
        wp_c2b = self._pk((key, 'c2b'), ops.K3S1, blk.conv2_k, f, f, role=ops.ROLE_BWD)
        both = None
        if self.fuse_gn1_bwd and n1._mode == ops.GN_SLAB and dc2_16.shape[-1] == f:
            # conv2's data gradient and GroupNorm-1's backward as one library call: on the layers the z-marching conv takes, the class
            # sums of GroupNorm's backward leave the conv's epilogue (no reduce pass over da and c1)
            both = lowp.conv_bwd_data_gn_bwd(code, self.tdt, dc2_16, wp_c2b, s['c1'], n1.gamma.t, n1.beta.t, s['m1'], s['r1'], self._gslot(n1.gamma),
                                             self._gslot(n1.beta), n1.groups, True, want_f32=not lp1,
                                             dbias=self._gslot(blk.conv1_b) if lp1 else None)
        if both is not None:
            da, dc1_16, dc1, _ = both
            self._db_done = lp1
            del dc2, dc2_16
        else:
            da = torch.empty_like(s['c1'])
            conv_bwd_data(ops.K3S1, code, dc2_16, wp_c2b, da, False)
            del dc2, dc2_16
            dc1_16, dc1 = self._gn_bwd(n1, s['c1'], da, s['m1'], s['r1'], want_f32=not lp1, dbias=self._gslot(blk.conv1_b) if lp1 else None)
        db1 = None if (lp1 and self._db_done) else self._gslot(blk.conv1_b)
        del da
        # gate branch (16-bit kernels; fp32 copies only where a weight gradient still runs on the fp32 kernels)
        if fused is not None:
            dres_16 = dres_fused
        else:
            dres_16 = lowp.se_bwd(code, self.tdt, dout, s['res'], s['sp'], s['gap'], s['hbuf'], s['ch'], blk.se_w1.t, blk.se_w2.t,
                                  blk.spatial_k.t.reshape(-1), self._gslot(blk.se_w1), self._gslot(blk.se_w2),
                                  self._gslot(blk.spatial_k).reshape(-1), dbias=self._gslot(blk.ptwise_b) if lp1 else None)
        dres = None if lp1 else self._f32(dres_16)
        # weight gradients of the two convolutions that read the block input
        if lp1 and pad_in:
            def wgrads():
                tk = torch.empty((3, 3, 3, x.shape[-1], f), dtype=torch.float32, device=x.device)
                tp = torch.empty((1, 1, 1, x.shape[-1], f), dtype=torch.float32, device=x.device)
                # (both from one pass over x where the streaming kernel takes the layer: see the branch below)
                if not lowp.conv_bwd_weight_pair(code, x, dc1_16, dres_16, tk, tp, db1, 0, 0, False):
                    _wgrad16(ops.K3S1, code, x, dc1_16, tk, db1, 0, 0, False)
                    _wgrad16(ops.K1, code, x, dres_16, tp, None, 0, 0, False)
                # live rows of the padded gradients into the real slots (library kernel: rows = taps, columns = the cin_slab x f prefix)
                ops.add_strided(self._gslot(blk.conv1_k).view(27, cin_slab * f), tk.view(27, -1)[:, :cin_slab * f], True)
                ops.add_strided(self._gslot(blk.ptwise_k).view(1, cin_slab * f), tp.view(1, -1)[:, :cin_slab * f], True)
            self._wg((x, dc1_16, dres_16), wgrads)
        elif lp1:
            def wgrads():
                # conv1's and the shortcut's weight gradients from ONE pass over the block input (round 6: the 1x1x1 gradient is one more
                # accumulator of the streaming 3x3x3 kernel; bts_lp_conv3d_bwd_weight_pair) where that kernel takes the layer
                if lowp.conv_bwd_weight_pair(code, x, dc1_16, dres_16, self._gslot(blk.conv1_k), self._gslot(blk.ptwise_k), db1, dup_start, dup_shift, True):
                    return
                if lowp.is_split(x):
                    raise RuntimeError('split concat input: the paired weight-gradient launch declined a shape its query accepted')
                _wgrad16(ops.K3S1, code, x, dc1_16, self._gslot(blk.conv1_k), db1, dup_start, dup_shift, True)
                _wgrad16(ops.K1, code, x, dres_16, self._gslot(blk.ptwise_k), None, dup_start, dup_shift, True)   # (bias: se_bwd)
            self._wg((x, dc1_16, dres_16), wgrads)
        else:   # fp32 kernels on the widened input view
            x32 = self._f32(x[..., :cin_slab])

            def wgrads():
                ops.conv_bwd_weight(ops.K3S1, x32, dc1, self._gslot(blk.conv1_k), self._gslot(blk.conv1_b), dup_start, dup_shift, accumulate=True)
                ops.conv_bwd_weight(ops.K1, x32, dres, self._gslot(blk.ptwise_k), self._gslot(blk.ptwise_b), dup_start, dup_shift, accumulate=True)
            self._wg((x32, dc1, dres), wgrads)
        if dx is not None:
            cin = xc
            assert dx.shape[-1] == cin or (dx.dim() == 6 and dx.shape[0] * 32 == cin)
            wpb1 = self._pk((key, 'c1b'), ops.K3S1, blk.conv1_k, blk.cin_ref, f, cin, dup_start, dup_shift, role=ops.ROLE_BWD)
            wpbp = self._pk((key, 'ptb'), ops.K1, blk.ptwise_k, blk.cin_ref, f, cin, dup_start, dup_shift, role=ops.ROLE_BWD)
            # conv1's and the shortcut's data gradients in ONE launch (round 6): the shortcut is an extra K-segment at the centre tap of
            # conv1's data-gradient kernel -- no second launch that read-modify-writes the Cin-wide dx (bts_lp_conv3d_bwd_data_sc; shapes the
            # fused kernels decline run as the two launches inside the same call)
            if dx.dim() == 6:           # the concat's gradient as dense 32-channel tensors (see step(): gsplit)
                assert first and dc1_16.shape == dres_16.shape
                if not lowp.conv_bwd_data_sc(code, dc1_16, wpb1, dres_16, wpbp, dx, False):
                    raise RuntimeError('split data gradient: the fused launch declined a shape its query accepted')
            elif dc1_16.shape == dres_16.shape:
                lowp.conv_bwd_data_sc(code, dc1_16, wpb1, dres_16, wpbp, dx, not first)
            else:
                conv_bwd_data(ops.K3S1, code, dc1_16, wpb1, dx, not first)
                conv_bwd_data(ops.K1, code, dres_16, wpbp, dx, True)
        self._written(blk.trainable_variables)

    def _sampler_fwd(self, lay, kind, x, out=None):
        if isinstance(lay, self.fwd._max):                   # MaxPooling3D(2) (downsample.py:51-70): no parameters, channels kept
            y, idx = lowp.maxpool2(self.code, x)
            return y, dict(lay=lay, kind='max', idx=idx)
        if isinstance(lay, self.fwd._linear):                # 1x1x1 conv, then nearest-neighbour repeat (upsample.py:49-79)
            wp = self._pk((id(lay), 'f'), ops.K1, lay.ptwise_k, lay.cin, lay.filters)
            c = conv(ops.K1, self.code, self.tdt, x, wp, lay.ptwise_b.t, lay.filters)
            y = lowp.upsample2(self.code, c, out=out)
            return y, dict(lay=lay, kind='linear', x=x)
        wp = self._pk((id(lay), 'f'), kind, lay.conv_k, lay.cin, lay.filters)
        if kind == ops.K3S2T:       # transposed conv with the statistics of its (fine) output from its epilogue
            c, m, r = lowp.convT_gn(self.code, self.tdt, x, wp, lay.conv_b.t, lay.filters, lay.norm)
        else:
            c = conv(kind, self.code, self.tdt, x, wp, lay.conv_b.t, lay.filters)
            m, r = gn_stats(self.code, c, lay.norm.groups, lay.norm._mode, lay.norm.epsilon)
        y = gn_apply(self.code, c, lay.norm.gamma.t, lay.norm.beta.t, m, r, lay.norm.groups, lay.norm._mode, True, out=out)
        return y, dict(lay=lay, kind=kind, x=x, c=c, m=m, r=r)

    def _sampler_bwd(self, s, dy, dx, accumulate, cin_live=None):
        """ConvDownsample / ConvUpsample backward (downsample.py:41-45, upsample.py:39-43): GN(+ReLU) gradient, weight gradient,
        data gradient into dx (None: not needed).  cin_live: real input channels when the input was zero-padded to 16"""
        lay, kind = s['lay'], s['kind']
        if kind == 'max':
            if dx is not None:
                lowp.maxpool2_bwd(self.code, dy, s['idx'], dx, accumulate)
            return self._written([])
        if kind == 'linear':
            self._linear_bwd(s, dy, dx, accumulate, cin_live)
            return self._written(lay.trainable_variables)
        nrm = lay.norm
        x = s['x'] if cin_live is None else s['x'][..., :cin_live]
        cout = s['c'].shape[-1]
        lp16 = cin_live is None and cout % 16 == 0 and lowp.wgrad_supported(kind, x.shape[-1], cout) and \
            not (kind == ops.K3S2 and any(v & 1 for v in x.shape[1:4]))
        dc16, dc = self._gn_bwd(nrm, s['c'], dy, s['m'], s['r'], want_f32=not lp16, dbias=self._gslot(lay.conv_b) if lp16 else None)
        if lp16 and dc16.is_contiguous():
            # 16-bit operands straight into the transposing-read weight-gradient kernel (no widened copies)
            dbs = None if self._db_done else self._gslot(lay.conv_b)
            self._wg((x, dc16), lambda: _wgrad16(kind, self.code, x, dc16, self._gslot(lay.conv_k), dbs, accumulate=True))
        else:
            x32 = self._f32(x)
            if dc is None:
                dc = self._f32(dc16)

            def wgrads():
                if kind == ops.K3S2T:
                    ops.conv_bwd_weight(kind, x32, dc, self._gslot(lay.conv_k), None, accumulate=True)
                    ops.colsum(dc, sum_over_n=True, out=self._gslot(lay.conv_b), accumulate=True)
                else:
                    ops.conv_bwd_weight(kind, x32, dc, self._gslot(lay.conv_k), self._gslot(lay.conv_b), accumulate=True)
            self._wg((x32, dc), wgrads)
        if dx is not None:
            wpb = self._pk((id(lay), 'b'), kind, lay.conv_k, lay.cin, lay.filters, role=ops.ROLE_BWD)
            conv_bwd_data(kind, self.code, dc16, wpb, dx, accumulate)
        self._written(lay.trainable_variables)

    def _linear_bwd(self, s, dy, dx, accumulate, cin_live):
        """LinearUpsample backward (upsample.py:49-79 under autodiff): the repeat's gradient (sums of 8 fine voxels, fp32 sums),
        then the 1x1x1 conv's weight, bias and data gradients"""
        lay = s['lay']
        f = lay.filters
        fp = (f + 15) // 16 * 16                              # (the contraction of the data gradient steps over 16 channels)
        x = s['x'] if cin_live is None else s['x'][..., :cin_live]
        n, d2, h2, w2 = dy.shape[:4]
        buf = (torch.zeros if fp != f else torch.empty)((n, d2 // 2, h2 // 2, w2 // 2, fp), dtype=self.tdt, device=dy.device)
        dc16 = lowp.upsample2_bwd(self.code, dy, dx=buf[..., :f])
        if cin_live is None and fp == f and lowp.wgrad_supported(ops.K1, x.shape[-1], f):
            self._wg((x, dc16), lambda: _wgrad16(ops.K1, self.code, x, dc16, self._gslot(lay.ptwise_k), self._gslot(lay.ptwise_b),
                                                             accumulate=True))
        else:
            x32, dc = self._f32(x), self._f32(dc16)
            self._wg((x32, dc), lambda: ops.conv_bwd_weight(ops.K1, x32, dc, self._gslot(lay.ptwise_k), self._gslot(lay.ptwise_b),
                                                            accumulate=True))
        if dx is not None:
            wpb = self._pk((id(lay), 'b'), ops.K1, lay.ptwise_k, lay.cin, f, role=ops.ROLE_BWD)
            conv_bwd_data(ops.K1, self.code, buf, wpb, dx, accumulate)

    # ================================================================================================================
    # the step
    # ================================================================================================================
    def step(self, optimizer, dice_fn, x, y):
        """train.py:140-152 with 16-bit storage -> (loss, macro_dice, micro_dice) as 1-element Tensors"""
        m = self.model
        code, tdt = self.code, self.tdt
        enc, dec, vae = m.encoder, m.decoder, m.vae
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        if not torch.is_tensor(y):
            y = torch.as_tensor(y)
        dev = torch.device('cuda', torch.cuda.current_device())
        fence = ops.step_fence('train')          # at most two steps in flight (see ops.step_fence)
        self._clock = None
        x, y = x.to(dev).float(), y.to(dev).float()
        cf = m.data_format == 'channels_first'
        if cf:          # raw NCDHW volumes -> the engine's NDHWC memory (tape.as_tensor does the same for the fp32 step)
            x, y = x.permute(0, 2, 3, 4, 1), y.permute(0, 2, 3, 4, 1)
        x, y = x.contiguous(), y.contiguous()
        n = x.shape[0]
        ops.fill(m.flat_grads, 0.0)
        # ------------------------------------------------ forward ------------------------------------------------
        xin = x
        cur = None
        if enc.dropout_rate > 0:                                                      # encoder.py:39,71
            if enc._mask is not None:
                msk = torch.as_tensor(enc._mask)
                if cf and msk.dim() == 5:                                             # (an injected mask is in the public layout)
                    msk = msk.permute(0, 2, 3, 4, 1)
                msk = (msk != 0).to(torch.uint8).to(dev).contiguous()
                enc._mask = None
                xin = ops.dropout_apply(x, msk, enc.dropout_rate)
            else:
                enc._seed += 1
                if x.shape[-1] <= 4:      # the draw, the scaling and the cast into the 16-channel matrix step in one pass (same generator, same seed)
                    cur = lowp.dropout_cast_pad16(code, tdt, x, enc.dropout_rate, enc._seed)
                else:
                    xin = ops.dropout_apply(x, ops.dropout_mask(x.shape, enc.dropout_rate, enc._seed, dev), enc.dropout_rate)
        if cur is not None:
            pass
        elif x.shape[-1] <= 4 and xin.is_contiguous():      # in_ch = 2 (model.py:18): one pass writes the whole 16-channel matrix step
            cur = lowp.cast_pad16(code, tdt, xin)
        else:
            cpad = (x.shape[-1] + 15) // 16 * 16
            cur = torch.zeros(tuple(x.shape[:4]) + (cpad,), dtype=tdt, device=dev)
            cast(code, tdt, xin, out=cur[..., :x.shape[-1]])
        del xin
        levels = []                     # per encoder level: (slab, used, [block saves], down save or None)
        for i, (convs, down) in enumerate(enc.levels):
            d, h, w = cur.shape[1:4]
            f = enc.base_filters * 2 ** i
            nb = len(convs)
            spare = f if i < enc.depth - 1 else 0
            # Level 0 of the CLI model: [o_0 (32) | up-sampled (32)] (decoder.py:75) as TWO dense tensors instead of a 64-wide slab (SURVEY K13:
            # virtual concat as a list of segments) where every reader of the pair takes the list -- conv1 + shortcut of the decoder's top
            # block ride on two z-marching passes, one per operand (bts_lp_conv3d_fwd_gn_shortcut, x_split), their weight gradients read
            # one 32-channel block of P per workgroup anyway (bts_lp_conv3d_bwd_weight_pair, x_split), the data gradient leaves split as well
            # (gsplit below).  The readers of ONE operand (the down-sampler, the skip level's kernels) then fetch whole 128-byte lines.
            split0 = i == 0 and self._level0_split_ok(n, d, h, w, nb, f, spare, dec)
            if split0:
                slab = torch.empty((2, n, d, h, w, 32), dtype=tdt, device=dev)
            else:
                slab = torch.empty((n, d, h, w, nb * f + spare), dtype=tdt, device=dev)
            saves = []
            for j, blk in enumerate(convs):
                out = slab[0] if split0 else slab[..., j * f:(j + 1) * f]
                if j == 0:
                    _, sv = self._block_fwd(blk, cur, out)
                else:
                    _, sv = self._block_fwd(blk, slab[..., :j * f], out, fold=((j - 1) * f, f))
                saves.append(sv)
            dsave = None
            if down is not None:
                cur, dsave = self._sampler_fwd(down, ops.K3S2, slab[0] if split0 else slab[..., :nb * f])
            levels.append((slab, nb * f, saves, dsave))
        top_slab, top_used = levels[-1][0], levels[-1][1]
        top = top_slab[..., :top_used]
        # decoder (decoder.py:65-83)
        yk = top
        dsaves = []
        for k, (up, blk) in enumerate(dec.levels):
            li = len(levels) - 2 - k
            slab, cres = levels[li][0], levels[li][1]
            f = up.filters
            if lowp.is_split(slab):       # (level 0 as two dense operands: see above)
                _, us = self._sampler_fwd(up, ops.K3S2T, yk, out=slab[1])
                yk, bs = self._block_fwd(blk, slab, None)
            else:
                _, us = self._sampler_fwd(up, ops.K3S2T, yk, out=slab[..., cres:cres + f])
                yk, bs = self._block_fwd(blk, slab[..., :cres + f], None)
            dsaves.append((us, bs, li, cres, f))
        y_last = yk
        y_pred = head(code, y_last, dec.out_k.t.reshape(dec.out_k.t.shape[-2], dec.out_k.t.shape[-1]), dec.out_b.t, True)
        # VAE branch (vae.py:114-143)
        hdn, vds = self._sampler_fwd(vae.downsample, ops.K3S2, top)
        flat = self._f32(hdn).reshape(n, -1)
        proj = ops.dense_fwd(flat, vae.proj_k.t, vae.proj_b.t, False)
        L = vae.latent_size
        if vae._eps is not None:
            eps = torch.as_tensor(vae._eps, dtype=torch.float32).to(dev).contiguous()
            vae._eps = None
        else:
            vae._seed += 1
            eps = ops.normal((n, L), vae._seed, dev)
        z = ops.vae_sample_fwd(proj, eps)
        u = ops.dense_fwd(z, vae.unproj_k.t, vae.unproj_b.t, True)
        u5 = u.reshape((n,) + tuple(vae._unflat))
        u16 = lowp.cast_pad16(code, tdt, u5.contiguous())      # 1 channel, zero-padded to a matrix step
        yv, vus = self._sampler_fwd(vae.upsample, ops.K3S2T, u16)
        vsaves = []
        for up, blk in vae.levels:
            yv, us = self._sampler_fwd(up, ops.K3S2T, yv)
            yv, bs = self._block_fwd(blk, yv, None)
            vsaves.append((us, bs))
        yv_last = yv
        if vae.out_ch < 8 and yv_last.shape[-1] % 16 == 0:
            # out_ch = in_ch = 2 (vae.py:92-99): fewer than the 8 couts a 16-byte store carries, which left this 128^3 layer to the
            # register-staged kernel (0.97 ms of the batch-8 step).  Kernel and bias zero-padded to 8 output channels -> the streaming
            # kernels take it (the pad columns multiply zeros; only the live ones are read back)
            cvv = yv_last.shape[-1]
            wpad = self._padded((id(vae), 'out_k8'), vae.out_k, 8)
            bpad = self._padded((id(vae), 'out_b8'), vae.out_b, 8)
            y8 = conv(ops.K3S1, code, tdt, yv_last, self._pk((id(vae), 'out8'), ops.K3S1, wpad, cvv, 8), bpad.t, 8)
            y_vae = self._f32(y8[..., :vae.out_ch])
            del y8
        else:
            wp_vo = self._pk((id(vae), 'out'), ops.K3S1, vae.out_k, yv_last.shape[-1], vae.out_ch)
            y_vae = self._f32(conv(ops.K3S1, code, tdt, yv_last, wp_vo, vae.out_b.t, vae.out_ch))
        # ------------------------------------------------ loss, metric (fp32: util.py:13-24,35-57, train.py:145-148) -------------
        c = y_pred.shape[-1]
        sums = ops.loss_sums(y_pred, y, x, y_vae, proj)
        parallel.all_reduce_sum(sums)
        lt, _ = ops.loss_value(sums, c, True)
        l2v = ops.l2_reg_fwd(m.flat_params, m._l2_ranges) if m._l2_ranges else None
        loss_t = ops.scalar_lincomb(lt, l2v, 1.0, 1.0) if l2v is not None else lt
        macro, micro = dice_fn(Tensor(y, requires_grad=False), Tensor(y_pred, requires_grad=False))     # (engine layout already)
        self.last_labels = dice_fn.last_labels
        # ------------------------------------------------ backward ------------------------------------------------
        # `one` seeds the backward: d loss / d loss, times the loss scale (float16: see the class docstring)
        one = torch.full((1,), self.loss_scale, dtype=torch.float32, device=dev)
        dyp = torch.empty_like(y_pred)
        dyv = torch.empty_like(y_vae)
        dproj = torch.empty_like(proj)
        # (through the decoder's sigmoid, decoder.py:60, in the same pass: dyp IS the gradient of the head's pre-activation)
        ops.loss_bwd(y_pred, y, x, y_vae, proj, sums, one, dyp, dyv, dproj, through_sigmoid=True)
        # slab gradients: uninitialised -- the first writer of each one writes, every later contribution accumulates.  Levels below the
        # top: the decoder block's conv1 data gradient (its view [0, cres + f) is the whole slab); the top level: the VAE's
        # down-sampling conv's data gradient (its view [0, top_used) is the whole slab, which has no spare channels)
        assert levels[-1][0].shape[-1] == top_used
        # Level 0 of the CLI model: the slab is [o_0 (32) | up-sampled (32)] and its gradient has two readers that each want ONE half -- the
        # encoder block's backward (dout = [0, 32)) and the up-sampler's GroupNorm backward (dy = [32, 64)) -- twice each (reduce + apply
        # pass): as channel slices of a 64-wide slab they fetch 64-byte halves of 128-byte lines.  Where the fused data-gradient launch can
        # write its 64 columns as two dense tensors, the gradient "slab" is a (2, N, D, H, W, 32) buffer instead (round 6)
        gsplit = None
        if len(levels) > 1 and dsaves:
            slab0, used0 = levels[0][0], levels[0][1]
            f0 = dsaves[-1][4]
            blk0 = dsaves[-1][1]['blk']
            n0, d0, h0, w0, width0 = lowp.xdims(slab0)[:5]
            if used0 == 32 and f0 == 32 and width0 == 64 and len(levels[0][2]) == 1 and \
                    lowp.conv_bwd_data_sc_split_ok(n, d0, h0, w0, 64, blk0.filters):
                gsplit = torch.empty((2, n, d0, h0, w0, 32), dtype=tdt, device=dev)
            assert gsplit is not None or not lowp.is_split(slab0)      # (_level0_split_ok asked the same question)
        gslabs = [None if (i == 0 and gsplit is not None) else torch.empty_like(lv[0]) for i, lv in enumerate(levels)]

        def gview(i, c0, c1):
            """channels [c0, c1) of level i's slab gradient"""
            if i == 0 and gsplit is not None:
                if c0 % 32 == 0 and c1 == c0 + 32:
                    return gsplit[c0 // 32]
                assert (c0, c1) == (0, 64)
                return gsplit
            return gslabs[i][..., c0:c1]
        # The backward walks vae -> decoder -> encoder level 3 .. 0: the order of the model's flat gradient buffer (model._backward_groups),
        # so that with a process group finished buckets are all-reduced from inside the backward pass, as the fp32
        # tape does (parallel.GradSync); the regulariser term goes into each bucket just before its exchange
        sync = parallel.grad_sync(m)
        if sync is not None:
            self._clock = _Progress(sync, self._backward_stages(len(vsaves), len(dsaves), [len(lv[2]) for lv in levels]))
            sync.begin(self._clock, l2_grad=one if l2v is not None else None, prefilled=True)
        # VAE branch backward.  Its output conv has out_ch = in_ch = 2 channels (vae.py:92-99): dy is stored zero-padded to one matrix step
        # (16 channels, like the input volume) so that the weight gradient (padded columns dropped afterwards) and the data gradient
        # (role-swapped image of the zero-padded kernel) run on the 16-bit kernels instead of the fp32 ones over widened copies
        cv, co = yv_last.shape[-1], vae.out_ch
        if cv % 16 == 0 and co <= 16 and lowp.wgrad_supported(ops.K3S1, cv, 16):
            if co <= 4 and dyv.is_contiguous():
                dyv16 = lowp.cast_pad16(code, tdt, dyv)
            else:
                dyv16 = torch.zeros(tuple(dyv.shape[:4]) + (16,), dtype=tdt, device=dev)
                cast(code, tdt, dyv, out=dyv16[..., :co])

            def wg_out():
                tk = torch.empty((3, 3, 3, cv, 16), dtype=torch.float32, device=dev)
                _wgrad16(ops.K3S1, code, yv_last, dyv16, tk, None, 0, 0, False)
                ops.add_strided(self._gslot(vae.out_k).view(-1, co), tk.view(-1, 16)[:, :co], True)      # the co live columns of the padded gradient
                ops.colsum(dyv, sum_over_n=True, out=self._gslot(vae.out_b), accumulate=True)
            self._wg((yv_last, dyv16, dyv), wg_out)
            wpb = self._pk((id(vae), 'out16b'), ops.K3S1, self._padded((id(vae), 'out_k16'), vae.out_k, 16), cv, 16, role=ops.ROLE_BWD)
            dv = torch.empty(yv_last.shape, dtype=tdt, device=dev)
            conv_bwd_data(ops.K3S1, code, dyv16, wpb, dv, False)
            del dyv16
        else:
            ylv32 = self._f32(yv_last)
            self._wg((ylv32, dyv), lambda: ops.conv_bwd_weight(ops.K3S1, ylv32, dyv, self._gslot(vae.out_k), self._gslot(vae.out_b),
                                                               accumulate=True))
            dv32 = torch.empty_like(ylv32)
            wpb = vae.packed('out_b', ops.K3S1, ops.ROLE_BWD, vae.out_k, yv_last.shape[-1], vae.out_ch)
            ops.conv_bwd_data(ops.K3S1, dyv, wpb, dv32, False)
            dv = self._b16(dv32)
            del dv32, ylv32
        self._written([vae.out_k, vae.out_b])
        for us, bs in reversed(vsaves):
            dblk_in = torch.empty(bs['x'].shape, dtype=tdt, device=dev)
            self._block_bwd(bs, dv, dblk_in, first=True)
            dv = torch.empty(us['x'].shape, dtype=tdt, device=dev)
            self._sampler_bwd(us, dblk_in, dv, False)
        du16 = torch.zeros_like(u16)
        self._sampler_bwd(vus, dv, du16, False, cin_live=1)
        du5 = self._f32(du16[..., :1]).reshape(u.shape)
        gz = ops.relu_bwd(u, du5)
        dz = torch.empty_like(z)
        ops.dense_bwd(z, vae.unproj_k.t, gz, dz, self._gslot(vae.unproj_k), self._gslot(vae.unproj_b), accumulate_dx=False,
                      accumulate_params=True)
        ops.vae_sample_bwd(proj, eps, dz, dproj)
        dflat = torch.empty_like(flat)
        ops.dense_bwd(flat, vae.proj_k.t, dproj, dflat, self._gslot(vae.proj_k), self._gslot(vae.proj_b), accumulate_dx=False,
                      accumulate_params=True)
        self._written([vae.unproj_k, vae.unproj_b, vae.proj_k, vae.proj_b])
        dhdn = self._b16(dflat.reshape(hdn.shape))
        self._sampler_bwd(vds, dhdn, gslabs[-1][..., :top_used], False)           # first writer of the top level's slab gradient
        # decoder head (decoder.py:55-63): sigmoid, 1x1x1 conv to out_ch -- dx, dW and db from one pass over the 16-bit activations
        dpre = dyp
        wk2 = dec.out_k.t.reshape(dec.out_k.t.shape[-2], dec.out_k.t.shape[-1])
        dcur = lowp.head_bwd(code, tdt, y_last, dpre, wk2, self._gslot(dec.out_k).reshape(wk2.shape), self._gslot(dec.out_b), True)
        if dcur is None:       # head outside the fused kernel's shapes: fp32 kernels on widened copies
            ylast32 = self._f32(y_last)
            self._wg((ylast32, dpre), lambda: ops.conv_bwd_weight(ops.K1, ylast32, dpre, self._gslot(dec.out_k), self._gslot(dec.out_b),
                                                                 accumulate=True))
            dlast32 = torch.empty_like(ylast32)
            wpb = dec.packed('out_b', ops.K1, ops.ROLE_BWD, dec.out_k, y_last.shape[-1], dec.out_ch)
            ops.conv_bwd_data(ops.K1, dpre, wpb, dlast32, False)
            dcur = self._b16(dlast32)
            del dlast32, ylast32
        del dpre
        self._written([dec.out_k, dec.out_b])
        for idx in range(len(dsaves) - 1, -1, -1):
            us, bs, li, cres, f = dsaves[idx]
            assert lowp.xdims(levels[li][0])[4] == cres + f
            self._block_bwd(bs, dcur, gview(li, 0, cres + f), first=True)    # skip part [0, cres) and the up-sampled part [cres, cres + f)
            if idx == 0:                                         # the first up layer read the top level's slab view
                self._sampler_bwd(us, gview(li, cres, cres + f), gslabs[-1][..., :top_used], True)
            else:                                                # the others read the previous decoder block's output
                dcur = torch.empty(us['x'].shape, dtype=tdt, device=dev)
                self._sampler_bwd(us, gview(li, cres, cres + f), dcur, False)
        # encoder backward (encoder.py:69-101 in reverse)
        for i in range(len(levels) - 1, -1, -1):
            slab, used, saves, dsave = levels[i]
            f = enc.base_filters * 2 ** i
            for j in range(len(saves) - 1, -1, -1):
                dout = gview(i, j * f, (j + 1) * f)
                if j > 0:
                    self._block_bwd(saves[j], dout, gview(i, 0, j * f))
                elif i > 0:
                    dprev = torch.empty(saves[0]['x'].shape, dtype=tdt, device=dev)
                    self._block_bwd(saves[0], dout, dprev, first=True)
                    pslab, pused, _, pds = levels[i - 1]
                    self._sampler_bwd(pds, dprev, gview(i - 1, 0, pused), True)
                else:
                    self._block_bwd(saves[0], dout, None)
        # regulariser (train.py:146), exchange, optimiser (train.py:151-152)
        ops.join_side_stream()
        if sync is not None:
            self._clock = None
            sync.finish()              # parameters no stage reported (none in this graph) and pad-only buckets; waits for the handles
            scale = 1.0
        else:
            if l2v is not None:
                k = parallel.l2_grad_scale()
                ops.l2_reg_bwd(m.flat_params, m.flat_grads, [(o, ln, cf * k) for o, ln, cf in m._l2_ranges], one)
            scale = parallel.all_reduce_gradients(m)
        grads = [p._gview for p in m.trainable_variables]
        if self.dynamic_scale:
            # after the exchange every rank holds the same summed gradient, so every rank's flag -- and later decision -- is the same
            if self._flags is None:
                self._flags = [(torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32).pin_memory()) for _ in range(2)]
                self._flag_turn = 0
            flag, mirror = self._flags[self._flag_turn]
            self._flag_turn ^= 1
            used = self.last_grad_scale = self.loss_scale          # the scale this step's backward was seeded with
            ops.grad_nonfinite(m.flat_grads, flag)
            self.settle()                               # the PREVIOUS step's decision (its copy is long done): may change loss_scale / iterations
            mirror.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, mirror, optimizer, used)
            # un-scaling rides on Adam's gradient read (1 / scale: a power of two, exact); the update is dropped on the device on overflow
            optimizer.apply_gradients(zip(grads, m.trainable_variables), model=m, grad_scale=scale / used, skip_flag=flag)
            ops.step_fence_done(fence)
            return Tensor(loss_t, requires_grad=False), macro, micro
        optimizer.apply_gradients(zip(grads, m.trainable_variables), model=m, grad_scale=scale)
        ops.step_fence_done(fence)
        return Tensor(loss_t, requires_grad=False), macro, micro

    def settle(self):
        """read the overflow flag of the last dynamic-scale step (if it has not been read yet) and apply its consequences: skipped
        steps are counted, the optimiser's step counter taken back, the loss scale halved / doubled.  Called by step() for the step
        before, and by whoever reads loss_scale / skipped_steps / the optimiser state (checkpoints, tests)."""
        if self._pending is None:
            return
        ev, mirror, optimizer, used = self._pending
        self._pending = None
        ev.synchronize()
        if int(mirror[0]) == 0:
            self._consecutive_skips = 0
            self._clean_steps += 1
            if self._clean_steps >= self.growth_interval and self.loss_scale < 2.0 ** 24:
                self.loss_scale *= 2.0
                self._clean_steps = 0
            return
        self.skipped_steps += 1
        self._consecutive_skips += 1
        self._clean_steps = 0
        optimizer.iterations -= 1                       # the device dropped that update: Adam's bias correction must not count it
        if used <= 1.0 or self._consecutive_skips >= 8:
            # nothing left to halve (or halving does not help): the gradient itself is not finite -- a NaN in the data or the weights.
            # Training would otherwise stall silently, every later step skipped.
            import warnings
            warnings.warn('LowPrecisionTrainer: step skipped on a non-finite gradient with loss_scale %g (%d in a row, %d in total); '
                          'the weights or the data hold Inf / NaN' % (used, self._consecutive_skips, self.skipped_steps), RuntimeWarning)
        self.loss_scale = max(1.0, min(self.loss_scale, used * 0.5))
