"""where the 16-bit forward leaves the fp32 engine, level by level of the encoder (debug aid)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bts_amd  # noqa
import test_lowp_train_gpu as T
from bts_amd.lowp import LowPrecisionForward, cast
for opts in (dict(data_format='channels_first', downsampling='max'), dict(downsampling='max'), dict(data_format='channels_first')):
    cf = opts.get('data_format') == 'channels_first'
    m, x, y, mask, eps = T._setup(seed=5, **opts)
    xin = T._public(x, cf)
    from bts_amd.tape import as_tensor
    res = m.encoder(as_tensor(xin, data_format=m.data_format), training=False)
    for dt in ('float16', 'bfloat16'):
        F = LowPrecisionForward(m, dt)
        xx = x.cuda().float().contiguous()
        cur = torch.zeros(tuple(xx.shape[:4]) + (16,), dtype=F.tdt, device='cuda')
        cast(F.code, F.tdt, xx, out=cur[..., :2])
        enc = m.encoder
        line = []
        for i, (convs, down) in enumerate(enc.levels):
            n, d, h, w = cur.shape[:4]
            f = enc.base_filters * 2 ** i
            nb = len(convs)
            slab = torch.empty((n, d, h, w, nb * f), dtype=F.tdt, device='cuda')
            for j, blk in enumerate(convs):
                out = slab[..., j * f:(j + 1) * f]
                if j == 0:
                    F._block(blk, cur, out)
                else:
                    F._block(blk, slab[..., :j * f], out, fold=((j - 1) * f, f))
                r = res[i].t[..., j * f:(j + 1) * f]
                line.append('L%dB%d %.2e/%.2e' % (i, j, float((out.float() - r).abs().max()), float(r.abs().max())))
            if down is not None:
                cur = F._down(down, slab)
        print(opts, dt, ' '.join(line))
