"""split (two dense operands) vs slab route of the 16-bit inference forward at batch 2 / channels_first / odd-but-legal sizes (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bts_amd  # noqa: F401,E402
from bts_amd import lowp  # noqa: E402
from bts_amd.layers import _base  # noqa: E402
from bts_amd.model import Model  # noqa: E402
from bts_amd.tape import bump_weights_epoch  # noqa: E402

dev = torch.device('cuda', 0)
for fmt, n, crop in (('channels_last', 2, (40, 16, 96)), ('channels_first', 1, (48, 32, 64)), ('channels_last', 3, (24, 16, 32)), ('channels_last', 1, (64, 64, 64))):
    _base.set_seed(3)
    m = Model(base_filters=32, groups=8, reduction=4, depth=3, data_format=fmt)
    m.build((1,) + crop + (2,))      # (the build shape is the internal NDHWC one)
    shape = (1,) + crop + (2,) if fmt == 'channels_last' else (1, 2) + crop
    g = torch.Generator().manual_seed(4)
    for p in m.trainable_variables:
        if p.name.endswith('gamma'):
            p.t.copy_((1.0 + 0.3 * torch.randn(p.t.shape, generator=g)).to(p.t.device))
    bump_weights_epoch()
    x = torch.randn((n,) + shape[1:], generator=g).to(dev)
    outs = {}
    for sw in ('1', '0'):
        os.environ['BTS_LP_INF_SPLIT'] = sw
        eng = lowp.LowPrecisionForward(m, 'float16')
        outs[sw] = eng(x)
        torch.cuda.synchronize()
    y32 = m(x, training=False, inference=True)[0].public()
    d = (outs['1'] - outs['0']).abs().max().item()
    e1 = (outs['1'] - y32).abs().max().item()
    e0 = (outs['0'] - y32).abs().max().item()
    print(fmt, n, crop, 'split vs slab %.2e ; vs fp32: split %.2e slab %.2e' % (d, e1, e0))
    assert d < 5e-3 and e1 < 2e-2 and e0 < 2e-2
print('ok')
