import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bts_amd
from bts_amd import lowp, ops
from bts_amd.model import Model
model = Model(base_filters=32, reduction=8, depth=4, groups=8)
model.build((1, 128, 128, 128, 2))
x = torch.randn((1, 160, 192, 160, 2)).cuda()
run = lowp.LowPrecisionForward(model, 'float16')
for _ in range(2): run(x)
torch.cuda.synchronize()
ops.profile_enable(True)
run(x)
torch.cuda.synchronize()
ops.profile_enable(False)
for sym, fl, ms in ops.profile_records():
    print('%-24s %8.2f GF %8.1f us %7.1f TF' % (sym, fl / 1e9, ms * 1e3, fl / ms / 1e9))
