#!/usr/bin/env python3
"""Per-CU timeline of wino_kernel on one layer (experiment build: `make -C 3d-brain-tumor-segmentation_amd/csrc stamps`, which
gpurun must carry: the .so lives under csrc/build/).  usage: wino_timeline.py D Cin Cout   (env BTS_WINO_T as usual)
Stamps per item (wave 0, 100 MHz wall clock): 0 top, 1 after commit + barrier, 2 after the first stage, 3 after the last
stage, 4 after the next item's requests were issued, 5 after the stores."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, '3d-brain-tumor-segmentation_amd', 'csrc', 'build', 'libbts_hip_stamps.so')
os.environ['BTS_HIP_LIB'] = LIB
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bts_amd  # noqa: E402,F401
from bts_amd import ops  # noqa: E402
from bts_amd._lib import lib  # noqa: E402

d, cin, cout = (int(v) for v in sys.argv[1:4])
D = torch.device('cuda:0')
x = torch.randn((1, d, d, d, cin), device=D)
w = torch.randn((3, 3, 3, cin, cout), device=D) * 0.1
b = torch.randn(cout, device=D)
wp = ops.conv_pack(1, ops.ROLE_FWD, w, cin, cout)
y = ops.conv_fwd(1, x, wp, b, cout)
for _ in range(3):
    ops.conv_fwd(1, x, wp, b, cout, out=y)
torch.cuda.synchronize()
n = 1 << 20
buf = np.zeros(n, dtype=np.int64)
fn = lib().cdll.bts_wino_stamps_copy_
fn.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert fn(buf.ctypes.data, n) == 0
st = buf.reshape(-1, 16)
st = st[st[:, 0] > 0]
hw = st[:, 6] & 0xffffffff
xcc = st[:, 6] >> 32
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0 = st[:, 0].min()
print('%d items on %d CUs; kernel span %.1f us' % (len(st), len(set(key.tolist())), (st[:, 5].max() - t0) / 100.0))
seg = np.stack([st[:, i + 1] - st[:, i] for i in range(5)], 1) / 100.0
print('per item (us): commit+barrier %.2f | first stage %.2f | other stages %.2f | issue next %.2f | transform+store %.2f | whole %.2f' %
      (*seg.mean(0), (st[:, 5] - st[:, 0]).mean() / 100.0))
clk = []
gaps = []
for k in sorted(set(key.tolist())):
    rows = st[key == k]
    rows = rows[np.argsort(rows[:, 0])]
    gaps.extend(((rows[1:, 0] - rows[:-1, 5]) / 100.0).tolist())
    clk.extend(((rows[1:, 7] - rows[:-1, 7]) / np.maximum(rows[1:, 0] - rows[:-1, 0], 1) * 100.0).tolist())
gaps = np.array(gaps)
sub = st[st[:, 9] > 0]
if len(sub):
    print('issue-next split (us): stage end -> bias read %.2f | advance + setup %.2f | requests %.2f' % (
        (sub[:, 8] - sub[:, 3]).mean() / 100.0, (sub[:, 9] - sub[:, 8]).mean() / 100.0, (sub[:, 4] - sub[:, 9]).mean() / 100.0))
print('gap between one item\'s last stamp and the next item\'s first on the same CU (us): mean %.2f median %.2f p90 %.2f max %.2f' %
      (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
print('shader clock over item-to-item intervals: mean %.0f MHz' % np.mean(clk))
k0 = sorted(set(key.tolist()))[0]
rows = st[key == k0]
rows = rows[np.argsort(rows[:, 0])]
print('first CU timeline (us since kernel start):')
for r in rows[:8]:
    print('   ' + ' '.join('%8.2f' % ((r[i] - t0) / 100.0) for i in range(6)))
