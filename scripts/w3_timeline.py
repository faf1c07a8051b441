#!/usr/bin/env python3
"""Per-workgroup timeline of w3_kernel on one layer (experiment build: `make -C 3d-brain-tumor-segmentation_amd/csrc stamps`).
usage: w3_timeline.py D Cin Cout.  Stamps (wave 0, 100 MHz wall clock): 0 entry, 1 first operands formed, 2 after the first
stage, 3 after the last stage, 4 after the exchange barrier, 5 after the stores."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('BTS_HIP_LIB', os.path.join(ROOT, '3d-brain-tumor-segmentation_amd', 'csrc', 'build', 'libbts_hip_stamps.so'))
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
fn = lib().cdll.bts_w3_stamps_copy_
fn.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert fn(buf.ctypes.data, n) == 0
st = buf.reshape(-1, 16)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print('%d workgroups; kernel span %.1f us' % (len(st), (st[:, 5].max() - t0) / 100.0))
seg = np.stack([st[:, i + 1] - st[:, i] for i in range(5)], 1) / 100.0
print('per workgroup (us): prologue %.2f | first stage %.2f | other stages %.2f | transform+exchange %.2f | combine+store %.2f | whole %.2f'
      % (*seg.mean(0), (st[:, 5] - st[:, 0]).mean() / 100.0))
cyc = (st[:, 8] - st[:, 7]) / np.maximum(st[:, 5] - st[:, 0], 1) * 100.0
print('shader clock inside a workgroup: mean %.0f MHz' % cyc.mean())
hw = st[:, 6] & 0xffffffff
key = (((st[:, 6] >> 32) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
gaps = []
for k in sorted(set(key.tolist())):
    rows = st[key == k]
    rows = rows[np.argsort(rows[:, 0])]
    gaps.extend(((rows[1:, 0] - rows[:-1, 5]) / 100.0).tolist())
gaps = np.array(gaps)
print('gap between workgroups on the same CU (us): mean %.2f median %.2f p90 %.2f' % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
