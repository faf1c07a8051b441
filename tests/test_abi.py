"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/bts_hip.h declares; argument
validation returns engine error codes before anything touches a GPU (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, '3d-brain-tumor-segmentation_amd', 'libbts_hip.so')
HEADER = os.path.join(ROOT, 'include', 'bts_hip.h')


@pytest.fixture(scope='module')
def cdll():
    if not os.path.exists(LIB):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, '3d-brain-tumor-segmentation_amd', 'csrc'), '-j8'])
    import torch  # noqa: F401  (HIP runtime load order, see _lib.py)
    return ctypes.CDLL(LIB)


def declared_symbols():
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    return re.findall(r'^\s*(?:int|long|const char\*)\s+(bts_\w+)\s*\(', src, flags=re.M)


def test_every_declared_symbol_is_exported(cdll):
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(cdll, n), 'libbts_hip.so does not export %s' % n
    out = subprocess.check_output(['nm', '-D', '--defined-only', LIB]).decode()
    exported = set(re.findall(r' T (bts_\w+)', out))
    assert set(names) <= exported
    assert exported - set(names) == set(), 'exported but undeclared: %s' % (exported - set(names))


def test_python_binding_matches_header():
    import bts_amd  # noqa: F401
    from bts_amd._lib import lib, parse_header
    protos = parse_header()
    assert set(protos) == set(declared_symbols())
    L = lib()
    assert L._bts_version().decode().endswith('gfx950')
    # every citation in the header points at the reference's files
    txt = open(HEADER).read()
    for f in ('layers/resnet.py', 'layers/group_norm.py', 'layers/downsample.py', 'layers/upsample.py', 'layers/vae.py',
              'util.py', 'train.py'):
        assert f in txt


def test_argument_validation_without_gpu(cdll):
    import bts_amd  # noqa: F401
    from bts_amd._lib import lib
    L = lib()
    # bad shapes are rejected with BTS_ERR_SHAPE (-1) before any HIP call
    assert L._bts_conv3d_fwd(1, None, None, None, None, None, 0, 0, 8, 8, 8, 4, 4, 4, 4, 0, None) == -1
    assert L._bts_conv3d_fwd(2, None, None, None, None, None, 0, 1, 7, 8, 8, 4, 4, 4, 4, 0, None) == -1   # odd size for s2
    assert L._bts_conv3d_fwd(1, None, None, None, None, None, 0, 1, 8, 8, 8, 8, 4, 4, 4, 0, None) == -1   # ld < C
    assert L._bts_conv_pack(7, 0, None, None, 4, 4, 4, 0, 0, None) == -3                          # unknown kind
    assert L._bts_gn_stats(None, None, None, None, 0, 1, 8, 6, 4, 0, 1e-5, None) == -1            # C % G != 0
    assert L._bts_gn_workspace(1, 512, 32, 8, 0) > 0
    assert L._bts_conv_packed_floats(1, 0, 32, 32) == (27 + 48 + 64) * 4 * 2 * 32 * 4   # implicit-GEMM image + the two Winograd images
    assert L._bts_conv_packed_floats(2, 0, 32, 32) == 27 * 4 * 2 * 32 * 4
    assert L._bts_conv3d_bwd_weight_workspace(1, 1, 128, 128, 128, 32, 32) > 0
    assert L._bts_adam_tf_step(None, None, None, None, 0, 1e-4, 0.9, 0.999, 1e-7, 1.0, None) == -1
    # kernel-symbol query used by bench.py
    assert L._bts_conv3d_fwd_config(1, 1, 128, 128, 128, 32, 32) == 0
    # split-K planning is host-only: the VAE's 1024->16 stride-2 conv on 16^3 needs a workspace, the 128^3 convs do not
    assert L._bts_conv3d_fwd_workspace(2, 1, 16, 16, 16, 1024, 16) > 0
    assert L._bts_conv3d_fwd_workspace(1, 1, 128, 128, 128, 32, 32) == 0
