"""ctypes binding of libbts_hip.so (the C ABI declared in include/bts_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the C side.
There is NO fallback: if the HIP library is missing or a call returns non-zero, a RuntimeError is raised.
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- MUST precede the dlopen below: torch ships its own libamdhip64/libhsa-runtime64; loading
#                      ours first would put a second HIP runtime in the process (kernel launches then fail with
#                      hipErrorNoDevice).  With torch's runtime resident, libbts_hip.so binds to the same one.

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, 'include', 'bts_hip.h')
LIBPATH = os.environ.get('BTS_HIP_LIB') or os.path.join(_HERE, 'libbts_hip.so')  # override: A/B timing of two builds

_CT = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'uint64_t': ctypes.c_uint64,
    'bts_stream_t': ctypes.c_void_p,
}

ERRORS = {-1: 'BTS_ERR_SHAPE', -2: 'BTS_ERR_ALIGN', -3: 'BTS_ERR_UNSUPPORTED', -4: 'BTS_ERR_WORKSPACE',
          -101: 'ncclUnhandledCudaError', -102: 'ncclSystemError', -103: 'ncclInternalError', -104: 'ncclInvalidArgument',
          -105: 'ncclInvalidUsage', -106: 'ncclRemoteError', -107: 'ncclInProgress'}


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(ctype_str, argname), ...])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'^\s*(int|long|const char\*)\s+(bts_\w+)\s*\((.*?)\)\s*;', src, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'^(.*?)(\w+)$', a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        out[name] = (ret, alist)
    return out


def _ctype(s):
    if '*' in s:
        return ctypes.c_void_p
    return _CT[s.replace('const ', '').strip()]


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise RuntimeError(
                'libbts_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(there is no CPU fallback for the product path)' % LIBPATH)
        self.cdll = ctypes.CDLL(LIBPATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so does not export a declared symbol
            fn.argtypes = [_ctype(t) for t, _ in args]
            fn.restype = ctypes.c_char_p if ret == 'const char*' else (ctypes.c_long if ret == 'long' else ctypes.c_int)
            setattr(self, '_' + name, fn)

    def call(self, name, *args):
        """status-checked call of an int-returning entry point"""
        r = getattr(self, '_' + name)(*args)
        if r != 0:
            raise RuntimeError('%s failed: %s' % (name, ERRORS.get(r, 'hipError %d' % r)))

    def query(self, name, *args):
        """long-returning size query"""
        r = getattr(self, '_' + name)(*args)
        if r < 0:
            raise RuntimeError('%s: unsupported shape' % name)
        return r

    def probe(self, name, *args):
        """unchecked value of a query whose negative / zero answer means "this form does not take the shape" (the caller then runs the
        general route), not an error"""
        return getattr(self, '_' + name)(*args)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
