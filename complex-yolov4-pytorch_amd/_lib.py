"""ctypes binding of libcyolo_hip.so.  The prototypes are read from include/cyolo_hip.h (single source
of truth for the C ABI); there is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os
import re

# PyTorch must load ITS HIP runtime first: libcyolo_hip.so links libamdhip64.so.7 by soname, and if the system copy
# under /opt/rocm is mapped before torch's bundled one, torch later finds "No HIP GPUs".  Importing torch here makes
# the loader resolve our dependency to the runtime torch already mapped (one runtime, shared streams).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, '..', 'include', 'cyolo_hip.h')
LIBPATH = os.path.join(_HERE, 'csrc', 'libcyolo_hip.so')

_SCALARS = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'cy_stream_t': ctypes.c_void_p,
            'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32}


class CyoloError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(argtype, argname), ...])} for every `cy_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int64_t|int)\s+(cy_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        parsed = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    parsed.append((ctypes.c_void_p, a.split('*')[-1].strip()))
                else:
                    toks = a.replace('const ', '').split()
                    parsed.append((_SCALARS[toks[0]], toks[-1]))
        protos[name] = (_SCALARS[ret], parsed)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise CyoloError('libcyolo_hip.so not built: run `python __graft_entry__.py` (build()) first: ' + LIBPATH)
        self._dll = ctypes.CDLL(LIBPATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self._dll, name)
            fn.restype = ret
            fn.argtypes = [a for a, _ in args]

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise CyoloError('%s failed with status %d' % (name, rc))
        return rc


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
