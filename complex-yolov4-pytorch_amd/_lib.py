"""ctypes binding of libcyolo_hip.so.  The prototypes are read from include/cyolo_hip.h (single source
of truth for the C ABI); there is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os
import re
import threading

# PyTorch must load ITS HIP runtime first: libcyolo_hip.so links libamdhip64.so.7 by soname, and if the system copy
# under /opt/rocm is mapped before torch's bundled one, torch later finds "No HIP GPUs".  Importing torch here makes
# the loader resolve our dependency to the runtime torch already mapped (one runtime, shared streams).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, '..', 'include', 'cyolo_hip.h')
LIBPATH = os.environ.get('CY_LIBPATH') or os.path.join(_HERE, 'csrc', 'libcyolo_hip.so')      # (CY_LIBPATH: A/B builds, tools/)

_SCALARS = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'cy_stream_t': ctypes.c_void_p,
            'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32}


_TRACE_SYNC = os.environ.get('CY_TRACE_SYNC') == '1'      # fault hunting only (see _Lib._call_traced)


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


class PlanRecorder:
    """Collects the calls of one eager pass as a program for cy_run_plan (include/cyolo_hip.h, csrc/plan_replay.cpp): while
    it is the library's ``recorder`` every ``_Lib.call`` is executed as usual AND appended as int64 words [fn, nargs, args...]
    (pointers / integers as themselves, floats as the bits of a double).  ``py(fn)`` cuts the C segment and puts a Python
    callable between two segments (a hook that has to run on the host at that point of the stream order)."""

    def __init__(self, lib):
        self.lib, self.items, self._words, self.calls = lib, [], None, 0

    def call(self, name, args):
        fn = self.lib.fn_index(name)
        if fn < 0:
            raise CyoloError('%s cannot be part of a recorded launch list (host-array arguments or not an int entry point)' % name)
        types = self.lib.protos[name][1]
        if self._words is None:
            self._words = []
            self.items.append(('c', self._words))
        w = self._words
        w.append(fn)
        w.append(len(types))
        for (ct, _), v in zip(types, args):
            if ct is ctypes.c_void_p:
                if v is None:
                    w.append(0)
                elif isinstance(v, int):
                    w.append(v)
                elif isinstance(v, ctypes.Array):       # a host array the caller keeps alive (per-model constants: anchors)
                    w.append(ctypes.addressof(v))
                else:
                    w.append(v.value or 0)
            elif ct is ctypes.c_float:
                w.append(_DBL.unpack(_DBLP.pack(float(v)))[0])
            else:
                w.append(int(v))
        self.calls += 1

    def py(self, fn):
        self.items.append(('py', fn))
        self._words = None

    def finish(self):
        return Program(self.lib, self.items, self.calls)


class Program:
    """A recorded launch list: ``run()`` re-issues it (one cy_run_plan call per C segment, Python hooks in between)."""

    def __init__(self, lib, items, calls):
        self.lib, self.calls = lib, calls
        self.items = [(k, (ctypes.c_int64 * len(v))(*v)) if k == 'c' else (k, v) for k, v in items if k == 'py' or v]
        self._failed = ctypes.c_int32(0)

    def run(self):
        run, failed = self.lib._dll.cy_run_plan, self._failed
        for kind, v in self.items:
            if kind == 'c':
                rc = run(v, len(v), ctypes.byref(failed))
                if rc != 0:
                    raise CyoloError('replayed launch list: call #%d failed with status %d' % (failed.value, rc))
            else:
                v()


import struct as _struct  # noqa: E402

_DBL, _DBLP = _struct.Struct('<q'), _struct.Struct('<d')


class _Lib:
    # a PlanRecorder while an engine records a pass -- of the recording THREAD only (a prefetch thread rasterising
    # the next batch, or an eval engine in another thread, must neither be recorded into the training program nor be refused)
    _rec = threading.local()

    @property
    def recorder(self):
        return getattr(self._rec, 'r', None)

    @recorder.setter
    def recorder(self, r):
        self._rec.r = r

    def fn_index(self, name):
        i = self._fn_index.get(name)
        if i is None:
            i = self._fn_index[name] = int(self._dll.cy_plan_fn_index(name.encode()))
        return i

    def __init__(self):
        self._fn_index = {}
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
        if self.recorder is not None:
            self.recorder.call(name, args)
        if _TRACE_SYNC:
            return self._call_traced(name, args)
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise CyoloError('%s failed with status %d' % (name, rc))
        return rc

    def _call_traced(self, name, args):
        """CY_TRACE_SYNC=1 (fault hunting): name and arguments of every call go to stderr BEFORE it is issued and the device is
        synchronised after it, so that a GPU memory-access fault is attributed to the last line printed."""
        import sys
        sys.stderr.write('cy> %s %s\n' % (name, ' '.join(
            ('%#x' % (a.value or 0)) if isinstance(a, ctypes.c_void_p) else ('host[]' if isinstance(a, ctypes.Array) else repr(a)) for a in args)))
        sys.stderr.flush()
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise CyoloError('%s failed with status %d' % (name, rc))
        torch.cuda.synchronize()
        return rc


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
