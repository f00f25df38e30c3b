"""Persisted kernel / tile / split-K choices of the conv launches (models/engine.py).

The engine picks, per launch shape, one of the conv kernels' tile variants and the split-K factor of every weight-gradient
launch by timing the candidates once.  Timing has two costs: every process pays for it again (about 0.8 s for
complex_yolov4.cfg at 608 x 608, more at 1216 x 1216), and two processes (two data-parallel ranks, a resumed run, another
box) may crown different winners -- different tiles sum in a different order, so ``deterministic=True`` was reproducible
inside one process only.  This module keeps the winners in a JSON table shipped with the
package:

    tune_cache/gfx950.json = {"kernel_sources_sha": <sha of csrc/*.hip, *.hpp>, "entries": {repr(key): [choice, ms]}}

* a lookup hit replaces the timing (any mode);
* the table is valid only for the kernel sources it was measured on (the sha): after a kernel edit it is ignored and
  the default mode falls back to timing, until ``tools/make_tune_cache.py`` has been re-run on an MI355X;
* ``deterministic=True`` NEVER times: a hit is used, a miss takes the library's shape-only heuristic -- both are functions
  of the shape alone, so two fresh processes launch identical kernels (tests/test_gpu_r3.py::test_deterministic_across_processes);
* ``CY_TUNE_CACHE=0`` ignores the table, ``CY_TUNE_RECORD=<path>`` writes every choice this process made (timed or looked
  up) to <path> at exit -- that is how the table is produced.
"""
import atexit
import glob
import hashlib
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CACHE_PATH = os.path.join(_HERE, 'tune_cache', 'gfx950.json')

# bump when the MEANING of an entry changes without a kernel edit (engine.py's split caps, hint numbering, key layout):
# folded into the hash the table is validated against, so an older table is ignored instead of misread
TABLE_VERSION = 3

_sha = None
_table = None
_recorded = {}


def sources_sha():
    """sha256 (16 hex digits) over the HIP sources every measured kernel is built from."""
    global _sha
    if _sha is None:
        h = hashlib.sha256()
        h.update(b'table-version %d\0' % TABLE_VERSION)
        for f in sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.h*'))):
            with open(f, 'rb') as fh:
                h.update(os.path.basename(f).encode() + b'\0' + fh.read())
        # the compiler flags are part of what was measured: build.py's per-file flags change the code of every kernel
        # (CY_BUILD_NO_SLP=1), and a library loaded from elsewhere (CY_LIBPATH) is not the one the table was measured on at all
        from . import build as _build
        h.update(repr((_build.FLAGS[:4], sorted((k, tuple(v)) for k, v in _build.EXTRA_FLAGS.items()))).encode())
        h.update(b'libpath ' + os.environ.get('CY_LIBPATH', '').encode())
        _sha = h.hexdigest()[:16]
    return _sha


def _load():
    global _table
    if _table is not None:
        return _table
    _table = {}
    if os.environ.get('CY_TUNE_CACHE', '1') == '0':
        return _table
    try:
        with open(os.environ.get('CY_TUNE_CACHE_PATH', CACHE_PATH)) as f:
            doc = json.load(f)
        if doc.get('kernel_sources_sha') == sources_sha():
            _table = {k: tuple(v) for k, v in doc.get('entries', {}).items()}
    except (OSError, ValueError):
        pass
    return _table


def valid():
    """True when a table measured on the current kernel sources is loaded."""
    return bool(_load())


def get(key):
    """-> (choice, ms or None) or None."""
    return _load().get(repr(key))


def put(key, choice, ms=None):
    """Remember a choice made in this process (written out under CY_TUNE_RECORD)."""
    _recorded[repr(key)] = [choice, None if ms is None else round(float(ms), 5)]


def save(path, merge=True):
    entries = {}
    if merge:
        try:
            with open(path) as f:
                doc = json.load(f)
            if doc.get('kernel_sources_sha') == sources_sha():
                entries = doc.get('entries', {})
        except (OSError, ValueError):
            pass
    entries.update(_recorded)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump({'kernel_sources_sha': sources_sha(), 'entries': dict(sorted(entries.items()))}, f, indent=0)
        f.write('\n')
    return len(entries)


if os.environ.get('CY_TUNE_RECORD'):
    atexit.register(lambda: save(os.environ['CY_TUNE_RECORD']))
