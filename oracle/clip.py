"""ORACLE (test infrastructure).  ctypes binding of oracle/c/convex_clip.c plus a pure-Python twin.

Role: the float64 ``Polygon.intersection(...).area`` the reference obtains from shapely
(reference src/utils/iou_rotated_boxes_utils.py:91,119-120; src/utils/evaluation_utils.py:36,214).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.cy_oracle_quad_inter_area.restype = ctypes.c_double
        lib.cy_oracle_quad_inter_area.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.cy_oracle_inter_pairs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        lib.cy_oracle_inter_matrix.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                               ctypes.c_void_p]
        _LIB = lib
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def inter_area(qa, qb):
    """Area (float) of quad qa[4,2] ∩ quad qb[4,2]; corners are rounded to float32 first."""
    qa, qb = _f32(qa).reshape(8), _f32(qb).reshape(8)
    return float(_lib().cy_oracle_quad_inter_area(qa.ctypes.data, qb.ctypes.data))


def inter_pairs(a, b):
    a, b = _f32(a).reshape(-1, 8), _f32(b).reshape(-1, 8)
    out = np.empty(a.shape[0], dtype=np.float64)
    _lib().cy_oracle_inter_pairs(a.ctypes.data, b.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def inter_matrix(a, b):
    a, b = _f32(a).reshape(-1, 8), _f32(b).reshape(-1, 8)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    _lib().cy_oracle_inter_matrix(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out


def inter_area_py(qa, qb):
    """Pure-Python twin of the C routine (small cases; used to cross-check the C build)."""
    subj = [(float(x), float(y)) for x, y in _f32(qa).reshape(4, 2)]
    clip = [(float(x), float(y)) for x, y in _f32(qb).reshape(4, 2)]

    def area2(p):
        return sum(p[i][0] * p[(i + 1) % len(p)][1] - p[i][1] * p[(i + 1) % len(p)][0] for i in range(len(p)))

    ab = area2(clip)
    if area2(subj) == 0.0 or ab == 0.0:
        return 0.0
    if ab < 0.0:
        clip = [clip[3], clip[2], clip[1], clip[0]]
    for e in range(4):
        if not subj:
            break
        (cx, cy), (dx, dy) = clip[e], clip[(e + 1) % 4]
        ex, ey = dx - cx, dy - cy
        nxt = []
        for i, (sx, sy) in enumerate(subj):
            tx, ty = subj[(i + 1) % len(subj)]
            ds = ex * (sy - cy) - ey * (sx - cx)
            dt = ex * (ty - cy) - ey * (tx - cx)
            if ds >= 0.0:
                nxt.append((sx, sy))
            if (ds > 0.0 and dt < 0.0) or (ds < 0.0 and dt > 0.0):
                u = ds / (ds - dt)
                nxt.append((sx + u * (tx - sx), sy + u * (ty - sy)))
        subj = nxt
    if len(subj) < 3:
        return 0.0
    return 0.5 * abs(area2(subj))


class QuadPolygon:
    """Stand-in for ``shapely.geometry.Polygon`` limited to what the reference touches
    (``Polygon(pts).buffer(0)``, ``.intersection(other).area``, ``.area``).  Used by
    tests/golden/make_golden.py to import the reference, and by the oracle restatements."""

    def __init__(self, pts=None, _area=None):
        self._area = _area
        self.pts = None if pts is None else np.asarray([(float(p[0]), float(p[1])) for p in pts], dtype=np.float32)

    def buffer(self, _d):
        return self

    @property
    def area(self):
        if self._area is not None:
            return self._area
        p = self.pts.astype(np.float64)
        q = np.roll(p, -1, axis=0)
        return 0.5 * abs(float((p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0]).sum()))

    def intersection(self, other):
        return QuadPolygon(_area=inter_area(self.pts, other.pts))
