"""ORACLE -- test infrastructure, NOT product code.

CPU restatement of the reference's algorithms for the hot path, used only as the
checker: by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``complex-yolov4-pytorch_amd/`` imports this package; the product path fails loudly
when the HIP library is missing instead of falling back to anything here.

Pinning: ``tests/golden/*.npz`` were produced by importing the reference itself from
/root/reference (script: tests/golden/make_golden.py, with a float64 convex-clip stand-in for the
absent ``shapely``); ``tests/test_oracle_golden.py`` checks every oracle function against them.
The GEOS polygon intersection itself is PARITY UNPINNED (dependency absent, version unlisted):
it is restated as exact convex clipping in float64 (oracle/c/convex_clip.c) and pinned only
against analytic known answers.
"""
