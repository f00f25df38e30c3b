"""Host logic of the persisted kernel / tile / split-K table (complex-yolov4-pytorch_amd/tune.py) and the consistency of the
files shipped with it -- no GPU needed.  A table whose stamp does not match the kernel sources is silently ignored at run time
(the default mode falls back to timing, the deterministic mode to the shape-only heuristics), and bench.py then also reports
`roofline.traffic: null`; these tests make a stale table or PMC file a test failure instead."""
import importlib
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402,F401  (registers the complex_yolov4_pytorch_amd alias)


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _restore_tune_module():
    """Every test reloads tune.py under its own environment; leave the module as a fresh default import for whoever runs next."""
    yield
    import complex_yolov4_pytorch_amd.tune as tune
    importlib.reload(tune)          # (monkeypatch has restored the caller's environment by now)


def _fresh_tune(monkeypatch, **env):
    for k in ('CY_TUNE_CACHE', 'CY_TUNE_CACHE_PATH', 'CY_TUNE_RECORD'):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    import complex_yolov4_pytorch_amd.tune as tune
    return importlib.reload(tune)


def test_shipped_table_matches_the_kernel_sources(monkeypatch):
    tune = _fresh_tune(monkeypatch)
    with open(tune.CACHE_PATH) as f:
        doc = json.load(f)
    assert doc['kernel_sources_sha'] == tune.sources_sha(), \
        'tune_cache/gfx950.json was measured on other kernel sources: re-run tools/make_tune_cache.py on an MI355X'
    assert tune.valid()
    kinds = {}
    for k, v in doc['entries'].items():
        key = eval(k)                       # keys are reprs of tuples of ints / bools / strs
        assert isinstance(key, tuple) and isinstance(key[0], str)
        assert isinstance(v, list) and len(v) == 2 and (v[0] is None or isinstance(v[0], int))
        if v[0] is None:                    # (a fused-sums candidate the pipelined kernel does not take: recorded as such)
            continue
        kinds[key[0]] = kinds.get(key[0], 0) + 1
        if key[0] == 'wgrad':               # split, + 1000 = 64 x 64 tiles, < 0 = atomic mode (never shipped)
            assert 1 <= v[0] % 1000 <= 2048 and v[0] > 0, (k, v)
        elif key[0] in ('fwd', 'dgrad', 'eval'):
            assert 0 <= v[0] <= 13, (k, v)   # tile hints of include/cyolo_hip.h (11-13: the slab kernels)
    for kind, least in (('fwd', 100), ('dgrad', 100), ('wgrad', 100), ('eval', 30), ('dgrad+sums', 50)):
        assert kinds.get(kind, 0) >= least, kinds
    assert tune.get(eval(next(iter(doc['entries'])))) is not None


def test_pmc_traffic_file_is_stamped_with_the_same_sources(monkeypatch):
    tune = _fresh_tune(monkeypatch)
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, bench.PMC_FILE)) as f:
        doc = json.load(f)
    assert doc['kernel_sources_sha'] == tune.sources_sha(), \
        '%s was collected on other kernel sources: re-run tools/pmc_traffic.sh (bench.py reports traffic: null until then)' % bench.PMC_FILE
    fam = {k: v for k, v in doc.items() if isinstance(v, dict) and 'launches' in v}
    assert {'igemm', 'wgrad', 'wgrad_reduce', 'bn_act_fwd', 'bn_bwd_apply', 'adam', 'pack_weights'} <= set(fam)
    step = sum(v['launches'] / doc['steps_counted'] * (v['fetch_bytes_per_launch_corrected'] + v['write_bytes_per_launch']) for v in fam.values())
    assert 40e9 < step < 80e9            # whole-step HBM traffic of the benchmarked configuration (round 3: 61.5 GB)


def test_sq_counter_file_is_stamped_with_the_same_sources(monkeypatch):
    """BASELINE's "MFMA utilisation % (rocprof)" on the line (roofline.mfma_busy) comes from profiles/r05_sq_counters.json: valid for
    exactly the kernel sources it was collected on, like the PMC traffic figure."""
    tune = _fresh_tune(monkeypatch)
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, bench.SQ_FILE)) as f:
        doc = json.load(f)
    assert doc['kernel_sources_sha'] == tune.sources_sha(), \
        '%s was collected on other kernel sources: re-run tools/pmc_sq.sh (bench.py reports mfma_busy: null until then)' % bench.SQ_FILE
    fam = doc['families']
    assert {'conv fwd/dgrad', 'wgrad', 'whole step', 'bn_act_fwd', 'bn_bwd_apply'} <= set(fam)
    assert 0.05 < fam['conv fwd/dgrad']['mfma_busy'] < 0.9 and fam['bn_act_fwd']['mfma_busy'] < 1e-3
    assert 500 < fam['whole step']['launches_per_step'] < 800

    class A:
        batch, size, dtype, config = 16, 608, 'f16', 'train608'
    sq = bench.sq_counters(A)
    assert sq['conv_fwd_dgrad']['mfma_busy'] == round(fam['conv fwd/dgrad']['mfma_busy'], 4) and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq['definition']
    A.batch = 8
    assert bench.sq_counters(A) is None          # any other workload: no figure


def test_stale_or_disabled_table_is_ignored(monkeypatch, tmp_path):
    stale = tmp_path / 'stale.json'
    stale.write_text(json.dumps({'kernel_sources_sha': '0' * 16, 'entries': {repr(('fwd', 1)): [7, 0.1]}}))
    tune = _fresh_tune(monkeypatch, CY_TUNE_CACHE_PATH=str(stale))
    assert not tune.valid() and tune.get(('fwd', 1)) is None
    good = tmp_path / 'good.json'
    tune = _fresh_tune(monkeypatch)
    good.write_text(json.dumps({'kernel_sources_sha': tune.sources_sha(), 'entries': {repr(('fwd', 1)): [7, 0.1]}}))
    tune = _fresh_tune(monkeypatch, CY_TUNE_CACHE_PATH=str(good))
    assert tune.valid() and tune.get(('fwd', 1)) == (7, 0.1)
    tune = _fresh_tune(monkeypatch, CY_TUNE_CACHE_PATH=str(good), CY_TUNE_CACHE='0')
    assert not tune.valid() and tune.get(('fwd', 1)) is None
    tune = _fresh_tune(monkeypatch, CY_TUNE_CACHE_PATH=str(tmp_path / 'missing.json'))
    assert not tune.valid()


def test_save_merges_into_a_table_of_the_same_sources_only(monkeypatch, tmp_path):
    tune = _fresh_tune(monkeypatch)
    out = tmp_path / 't.json'
    out.write_text(json.dumps({'kernel_sources_sha': tune.sources_sha(), 'entries': {repr(('a',)): [1, 0.5], repr(('b',)): [2, None]}}))
    tune.put(('b',), 3, 0.123456789)
    tune.put(('c',), 4)
    assert tune.save(str(out)) == 3
    doc = json.loads(out.read_text())
    assert doc['entries'] == {repr(('a',)): [1, 0.5], repr(('b',)): [3, 0.12346], repr(('c',)): [4, None]}
    other = tmp_path / 'o.json'
    other.write_text(json.dumps({'kernel_sources_sha': 'f' * 16, 'entries': {repr(('z',)): [9, 1.0]}}))
    assert tune.save(str(other)) == 2          # entries measured on other sources are dropped, not carried over
    assert repr(('z',)) not in json.loads(other.read_text())['entries']
    assert json.loads(other.read_text())['kernel_sources_sha'] == tune.sources_sha()


def test_sources_sha_covers_every_kernel_source(monkeypatch, tmp_path):
    """The stamp changes with any byte of csrc/*.hip, *.hpp (that is what invalidates a table after a kernel edit), with the
    compiler flags the kernels are built with (build.py: CY_BUILD_NO_SLP=1 changes every kernel's code) and with CY_LIBPATH (a
    library loaded from elsewhere is not the one the table was measured on) -- ADVICE r5."""
    import glob
    import hashlib
    tune = _fresh_tune(monkeypatch)
    from complex_yolov4_pytorch_amd import build
    here = os.path.dirname(tune.__file__)
    files = sorted(glob.glob(os.path.join(here, 'csrc', '*.h*')))
    names = {os.path.basename(f) for f in files}
    assert {'conv_pipe.hip', 'conv_igemm.hip', 'conv_wgrad.hip', 'conv_direct.hip', 'elementwise.hip', 'yolo_head.hip',
            'igemm_common.hpp', 'common.hpp', 'geometry.hpp'} <= names
    h = hashlib.sha256()
    h.update(b'table-version %d\0' % tune.TABLE_VERSION)      # (the meaning of the entries: engine.py's caps / hint numbering)
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    h.update(repr((build.FLAGS[:4], sorted((k, tuple(v)) for k, v in build.EXTRA_FLAGS.items()))).encode())
    h.update(b'libpath ')
    base = tune.sources_sha()
    assert base == h.hexdigest()[:16]
    assert not any('/' in f for f in build.FLAGS[:4])          # no path of this checkout in the stamp: the GPU box sees another root
    assert _fresh_tune(monkeypatch, CY_LIBPATH='/somewhere/else.so').sources_sha() != base
    monkeypatch.delenv('CY_LIBPATH')
    monkeypatch.setattr(build, 'EXTRA_FLAGS', {k: list(v) + ['-fno-slp-vectorize'] for k, v in build.EXTRA_FLAGS.items()})
    assert importlib.reload(tune).sources_sha() != base
