"""Build libcyolo_hip.so in-tree with hipcc for gfx950 (no torch extension machinery: the library
is a plain C ABI, see include/cyolo_hip.h).  Objects are cached by source mtime."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(HERE, '..', 'include')
LIB = os.path.join(CSRC, 'libcyolo_hip.so')
SOURCES = ['conv_igemm.hip', 'conv_pipe.hip', 'conv_direct.hip', 'conv_wgrad.hip', 'elementwise.hip', 'yolo_head.hip', 'riou_nms.hip', 'bev.hip']
HEADERS = ['common.hpp', 'igemm_common.hpp', 'geometry.hpp', os.path.join(INCLUDE, 'cyolo_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC, '-Wno-unused-value']
# Per-file flags.  yolo_head.hip / riou_nms.hip: the per-target / per-pair kernels index small polygon arrays dynamically; with the
# default promote-alloca budget they lived in scratch memory (368 / 880 bytes per lane) and assign / pairs were the only kernels of
# the step whose results changed when another kernel ran beside them (tools/head_race_probe.py).  With this budget (and their
# 64-thread launch bounds) every array is a register vector: ScratchSize 0.
_ALLOCA = ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=1024']
_REMARKS = ['-Rpass-analysis=kernel-resource-usage']
# -fno-slp-vectorize for the geometry / head / rasteriser kernels (round 5, profiles/r05_head_race.txt): hipcc's SLP vectoriser turns
# their float32 chains into PACKED-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel shuffles: 2703 of
# them in yolo_head.hip), and on the MI355X those returned wrong values in lanes 48-63 of a wave in 0.7 % of launches whenever an
# MFMA-heavy wave of ANOTHER kernel shared the SIMD (igemm_fast<192,128> or the 384 x 128 pipelined tile on a second stream; never
# alone) -- the "head race" of rounds 2-4, whose mechanism was unknown.  Built without SLP (4 packed instructions left) the same
# probe is clean: 0 of 2999 repeats against 8 / 1499, 11 / 1499, 10 / 999 with it, results bit-identical.  These kernels are
# latency-bound one-wave chains: the scalar forms cost nothing.  The BatchNorm / activation passes and the conv epilogues keep the
# packed forms (0.4-0.6 % of the step, tools/r5_ab_lib.sh): they were never seen to differ beside the same aggressors
# (tools/victim_probe.py, 0 / 1499 each) nor in 5000 bit-identical repeats of the whole two-stream step; CY_BUILD_NO_SLP=1 builds
# every file without SLP for whoever wants the belt as well as the braces.
_NO_SLP = ['-fno-slp-vectorize']
_NO_SLP_FILES = ('yolo_head.hip', 'riou_nms.hip', 'bev.hip')
EXTRA_FLAGS = {src: _REMARKS + (_ALLOCA if src in ('yolo_head.hip', 'riou_nms.hip') else []) +
               (_NO_SLP if (src in _NO_SLP_FILES or os.environ.get('CY_BUILD_NO_SLP') == '1') else []) for src in SOURCES}
# NO kernel of the library may come out of the compiler with a private segment (scratch): checked on every compile of every
# file (check_scratch).  Round 2 found the head kernels' results to depend on what ran beside them while they spilled; round 3
# found two conv instantiations that had quietly acquired spills (a 384 x 128 direct-store epilogue, 196 bytes per lane) while
# hunting a 1-in-10^4 gradient difference of the deterministic mode.  A different hipcc or a small edit can bring a spill
# back silently -- with this check it fails the build instead.  '*' = every kernel of the file.
NO_SCRATCH = {src: ('*',) for src in SOURCES}
RESOURCES = os.path.join(CSRC, 'kernel_resources.json')


def parse_resources(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {'scratch': bytes/lane, 'vgprs': n, 'sgprs': n}}."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('vgprs', r' VGPRs: (\d+)'), ('sgprs', r'TotalSGPRs: (\d+)')):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def check_scratch(src, text):
    """Raise when a kernel named in NO_SCRATCH[src] was compiled with a private segment; records every kernel's figures."""
    import json
    res = parse_resources(text)
    try:
        with open(RESOURCES) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {}
    doc[src] = res
    with open(RESOURCES, 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    wanted = NO_SCRATCH.get(src, ())
    seen = {w: [k for k in res if w == '*' or w in k] for w in wanted}
    missing = [w for w, ks in seen.items() if not ks]
    if missing:
        raise RuntimeError('%s: no resource-usage remark for %s (compiler output format changed?)' % (src, missing))
    bad = {k: res[k].get('scratch') for ks in seen.values() for k in ks if res[k].get('scratch', -1) != 0}
    if bad:
        raise RuntimeError('%s: kernels must have ScratchSize 0 (no private segment anywhere in the library, see build.py): %s' % (src, bad))
    return res


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


PLAN_SRC, PLAN_INC = os.path.join(CSRC, 'plan_replay.cpp'), os.path.join(CSRC, 'plan_tramp.inc')
# entry points that cannot be part of a recorded launch list: the plan machinery itself, and calls whose HOST-array arguments
# change from call to call (learning rates per step, augmentation rectangles).  The head entry points read host arrays too
# (anchors, the heads table), but those are per-model constants the operator layer keeps alive (ops.py).
_PLAN_SKIP = ('cy_run_plan', 'cy_plan_fn_index', 'cy_plan_fn_nargs', 'cy_event_create', 'cy_event_destroy', 'cy_adam_multi',
              'cy_adam_multi_dev', 'cy_sgd_multi', 'cy_bev_mosaic', 'cy_bev_mosaic_targets', 'cy_bev_flip_cutout')


def gen_plan_trampolines(header=None, out=None):
    """include/cyolo_hip.h -> csrc/plan_tramp.inc: one trampoline per int-returning entry point, unpacking cy_run_plan's
    int64 words (pointers / integers as themselves, floats as double bits) into the C signature, and the name table."""
    import re
    text = open(header or os.path.join(INCLUDE, 'cyolo_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    tramps, table = [], []
    for m in re.finditer(r'\b(int64_t|int)\s+(cy_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        if ret != 'int' or name in _PLAN_SKIP:
            continue
        casts = []
        if args and args != 'void':
            for i, a in enumerate(args.split(',')):
                a = a.strip()
                ty = a[:re.search(r'[A-Za-z_0-9]+$', a).start()].strip()
                if '*' in ty or ty == 'cy_stream_t':
                    casts.append('(%s)a[%d].p' % (ty, i))
                elif ty in ('float', 'double'):
                    casts.append('(%s)a[%d].d' % (ty, i))
                else:
                    casts.append('(%s)a[%d].i' % (ty, i))
        tramps.append('static int t_%s(const Word* a) { (void)a; return %s(%s); }' % (name, name, ', '.join(casts)))
        table.append('    {"%s", t_%s, %d},' % (name, name, len(casts)))
    body = ('// GENERATED by build.py::gen_plan_trampolines from include/cyolo_hip.h -- do not edit\n' + '\n'.join(tramps) +
            '\nstatic const Entry kEntries[] = {\n' + '\n'.join(table) + '\n};\n')
    out = out or PLAN_INC
    if not os.path.exists(out) or open(out).read() != body:
        with open(out, 'w') as f:
            f.write(body)
    return len(table)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdr_time = max(_mtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS)
    objs, procs = [], []
    # the launch-list replayer (host code only; not a kernel source: outside tune.sources_sha's glob)
    gen_plan_trampolines()
    plan_o = os.path.join(CSRC, 'plan_replay.o')
    if force or _mtime(plan_o) < max(_mtime(PLAN_SRC), _mtime(PLAN_INC), _mtime(os.path.join(INCLUDE, 'cyolo_hip.h'))):
        subprocess.check_call([hipcc, '-O2', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC,
                               '-I' + os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), 'include'),
                               '-D__HIP_PLATFORM_AMD__', '-x', 'c++',
                               '-c', PLAN_SRC, '-o', plan_o])
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _mtime(o) < max(_mtime(s), hdr_time, _mtime(os.path.abspath(__file__))):      # (this file holds the per-file flags)
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if src in NO_SCRATCH:
            try:
                check_scratch(src, out.decode())
            except RuntimeError:
                os.remove(os.path.join(CSRC, src.replace('.hip', '.o')))     # never link an object that failed the check
                raise
    objs.append(plan_o)
    if procs or force or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        # dlopen it: a kernel stub the host pass dropped shows up as an undefined symbol only at load time
        import ctypes
        ctypes.CDLL(LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
