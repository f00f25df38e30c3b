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
# Per-file flags.  yolo_head.hip: the per-target kernels (assign, pairs) index small polygon arrays dynamically; with the default
# promote-alloca budget they lived in scratch memory (368 / 880 bytes per lane) and were the only kernels of the step whose
# results changed when another kernel ran beside them (tools/head_race_probe.py).  With this budget (and their 64-thread
# launch bounds) every array is a register vector: ScratchSize 0.
EXTRA_FLAGS = {'yolo_head.hip': ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=1024']}


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdr_time = max(_mtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS)
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _mtime(o) < max(_mtime(s), hdr_time):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
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
