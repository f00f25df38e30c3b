"""Test-only HIP instruments (tests/csrc/probes.hip -> tests/_build/libcyolo_probes.so, built by __graft_entry__.build() beside the
oracle's C checker).  They used to sit in the product library and its public header; nothing under
complex-yolov4-pytorch_amd/ knows about them."""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'probes.hip')
LIB = os.path.join(HERE, '_build', 'libcyolo_probes.so')
_dll = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                               '-shared', '-o', LIB, SRC])
    build_head_slp(force)
    return LIB


# The reproducer of "packed f32 beside MFMA loses a result" (profiles/r05_head_race.txt): the product's yolo_head.hip compiled
# WITH hipcc's SLP vectoriser -- the default -O3 flags, i.e. what the library shipped until round 5 -- into a library of its own.
# Its cy_yolo_loss beside an MFMA-issuing conv kernel on another stream is the victim of tests/test_gpu_r6.py; the shipped build
# (-fno-slp-vectorize for this file, build.py) is the control.
PKG_CSRC = os.path.join(HERE, '..', 'complex-yolov4-pytorch_amd', 'csrc')
HEAD_SLP_LIB = os.path.join(HERE, '_build', 'libcyolo_head_slp.so')


def build_head_slp(force=False):
    src = os.path.join(PKG_CSRC, 'yolo_head.hip')
    deps = [src, os.path.join(PKG_CSRC, 'geometry.hpp'), os.path.join(PKG_CSRC, 'common.hpp')]
    if force or not os.path.exists(HEAD_SLP_LIB) or os.path.getmtime(HEAD_SLP_LIB) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(HEAD_SLP_LIB), exist_ok=True)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                               '-I' + os.path.join(HERE, '..', 'include'), '-I' + PKG_CSRC, '-Wno-unused-value',
                               '-mllvm', '-amdgpu-promote-alloca-to-vector-limit=1024', '-shared', '-o', HEAD_SLP_LIB, src])
    return HEAD_SLP_LIB


def head_slp_lib():
    """ctypes handle of the SLP build of the head kernels (cy_yolo_loss with the product's signature)."""
    import complex_yolov4_pytorch_amd._lib  # noqa: F401  (torch's HIP runtime first)
    dll = ctypes.CDLL(build_head_slp())
    dll.cy_yolo_loss.restype = ctypes.c_int
    dll.cy_yolo_loss.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p]
    return dll


def _lib():
    global _dll
    if _dll is None:
        import complex_yolov4_pytorch_amd._lib  # noqa: F401  (torch's HIP runtime is mapped first, see _lib.py)
        _dll = ctypes.CDLL(build())
        _dll.cyt_probe_dirty.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _dll.cyt_probe_tr16.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return _dll


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def probe_dirty(pattern, blocks=2048, lds_bytes=64 * 1024):
    """Dirties VGPRs and LDS with ``pattern`` on every CU (on torch's current stream)."""
    rc = _lib().cyt_probe_dirty(int(pattern) & 0xFFFFFFFF, int(blocks), int(lds_bytes), None, _stream())
    assert rc == 0, rc


def probe_tr16():
    out = torch.zeros(64, 4, dtype=torch.int16, device='cuda')
    rc = _lib().cyt_probe_tr16(ctypes.c_void_p(out.data_ptr()), _stream())
    assert rc == 0, rc
    return out
