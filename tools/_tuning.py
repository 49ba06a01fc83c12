"""Load the tuning build of the library (tools/libstabstitch_hip_tuning.so, `csrc/build.sh tuning`) in place of the
product library for the A/B and diagnosis scripts in this directory.  The product library has no tuning knobs."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, 'tools', 'libstabstitch_hip_tuning.so')


def lib():
    from stabstitch2_amd import _hip
    if not os.path.exists(PATH):
        subprocess.run(['bash', os.path.join(ROOT, 'stabstitch2_amd', 'csrc', 'build.sh'), 'tuning'], check=True)
    h = ctypes.CDLL(PATH)
    for name, (res, args) in _hip.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    h.ss_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
    h.ss_debug_set.restype = None
    h.ss_debug_ptr.argtypes = [ctypes.c_void_p]
    h.ss_debug_ptr.restype = None
    _hip._lib = h              # every ops.* call of this process now goes to the tuning build
    return h
