"""libav_amd — MI355X (gfx950) backend for libav's H.264/HEVC DSP hot path and
libswscale's inner loops.

The product is the C-ABI shared library `libav_amd/libmi355dsp.so` (HIP kernels +
`extern "C"` entry points declared in include/*.h).  This Python package is only a
thin loader used by bench.py and the tests; it never computes anything itself and
it has NO CPU fallback: `load()` raises if the HIP library or an MI355X is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355dsp.so")
_lib = None


class BackendUnavailable(RuntimeError):
    pass


def load(device=0):
    """Load libmi355dsp.so and bind it to GPU `device`.  Raises BackendUnavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.mi355_init.restype = ctypes.c_int
    rc = lib.mi355_init(ctypes.c_int(device))
    if rc != 0:
        raise BackendUnavailable("mi355_init(%d) failed with %d: no usable gfx950 device" % (device, rc))
    _lib = lib
    return lib
