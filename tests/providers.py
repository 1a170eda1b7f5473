"""Interchangeable providers of the reference's DSP tables.

  ref     oracle/_ref/libref.so      the reference's own C objects (only where /root/reference exists)
  oracle  oracle/liboracle.so        our CPU restatement (the checker that travels to the GPU box)
  mi355   libav_amd/libmi355dsp.so   the product: C-ABI shim + HIP kernels (needs an MI355X)
  emu     tests/_emu/libmi355dsp_emu.so  the SAME product sources compiled against the SIMT
          emulator in tools/simt_emu (test-only; lets the CPU suite exercise kernel logic)
"""
import ctypes as C
import os
import subprocess

import abi_ctypes as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REFERENCE = os.path.isdir("/root/reference/libavcodec")


class Provider:
    def __init__(self, name, lib, fmt):
        self.name, self.lib, self.fmt = name, lib, fmt

    def _init(self, base, ctx, *args):
        fn = getattr(self.lib, self.fmt.format(base))
        fn.restype = None
        fn(C.byref(ctx), *args)
        return ctx

    def has(self, base):
        return hasattr(self.lib, self.fmt.format(base))

    def h264dsp(self, bit_depth=8, chroma_format_idc=1):
        return self._init("h264dsp_init", A.H264DSPContext(), C.c_int(bit_depth), C.c_int(chroma_format_idc))

    def h264qpel(self, bit_depth=8):
        return self._init("h264qpel_init", A.H264QpelContext(), C.c_int(bit_depth))

    def h264chroma(self, bit_depth=8):
        return self._init("h264chroma_init", A.H264ChromaContext(), C.c_int(bit_depth))

    def h264pred(self, bit_depth=8, chroma_format_idc=1):
        return self._init("h264_pred_init", A.H264PredContext(), C.c_int(A.AV_CODEC_ID_H264),
                          C.c_int(bit_depth), C.c_int(chroma_format_idc))

    def videodsp(self, bpc=8):
        return self._init("videodsp_init", A.VideoDSPContext(), C.c_int(bpc))

    def hevcdsp(self, bit_depth=8):
        return self._init("hevc_dsp_init", A.HEVCDSPContext(), C.c_int(bit_depth))

    def hevcpred(self, bit_depth=8):
        return self._init("hevc_pred_init", A.HEVCPredContext(), C.c_int(bit_depth))


def _make(target, cwd):
    subprocess.run(["make", "-s", target], cwd=cwd, check=True)


def oracle():
    _make("liboracle.so", os.path.join(ROOT, "oracle"))
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    return Provider("oracle", lib, "oracle_{}")


def ref():
    if not HAVE_REFERENCE:
        return None
    _make("_ref/libref.so", os.path.join(ROOT, "oracle"))
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
    return Provider("ref", lib, "ff_{}")


class Mi355Provider(Provider):
    """ff_<table>_init_mi355x(ctx, ...) — same argument lists as the reference's ff_<table>_init."""

    def _init(self, base, ctx, *args):
        fn = getattr(self.lib, "ff_" + base + "_mi355x")
        fn.restype = None
        fn(C.byref(ctx), *args)
        return ctx

    def has(self, base):
        return hasattr(self.lib, "ff_" + base + "_mi355x")


def mi355():
    """The product library.  Fails loudly if it is missing or no GPU is usable."""
    path = os.path.join(ROOT, "libav_amd", "libmi355dsp.so")
    if not os.path.exists(path):
        raise RuntimeError("libav_amd/libmi355dsp.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.mi355_init.restype = C.c_int
    rc = lib.mi355_init(C.c_int(0))
    if rc != 0:
        raise RuntimeError("mi355_init failed (%d): no usable MI355X" % rc)
    return Mi355Provider("mi355", lib, "")


def emu():
    d = os.path.join(ROOT, "tools", "simt_emu")
    _make("emu", d)
    lib = C.CDLL(os.path.join(ROOT, "tests", "_emu", "libmi355dsp_emu.so"))
    lib.mi355_init.restype = C.c_int
    assert lib.mi355_init(C.c_int(0)) == 0
    return Mi355Provider("emu", lib, "")
