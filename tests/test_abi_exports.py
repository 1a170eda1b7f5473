"""CPU: the product library loads without a GPU and exports every entry point that include/*.h
declares (no compute call is made here); and it refuses to work without a device."""
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libav_amd", "libmi355dsp.so")


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for m in re.finditer(r"^\s*(?!static)(?:[A-Za-z_][\w\s\*]*?)\b((?:mi355_|ff_)\w+)\s*\(", text, flags=re.M):
            if "static" in text[text.rfind("\n", 0, m.start()) + 1:m.start(1)]:
                continue
            names.add(m.group(1))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return C.CDLL(LIB)


def test_headers_declare_something():
    names = declared_symbols()
    assert len(names) >= 20, names
    for must in ("mi355_init", "ff_h264dsp_init_mi355x", "ff_hevc_dsp_init_mi355x", "mi355_h264_decode_frames_dev"):
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback(lib):
    """Without a gfx950 device mi355_init reports failure (and libav_amd.load raises)."""
    import sys
    sys.path.insert(0, ROOT)
    import libav_amd
    lib.mi355_init.restype = C.c_int
    rc = lib.mi355_init(C.c_int(0))
    if rc == 0:
        pytest.skip("a GPU is present")
    with pytest.raises(libav_amd.BackendUnavailable):
        libav_amd.load(0)
