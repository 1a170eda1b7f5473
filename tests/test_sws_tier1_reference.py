"""CPU, only where /root/reference exists: the reference's own sws_scale() with the inner loops of its generic
scaler forwarded, call by call, to the product's Tier-1 swscale entry points (oracle/ref_sws_tier1_glue.c:
ff_getSwsFunc interposed by the linker, product sources in their emulated build).  The line-pull loop, ring
buffers and filter banks are the reference's; the pictures must equal the plain reference's golden sha1s."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import pytest

import sws_support as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "sws_ref_sha1.json")))
GENERIC = [k for k in S.SMALL if not k.startswith("special")]


@pytest.fixture(scope="module")
def hooked(emu):
    if not S.HAVE_REFERENCE:
        pytest.skip("/root/reference not present")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libswsref_tier1.so"], check=True)
    ref = S.Reference.__new__(S.Reference)
    ref.lib = lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libswsref_tier1.so"))
    lib.sws_getContext.restype = C.c_void_p
    lib.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
    lib.sws_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.sws_freeContext.argtypes = [C.c_void_p]
    lib.ref_sws_tier1_calls.restype = C.c_ulong
    ref.name = "ref+tier1"
    return ref


@pytest.mark.parametrize("name", GENERIC)
def test_reference_sws_scale_through_tier1_inner_loops(hooked, name):
    before = hooked.lib.ref_sws_tier1_calls()
    out = hooked.scale(name, S.picture(name), dst_pad=8)
    assert hooked.lib.ref_sws_tier1_calls() > before          # the shims really ran
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]
