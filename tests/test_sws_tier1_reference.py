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


def bind(ref, path):
    ref.lib = lib = C.CDLL(path)
    lib.sws_getContext.restype = C.c_void_p
    lib.sws_getContext.argtypes = [C.c_int] * 7 + [C.c_void_p] * 3
    lib.sws_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.sws_freeContext.argtypes = [C.c_void_p]
    lib.ref_sws_tier1_calls.restype = C.c_ulong
    lib.ref_sws_pictures.restype = C.c_ulong
    return ref


@pytest.fixture(scope="module")
def hooked(emu):
    if not S.HAVE_REFERENCE:
        pytest.skip("/root/reference not present")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libswsref_tier1.so"], check=True)
    ref = bind(S.Reference.__new__(S.Reference), os.path.join(ROOT, "oracle", "_ref", "libswsref_tier1.so"))
    ref.name = "ref+tier1"
    return ref


@pytest.mark.parametrize("name", GENERIC)
def test_reference_sws_scale_through_tier1_inner_loops(hooked, name, monkeypatch):
    monkeypatch.setenv("MI355_SWS_LINES", "1")                # the inner-loop form of the binding (read when the context is made)
    before, pics = hooked.lib.ref_sws_tier1_calls(), hooked.lib.ref_sws_pictures()
    out = hooked.scale(name, S.picture(name), dst_pad=8)
    assert hooked.lib.ref_sws_tier1_calls() > before          # the shims really ran
    assert hooked.lib.ref_sws_pictures() == pics
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]


@pytest.mark.parametrize("name", S.SMALL)
def test_reference_sws_scale_whole_picture_binding(hooked, name, monkeypatch):
    """the default form: SwsContext.swscale replaced through both selectors (ff_getSwsFunc, ff_yuv2rgb_get_func_ptr) — the
    reference's sws_scale() of a whole picture is one pass of the product's fused kernel"""
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    before, calls = hooked.lib.ref_sws_pictures(), hooked.lib.ref_sws_tier1_calls()
    out = hooked.scale(name, S.picture(name), dst_pad=8)
    assert hooked.lib.ref_sws_pictures() == before + 1
    assert hooked.lib.ref_sws_tier1_calls() == calls
    assert (out[:, -8:] == 0x5A).all()
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]


def test_sliced_calls_keep_the_reference_function(hooked, monkeypatch):
    """a picture handed over in two slices goes to the function the reference had chosen (same picture as in one piece)"""
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    name = "generic_64x48"
    sw, sh, dw, dh = S.CONFIGS[name][:4]
    planes = S.picture(name)
    c = hooked.open(name)
    import numpy as np
    out = np.full((dh, dw * 3 + 8), 0x5A, np.uint8)
    strides = (C.c_int * 4)(*[p.strides[0] for p in planes], 0)
    dst = (C.c_void_p * 4)(out.ctypes.data, None, None, None)
    dstrides = (C.c_int * 4)(out.strides[0], 0, 0, 0)
    before, n = hooked.lib.ref_sws_pictures(), 0
    for y0, hh in ((0, sh // 2), (sh // 2, sh - sh // 2)):
        src = (C.c_void_p * 4)(planes[0][y0:].ctypes.data, planes[1][y0 // 2:].ctypes.data, planes[2][y0 // 2:].ctypes.data, None)
        n += hooked.lib.sws_scale(c, src, strides, y0, hh, dst, dstrides)
    hooked.close(c)
    assert n == dh and hooked.lib.ref_sws_pictures() == before
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]


def test_bottom_up_pictures_go_to_the_reference_function(hooked, monkeypatch):
    """negative line sizes are legal for sws_scale() (vf_vflip makes them): the whole-picture binding must hand such a call to the function
    the reference chose instead of a 2-D copy with a negative pitch (that aborted the process)"""
    import numpy as np
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    name = "generic_64x48"
    sw, sh, dw, dh = S.CONFIGS[name][:4]
    planes = S.picture(name)
    outs = []
    for flip in (False, True):
        c = hooked.open(name)
        out = np.full((dh, dw * 3), 0x5A, np.uint8)
        src_planes = [np.ascontiguousarray(p[::-1]) for p in planes] if not flip else planes          # not flipped: an upside-down copy, read top-down
        if flip:
            src = (C.c_void_p * 4)(*[p.ctypes.data + (p.shape[0] - 1) * p.strides[0] for p in planes], None)
            strides = (C.c_int * 4)(*[-p.strides[0] for p in planes], 0)
        else:
            src = (C.c_void_p * 4)(*[p.ctypes.data for p in src_planes], None)
            strides = (C.c_int * 4)(*[p.strides[0] for p in src_planes], 0)
        dst = (C.c_void_p * 4)(out.ctypes.data, None, None, None)
        dstrides = (C.c_int * 4)(out.strides[0], 0, 0, 0)
        before = hooked.lib.ref_sws_pictures()
        assert hooked.lib.sws_scale(c, src, strides, 0, sh, dst, dstrides) == dh
        assert hooked.lib.ref_sws_pictures() == before + (0 if flip else 1)       # the bottom-up call stayed with the reference
        hooked.close(c)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])


def test_freed_contexts_give_their_device_side_back(hooked, monkeypatch):
    """sws_freeContext (wrapped: --wrap=sws_freeContext, what a statically linked caller gets) releases the context's device side; 300 contexts
    in a row neither leak nor run the table full"""
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    lib = hooked.lib
    lib.mi355_sws_glue_live_contexts.restype = C.c_int
    getattr(lib, "__wrap_sws_freeContext").argtypes = [C.c_void_p]
    name = "generic_64x48"
    base = lib.mi355_sws_glue_live_contexts()
    import numpy as np
    sw, sh, dw, dh = S.CONFIGS[name][:4]
    planes = S.picture(name)
    src = (C.c_void_p * 4)(*[p.ctypes.data for p in planes], None)
    strides = (C.c_int * 4)(*[p.strides[0] for p in planes], 0)
    out = np.zeros((dh, dw * 3), np.uint8)
    dst = (C.c_void_p * 4)(out.ctypes.data, None, None, None)
    dstrides = (C.c_int * 4)(out.strides[0], 0, 0, 0)
    for i in range(300):
        c = hooked.open(name)
        before = lib.ref_sws_pictures()
        assert lib.sws_scale(c, src, strides, 0, sh, dst, dstrides) == dh
        live = lib.mi355_sws_glue_live_contexts()
        # (the table is never full here: the contexts earlier tests of this process left behind — ctypes callers do not pass through the
        # wrapper — are far fewer than its 256 entries; a full table would leave this context unbound, never take an entry from another)
        assert lib.ref_sws_pictures() == before + 1 and base <= live <= base + 1
        getattr(lib, "__wrap_sws_freeContext")(C.c_void_p(c))
        assert lib.mi355_sws_glue_live_contexts() == live - 1
        base = live - 1
    out = hooked.scale(name, S.picture(name), dst_pad=8)                          # and the binding still takes pictures afterwards
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == GOLD["pictures"][name]


def test_full_table_leaves_new_contexts_to_the_reference(hooked, monkeypatch):
    """with every entry of the binding's table taken by a context that may be alive, a new context is NOT given another's entry (the other
    one's next call would have converted nothing and returned 0 lines): it stays with the reference's function.  270 live contexts, every one of
    them converts its picture — the ones past the table's size on the CPU — and the first ones still do afterwards."""
    monkeypatch.delenv("MI355_SWS_LINES", raising=False)
    lib = hooked.lib
    lib.mi355_sws_glue_live_contexts.restype = C.c_int
    getattr(lib, "__wrap_sws_freeContext").argtypes = [C.c_void_p]
    name = "generic_64x48"
    import numpy as np
    sw, sh, dw, dh = S.CONFIGS[name][:4]
    planes = S.picture(name)
    src = (C.c_void_p * 4)(*[p.ctypes.data for p in planes], None)
    strides = (C.c_int * 4)(*[p.strides[0] for p in planes], 0)

    def convert(c):
        out = np.zeros((dh, dw * 3), np.uint8)
        dst = (C.c_void_p * 4)(out.ctypes.data, None, None, None)
        dstrides = (C.c_int * 4)(out.strides[0], 0, 0, 0)
        assert lib.sws_scale(c, src, strides, 0, sh, dst, dstrides) == dh
        return hashlib.sha1(out.tobytes()).hexdigest()[:20]
    ctxs = [hooked.open(name) for _ in range(270)]
    try:
        on_device = lib.ref_sws_pictures()
        sums = [convert(c) for c in ctxs]
        assert len(set(sums)) == 1                                                   # device and reference pictures alike (bit-exact path)
        taken = lib.ref_sws_pictures() - on_device
        assert 0 < taken <= 256 and lib.mi355_sws_glue_live_contexts() <= 256      # the rest ran the reference's function
        assert [convert(c) for c in ctxs[:8]] == sums[:8]                          # nobody lost its entry to a later context
    finally:
        for c in ctxs:
            getattr(lib, "__wrap_sws_freeContext")(C.c_void_p(c))
