"""CPU (emulated kernels): one host process, several GPUs — the emulator pretends to have two devices (MI355_EMU_DEVICES=2, one
address space, launches and allocations counted per device).  A context lives on ONE device and its entry points take the calling
thread there: two sessions on different devices driven from the same thread decode the same pictures, each device sees its share of
the launches, and the thread's own device is what it was afterwards (the reference's model is one process with
a thread per stream, pthread_frame.c:502-541)."""
import ctypes as C

import numpy as np

import session_cases as SC
import sws_support as S


def _stats(lib, dev):
    out = (C.c_long * 2)()
    lib.simt_emu_device_stats(dev, out)
    return out[0], out[1]


def test_two_devices_in_one_process_emulated(emu, monkeypatch):
    lib = emu.lib
    monkeypatch.setenv("MI355_EMU_DEVICES", "2")
    for f in ("mi355_device_count", "mi355_get_device", "mi355_set_device"):
        getattr(lib, f).restype = C.c_int
    assert lib.mi355_device_count() == 2 and lib.mi355_get_device() == 0
    assert lib.mi355_set_device(2) < 0                                    # no such device
    before = [_stats(lib, d) for d in (0, 1)]
    # the same stream on device 0 (the thread's) and on device 1 (named in the parameters): both give the reference decoder's pictures
    assert SC.run_stream(emu, SC.SF_NPZ, 0, 4, tiled=True) == 4
    mid = [_stats(lib, d) for d in (0, 1)]
    assert SC.run_stream(emu, SC.SF_NPZ, 0, 4, tiled=True, device=2) == 4
    after = [_stats(lib, d) for d in (0, 1)]
    assert lib.mi355_get_device() == 0                                     # the thread is back on its own device
    l0, l1 = mid[0][0] - before[0][0], after[1][0] - mid[1][0]
    assert l0 > 0 and l0 == l1, (before, mid, after)                      # the same launches, on the other device
    assert mid[1] == before[1] and after[0] == mid[0]                      # and nothing on the wrong one
    assert after[1][1] - mid[1][1] == mid[0][1] - before[0][1]             # the session's memory too
    # a thread that moves to device 1: its contexts are made there; a context made on device 0 is still served there
    assert lib.mi355_set_device(1) == 0 and lib.mi355_get_device() == 1
    try:
        b = _stats(lib, 1)[0]
        assert SC.run_stream(emu, SC.SF_NPZ, 0, 2) == 2
        assert _stats(lib, 1)[0] > b and _stats(lib, 0) == after[0]
        # sessions of a group must live on the group's device
        g = SC.Group(lib)
        try:
            pics = SC.SF.load_npz(SC.SF_NPZ)
            h = C.c_void_p()
            p = SC.SessionParams(pics[0]["mb_w"], pics[0]["mb_h"], 3, 0, 0, 1)        # device 0 named, the group is on 1
            assert lib.mi355_h264_session_open_grouped(C.byref(h), C.byref(p), g.h) == -1
        finally:
            g.destroy()
    finally:
        assert lib.mi355_set_device(-1) == 0 and lib.mi355_get_device() == 0


def test_swscale_context_stays_on_its_device_emulated(emu, monkeypatch):
    lib = emu.lib
    monkeypatch.setenv("MI355_EMU_DEVICES", "2")
    lib.mi355_set_device.restype = C.c_int
    name = "generic_64x48"
    ctx = S.load_context(name)
    planes = S.picture(name)
    lib.mi355_sws_create.restype = C.c_void_p
    assert lib.mi355_set_device(1) == 0
    try:
        h = lib.mi355_sws_create(C.byref(ctx.desc))
        assert h
    finally:
        assert lib.mi355_set_device(-1) == 0
    b0, b1 = _stats(lib, 0)[0], _stats(lib, 1)[0]
    out = np.full((ctx.desc.dstH, ctx.desc.dstW * 3 + 8), 0x5A, np.uint8)       # the golden pictures carry 8 bytes of row padding
    src = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    st = (C.c_int * 3)(*[p.strides[0] for p in planes])
    lib.mi355_sws_scale.restype = C.c_int
    assert lib.mi355_sws_scale(C.c_void_p(h), src, st, C.c_void_p(out.ctypes.data), out.strides[0]) == ctx.desc.dstH
    lib.mi355_sws_destroy(C.c_void_p(h))
    assert _stats(lib, 1)[0] > b1 and _stats(lib, 0)[0] == b0              # called from a device-0 thread, ran on device 1
    import hashlib, json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sws_ref_sha1.json")))
    assert hashlib.sha1(out.tobytes()).hexdigest()[:20] == gold["pictures"][name]
