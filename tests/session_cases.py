"""Whole-frame sessions (include/mi355_h264_session.h: start_frame / decode_slice / end_frame over Tier 2).
 * a real stream: the 36 pictures of realshort.mp4 (records exported from the reference decoder's own run,
   tests/golden/h264_stream_realshort.npz) decoded IN SEQUENCE — each picture's reference is the surface the session decoded
   before, nothing is reloaded from the fixture — must equal the reference decoder's pictures;
 * synthetic pictures (B, weighted, several references and slices): references loaded with put_frame, result = the oracle's."""
import ctypes as C

import numpy as np

import h264_frames as HF
import stream_fixture as SF

MAX_SLOTS = HF.MAX_SLOTS


class SessionParams(C.Structure):
    _fields_ = [("mb_width", C.c_int32), ("mb_height", C.c_int32), ("num_surfaces", C.c_int32), ("max_slices", C.c_int32),
                ("surface_layout", C.c_int32), ("device", C.c_int32)]


class PictureParams(C.Structure):
    _fields_ = [("surface", C.c_int32), ("nslots", C.c_int32), ("ref_surface", C.c_int32 * MAX_SLOTS), ("two_lists", C.c_int32),
                ("field", C.c_int32), ("ref_parity", C.c_int32 * MAX_SLOTS)]


def _bind(lib):
    lib.mi355_h264_session_open.argtypes = [C.POINTER(C.c_void_p), C.POINTER(SessionParams)]
    lib.mi355_h264_session_close.argtypes = [C.c_void_p]
    lib.mi355_h264_session_close.restype = None
    lib.mi355_h264_start_frame.argtypes = [C.c_void_p, C.POINTER(PictureParams)]
    lib.mi355_h264_decode_slice.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi355_h264_end_frame.argtypes = [C.c_void_p]
    lib.mi355_h264_get_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.mi355_h264_put_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.mi355_h264_group_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.mi355_h264_group_destroy.argtypes = [C.c_void_p]
    lib.mi355_h264_group_destroy.restype = None
    lib.mi355_h264_group_flush.argtypes = [C.c_void_p]
    lib.mi355_h264_session_open_grouped.argtypes = [C.POINTER(C.c_void_p), C.POINTER(SessionParams), C.c_void_p]
    for f in ("session_open", "start_frame", "decode_slice", "end_frame", "get_frame", "put_frame", "group_create", "group_flush", "session_open_grouped"):
        getattr(lib, "mi355_h264_" + f).restype = C.c_int


class Group:
    def __init__(self, lib):
        _bind(lib)
        self.lib = lib
        self.h = C.c_void_p()
        assert lib.mi355_h264_group_create(C.byref(self.h)) == 0

    def flush(self):
        return self.lib.mi355_h264_group_flush(self.h)

    def destroy(self):
        self.lib.mi355_h264_group_destroy(self.h)


class Session:
    def __init__(self, lib, mb_w, mb_h, nsurf, max_slices=0, group=None, tiled=False, device=0):
        """tiled: the session keeps its surfaces macroblock-tiled (frame pictures only); device: 0 = the calling thread's, n = device n - 1"""
        _bind(lib)
        self.lib, self.mb_w, self.mb_h = lib, mb_w, mb_h
        self.h = C.c_void_p()
        p = SessionParams(mb_w, mb_h, nsurf, max_slices, 1 if tiled else 0, device)
        rc = lib.mi355_h264_session_open_grouped(C.byref(self.h), C.byref(p), group.h) if group else lib.mi355_h264_session_open(C.byref(self.h), C.byref(p))
        assert rc == 0, rc

    def close(self):
        self.lib.mi355_h264_session_close(self.h)

    def start(self, surface, refs, two_lists, field=0, ref_parity=()):
        pp = PictureParams()
        pp.surface, pp.nslots, pp.two_lists, pp.field = surface, len(refs), int(two_lists), field
        for i, r in enumerate(refs):
            pp.ref_surface[i] = r
        for i, rp in enumerate(ref_parity):
            pp.ref_parity[i] = rp
        return self.lib.mi355_h264_start_frame(self.h, C.byref(pp))

    def slice(self, hdr, first, mb, mv0, mv1, coef, addr=None):
        """hdr: one SLICE_DT record; mb / mv0 / mv1 / coef: the macroblocks of this call"""
        keep = [np.ascontiguousarray(a) for a in (hdr, mb, mv0, coef)]
        m1 = np.ascontiguousarray(mv1) if mv1 is not None else None
        ad = np.ascontiguousarray(addr, dtype=np.int32) if addr is not None else None
        return self.lib.mi355_h264_decode_slice(self.h, keep[0].ctypes.data, first, len(keep[1]), ad.ctypes.data if ad is not None else None,
                                                keep[1].ctypes.data, keep[2].ctypes.data, m1.ctypes.data if m1 is not None else None, keep[3].ctypes.data)

    def end(self):
        return self.lib.mi355_h264_end_frame(self.h)

    def planes(self, fn, surface, arrs):
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs])
        strides = (C.c_int * 3)(*[a.strides[0] for a in arrs])
        return fn(self.h, surface, ptrs, strides)

    def get(self, surface):
        H, W = 16 * self.mb_h, 16 * self.mb_w
        out = [np.full((H, W), 0xA5, np.uint8), np.full((H // 2, W // 2), 0xA5, np.uint8), np.full((H // 2, W // 2), 0xA5, np.uint8)]
        rc = self.planes(self.lib.mi355_h264_get_frame, surface, out)
        assert rc == 0, rc
        return out

    def put(self, surface, planes):
        arrs = [np.ascontiguousarray(p) for p in planes]
        rc = self.planes(self.lib.mi355_h264_put_frame, surface, arrs)
        assert rc == 0, rc


def send_picture(ss, mb, mv0, mv1, coef, slices, how):
    """the picture's macroblocks, slice by slice (records carry the slice index the exporter / generator gave them).
    how: 'runs' (first_mb + count), 'addr' (explicit addresses, in reverse order), 'split' (each slice in two calls: the second
    half becomes a slice of its own with the same header — same pictures)"""
    sid = mb["slice_id"]
    for k in range(len(slices)):
        idx = np.nonzero(sid == k)[0]
        if not len(idx):
            continue
        parts = [idx]
        if how == "split" and len(idx) > 1:
            parts = [idx[:len(idx) // 2], idx[len(idx) // 2:]]
        for part in parts:
            contiguous = bool((np.diff(part) == 1).all())
            m1 = mv1[part] if mv1 is not None else None
            if how == "addr" or not contiguous:
                order = part[::-1]
                m1 = mv1[order] if mv1 is not None else None
                rc = ss.slice(slices[k:k + 1], 0, mb[order], mv0[order], m1, coef[order], addr=order)
            else:
                rc = ss.slice(slices[k:k + 1], int(part[0]), mb[part], mv0[part], m1, coef[part])
            assert rc == 0, rc


def run_stream(prov, npz, first=0, count=None, nsurf=3, sync_each=True, tiled=False, device=0):
    pics = SF.load_npz(npz)
    count = len(pics) - first if count is None else count
    ss = Session(prov.lib, pics[0]["mb_w"], pics[0]["mb_h"], nsurf, tiled=tiled, device=device)
    try:
        if first > 0:      # join the stream in the middle: the references of the first picture come from the fixture
            for s_ in pics[first]["slots"]:
                r = pics[s_]
                ss.put(s_ % nsurf, (r["y"], r["cb"], r["cr"]))
        pending = []
        for i in range(first, first + count):
            pc = pics[i]
            assert all(0 < i - s_ < nsurf for s_ in pc["slots"])
            assert ss.start(i % nsurf, [s_ % nsurf for s_ in pc["slots"]], pc["use_l1"]) == 0
            send_picture(ss, pc["mb"], pc["mv0"].reshape(-1, 32), pc["mv1"].reshape(-1, 32) if pc["use_l1"] else None, pc["coef"], pc["slices"],
                         ("runs", "addr", "split")[i % 3])
            assert ss.end() == 0
            pending.append(i)
            if sync_each or len(pending) == nsurf - 1 or i == first + count - 1:
                # every picture still resident (a surface is overwritten nsurf pictures later)
                for j in pending:
                    got = ss.get(j % nsurf)
                    for g, key in zip(got, ("y", "cb", "cr")):
                        assert np.array_equal(g, pics[j][key]), "picture %d plane %s differs from the reference decoder" % (j, key)
                pending = []
    finally:
        ss.close()
    return count


def run_group(prov, npzs, nsurf=8, explicit_flush=True, tiled=()):
    """several streams (different picture sizes), one session each, all in ONE group: picture i of every stream that still has
    one goes out in the same launch set.  explicit_flush False: nothing calls group_flush — the next start_frame of a session
    whose picture still waits, and get_frame, flush by themselves."""
    streams = [SF.load_npz(p) for p in npzs]
    g = Group(prov.lib)
    sess = [Session(prov.lib, pics[0]["mb_w"], pics[0]["mb_h"], nsurf, group=g, tiled=k in tiled) for k, pics in enumerate(streams)]
    try:
        for i in range(max(len(p) for p in streams)):
            live = [(ss, pics) for ss, pics in zip(sess, streams) if i < len(pics)]
            for ss, pics in live:
                pc = pics[i]
                assert all(0 < i - s_ < nsurf for s_ in pc["slots"])
                assert ss.start(i % nsurf, [s_ % nsurf for s_ in pc["slots"]], pc["use_l1"]) == 0
                send_picture(ss, pc["mb"], pc["mv0"].reshape(-1, 32), pc["mv1"].reshape(-1, 32) if pc["use_l1"] else None, pc["coef"], pc["slices"],
                             ("runs", "addr", "split")[i % 3])
                assert ss.end() == 0
            if explicit_flush:
                assert g.flush() == 0
            if explicit_flush or i % 3 == 2 or i == max(len(p) for p in streams) - 1:
                lo = i if explicit_flush else max(0, i - 2)
                for ss, pics in live:
                    for j in range(lo, i + 1):
                        got = ss.get(j % nsurf)
                        for gp, key in zip(got, ("y", "cb", "cr")):
                            assert np.array_equal(gp, pics[j][key]), "picture %d plane %s differs from the reference decoder" % (j, key)
    finally:
        for ss in sess:
            ss.close()
        g.destroy()
    return sum(len(p) for p in streams)


def run_synth(prov, oracle, name, how="runs", tiled=False):
    import frame_cases
    fs = HF.synth_frames(**frame_cases.CASES[name])
    _, want = HF.run_oracle(oracle, fs)
    nref = fs.nrefs
    ss = Session(prov.lib, fs.mb_w, fs.mb_h, nref + 1, 8, tiled=tiled)
    try:
        for f in range(fs.F):
            for s_ in range(nref):
                ss.put(s_, fs.refs[f][s_])
            assert ss.start(nref, list(range(nref)), fs.use_l1) == 0
            send_picture(ss, fs.mb[f], fs.mv[0, f].reshape(-1, 32), fs.mv[1, f].reshape(-1, 32) if fs.use_l1 else None, fs.coef[f], fs.slices[f], how)
            assert ss.end() == 0
            got = ss.get(nref)
            for p in range(3):
                assert np.array_equal(got[p], want[p][f]), "%s picture %d plane %d" % (name, f, p)
    finally:
        ss.close()


def run_errors(prov):
    """state and argument checks; an incomplete picture is refused at end_frame and the session goes on"""
    pics = SF.load_npz(SF_NPZ)
    pc = pics[0]
    bad = C.c_void_p()
    _bind(prov.lib)
    assert prov.lib.mi355_h264_session_open(C.byref(bad), C.byref(SessionParams(pc["mb_w"], pc["mb_h"], 3, 0, 7, 0))) == -1   # no such layout
    if pc["mb_h"] % 2 == 0:
        st = Session(prov.lib, pc["mb_w"], pc["mb_h"], 3, tiled=True)
        try:
            assert st.start(0, [], False, field=1) == -1                 # tiled surfaces hold frame pictures only
        finally:
            st.close()
    ss = Session(prov.lib, pc["mb_w"], pc["mb_h"], 3)
    try:
        assert ss.end() == -1                                            # no open picture
        assert ss.start(5, [], False) == -1                              # no such surface
        assert ss.start(0, [1], False) == -1                             # reference never decoded
        assert ss.start(0, [], False) == 0
        assert ss.start(1, [], False) == -1                              # already open
        n = len(pc["mb"])
        assert ss.slice(pc["slices"][:1], 0, pc["mb"][:n // 2], pc["mv0"].reshape(-1, 32)[:n // 2], None, pc["coef"][:n // 2]) == 0
        assert ss.slice(pc["slices"][:1], n - 3, pc["mb"][:8], pc["mv0"].reshape(-1, 32)[:8], None, pc["coef"][:8]) == -1   # runs past the picture
        assert ss.end() == -4                                            # half the picture missing
        assert ss.start(0, [], False) == 0                               # ... and the session is usable
        send_picture(ss, pc["mb"], pc["mv0"].reshape(-1, 32), None, pc["coef"], pc["slices"], "runs")
        assert ss.end() == 0
        got = ss.get(0)
        assert np.array_equal(got[0], pc["y"]) and np.array_equal(got[1], pc["cb"]) and np.array_equal(got[2], pc["cr"])
    finally:
        ss.close()


import os
SF_NPZ = os.path.join(os.path.dirname(__file__), "golden", "h264_stream_realshort.npz")
