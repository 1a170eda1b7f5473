"""Field pictures at the Tier-2 API (mi355_h264_frame.field_picture, mi355_h264_mb.u.inter.chroma_dy): the pictures of a
FrameSet are decoded as FIELDS — picture f into the lines of parity f & 1 of a frame surface of twice the height (pointer at
the field's first line, strides doubled), each reference slot s a field of parity (s + f) & 1 of its own frame surface, the
chroma vector offset 2 * (parity - reference parity) in the records.  Device (emulated or real) against the oracle, whole frame
surfaces compared: the other field's lines must come back untouched.  (Against the REFERENCE decoder field pictures are pinned
through the bridge on generated PAFF streams: tests/test_synth_streams*.py.)"""
import ctypes as C

import numpy as np

import h264_frames as HF
from rng import SplitMix64


def _surfaces(fs, f, r, par=None, dst=None):
    """frame surfaces (double height) for picture f: destination (random), references (the FrameSet's pictures on one parity).
    dst given: the field goes into THAT frame, and reference slot 0 is the frame's other field (second field of a frame
    predicting from the first)"""
    H, W = fs.H, fs.W
    par = f & 1 if par is None else par
    own = dst is not None
    if not own:
        dst = [r.u8((2 * H, W)), r.u8((H, W // 2)), r.u8((H, W // 2))]
    refs, rpar = [], []
    for s_ in range(fs.nrefs):
        if own and s_ == 0:
            refs.append(dst)
            rpar.append(1 - par)
            continue
        rp = (s_ + f) & 1
        planes = []
        for p in range(3):
            h = H if p == 0 else H // 2
            a = r.u8((2 * h, W if p == 0 else W // 2))
            a[rp::2] = fs.refs[f][s_][p]
            planes.append(a)
        refs.append(planes)
        rpar.append(rp)
    return par, dst, refs, rpar


def _records(fs, f, par, rpar):
    """the picture's records with chroma_dy filled in (inter macroblocks: ref_pic[2][4] at bytes 0..7 of the union, chroma_dy at 8..15)"""
    mb = fs.mb[f].copy()
    inter = (mb["mb_type"] & 7) == 0
    slot = mb["i4mode"][:, :8].astype(np.int64)
    used = slot >= 0
    dy = np.where(used, 2 * (par - np.array(rpar + [0] * 256)[np.clip(slot, 0, 255)]), 0).astype(np.int8)
    mb["i4mode"][inter, 8:] = dy[inter]
    return mb


def _frame(fs, f, par, mb_ptr, mv0, mv1, coef, slices, ilist, istart, dst, recon, refs, rpar, ys, cs):
    fr = HF.Frame()
    fr.mb_width, fr.mb_height = fs.mb_w, fs.mb_h
    for p in range(3):
        st = ys if p == 0 else cs
        fr.dst[p] = dst[p] + par * st
        fr.recon[p] = recon[p]
    fr.dst_stride[0], fr.dst_stride[1] = 2 * ys, 2 * cs
    fr.recon_stride[0], fr.recon_stride[1] = ys, cs
    for s_ in range(fs.nrefs):
        for p in range(3):
            fr.ref[s_][p] = refs[s_][p] + rpar[s_] * (ys if p == 0 else cs)
    fr.mb, fr.coef, fr.slices = mb_ptr, coef, slices
    fr.mv[0] = mv0
    fr.mv[1] = mv1
    fr.nslices = fs.slices.shape[1]
    fr.max_intra_level = int(fs.intra_start[f].shape[0]) - 1
    fr.intra_list, fr.intra_level_start = ilist, istart
    fr.max_level_width = fs.max_level_width
    fr.reserved = 1                                          # field_picture
    return fr


def oracle_picture(fs, f, olib, r, par=None, dst=None):
    """picture f as a field: inputs and what the oracle makes of them (whole frame planes)"""
    ys, cs = fs.W, fs.W // 2
    par, dst0, refs, rpar = _surfaces(fs, f, r, par, dst)
    mb = _records(fs, f, par, rpar)
    il = fs.intra_list[f] if len(fs.intra_list[f]) else np.zeros(1, np.uint32)
    odst = dst0 if dst is not None else [a.copy() for a in dst0]       # a frame's second field: decoded in place, next to the first
    if dst is not None:
        refs = [odst if pl is dst0 else pl for pl in refs]
    orec = [np.zeros((fs.H, ys), np.uint8), np.zeros((fs.H // 2, cs), np.uint8), np.zeros((fs.H // 2, cs), np.uint8)]
    ofr = _frame(fs, f, par, mb.ctypes.data, fs.mv[0, f].ctypes.data, fs.mv[1, f].ctypes.data if fs.use_l1 else None, fs.coef[f].ctypes.data,
                 fs.slices[f].ctypes.data, il.ctypes.data, fs.intra_start[f].ctypes.data, [a.ctypes.data for a in odst], [a.ctypes.data for a in orec],
                 [[a.ctypes.data for a in pl] for pl in refs], rpar, ys, cs)
    olib.oracle_h264_recon_frame(C.byref(ofr))
    olib.oracle_h264_deblock_frame(C.byref(ofr))
    return par, dst0, refs, rpar, mb, il, orec, odst


def run(backend, oracle, fs, seed=7):
    lib, olib = backend.lib, oracle.lib
    olib.oracle_h264_recon_frame.restype = None
    olib.oracle_h264_deblock_frame.restype = None
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    lib.mi355_free.argtypes = [C.c_void_p]
    lib.mi355_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.mi355_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn = lib.mi355_h264_decode_frames_levels_dev
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    r = SplitMix64(seed)
    ys, cs = fs.W, fs.W // 2
    changed = 0
    for f in range(fs.F):
        par, dst0, refs, rpar, mb, il, orec, odst = oracle_picture(fs, f, olib, r)
        allocs = []

        def up(a):
            a = np.ascontiguousarray(a)
            p = lib.mi355_malloc(max(a.nbytes, 64))
            assert p and lib.mi355_memcpy_h2d(p, a.ctypes.data, a.nbytes) == 0
            allocs.append(p)
            return p
        ddst = [up(a) for a in dst0]
        drec = [up(np.zeros_like(a)) for a in orec]
        dfr = _frame(fs, f, par, up(mb), up(fs.mv[0, f]), up(fs.mv[1, f]) if fs.use_l1 else None, up(fs.coef[f]), up(fs.slices[f]), up(il), up(fs.intra_start[f]),
                     ddst, drec, [[up(a) for a in pl] for pl in refs], rpar, ys, cs)
        d_desc = up(np.frombuffer(bytes(dfr), np.uint8))
        levels = dfr.max_intra_level
        widths = np.diff(fs.intra_start[f]).astype(np.int32) if levels else np.zeros(1, np.int32)
        assert fn(d_desc, 1, fs.mb_w, fs.mb_h, levels, widths.ctypes.data, None) == 0
        assert lib.mi355_sync(None) == 0
        for p in range(3):
            got = np.empty_like(dst0[p])
            lib.mi355_memcpy_d2h(got.ctypes.data, ddst[p], got.nbytes)
            assert np.array_equal(got[1 - par::2], dst0[p][1 - par::2]), "picture %d plane %d: the other field's lines were touched" % (f, p)
            assert np.array_equal(got, odst[p]), "picture %d plane %d differs from the oracle" % (f, p)
            changed += int((got[par::2] != dst0[p][par::2]).sum())
        for a in allocs:
            lib.mi355_free(a)
    return changed


def run_session(backend, oracle, fs, seed=9, how="runs"):
    """the same through a whole-frame session (mi355_h264_session.h, picture parameter `field`): references and the frame the
    field goes into are loaded with put_frame, the field is decoded with start_frame / decode_slice / end_frame"""
    import session_cases as SC
    olib = oracle.lib
    olib.oracle_h264_recon_frame.restype = None
    olib.oracle_h264_deblock_frame.restype = None
    r = SplitMix64(seed)
    nref = fs.nrefs
    ss = SC.Session(backend.lib, fs.mb_w, 2 * fs.mb_h, nref + 1, 8)
    try:
        for f in range(fs.F):
            par, dst0, refs, rpar, mb, il, orec, odst = oracle_picture(fs, f, olib, r)
            for s_ in range(nref):
                ss.put(s_, refs[s_])
            ss.put(nref, dst0)
            assert ss.start(nref, list(range(nref)), fs.use_l1, field=1 + par, ref_parity=rpar) == 0
            SC.send_picture(ss, mb, fs.mv[0, f].reshape(-1, 32), fs.mv[1, f].reshape(-1, 32) if fs.use_l1 else None, fs.coef[f], fs.slices[f], how)
            assert ss.end() == 0
            got = ss.get(nref)
            for p in range(3):
                assert np.array_equal(got[p], odst[p]), "picture %d plane %d differs from the oracle" % (f, p)
    finally:
        ss.close()
    return fs.F


def run_session_pairs(backend, oracle, fs, seed=11):
    """frames as field PAIRS in one surface: picture 2k is decoded as the top field of surface `nref`, picture 2k + 1 as its
    bottom field with reference slot 0 = the top field just decoded (same surface, other parity)"""
    import session_cases as SC
    olib = oracle.lib
    olib.oracle_h264_recon_frame.restype = None
    olib.oracle_h264_deblock_frame.restype = None
    r = SplitMix64(seed)
    nref = fs.nrefs
    ss = SC.Session(backend.lib, fs.mb_w, 2 * fs.mb_h, nref + 1, 8)
    try:
        for k in range(fs.F // 2):
            frame = None
            for par in (0, 1):
                f = 2 * k + par
                _, dst0, refs, rpar, mb, il, orec, odst = oracle_picture(fs, f, olib, r, par=par, dst=frame)
                for s_ in range(nref):
                    if not (par == 1 and s_ == 0):
                        ss.put(s_, refs[s_])
                if par == 0:
                    ss.put(nref, dst0)
                surf = [nref if (par == 1 and s_ == 0) else s_ for s_ in range(nref)]
                assert ss.start(nref, surf, fs.use_l1, field=1 + par, ref_parity=rpar) == 0
                SC.send_picture(ss, mb, fs.mv[0, f].reshape(-1, 32), fs.mv[1, f].reshape(-1, 32) if fs.use_l1 else None, fs.coef[f], fs.slices[f], "runs")
                assert ss.end() == 0
                frame = odst
            got = ss.get(nref)
            for p in range(3):
                assert np.array_equal(got[p], frame[p]), "frame %d plane %d differs from the oracle" % (k, p)
    finally:
        ss.close()
    return fs.F // 2
