"""Output-stage chaining: pictures decoded by the batched H.264 path are converted to RGB24 by the
swscale path without leaving device memory — the deblocked planes of mi355_h264_frame are the mi355_sws_frame
sources.  Checked against oracle decode + oracle conversion."""
import ctypes as C

import numpy as np

import h264_frames as HF
import sws_support as S


def run(backend, oracle, nframes=3, mb_w=9, mb_h=6, seed=31):
    fs = HF.synth_frames(nframes=nframes, mb_w=mb_w, mb_h=mb_h, seed=seed, mix="mixed", intra_frac=0.2, dct8_frac=0.3, refs="smooth", coef_b=8)
    _, dst_o = HF.run_oracle(oracle, fs)
    W, H = 16 * mb_w, 16 * mb_h
    # the unscaled special converter for this picture size, with the LUTs the reference built
    base = S.load_context("special_64x48")
    ints = dict(base.ints, srcW=W, srcH=H, dstW=W, dstH=H, chrSrcW=W // 2, chrSrcH=H // 2, chrDstW=W // 2, unscaled_special=1)
    ctx = S.Context(ints, {k: (np.zeros(0, np.int16), np.zeros(0, np.int32)) for k in S.BANKS}, base.luts)
    want = [S.oracle_backend(oracle).scale(ctx, [dst_o[0][f], dst_o[1][f], dst_o[2][f]]) for f in range(nframes)]

    lib = backend.lib
    d = HF.DeviceFrames(backend, fs)
    lib.mi355_sws_create.restype = C.c_void_p
    lib.mi355_malloc.restype = C.c_void_p
    try:
        d.decode()
        rgb_stride = W * 3
        p_rgb = lib.mi355_malloc(C.c_size_t(nframes * rgb_stride * H + 64))
        frames = (S.SwsFrame * nframes)()
        for f in range(nframes):
            fr = d.host_desc[f]
            for p in range(3):
                frames[f].src[p] = fr.dst[p]
                frames[f].src_stride[p] = fr.dst_stride[0] if p == 0 else fr.dst_stride[1]
            frames[f].dst = p_rgb + f * rgb_stride * H
            frames[f].dst_stride = rgb_stride
        p_frames = lib.mi355_malloc(C.c_size_t(C.sizeof(frames)))
        lib.mi355_memcpy_h2d(C.c_void_p(p_frames), C.addressof(frames), C.c_size_t(C.sizeof(frames)))
        h = lib.mi355_sws_create(C.byref(ctx.desc))
        assert h
        lib.mi355_sws_scale_frames_dev(C.c_void_p(h), C.c_void_p(p_frames), nframes, None)
        lib.mi355_sync(None)
        got = np.empty((nframes, H, rgb_stride), np.uint8)
        lib.mi355_memcpy_d2h(C.c_void_p(got.ctypes.data), C.c_void_p(p_rgb), C.c_size_t(got.nbytes))
        lib.mi355_sws_destroy(C.c_void_p(h))
        lib.mi355_free(C.c_void_p(p_rgb))
        lib.mi355_free(C.c_void_p(p_frames))
    finally:
        d.free()
    for f in range(nframes):
        assert np.array_equal(got[f], want[f]), "RGB picture %d differs" % f
    return nframes


def run_session(backend, oracle, first=3, count=3, tiled=False):
    """The same chain through a whole-frame session (mi355_h264_session.h): pictures `first` .. of the real stream are decoded
    on the session's surfaces and converted from those surfaces (mi355_h264_surface_dev) by work enqueued on the session's
    stream right behind end_frame() — nothing waits in between.  tiled: the session keeps macroblock-tiled surfaces and the
    converter reads lines that mi355_h264_export_frame_dev() wrote (one launch per picture, same stream).  Expected: the reference decoder's picture through the
    oracle's conversion."""
    import session_cases as SC
    import stream_fixture as SF
    pics = SF.load_npz(SC.SF_NPZ)
    mb_w, mb_h = pics[0]["mb_w"], pics[0]["mb_h"]
    W, H = 16 * mb_w, 16 * mb_h
    base = S.load_context("special_64x48")
    ints = dict(base.ints, srcW=W, srcH=H, dstW=W, dstH=H, chrSrcW=W // 2, chrSrcH=H // 2, chrDstW=W // 2, unscaled_special=1)
    ctx = S.Context(ints, {k: (np.zeros(0, np.int16), np.zeros(0, np.int32)) for k in S.BANKS}, base.luts)
    lib = backend.lib
    lib.mi355_sws_create.restype = C.c_void_p
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_h264_surface_dev.restype = C.c_void_p
    lib.mi355_h264_surface_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.mi355_h264_session_stream.restype = C.c_void_p
    lib.mi355_h264_session_stream.argtypes = [C.c_void_p]
    lib.mi355_malloc.argtypes = [C.c_size_t]
    lib.mi355_free.argtypes = [C.c_void_p]
    lib.mi355_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.mi355_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    nsurf = count + 2
    ss = SC.Session(lib, mb_w, mb_h, nsurf, tiled=tiled)
    lib.mi355_h264_export_frame_dev.restype = C.c_int
    lib.mi355_h264_export_frame_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
    lys, lcs = (W + 63) & ~63, ((W + 63) & ~63) // 2
    lin_bytes = lys * H + 2 * lcs * (H // 2)
    p_lin = lib.mi355_malloc(C.c_size_t(count * lin_bytes)) if tiled else None
    rgb_stride = W * 3
    p_rgb = lib.mi355_malloc(C.c_size_t(count * rgb_stride * H + 64))
    p_frames = lib.mi355_malloc(C.c_size_t(C.sizeof(S.SwsFrame) * count))
    h = lib.mi355_sws_create(C.byref(ctx.desc))
    assert h and p_rgb and p_frames
    try:
        for s_ in pics[first]["slots"]:
            r = pics[s_]
            ss.put(s_ % nsurf, (r["y"], r["cb"], r["cr"]))
        stream = lib.mi355_h264_session_stream(ss.h)
        frames = (S.SwsFrame * count)()
        for k in range(count):
            i = first + k
            pc = pics[i]
            assert ss.start(i % nsurf, [s_ % nsurf for s_ in pc["slots"]], pc["use_l1"]) == 0
            SC.send_picture(ss, pc["mb"], pc["mv0"].reshape(-1, 32), None, pc["coef"], pc["slices"], "runs")
            assert ss.end() == 0
            if tiled:
                base_ = p_lin + k * lin_bytes
                planes = (C.c_void_p * 3)(base_, base_ + lys * H, base_ + lys * H + lcs * (H // 2))
                strides = (C.c_int * 3)(lys, lcs, lcs)
                assert lib.mi355_h264_export_frame_dev(ss.h, i % nsurf, planes, strides, None) == 0
                for p in range(3):
                    frames[k].src[p] = planes[p]
                    frames[k].src_stride[p] = strides[p]
            else:
                for p in range(3):
                    st = C.c_int(0)
                    frames[k].src[p] = lib.mi355_h264_surface_dev(ss.h, i % nsurf, p, C.byref(st))
                    frames[k].src_stride[p] = st.value
            frames[k].dst = p_rgb + k * rgb_stride * H
            frames[k].dst_stride = rgb_stride
        lib.mi355_memcpy_h2d(C.c_void_p(p_frames), C.addressof(frames), C.c_size_t(C.sizeof(frames)))
        assert lib.mi355_sws_scale_frames_dev(C.c_void_p(h), C.c_void_p(p_frames), count, C.c_void_p(stream)) == 0
        assert lib.mi355_sync(C.c_void_p(stream)) == 0
        got = np.empty((count, H, rgb_stride), np.uint8)
        lib.mi355_memcpy_d2h(C.c_void_p(got.ctypes.data), C.c_void_p(p_rgb), C.c_size_t(got.nbytes))
    finally:
        lib.mi355_sws_destroy(C.c_void_p(h))
        lib.mi355_free(C.c_void_p(p_rgb))
        lib.mi355_free(C.c_void_p(p_frames))
        if p_lin:
            lib.mi355_free(C.c_void_p(p_lin))
        ss.close()
    ob = S.oracle_backend(oracle)
    for k in range(count):
        pc = pics[first + k]
        assert np.array_equal(got[k], ob.scale(ctx, [pc["y"], pc["cb"], pc["cr"]])), "RGB picture %d differs" % (first + k)
    return count
