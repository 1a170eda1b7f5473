"""Tier-2 parity cases: synthetic pictures decoded by a backend (GPU or emulator) through
the C ABI vs the CPU oracle, compared sample for sample (bit-exact)."""
import numpy as np

import h264_frames as HF

# name -> synth_frames kwargs.  Sizes are small enough for the oracle to finish in seconds.
CASES = {
    "p16_noise":        dict(nframes=2, mb_w=8, mb_h=5, seed=0x264),
    "p16_smooth":       dict(nframes=2, mb_w=8, mb_h=5, seed=11, refs="smooth", coef_b=6, offsets=True),
    "mixed_dct8":       dict(nframes=2, mb_w=7, mb_h=5, seed=12, mix="mixed", dct8_frac=0.3, refs="smooth", coef_b=8),
    "mixed_intra":      dict(nframes=2, mb_w=7, mb_h=6, seed=13, mix="mixed", intra_frac=0.3, dct8_frac=0.3,
                             pcm_frac=0.05, offsets=True, refs="smooth", coef_b=8),
    "all_intra":        dict(nframes=1, mb_w=6, mb_h=5, seed=14, intra_frac=1.0, pcm_frac=0.05, coef_b=10),
    "b_mixed":          dict(nframes=2, mb_w=6, mb_h=4, seed=15, mix="mixed", bframes=True, intra_frac=0.1, refs="smooth", coef_b=6),
    "b_weight_explicit": dict(nframes=1, mb_w=6, mb_h=4, seed=16, mix="mixed", bframes=True, weighted=1),
    "b_weight_implicit": dict(nframes=1, mb_w=6, mb_h=4, seed=17, mix="mixed", bframes=True, weighted=2, refs="smooth"),
    "p_weight_farmv":   dict(nframes=1, mb_w=5, mb_h=4, seed=18, mix="mixed", weighted=1, mv_range=300),
    "one_mb":           dict(nframes=3, mb_w=1, mb_h=1, seed=19, mix="mixed", intra_frac=0.3),
    "one_row":          dict(nframes=1, mb_w=9, mb_h=1, seed=20, mix="mixed", intra_frac=0.3, refs="smooth"),
    "one_col":          dict(nframes=1, mb_w=1, mb_h=7, seed=21, mix="mixed", intra_frac=0.3, refs="smooth"),
    # several 8-MB chunks and 4-row bands with remainders on both axes, slices switching deblocking off / offsets
    "wide_mixed":       dict(nframes=2, mb_w=37, mb_h=9, seed=22, mix="mixed", intra_frac=0.15, dct8_frac=0.3, refs="smooth",
                             coef_b=8, offsets=True),
    "wide_b":           dict(nframes=1, mb_w=19, mb_h=6, seed=23, mix="mixed", bframes=True, intra_frac=0.1, refs="smooth", coef_b=6),
    # ~900 macroblocks each: enough draws to reach the rare value combinations (large coefficients, far vectors)
    "mid_b_bigcoef":    dict(nframes=1, mb_w=40, mb_h=22, seed=102, mix="mixed", bframes=True, intra_frac=0.15, dct8_frac=0.3, coef_b=200),
    "mid_b_weighted":   dict(nframes=1, mb_w=40, mb_h=22, seed=103, mix="mixed", bframes=True, weighted=1, intra_frac=0.1, coef_b=100, mv_range=200),
    "mid_intra_pcm":    dict(nframes=1, mb_w=33, mb_h=19, seed=105, intra_frac=1.0, pcm_frac=0.05, dct8_frac=0.4, coef_b=300),
    # 268 intra dependency levels: more than the record's 8-bit intra_level field can hold (an all-intra picture
    # above 1080p does this); the schedule must not wrap
    "tall_all_intra":   dict(nframes=1, mb_w=10, mb_h=130, seed=107, intra_frac=1.0, pcm_frac=0.02, coef_b=10),
    # levels up to +-32767: first-pass values beyond +-8191 (k_recon_inter's 16-bit second pass hands the macroblock to the
    # int form) and the reference's own int16 wrap of the first pass
    "mid_wrapcoef":     dict(nframes=1, mb_w=20, mb_h=9, seed=108, mix="mixed", intra_frac=0.2, dct8_frac=0.2, coef_b=6000, coef_clip=32767),
    "p16_wrapcoef":     dict(nframes=1, mb_w=12, mb_h=5, seed=109, intra_frac=0.1, coef_b=3000, coef_clip=32767),
    "mid_hugecoef":     dict(nframes=1, mb_w=33, mb_h=19, seed=106, mix="mixed", intra_frac=0.5, dct8_frac=0.4, coef_b=1000),
}


def run_case(backend, oracle, name, pad=0, per_level=True, sparse=False, tiled=False, by_layout=False, wide=False):
    fs = HF.synth_frames(**CASES[name])
    if sparse:
        # real P / B pictures: many inter macroblocks carry no residual at all (cbp 0) — about every other one here
        inter = (fs.mb["mb_type"] & 7) == 0
        pick = inter & (np.random.default_rng(len(name)).random(inter.shape) < 0.5)
        fs.mb["cbp"][pick] = 0
        fs.mb["nnz_mask"][pick] = 0
        fs.coef[pick] = 0
        assert pick.any() or not inter.any()
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    if wide:
        # the second kernel set takes an I_PCM macroblock's samples one per coefficient slot (mi355_h264_frame.h), not as 384 bytes
        pcm = (fs.mb["mb_type"] & 4) != 0
        fs.coef[pcm] = fs.coef[pcm].view(np.uint8)[:, :384].astype(np.int16)
    d = HF.DeviceFrames(backend, fs, pad=pad, tiled=tiled)
    try:
        if sparse:
            d.decode_sparse()
        elif by_layout:
            d.decode_by_layout()
        elif wide:
            d.decode_wide()                  # the second kernel set (h264_frame_wide.hip) on the pictures the first one decodes
        else:
            d.decode(per_level=per_level)
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p]), "%s: reconstruction differs in plane %d" % (name, p)
        assert np.array_equal(dst_o[p], dst_g[p]), "%s: deblocked picture differs in plane %d" % (name, p)
    return sum(int((a != b).sum()) for a, b in zip(dst_o, recon_o))   # samples the loop filter changed


def smoke(gpu, oracle):
    assert run_case(gpu, oracle, "mixed_intra") > 0


def run_mixed_batch(backend, oracle, names=("mixed_intra", "wide_b", "one_col", "tall_all_intra"), tiled=()):
    """pictures of DIFFERENT geometry in ONE call (what the bridge's dispatcher produces when streams of different size
    decode at the same time): the descriptors of several cases are concatenated on the device and go through
    mi355_h264_decode_frames_levels_dev with the largest width / height and the per-level maxima; every picture must
    come out as in its own single-geometry run (= the oracle's)."""
    import ctypes as C
    sets = [HF.synth_frames(**CASES[n]) for n in names]
    devs = [HF.DeviceFrames(backend, fs, tiled=n in tiled) for n, fs in zip(names, sets)]      # layouts may be mixed in one call
    lib = backend.lib
    try:
        fsz = C.sizeof(devs[0].host_desc) // devs[0].F
        total = sum(d.F for d in devs)
        lib.mi355_malloc.restype = C.c_void_p
        lib.mi355_malloc.argtypes = [C.c_size_t]
        d_all = lib.mi355_malloc(total * fsz)
        assert d_all
        off = 0
        for d in devs:
            assert lib.mi355_memcpy_h2d(C.c_void_p(d_all + off), C.c_void_p(C.addressof(d.host_desc)), C.c_size_t(d.F * fsz)) == 0
            off += d.F * fsz
        mw, mh = max(fs.mb_w for fs in sets), max(fs.mb_h for fs in sets)
        ml = max(fs.max_intra_level for fs in sets)
        widths = [0] * max(1, ml)
        for fs in sets:
            for i, w in enumerate(fs.level_widths[:fs.max_intra_level]):
                widths[i] = max(widths[i], w)
        lw = (C.c_int32 * len(widths))(*widths)
        fn = lib.mi355_h264_decode_frames_levels_dev
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        assert fn(d_all, total, mw, mh, ml, lw, None) == 0
        assert lib.mi355_sync(None) == 0
        for name, fs, d in zip(names, sets, devs):
            recon_o, dst_o = HF.run_oracle(oracle, fs)
            recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
            for p in range(3):
                assert np.array_equal(recon_o[p], recon_g[p]), "%s: reconstruction differs in plane %d" % (name, p)
                assert np.array_equal(dst_o[p], dst_g[p]), "%s: deblocked picture differs in plane %d" % (name, p)
        lib.mi355_free(C.c_void_p(d_all))
    finally:
        for d in devs:
            d.free()
    return total


def run_surface_convert(backend, cases=((5, 3, 0, 0), (1, 1, 0, 256), (9, 4, 24, 512), (120, 68, 0, 0))):
    """mi355_h264_surface_convert_dev against the numpy statement of the tiled layout (HF.tile_planes), both directions, several
    pictures of different size in one launch; (mb_w, mb_h, linear pad, tile-row pad)"""
    import ctypes as C
    lib = backend.lib

    class Job(C.Structure):
        _fields_ = [("lin", C.c_void_p * 3), ("tiled", C.c_void_p * 2), ("lin_stride", C.c_int32 * 2), ("tiled_stride", C.c_int32 * 2),
                    ("mb_width", C.c_int32), ("mb_height", C.c_int32), ("to_tiled", C.c_int32), ("reserved0", C.c_int32)]
    assert C.sizeof(Job) == 72
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    lib.mi355_free.argtypes = [C.c_void_p]
    lib.mi355_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.mi355_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn = lib.mi355_h264_surface_convert_dev
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(7)
    bufs, jobs_fwd, jobs_back, meta = [], [], [], []

    def dev(nbytes, init=None):
        p = lib.mi355_malloc(nbytes)
        assert p
        bufs.append(p)
        if init is not None:
            init = np.ascontiguousarray(init)
            assert lib.mi355_memcpy_h2d(p, init.ctypes.data, init.nbytes) == 0
        return p
    try:
        for (mw, mh, lpad, tpad) in cases:
            W, H = 16 * mw, 16 * mh
            ys, cs = W + lpad, W // 2 + lpad // 2
            lin = [np.zeros((H, ys), np.uint8), np.zeros((H // 2, cs), np.uint8), np.zeros((H // 2, cs), np.uint8)]
            planes = [rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8),
                      rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8)]
            for a, b in zip(lin, planes):
                a[:, :b.shape[1]] = b
            d_lin = [dev(a.nbytes, a) for a in lin]
            tys, tcs = mw * 256 + tpad, mw * 128 + tpad // 2
            d_t = [dev(mh * tys, np.full(mh * tys, 0xA5, np.uint8)), dev(mh * tcs, np.full(mh * tcs, 0xA5, np.uint8))]
            d_back = [dev(a.nbytes, np.full(a.shape, 0x5A, np.uint8)) for a in lin]
            for to_tiled, lins, lst in ((1, d_lin, jobs_fwd), (0, d_back, jobs_back)):
                j = Job()
                for k in range(3):
                    j.lin[k] = lins[k]
                j.tiled[0], j.tiled[1] = d_t
                j.lin_stride[0], j.lin_stride[1] = ys, cs
                j.tiled_stride[0], j.tiled_stride[1] = tys, tcs
                j.mb_width, j.mb_height, j.to_tiled = mw, mh, to_tiled
                lst.append(j)
            meta.append((mw, mh, ys, cs, tys, tcs, planes, d_t, d_back))
        maxw, maxh = max(c[0] for c in cases), max(c[1] for c in cases)
        for lst in (jobs_fwd, jobs_back):
            arr = (Job * len(lst))(*lst)
            d_jobs = dev(C.sizeof(arr))
            assert lib.mi355_memcpy_h2d(d_jobs, C.addressof(arr), C.sizeof(arr)) == 0
            assert fn(d_jobs, len(lst), maxw, maxh, None) == 0
            assert lib.mi355_sync(None) == 0
        for (mw, mh, ys, cs, tys, tcs, planes, d_t, d_back) in meta:
            ty, tc = HF.tile_planes(*planes)
            got_y, got_c = np.empty(mh * tys, np.uint8), np.empty(mh * tcs, np.uint8)
            lib.mi355_memcpy_d2h(got_y.ctypes.data, d_t[0], got_y.nbytes)
            lib.mi355_memcpy_d2h(got_c.ctypes.data, d_t[1], got_c.nbytes)
            gy, gc = got_y.reshape(mh, tys)[:, :mw * 256].reshape(-1), got_c.reshape(mh, tcs)[:, :mw * 128].reshape(-1)
            assert np.array_equal(gy, ty), "luma tiles differ (%d x %d, strides %d / %d): first at byte %d" % (mw, mh, ys, tys, int(np.argmax(gy != ty)))
            assert np.array_equal(gc, tc), "chroma tiles differ (%d x %d, strides %d / %d): %d bytes, first at %d: %s" % (
                mw, mh, cs, tcs, int((gc != tc).sum()), int(np.argmax(gc != tc)), np.flatnonzero(gc != tc)[:24].tolist())
            assert (got_y.reshape(mh, tys)[:, mw * 256:] == 0xA5).all()          # padding untouched
            for k, (st, pl) in enumerate(zip((ys, cs, cs), planes)):
                back = np.empty((pl.shape[0], st), np.uint8)
                lib.mi355_memcpy_d2h(back.ctypes.data, d_back[k], back.nbytes)
                assert np.array_equal(back[:, :pl.shape[1]], pl)
                assert (back[:, pl.shape[1]:] == 0x5A).all()
    finally:
        for p in bufs:
            lib.mi355_free(p)
    return len(cases)


def run_fast_workload_by_layout(backend, oracle, nframes, mb_w, mb_h, seed, replicate=None, pipelined=None, **kw):
    """the bench generator's pictures through the single-layout entry points on tiled surfaces (k_recon_inter_tiled: the run kernel of
    h264_recon_fast.h), every sample of both surfaces of every picture against the oracle"""
    fs = HF.synth_frames_fast(nframes, mb_w, mb_h, seed=seed, lib=backend.lib, **kw)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    F = replicate or nframes
    d = HF.DeviceFrames(backend, fs, tiled=True, replicate=replicate) if replicate else HF.DeviceFrames(backend, fs, tiled=True)
    try:
        if pipelined:
            d.decode_pipelined(*pipelined)                 # (shares, turns, calls): mi355_h264_pipelines_*
        else:
            d.decode_by_layout()
        for first in range(0, F, 32):
            n = min(32, F - first)
            recon_g, dst_g = (d.fetch(d.recon, first, n), d.fetch(d.dst, first, n)) if replicate else (d.fetch(d.recon), d.fetch(d.dst))
            for i in range(n):
                g = (first + i) % fs.F
                for p in range(3):
                    assert np.array_equal(recon_o[p][g], recon_g[p][i]), "picture %d: reconstruction differs in plane %d" % (first + i, p)
                    assert np.array_equal(dst_o[p][g], dst_g[p][i]), "picture %d: deblocked picture differs in plane %d" % (first + i, p)
    finally:
        d.free()


def run_case_hbd(backend, oracle, name, bit_depth=10, fs=None, replicate=None, idc=1):
    """A case (or a given picture set) as a High 10 / 9-bit batch through the SECOND kernel set (mi355_h264_decode_frames_wide_dev) against the
    frame-level checker above 8 bits — oracle/oracle_h264frame_hbd.c: the restated macroblock drivers on the reference's own tables of that bit
    depth (oracle/_ref/libref.so) — every sample of both surfaces.  Returns False when libref.so is not there."""
    fs = HF.synth_frames(**CASES[name]) if fs is None else fs
    ref = HF.run_oracle_hbd(oracle, fs, bit_depth, idc=idc)      # idc 2: the 4:2:2 variant of the set (chroma planes of the luma's height, eight chroma blocks a plane)
    if ref is None:
        return False
    recon_o, dst_o = ref
    d = HF.DeviceFrames(backend, fs, bit_depth=bit_depth, replicate=replicate, idc=idc)
    try:
        d.decode_wide(bit_depth=bit_depth, idc=idc)
        n = d.F
        for first in range(0, n, 16):
            cnt = min(16, n - first)
            recon_g, dst_g = d.fetch(d.recon, first, cnt), d.fetch(d.dst, first, cnt)
            for i in range(cnt):
                g = (first + i) % fs.F
                for p in range(3):
                    assert np.array_equal(recon_o[p][g], recon_g[p][i]), "%s at %d bits (chroma_format_idc %d), picture %d: reconstruction differs in plane %d" % (name, bit_depth, idc, first + i, p)
                    assert np.array_equal(dst_o[p][g], dst_g[p][i]), "%s at %d bits, picture %d: deblocked picture differs in plane %d" % (name, bit_depth, first + i, p)
    finally:
        d.free()
    return True
