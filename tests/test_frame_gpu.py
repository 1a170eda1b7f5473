"""GPU: Tier-2 batched reconstruction + deblocking through the C ABI vs the oracle, bit-exact."""
import numpy as np
import pytest

import frame_cases
import h264_frames as HF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_gpu(mi355, oracle, name):
    frame_cases.run_case(mi355, oracle, name)


def test_frame_pipeline_gpu_is_deterministic(mi355):
    fs = HF.synth_frames(nframes=4, mb_w=20, mb_h=12, seed=99, mix="mixed", intra_frac=0.1, refs="smooth", coef_b=6)
    outs = []
    for _ in range(2):
        d = HF.DeviceFrames(mi355, fs)
        d.decode()
        outs.append(d.fetch(d.dst))
        d.free()
    for p in range(3):
        assert np.array_equal(outs[0][p], outs[1][p])


def test_full_size_1080p_batch_matches_oracle(mi355, oracle):
    """BASELINE.json config 2 at its real size: three 1080p P pictures (the bench generator: 16x16
    partitions, random quarter-pel vectors over 4 references, ~50 % coded blocks, 5 % Intra16x16),
    bit-exact against the oracle on every sample of both surfaces."""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x264, lib=mi355.lib)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])
    assert sum(int((a != b).sum()) for a, b in zip(dst_o, recon_o)) > 10000      # the loop filter did real work


def test_decode_then_convert_on_device(mi355, oracle):
    """H.264 Tier-2 output feeds the swscale kernels without a host round trip"""
    import chain_check
    assert chain_check.run(mi355, oracle, nframes=4, mb_w=20, mb_h=12, seed=32) == 4


@pytest.mark.parametrize("pad", (8, 24))
@pytest.mark.parametrize("name", ("mixed_intra", "wide_mixed", "mid_b_weighted"))
def test_frame_pipeline_gpu_unaligned_strides(mi355, oracle, name, pad):
    frame_cases.run_case(mi355, oracle, name, pad=pad)


def test_all_intra_picture_above_1080p(mi355, oracle):
    """an all-intra 2560x1440 picture has 338 dependency levels (> 255): every macroblock must still be reconstructed
    after its neighbours"""
    fs = HF.synth_frames_fast(1, 160, 90, seed=7, intra_frac=1.0, lib=mi355.lib)
    assert fs.max_intra_level == 338
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


def test_mixed_geometry_batch_gpu(mi355, oracle):
    """pictures of different size in one call (largest geometry and per-level maxima passed): each comes out as the oracle's"""
    assert frame_cases.run_mixed_batch(mi355, oracle) >= 4


def test_copy_batch_gpu(mi355):
    import copy_batch_cases
    assert copy_batch_cases.run(mi355.lib) == 6


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_gpu_sparse_coefficients(mi355, oracle, name):
    """mi355_h264_recon_inter_sparse_dev: same pictures, and the coefficient blocks of cbp-0 inter macroblocks are never read"""
    frame_cases.run_case(mi355, oracle, name, sparse=True)


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_gpu_tiled_surfaces(mi355, oracle, name):
    """the same pictures with dst / recon / reference surfaces in the macroblock-tiled layout (mi355_h264_frame.surface_layout)"""
    frame_cases.run_case(mi355, oracle, name, tiled=True)


@pytest.mark.parametrize("name", ("mixed_intra", "wide_b", "mid_b_weighted"))
def test_frame_pipeline_gpu_tiled_surfaces_padded_rows(mi355, oracle, name):
    frame_cases.run_case(mi355, oracle, name, tiled=True, pad=512)


@pytest.mark.parametrize("name", ("p16_smooth", "mixed_intra", "b_weight_implicit", "mid_b_bigcoef"))
def test_frame_pipeline_gpu_tiled_sparse(mi355, oracle, name):
    frame_cases.run_case(mi355, oracle, name, tiled=True, sparse=True)


def test_mixed_layout_batch_gpu(mi355, oracle):
    """linear and tiled pictures of different geometry in one call"""
    assert frame_cases.run_mixed_batch(mi355, oracle, tiled=("wide_b", "tall_all_intra")) >= 4


def test_surface_convert_gpu(mi355):
    assert frame_cases.run_surface_convert(mi355) == 4


def test_full_size_1080p_batch_tiled_matches_oracle(mi355, oracle):
    """BASELINE.json config 2 at its real size on macroblock-tiled surfaces (what bench.py times by default): three 1080p P
    pictures of the bench generator, every sample of both surfaces against the oracle"""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x264, lib=mi355.lib)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs, tiled=True)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


@pytest.mark.parametrize("tiled", (True, False))
def test_full_size_1080p_mixed_partitions_matches_oracle(mi355, oracle, tiled):
    """SURVEY 8d's second run of config 2 (bench.py's extra point config2_mixed_partitions): 16x16 / 16x8 / 8x16 / 8x8 macroblocks
    with 8x8 / 8x4 / 4x8 / 4x4 quadrants, one vector per partition — two 1080p pictures, every sample against the oracle"""
    fs = HF.synth_frames_fast(2, 120, 68, seed=0x2640, lib=mi355.lib, partitions="mixed")
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs, tiled=tiled)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


@pytest.mark.parametrize("F,tiled", ((640, True), (96, True), (160, False)))
def test_loop_filter_bands_hand_down_under_load(mi355, oracle, F, tiled):
    """The single-launch loop filter (k_deblock2) hands rows from band to band THROUGH MEMORY inside one launch (agent-scope
    write-through stores, progress counters: h264_deblock.hip).  Many 1080p pictures at once — hundreds of bands in flight on
    every XCD, band pairs on different XCDs — on content where the filter changes most lines, decoded twice with `dst`
    overwritten in between (a line served stale from a cache would hold the scribble); EVERY picture compared with the oracle's."""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x2264, lib=mi355.lib, refs="smooth", coef_b=4)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs, replicate=F, tiled=tiled)
    try:
        d.decode_by_layout()
        assert mi355.lib.mi355_memcpy_d2d(d.dst, d.recon, F * d.fsz) == 0        # scribble: the unfiltered pictures
        d.decode_by_layout()
        bad = []
        for first in range(0, F, 32):
            n = min(32, F - first)
            got = d.fetch(d.dst, first, n)
            for i in range(n):
                g = (first + i) % fs.F
                if not all(np.array_equal(dst_o[p][g], got[p][i]) for p in range(3)):
                    bad.append(first + i)
        assert not bad, "%d of %d pictures differ from the oracle, first: %s" % (len(bad), F, bad[:8])
    finally:
        d.free()


def test_single_launch_loop_filter_with_every_cu_taken(mi355, oracle):
    """k_deblock_tiled hands rows from band to band inside ONE launch: a band spins on the progress counter of the band above.  That is safe only
    because bands are taken in ticket order — the band waited for has always started.  Here the launch gets the device the hard way: 2048 1080p
    pictures (34816 bands) on one stream while a second stream keeps every CU's LDS and wave slots full with reconstruction launches of another
    batch (k_recon_inter_tiled: eight waves per SIMD, all 160 KB of a CU's LDS), so the filter's workgroups start a few at a time, in whatever
    slots fall free.  Must finish (polled with a deadline: no blocking wait that a hang would turn into a dead test process) and every picture
    must equal the oracle's."""
    import ctypes as C
    import time
    lib = mi355.lib
    F, FB = 2048, 256
    fs = HF.synth_frames_fast(4, 120, 68, seed=0x2264, lib=lib, refs="smooth", coef_b=4)
    other = HF.synth_frames_fast(2, 120, 68, seed=0x264, lib=lib)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    for name, res, at in (("mi355_h264_recon_inter_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_h264_deblock_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_stream_create", C.c_void_p, []), ("mi355_stream_destroy", None, [C.c_void_p]),
                          ("mi355_event_create", C.c_void_p, []), ("mi355_event_destroy", None, [C.c_void_p]),
                          ("mi355_event_record", C.c_int, [C.c_void_p, C.c_void_p]), ("mi355_event_query", C.c_int, [C.c_void_p]),
                          ("mi355_event_elapsed_ms", C.c_float, [C.c_void_p, C.c_void_p])):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = at
    d = HF.DeviceFrames(mi355, fs, replicate=F, tiled=True)
    busy = HF.DeviceFrames(mi355, other, replicate=FB, tiled=True)
    sa, sb = C.c_void_p(lib.mi355_stream_create()), C.c_void_p(lib.mi355_stream_create())
    ev = [C.c_void_p(lib.mi355_event_create()) for _ in range(4)]
    try:
        d.decode_by_layout()                                                   # reconstruction (and a first, undisturbed run of the filter)
        assert lib.mi355_memcpy_d2d(d.dst, d.recon, F * d.fsz) == 0             # scribble: the unfiltered pictures
        assert lib.mi355_sync(None) == 0
        # ~60 ms of reconstruction launches queued on stream b, then the loop filter of the 2048 pictures on stream a
        lib.mi355_event_record(ev[0], sb)
        for _ in range(80):
            assert lib.mi355_h264_recon_inter_layouts_dev(busy.d_desc, FB, fs.mb_w, fs.mb_h, 2, sb) == 0
        lib.mi355_event_record(ev[1], sb)
        lib.mi355_event_record(ev[2], sa)
        assert lib.mi355_h264_deblock_layouts_dev(d.d_desc, F, fs.mb_w, fs.mb_h, 2, sa) == 0
        lib.mi355_event_record(ev[3], sa)
        deadline = time.time() + 120
        while not (lib.mi355_event_query(ev[3]) == 1 and lib.mi355_event_query(ev[1]) == 1):
            assert lib.mi355_event_query(ev[3]) >= 0 and lib.mi355_event_query(ev[1]) >= 0
            assert time.time() < deadline, "the loop filter did not finish within 120 s beside a device full of other work"
            time.sleep(0.002)
        t_busy, t_filter = lib.mi355_event_elapsed_ms(ev[0], ev[1]), lib.mi355_event_elapsed_ms(ev[2], ev[3])
        print("busy stream %.1f ms, loop filter of %d pictures beside it %.1f ms" % (t_busy, F, t_filter))
        # the other stream's launches were running before the filter started and still running when it ended
        assert lib.mi355_event_elapsed_ms(ev[0], ev[2]) > 0 and lib.mi355_event_elapsed_ms(ev[3], ev[1]) > 0, "the two streams did not overlap"
        bad = []
        for first in range(0, F, 32):
            got = d.fetch(d.dst, first, 32)
            for i in range(32):
                g = (first + i) % fs.F
                if not all(np.array_equal(dst_o[p][g], got[p][i]) for p in range(3)):
                    bad.append(first + i)
        assert not bad, "%d of %d pictures differ from the oracle, first: %s" % (len(bad), F, bad[:8])
    finally:
        lib.mi355_sync(None)
        for e in ev:
            lib.mi355_event_destroy(e)
        lib.mi355_stream_destroy(sa)
        lib.mi355_stream_destroy(sb)
        busy.free()
        d.free()


def test_single_launch_intra_pass_with_every_cu_taken(mi355, oracle):
    """k_recon_intra_all (the intra levels of a batch of I pictures in ONE launch): a macroblock waits for the flag bytes of its intra neighbours, which is safe
    because workgroups start in the order of their numbers and a neighbour's number is lower.  192 all-intra 1080p pictures (254 levels each) on one stream
    while a second stream fills every CU with reconstruction launches of another batch: must finish well inside the waits' own bound and every picture must
    equal the oracle's."""
    import ctypes as C
    import time
    lib = mi355.lib
    F, FB = 192, 256
    fs = HF.synth_frames_fast(2, 120, 68, seed=0x1264, lib=lib, intra_frac=1.0)
    other = HF.synth_frames_fast(2, 120, 68, seed=0x264, lib=lib)
    assert fs.max_intra_level >= 200
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    for name, res, at in (("mi355_h264_recon_inter_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                          ("mi355_h264_recon_intra_all_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                          ("mi355_stream_create", C.c_void_p, []), ("mi355_stream_destroy", None, [C.c_void_p]),
                          ("mi355_event_create", C.c_void_p, []), ("mi355_event_destroy", None, [C.c_void_p]),
                          ("mi355_event_record", C.c_int, [C.c_void_p, C.c_void_p]), ("mi355_event_query", C.c_int, [C.c_void_p]),
                          ("mi355_event_elapsed_ms", C.c_float, [C.c_void_p, C.c_void_p])):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = at
    d = HF.DeviceFrames(mi355, fs, replicate=F, tiled=True)
    busy = HF.DeviceFrames(mi355, other, replicate=FB, tiled=True)
    sa, sb = C.c_void_p(lib.mi355_stream_create()), C.c_void_p(lib.mi355_stream_create())
    ev = [C.c_void_p(lib.mi355_event_create()) for _ in range(4)]
    lw = (C.c_int32 * fs.max_intra_level)(*fs.level_widths[:fs.max_intra_level])
    try:
        lib.mi355_event_record(ev[0], sb)
        for _ in range(60):
            assert lib.mi355_h264_recon_inter_layouts_dev(busy.d_desc, FB, fs.mb_w, fs.mb_h, 2, sb) == 0
        lib.mi355_event_record(ev[1], sb)
        lib.mi355_event_record(ev[2], sa)
        assert lib.mi355_h264_recon_intra_all_dev(d.d_desc, F, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, sa) == 0
        lib.mi355_event_record(ev[3], sa)
        deadline = time.time() + 120
        while not (lib.mi355_event_query(ev[3]) == 1 and lib.mi355_event_query(ev[1]) == 1):
            assert lib.mi355_event_query(ev[3]) >= 0 and lib.mi355_event_query(ev[1]) >= 0
            assert time.time() < deadline, "the intra pass did not finish within 120 s beside a device full of other work"
            time.sleep(0.002)
        t_intra = lib.mi355_event_elapsed_ms(ev[2], ev[3])
        print("busy stream %.1f ms, intra pass of %d I pictures beside it %.1f ms" % (lib.mi355_event_elapsed_ms(ev[0], ev[1]), F, t_intra))
        assert t_intra < 500, "a wave gave up waiting (%.0f ms): workgroups did not start in order" % t_intra
        bad = []
        for first in range(0, F, 32):
            got = d.fetch(d.recon, first, 32)
            for i in range(32):
                g = (first + i) % fs.F
                if not all(np.array_equal(recon_o[p][g], got[p][i]) for p in range(3)):
                    bad.append(first + i)
        assert not bad, "%d of %d pictures differ from the oracle, first: %s" % (len(bad), F, bad[:8])
    finally:
        lib.mi355_sync(None)
        for e in ev:
            lib.mi355_event_destroy(e)
        lib.mi355_stream_destroy(sa)
        lib.mi355_stream_destroy(sb)
        busy.free()
        d.free()


@pytest.mark.parametrize("tiled", (True, False))
@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_gpu_layout_entry_points(mi355, oracle, name, tiled):
    """mi355_h264_recon_inter_layouts_dev / mi355_h264_deblock_layouts_dev with the batch's one layout named: the single-layout kernel instances"""
    frame_cases.run_case(mi355, oracle, name, tiled=tiled, by_layout=True)


@pytest.mark.parametrize("pad", (0, 8))
@pytest.mark.parametrize("name", [n for n in frame_cases.CASES if n != "tall_all_intra"])
def test_second_kernel_set_on_8bit_420_pictures(mi355, oracle, name, pad):
    """mi355_h264_decode_frames_wide_dev (the High 10 / High 4:2:2 kernels of h264_frame_wide.hip, instantiated for 8-bit 4:2:0) against
    the oracle on the cases of the first kernel set — the second set's partition walk, weights, intra predictors, transforms, I_PCM and
    loop filter on hardware (its 16-bit / 4:2:2 instances: the generated streams, tests/test_synth_streams_gpu.py)"""
    frame_cases.run_case(mi355, oracle, name, pad=pad, wide=True)


def test_second_kernel_set_full_size_1080p_matches_oracle(mi355, oracle):
    """three 1080p P pictures of the bench generator through the second kernel set (8-bit 4:2:0 instance): every sample of both surfaces —
    376 anti-diagonal launches of the loop filter over 24 480 macroblocks"""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x264, lib=mi355.lib)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs)
    try:
        d.decode_wide()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


def test_second_kernel_set_10bit_1080p_is_deterministic_and_in_range(mi355):
    """the bench point's input (DeviceFrames(bit_depth=10): config 2 scaled to 10 bits) through mi355_h264_decode_frames_wide_dev(10, 1):
    two runs agree sample for sample, samples stay inside 10 bits (parity of the 10-bit kernels: the generated High 10 streams)"""
    fs = HF.synth_frames_fast(2, 120, 68, seed=0x264, lib=mi355.lib)
    outs = []
    for _ in range(2):
        d = HF.DeviceFrames(mi355, fs, bit_depth=10)
        try:
            d.decode_wide(10, 1)
            outs.append(d.fetch(d.dst))
        finally:
            d.free()
    for p in range(3):
        assert np.array_equal(outs[0][p], outs[1][p]) and outs[0][p].max() <= 1023 and outs[0][p].std() > 10


@pytest.mark.parametrize("unit,F", (("4", 320), ("3", 36), ("1", 8), ("", 128)))
def test_second_kernel_set_loop_filter_units_under_load(mi355, oracle, monkeypatch, unit, F):
    """the second kernel set's loop filter with 1, 3, 4 macroblocks per group and launch (MI355_WIDE_UNIT; "": the launcher's own choice — a group
    filters a run of its row's macroblocks, the anti-diagonals count runs, the next macroblock's loads are in flight during the filter) on
    replicated 1080p pictures (8-bit 4:2:0 instance; 120 macroblocks a row: not a multiple of 3): every picture equals the oracle's"""
    if unit:
        monkeypatch.setenv("MI355_WIDE_UNIT", unit)
    else:
        monkeypatch.delenv("MI355_WIDE_UNIT", raising=False)
    fs = HF.synth_frames_fast(4, 120, 68, seed=0x2264, lib=mi355.lib, refs="smooth", coef_b=4)
    _, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(mi355, fs, replicate=F)
    try:
        for _ in range(2):
            d.decode_wide()
            got = d.fetch(d.dst)
            for p in range(3):
                for f in range(F):
                    assert np.array_equal(got[p][f], dst_o[p][f % fs.F]), (unit, f, p)
    finally:
        d.free()


def test_full_size_1080p_batch_run_kernel(mi355, oracle):
    """three 1080p pictures of the headline workload through k_recon_inter_tiled (the run kernel: raw LDS-DMA windows, the 6-tap filters as
    v_mfma_i32_16x16x32_i8 products, coefficients of the next macroblock in flight), every sample of both surfaces"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 3, 120, 68, 0x264)


@pytest.mark.parametrize("mv_range", (64, 200, 1200))
@pytest.mark.parametrize("mb_w,mb_h", ((7, 5), (1, 1), (2, 3), (3, 1), (5, 9)))
def test_run_kernel_windows_over_every_border(mi355, oracle, mb_w, mb_h, mv_range):
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 2, mb_w, mb_h, 0x2650 + mv_range + mb_w, mv_range=mv_range)


@pytest.mark.parametrize("shares,turns,replicate", ((3, 1, 8), (2, 0, 5), (5, 1, 3), (1, 1, 4)))
def test_pipelines_object_small(mi355, oracle, shares, turns, replicate):
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 2, 9, 5, 0x2670 + shares, replicate=replicate, pipelined=(shares, turns, 2), partitions="mixed", intra_frac=0.2)


def test_pipelines_object_under_load(mi355, oracle):
    """mi355_h264_pipelines_*: 256 1080p pictures as three shares whose reconstruction launches take turns, three batches one behind the other on the object's streams
    (the passes of different shares and of consecutive calls overlap on the device): every picture equals the oracle's"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 3, 120, 68, 0x2267, replicate=256, pipelined=(3, 1, 3))


def test_run_kernel_mixed_partitions(mi355, oracle):
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 2, 9, 5, 0x2641, partitions="mixed", intra_frac=0.2)


@pytest.mark.parametrize("mv_range", (64, 200, 1200))
@pytest.mark.parametrize("mb_w,mb_h", ((7, 5), (1, 1), (2, 3), (5, 9)))
def test_two_partition_path_over_every_border(mi355, oracle, mb_w, mb_h, mv_range):
    """16x8 / 8x16 macroblocks through fq_two (k_recon_inter_rest) beside plain and 8x8 ones, windows over the borders by any distance"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 2, mb_w, mb_h, 0x2660 + mv_range + mb_w, partitions="mixed", mv_range=mv_range)


def test_two_partition_path_many_pictures(mi355, oracle):
    """mixed partitions under load: 192 1080p pictures in one launch pair, all compared"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 3, 120, 68, 0x2266, replicate=192, partitions="mixed")


def test_run_kernel_many_pictures(mi355, oracle):
    """the same under load: 256 pictures in one launch (every SIMD at eight waves, requests of several macroblocks in flight per wave), all compared"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 3, 120, 68, 0x2265, replicate=256)


HBD_CASES = [n for n in frame_cases.CASES if n not in ("tall_all_intra", "mid_hugecoef", "mid_wrapcoef", "p16_wrapcoef")]


@pytest.mark.parametrize("name", HBD_CASES)
def test_second_kernel_set_against_the_frame_checker_at_10_bits(mi355, oracle, name):
    """the High 10 instantiation of the second kernel set, frame level: 16-bit samples, 32-bit coefficients against oracle/oracle_h264frame_hbd.c on the reference's
    own 10-bit tables (oracle/_ref/libref.so, built by __graft_entry__.build() where /root/reference exists and shipped with the tree)"""
    assert frame_cases.run_case_hbd(mi355, oracle, name, 10), "oracle/_ref/libref.so missing: __graft_entry__.build() makes it where /root/reference exists"


@pytest.mark.parametrize("name", ("mixed_intra", "b_weight_explicit", "wide_b"))
def test_second_kernel_set_against_the_frame_checker_at_9_bits(mi355, oracle, name):
    assert frame_cases.run_case_hbd(mi355, oracle, name, 9), "oracle/_ref/libref.so missing"


@pytest.mark.parametrize("name", HBD_CASES)
def test_second_kernel_set_against_the_frame_checker_at_422_10_bits(mi355, oracle, name):
    """the High 4:2:2 instantiation (10 bit) of the second kernel set, frame level: the 4:2:2 variant of every case against oracle/oracle_h264frame_hbd.c on the
    reference's tables initialised with chroma_format_idc 2 (the 2x4 chroma DC transform, idct_add8_422, the 8x16 predictors, the sixteen-line chroma edge)"""
    assert frame_cases.run_case_hbd(mi355, oracle, name, 10, idc=2), "oracle/_ref/libref.so missing: __graft_entry__.build() makes it where /root/reference exists"


@pytest.mark.parametrize("name", ("mixed_intra", "b_weight_explicit", "wide_b"))
def test_second_kernel_set_against_the_frame_checker_at_422_9_bits(mi355, oracle, name):
    assert frame_cases.run_case_hbd(mi355, oracle, name, 9, idc=2), "oracle/_ref/libref.so missing"


def test_config2_422_full_size_matches_the_frame_checker(mi355, oracle):
    """the headline generator's 1080p pictures as a High 4:2:2 (10-bit) batch, replicated to 24 in the launch: every sample of both surfaces of every picture"""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x2642, lib=mi355.lib)
    assert frame_cases.run_case_hbd(mi355, oracle, "config2_high422", 10, fs=fs, replicate=24, idc=2), "oracle/_ref/libref.so missing"


def test_config2_high10_full_size_matches_the_frame_checker(mi355, oracle):
    """bench.py's config2_high10 workload at its real size: three 1080p pictures of the headline generator as a High 10 batch (replicated to 24 in the launch),
    every sample of both surfaces of every picture"""
    fs = HF.synth_frames_fast(3, 120, 68, seed=0x264, lib=mi355.lib)
    assert frame_cases.run_case_hbd(mi355, oracle, "config2_high10", 10, fs=fs, replicate=24), "oracle/_ref/libref.so missing"


def test_pipelines_object_at_the_bench_share(mi355, oracle):
    """what bench.py's headline times, at its size: 2049 1080p pictures (two distinct ones) through mi355_h264_pipelines_* as three shares of 683 whose reconstruction
    launches take turns, two calls one behind the other: every sample of both surfaces of every picture against the oracle"""
    frame_cases.run_fast_workload_by_layout(mi355, oracle, 2, 120, 68, 0x2268, replicate=2049, pipelined=(3, 1, 2))


def test_intra_single_launch_wait_that_runs_out_is_reported(mi355):
    """k_recon_intra_all's bounded wait (h264_frame.hip): with the bound at zero every macroblock that has an intra neighbour gives up at once — the entry point still
    returns 0 (the launch was made), the wait behind it returns MI355_E_DEVICE_FAULT and the word says MI355_ERR_WAIT_EXPIRED; a process of its own: the bound is read once"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import providers, frame_cases, h264_frames as HF\n"
        "emu = providers.mi355()\n"
        "fs = HF.synth_frames(**frame_cases.CASES['tall_all_intra'])\n"
        "d = HF.DeviceFrames(emu, fs, tiled=True)\n"
        "lib = emu.lib\n"
        "lib.mi355_error_word_take.restype = C.c_uint\n"
        "lib.mi355_error_word_take()\n"
        "lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])\n"
        "lib.mi355_h264_recon_intra_all_dev.restype = C.c_int\n"
        "lib.mi355_h264_recon_intra_all_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]\n"
        "assert lib.mi355_h264_recon_intra_all_dev(d.d_desc, d.F, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, None) == 0\n"
        "lib.mi355_sync.restype = C.c_int\n"
        "rc = lib.mi355_sync(None)\n"
        "word = lib.mi355_error_word_take()\n"
        "print('RESULT', rc, word, lib.mi355_sync(None))\n"
    ) % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MI355_INTRA_NAPS_MAX="0", MI355_INTRA_SINGLE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1:] == ["-5", "1", "0"], line
