"""CPU: Tier-2 kernels under the SIMT emulator (test tooling) vs the oracle, bit-exact."""
import numpy as np
import pytest

import frame_cases
import h264_frames as HF


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_emulated(emu, oracle, name):
    changed = frame_cases.run_case(emu, oracle, name)
    if "smooth" in str(frame_cases.CASES[name]) and frame_cases.CASES[name]["mb_w"] > 1:
        assert changed > 100, "loop filter barely exercised"


def test_full_size_1080p_picture_emulated(emu, oracle):
    """One 1080p picture of the bench workload (BASELINE.json config 2's real size) through the emulated
    kernels: 8160 macroblocks reach value combinations the small cases do not (this is the case that
    caught the DC-only transform of Intra16x16 blocks wrapping at int16)."""
    import numpy as np
    import h264_frames as HF
    fs = HF.synth_frames_fast(1, 120, 68, seed=0x264, lib=emu.lib)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(emu, fs)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


def test_decode_then_convert_on_device_emulated(emu, oracle):
    import chain_check
    assert chain_check.run(emu, oracle) == 3


@pytest.mark.parametrize("name", ("mixed_intra", "tall_all_intra"))
def test_frame_pipeline_emulated_single_level_bound(emu, oracle, name):
    """mi355_h264_decode_frames_dev: every intra level sized by the one bound max_level_width (the other cases go through
    mi355_h264_decode_frames_levels_dev with per-level widths)"""
    frame_cases.run_case(emu, oracle, name, per_level=False)


@pytest.mark.parametrize("pad", (8, 24))
@pytest.mark.parametrize("name", ("mixed_intra", "wide_b", "one_col"))
def test_frame_pipeline_emulated_unaligned_strides(emu, oracle, name, pad):
    """strides that are multiples of 8 / 4 only: the dword paths instead of the 16-byte ones"""
    frame_cases.run_case(emu, oracle, name, pad=pad)


def test_intra_schedule_helper_above_255_levels(emu):
    """mi355_h264_intra_schedule on an all-intra 2560x1440 picture (338 levels): same lists as the python twin, every
    macroblock listed once, after all four neighbours it predicts from (the 8-bit level field wrapped)."""
    import numpy as np
    import h264_frames as HF
    fs = HF.synth_frames_fast(1, 160, 90, seed=5, intra_frac=1.0, lib=emu.lib)
    lst, start = fs.intra_list[0].copy(), fs.intra_start[0].copy()
    assert fs.max_intra_level == 160 + 2 * 89 == len(start) - 1
    assert sorted(lst.tolist()) == list(range(160 * 90))
    twin = HF.FrameSet(1, 160, 90, 4)
    twin.mb[:] = fs.mb
    assert HF.intra_schedule(twin, 0) == fs.max_intra_level
    assert np.array_equal(twin.intra_list[0], lst) and np.array_equal(twin.intra_start[0], start)
    level = np.zeros(160 * 90, np.int64)
    for l in range(1, len(start)):
        level[lst[start[l - 1]:start[l]]] = l
    lv = level.reshape(90, 160)
    assert (lv[:, 1:] > lv[:, :-1]).all() and (lv[1:, :] > lv[:-1, :]).all() and (lv[1:, :-1] > lv[:-1, 1:]).all()
    assert fs.mb[0]["intra_level"].max() == 255                      # informational field saturates


def test_mixed_geometry_batch_emulated(emu, oracle):
    assert frame_cases.run_mixed_batch(emu, oracle) >= 4


def test_copy_batch_emulated(emu):
    import copy_batch_cases
    assert copy_batch_cases.run(emu.lib) == 6


@pytest.mark.parametrize("name", ("p16_smooth", "mixed_intra", "wide_b", "b_weight_implicit"))
def test_frame_pipeline_emulated_sparse_coefficients(emu, oracle, name):
    """mi355_h264_recon_inter_sparse_dev: same pictures, and the coefficient blocks of cbp-0 inter macroblocks are never read"""
    frame_cases.run_case(emu, oracle, name, sparse=True)


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_emulated_tiled_surfaces(emu, oracle, name):
    """the same pictures with dst / recon / reference surfaces in the macroblock-tiled layout (mi355_h264_frame.surface_layout)"""
    frame_cases.run_case(emu, oracle, name, tiled=True)


@pytest.mark.parametrize("name", ("mixed_intra", "wide_b"))
def test_frame_pipeline_emulated_tiled_surfaces_padded_rows(emu, oracle, name):
    frame_cases.run_case(emu, oracle, name, tiled=True, pad=512)


@pytest.mark.parametrize("name", ("p16_smooth", "mixed_intra", "b_weight_implicit"))
def test_frame_pipeline_emulated_tiled_sparse(emu, oracle, name):
    frame_cases.run_case(emu, oracle, name, tiled=True, sparse=True)


def test_mixed_layout_batch_emulated(emu, oracle):
    """linear and tiled pictures of different geometry in one call"""
    assert frame_cases.run_mixed_batch(emu, oracle, tiled=("wide_b", "tall_all_intra")) >= 4


def test_surface_convert_emulated(emu):
    assert frame_cases.run_surface_convert(emu, cases=((5, 3, 0, 0), (1, 1, 0, 256), (9, 4, 24, 512))) == 3


@pytest.mark.parametrize("tiled", (True, False))
def test_mixed_partition_workload_matches_oracle_emulated(emu, oracle, tiled):
    """the mixed-partition variant of the bench generator (16x16 / 16x8 / 8x16 / 8x8 with every sub-partition shape) on a
    small picture whose windows reach over every border"""
    import numpy as np
    import h264_frames as HF
    fs = HF.synth_frames_fast(2, 7, 5, seed=0x2640, lib=emu.lib, partitions="mixed")
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(emu, fs, tiled=tiled)
    try:
        d.decode()
        recon_g, dst_g = d.fetch(d.recon), d.fetch(d.dst)
    finally:
        d.free()
    for p in range(3):
        assert np.array_equal(recon_o[p], recon_g[p])
        assert np.array_equal(dst_o[p], dst_g[p])


@pytest.mark.parametrize("tiled", (True, False))
@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_emulated_layout_entry_points(emu, oracle, name, tiled):
    """mi355_h264_recon_inter_layouts_dev / mi355_h264_deblock_layouts_dev with the batch's one layout named: the single-layout kernel instances"""
    frame_cases.run_case(emu, oracle, name, tiled=tiled, by_layout=True)


@pytest.mark.parametrize("pad", (0, 8))
@pytest.mark.parametrize("name", [n for n in frame_cases.CASES if n != "tall_all_intra"])
def test_second_kernel_set_on_8bit_420_pictures_emulated(emu, oracle, name, pad):
    """mi355_h264_decode_frames_wide_dev (the High 10 / High 4:2:2 kernels, instantiated for 8-bit 4:2:0) against the oracle on the
    cases of the first kernel set: partitions, weights, intra modes, I_PCM, the 8x8 transform, slices with the filter off"""
    frame_cases.run_case(emu, oracle, name, pad=pad, wide=True)


@pytest.mark.parametrize("unit", ("2", "3", "4"))
@pytest.mark.parametrize("name", ("mixed_intra", "wide_b", "one_col"))
def test_second_kernel_set_loop_filter_units_emulated(emu, oracle, monkeypatch, name, unit):
    """the second kernel set's loop filter with 2, 3 and 4 macroblocks per group and launch (MI355_WIDE_UNIT; the launcher picks 1 for batches
    this small): a group filters a run of its row — the left neighbour's columns stay in its tile, the next macroblock's loads are issued before
    the filter — and the anti-diagonals count runs; pictures whose width is not a multiple of the run, one macroblock wide, with slices"""
    monkeypatch.setenv("MI355_WIDE_UNIT", unit)
    frame_cases.run_case(emu, oracle, name, pad=0, wide=True)


@pytest.mark.parametrize("name", ("p16_noise", "p16_smooth", "b_weight_explicit"))
def test_second_kernel_set_10bit_sanity_emulated(emu, oracle, name):
    """h264_frames.DeviceFrames(bit_depth=10): the 8-bit case scaled to 10 bits (samples and coefficients shifted by two, QPs raised by 12)
    through mi355_h264_decode_frames_wide_dev(10, 1).  NOT a parity test (parity of the 10-bit kernels: the generated High 10 streams
    against the reference decoder, tests/test_synth_streams*.py) — it pins that this measurement input decodes to what the 8-bit picture
    is, four times as large: samples inside 10 bits, nine samples in ten within one 8-bit step of the 8-bit oracle's picture (the rest:
    DC levels that wrap in the 8-bit path's 16-bit coefficients and do not in 32 bits, filter decisions at a threshold), and two runs
    agree sample for sample"""
    fs = HF.synth_frames(**frame_cases.CASES[name])
    _, dst8 = HF.run_oracle(oracle, fs)
    runs = []
    for _ in range(2):
        d = HF.DeviceFrames(emu, fs, bit_depth=10)
        try:
            d.decode_wide(10, 1)
            runs.append(d.fetch(d.dst))
        finally:
            d.free()
    for p in range(3):
        assert np.array_equal(runs[0][p], runs[1][p])
        assert runs[0][p].max() <= 1023
        assert (np.abs(runs[0][p].astype(np.int32) / 4.0 - dst8[p]) <= 1.0).mean() > 0.9, (name, p)


_fast_workload_by_layout = frame_cases.run_fast_workload_by_layout


def test_full_size_1080p_picture_emulated_run_kernel(emu, oracle):
    """one 1080p picture of the headline workload through the run kernel (raw LDS windows, filters as matrix products)"""
    _fast_workload_by_layout(emu, oracle, 1, 120, 68, 0x264)


@pytest.mark.parametrize("mv_range", (64, 200, 1200))
@pytest.mark.parametrize("mb_w,mb_h", ((7, 5), (1, 1), (2, 3), (3, 1), (5, 9)))
def test_run_kernel_windows_over_every_border_emulated(emu, oracle, mb_w, mb_h, mv_range):
    """plain P macroblocks whose windows reach over the borders by a little, by more than a window, and by more than the picture:
    the fast path replicates the edge column / row for any distance (emulated_edge_mc)"""
    _fast_workload_by_layout(emu, oracle, 2, mb_w, mb_h, 0x2650 + mv_range + mb_w, mv_range=mv_range)


@pytest.mark.parametrize("shares,turns,replicate", ((3, 1, 8), (2, 0, 5), (5, 1, 3), (1, 1, 4)))
def test_pipelines_object_emulated(emu, oracle, shares, turns, replicate):
    """mi355_h264_pipelines_*: uneven shares, more shares than pictures, two calls one behind the other — the same pictures as the one-stream entry points
    (streams and events do nothing in the emulator: this checks the shares' arithmetic and the entry point's plumbing; the ordering is the device test's)"""
    _fast_workload_by_layout(emu, oracle, 2, 9, 5, 0x2670 + shares, replicate=replicate, pipelined=(shares, turns, 2), partitions="mixed", intra_frac=0.2)


def test_run_kernel_mixed_partitions_emulated(emu, oracle):
    """runs that hold fast macroblocks and deferred ones (partitions) side by side"""
    _fast_workload_by_layout(emu, oracle, 2, 9, 5, 0x2641, partitions="mixed", intra_frac=0.2)


@pytest.mark.parametrize("mv_range", (64, 200, 1200))
@pytest.mark.parametrize("mb_w,mb_h", ((7, 5), (1, 1), (2, 3), (5, 9)))
def test_two_partition_path_over_every_border_emulated(emu, oracle, mb_w, mb_h, mv_range):
    """16x8 / 8x16 macroblocks through fq_two (the fast path's code once per partition, in the second launch) beside plain ones and 8x8 ones (general code),
    windows of either partition reaching over the borders by any distance"""
    _fast_workload_by_layout(emu, oracle, 2, mb_w, mb_h, 0x2660 + mv_range + mb_w, partitions="mixed", mv_range=mv_range)


HBD_CASES = [n for n in frame_cases.CASES if n not in ("tall_all_intra", "mid_hugecoef", "mid_wrapcoef", "p16_wrapcoef")]     # (levels up to +-32767 << 2: beyond what a High 10 stream can carry)


@pytest.mark.parametrize("bit_depth", (10, 9))
@pytest.mark.parametrize("name", HBD_CASES)
def test_second_kernel_set_against_the_frame_checker_above_8_bits_emulated(emu, oracle, name, bit_depth):
    """VERDICT r4 "missing 1": the High 10 / 9-bit instantiations of the second kernel set on FRAME level — every macroblock kind of the 8-bit cases
    (partitions, weights, intra modes, I_PCM, the 8x8 transform) with 16-bit samples and 32-bit coefficients against the restated drivers calling
    the reference's own ff_h264dsp_init(c, 10 / 9, 1) tables"""
    if bit_depth == 9 and name not in ("mixed_intra", "b_weight_explicit", "wide_b"):
        pytest.skip("9 bits: three cases")
    if not frame_cases.run_case_hbd(emu, oracle, name, bit_depth):
        pytest.skip("oracle/_ref/libref.so not built (no /root/reference)")


@pytest.mark.parametrize("bit_depth", (10, 9))
@pytest.mark.parametrize("name", HBD_CASES)
def test_second_kernel_set_against_the_frame_checker_at_422_emulated(emu, oracle, name, bit_depth):
    """VERDICT r5 "missing 2": the High 4:2:2 instantiations (k_wide_*<10, 2> / <9, 2>) on FRAME level — the 4:2:2 variant of every case (chroma planes of the luma's
    height, eight chroma blocks and a 2x4 DC transform per plane, 8x16 chroma prediction, the sixteen-line chroma edge and the four horizontal chroma edges)
    against the restated drivers on the reference's tables initialised with chroma_format_idc 2 (ff_h264dsp_init(c, bd, 2), ff_h264_pred_init(.., 2):
    libavcodec/h264dsp.c:57-137, h264idct_template.c:216-310)"""
    if bit_depth == 9 and name not in ("mixed_intra", "b_weight_explicit", "wide_b"):
        pytest.skip("9 bits: three cases")
    if not frame_cases.run_case_hbd(emu, oracle, name, bit_depth, idc=2):
        pytest.skip("oracle/_ref/libref.so not built (no /root/reference)")


def test_config2_high10_workload_against_the_frame_checker_emulated(emu, oracle):
    """bench.py's config2_high10 generator (the headline workload as a High 10 batch), one small and one 1080p-wide picture"""
    fs = HF.synth_frames_fast(2, 12, 7, seed=0x264, lib=emu.lib)
    if not frame_cases.run_case_hbd(emu, oracle, "config2_high10", 10, fs=fs):
        pytest.skip("oracle/_ref/libref.so not built (no /root/reference)")


def test_error_word_round_trip_emulated(emu):
    """the device's error word (include/mi355dsp.h): a kernel's bits reach the host, the waits return MI355_E_DEVICE_FAULT until the caller takes them"""
    import ctypes as C
    lib = emu.lib
    lib.mi355_error_word_take.restype = C.c_uint
    lib.mi355_error_word_peek.restype = C.c_uint
    lib.mi355_error_word_inject.argtypes = [C.c_uint, C.c_void_p]
    lib.mi355_sync.restype = C.c_int
    lib.mi355_error_word_take()
    assert lib.mi355_sync(None) == 0
    assert lib.mi355_error_word_inject(0x40000000, None) == 0
    assert lib.mi355_sync(None) == -5 and lib.mi355_error_word_peek() == 0x40000000
    assert lib.mi355_error_word_take() == 0x40000000 and lib.mi355_sync(None) == 0


def test_intra_single_launch_wait_that_runs_out_is_reported_emulated(tmp_path):
    """k_recon_intra_all's bounded wait (h264_frame.hip): with the bound at zero every macroblock that has an intra neighbour gives up at once — the entry point still
    returns 0 (the launch was made), the wait behind it returns MI355_E_DEVICE_FAULT and the word says MI355_ERR_WAIT_EXPIRED; a process of its own: the bound is read once"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import providers, frame_cases, h264_frames as HF\n"
        "emu = providers.emu()\n"
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


def test_pipelines_object_joins_calls_on_different_batches_emulated(emu, oracle):
    """mi355_h264_pipelines_join: calls on another batch (count / array) and joined-always mode go through the waits on every share's loop filter of the call before;
    the pictures are the same (what the waits order is the device test's business)"""
    import ctypes as C
    fs = HF.synth_frames_fast(2, 9, 5, seed=0x2671, lib=emu.lib, partitions="mixed", intra_frac=0.2)
    recon_o, dst_o = HF.run_oracle(oracle, fs)
    d = HF.DeviceFrames(emu, fs, tiled=True, replicate=7)
    lib = emu.lib
    try:
        lib.mi355_h264_pipelines_create.restype = C.c_void_p
        lib.mi355_h264_pipelines_create.argtypes = [C.c_int, C.c_int]
        lib.mi355_h264_pipelines_decode_dev.restype = C.c_int
        lib.mi355_h264_pipelines_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.mi355_h264_pipelines_sync.argtypes = [C.c_void_p]
        lib.mi355_h264_pipelines_join.argtypes = [C.c_void_p, C.c_int]
        lib.mi355_h264_pipelines_join.restype = None
        lib.mi355_h264_pipelines_destroy.argtypes = [C.c_void_p]
        lib.mi355_h264_pipelines_destroy.restype = None
        lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
        p = lib.mi355_h264_pipelines_create(3, 1)
        assert p
        for mode, counts in ((1, (7, 5, 7)), (2, (7, 7)), (0, (7, 4))):
            lib.mi355_h264_pipelines_join(p, mode)
            for n in counts:
                assert lib.mi355_h264_pipelines_decode_dev(p, d.d_desc, n, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, 2) == 0
        assert lib.mi355_h264_pipelines_sync(p) == 0
        lib.mi355_h264_pipelines_destroy(p)
        recon_g, dst_g = d.fetch(d.recon, 0, 7), d.fetch(d.dst, 0, 7)
        for i in range(7):
            for pl in range(3):
                assert np.array_equal(recon_o[pl][i % fs.F], recon_g[pl][i]) and np.array_equal(dst_o[pl][i % fs.F], dst_g[pl][i]), (i, pl)
    finally:
        d.free()
