"""Config 1 (plumbing) pin: a REAL H.264 bitstream.

tests/golden/h264_stream_realshort.npz holds, for each of the 36 pictures of realshort.mp4, the
Tier-2 records exported from the reference decoder's own run and the reference's decoded picture
(tests/golden/make_stream_golden.py).  Reconstruction + loop filter of every picture must
reproduce the reference's output sample for sample:
  * by the CPU oracle (pins oracle_h264frame.c to the reference decoder),
  * by the product's kernels under the SIMT emulator (CPU) and on the MI355X (-m gpu).
"""
import os

import numpy as np
import pytest

import h264_frames as HF
import stream_fixture as SF

NPZ = os.path.join(os.path.dirname(__file__), "golden", "h264_stream_realshort.npz")


@pytest.fixture(scope="module")
def pics():
    return SF.load_npz(NPZ)


def _check(pics, planes, first=0):
    for f in range(planes[0].shape[0]):
        pc = pics[first + f]
        for p, key in enumerate(("y", "cb", "cr")):
            assert np.array_equal(planes[p][f], pc[key]), "picture %d plane %s differs from the reference decoder" % (first + f, key)


def test_fixture_shape(pics):
    assert len(pics) == 36 and pics[0]["mb_w"] == 20 and pics[0]["mb_h"] == 15
    assert pics[0]["pict_type"] == 1 and sum(p["pict_type"] == 2 for p in pics) >= 30      # I then P pictures
    assert any((p["mb"]["mb_type"] & 0x01000000).any() for p in pics)                      # 8x8 transform present
    assert any(((p["mb"]["mb_type"] & 7) != 0).any() and p["pict_type"] == 2 for p in pics)  # intra MBs inside P pictures


def test_oracle_reproduces_reference_decoder(oracle, pics):
    fs = SF.frameset_all(pics)
    _, dst = HF.run_oracle(oracle, fs)
    _check(pics, dst)


def test_emulated_kernels_reproduce_reference_decoder(emu, pics):
    fs = SF.frameset_all(pics, 0, 12)
    d = HF.DeviceFrames(emu, fs)
    try:
        d.decode()
        _check(pics, d.fetch(d.dst))
    finally:
        d.free()


@pytest.mark.gpu
def test_gpu_reproduces_reference_decoder(mi355, pics):
    fs = SF.frameset_all(pics)
    d = HF.DeviceFrames(mi355, fs)
    try:
        d.decode()
        _check(pics, d.fetch(d.dst))
    finally:
        d.free()
