"""Whole-frame sessions (mi355_h264_session.h) with the product sources under the SIMT emulator."""
import pytest

import session_cases as SC


@pytest.mark.parametrize("tiled", (False, True))
def test_session_real_stream_in_sequence_emulated(emu, tiled):
    """the first pictures of realshort.mp4 (I, then P pictures each predicted from the surface decoded before); on surfaces with
    line strides and on macroblock-tiled ones"""
    assert SC.run_stream(emu, SC.SF_NPZ, 0, 7, tiled=tiled) == 7


def test_session_joined_in_the_middle_pipelined_emulated(emu):
    """pictures 20..25: the first reference loaded with put_frame, pictures fetched two at a time (end_frame does not wait)"""
    SC.run_stream(emu, SC.SF_NPZ, 20, 6, nsurf=3, sync_each=False)
    SC.run_stream(emu, SC.SF_NPZ, 20, 4, nsurf=3, sync_each=False, tiled=True)      # put_frame into tiles


@pytest.mark.parametrize("name,how,tiled", (("b_mixed", "runs", False), ("mixed_intra", "addr", False), ("wide_b", "split", False), ("b_mixed", "runs", True), ("mixed_intra", "split", True)))
def test_session_synthetic_pictures_emulated(emu, oracle, name, how, tiled):
    SC.run_synth(emu, oracle, name, how, tiled=tiled)


def test_session_argument_and_state_checks_emulated(emu):
    SC.run_errors(emu)


def test_session_decode_then_convert_on_device_emulated(emu, oracle):
    """f2 through a session: surfaces of the session are the converter's sources, on the session's stream"""
    import chain_check
    assert chain_check.run_session(emu, oracle, first=2, count=2) == 2
    assert chain_check.run_session(emu, oracle, first=2, count=2, tiled=True) == 2


@pytest.mark.parametrize("explicit_flush", (True, False))
def test_session_group_one_launch_set_for_several_streams_emulated(emu, explicit_flush):
    """three generated streams of different picture sizes (slices, I_PCM, B pictures, four references) decoded in step by
    three sessions of one group: one launch set per step for all of them"""
    import synth_streams as SY
    n = SC.run_group(emu, [SY.npz("420_8_slices"), SY.npz("420_8_b_implicit"), SY.npz("420_8_qcif")], explicit_flush=explicit_flush, tiled=(0, 2))      # both layouts in one launch set
    assert n == 7 + 9 + 10
