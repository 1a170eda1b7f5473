"""Whole-frame sessions (mi355_h264_session.h) on the MI355X."""
import pytest

import frame_cases
import session_cases as SC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tiled", (False, True))
@pytest.mark.parametrize("sync_each", (True, False))
def test_session_real_stream_in_sequence_gpu(mi355, sync_each, tiled):
    """all 36 pictures of realshort.mp4, every picture compared with the reference decoder's; surfaces with line strides and
    macroblock-tiled surfaces"""
    assert SC.run_stream(mi355, SC.SF_NPZ, 0, None, nsurf=4, sync_each=sync_each, tiled=tiled) == 36


def test_session_joined_in_the_middle_gpu(mi355):
    SC.run_stream(mi355, SC.SF_NPZ, 17, 12, nsurf=3, sync_each=False)
    SC.run_stream(mi355, SC.SF_NPZ, 17, 12, nsurf=3, sync_each=False, tiled=True)


@pytest.mark.parametrize("how", ("runs", "addr", "split"))
@pytest.mark.parametrize("name", [n for n in frame_cases.CASES if not n.startswith(("tall", "one_"))])
def test_session_synthetic_pictures_gpu(mi355, oracle, name, how):
    SC.run_synth(mi355, oracle, name, how)
    SC.run_synth(mi355, oracle, name, how, tiled=True)


def test_session_argument_and_state_checks_gpu(mi355):
    SC.run_errors(mi355)


def test_session_decode_then_convert_on_device_gpu(mi355, oracle):
    import chain_check
    assert chain_check.run_session(mi355, oracle, first=5, count=6) == 6
    assert chain_check.run_session(mi355, oracle, first=5, count=6, tiled=True) == 6


@pytest.mark.parametrize("explicit_flush", (True, False))
def test_session_group_one_launch_set_for_several_streams_gpu(mi355, explicit_flush):
    import synth_streams as SY
    names = ("420_8_slices", "420_8_b_implicit", "420_8_qcif", "420_8_t8x8", "420_8_cip_mixed", "420_8_b_average")
    SC.run_group(mi355, [SY.npz(n) for n in names] + [SC.SF_NPZ], explicit_flush=explicit_flush, tiled=(0, 2, 3, 6))
