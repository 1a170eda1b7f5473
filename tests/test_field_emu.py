"""CPU: field pictures at the Tier-2 API (field_cases.py), emulated kernels against the oracle."""
import pytest

import field_cases
import frame_cases
import h264_frames as HF


@pytest.mark.parametrize("name", ("mixed_intra", "b_mixed", "wide_b", "p16_smooth"))
def test_field_pictures_emulated(emu, oracle, name):
    assert field_cases.run(emu, oracle, HF.synth_frames(**frame_cases.CASES[name])) > 1000


@pytest.mark.parametrize("name,how", (("b_mixed", "runs"), ("mixed_intra", "split")))
def test_field_pictures_through_sessions_emulated(emu, oracle, name, how):
    """picture parameter `field`: the field goes into every other line of its surface, references are (surface, parity)"""
    fs = HF.synth_frames(**frame_cases.CASES[name])
    assert field_cases.run_session(emu, oracle, fs, how=how) == fs.F


@pytest.mark.parametrize("name", ("b_mixed", "p16_smooth"))
def test_field_pairs_in_one_surface_emulated(emu, oracle, name):
    """the second field of a frame predicts from the first field of its own surface"""
    fs = HF.synth_frames(**frame_cases.CASES[name])
    assert field_cases.run_session_pairs(emu, oracle, fs) == fs.F // 2
