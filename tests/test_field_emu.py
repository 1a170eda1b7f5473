"""CPU: field pictures at the Tier-2 API (field_cases.py), emulated kernels against the oracle."""
import pytest

import field_cases
import frame_cases
import h264_frames as HF


@pytest.mark.parametrize("name", ("mixed_intra", "b_mixed", "wide_b", "p16_smooth"))
def test_field_pictures_emulated(emu, oracle, name):
    assert field_cases.run(emu, oracle, HF.synth_frames(**frame_cases.CASES[name])) > 1000
