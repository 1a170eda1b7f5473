"""GPU: field pictures at the Tier-2 API (field_cases.py) against the oracle."""
import pytest

import field_cases
import frame_cases
import h264_frames as HF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in frame_cases.CASES if not n.startswith(("tall", "one_", "mid_intra"))])
def test_field_pictures_gpu(mi355, oracle, name):
    assert field_cases.run(mi355, oracle, HF.synth_frames(**frame_cases.CASES[name])) > 1000


@pytest.mark.parametrize("name", ("b_mixed", "mixed_intra", "wide_b", "b_weight_implicit", "mid_b_weighted"))
def test_field_pictures_through_sessions_gpu(mi355, oracle, name):
    fs = HF.synth_frames(**frame_cases.CASES[name])
    assert field_cases.run_session(mi355, oracle, fs) == fs.F


@pytest.mark.parametrize("name", ("b_mixed", "p16_smooth", "mixed_intra"))
def test_field_pairs_in_one_surface_gpu(mi355, oracle, name):
    fs = HF.synth_frames(**frame_cases.CASES[name])
    assert field_cases.run_session_pairs(mi355, oracle, fs) == fs.F // 2
