"""GPU: field pictures at the Tier-2 API (field_cases.py) against the oracle."""
import pytest

import field_cases
import frame_cases
import h264_frames as HF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in frame_cases.CASES if not n.startswith(("tall", "one_", "mid_intra"))])
def test_field_pictures_gpu(mi355, oracle, name):
    assert field_cases.run(mi355, oracle, HF.synth_frames(**frame_cases.CASES[name])) > 1000
