"""CPU: Tier-2 kernels under the SIMT emulator (test tooling) vs the oracle, bit-exact."""
import pytest

import frame_cases


@pytest.mark.parametrize("name", list(frame_cases.CASES))
def test_frame_pipeline_emulated(emu, oracle, name):
    changed = frame_cases.run_case(emu, oracle, name)
    if "smooth" in str(frame_cases.CASES[name]) and frame_cases.CASES[name]["mb_w"] > 1:
        assert changed > 100, "loop filter barely exercised"
