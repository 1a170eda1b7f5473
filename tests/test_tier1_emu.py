"""CPU: the product's kernels, compiled against the SIMT emulator (tools/simt_emu —
test tooling, not a product path), must reproduce the oracle through the C ABI."""
import pytest

import cases_h264

EXPECT_MISSING = {"startcode"}   # table slots the backend leaves at the C default


@pytest.mark.parametrize("group", list(cases_h264.GROUPS))
def test_emulated_kernels_match_oracle(emu, oracle, group):
    got = cases_h264.run_group(emu, group)
    want = cases_h264.run_group(oracle, group)
    if group in EXPECT_MISSING:
        assert not got
        return
    assert set(got) == set(want), sorted(set(want) - set(got))[:10]
    bad = [k for k in got if got[k] != want[k]]
    assert not bad, bad[:20]
