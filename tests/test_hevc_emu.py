"""CPU: the HEVC Tier-1 kernels compiled against the SIMT emulator (test tooling) reproduce the
oracle through HEVCDSPContext / HEVCPredContext, bit depths 8, 9 and 10."""
import pytest

import cases_hevc


@pytest.mark.parametrize("bd", (8, 9, 10))
@pytest.mark.parametrize("group", list(cases_hevc.GROUPS))
def test_emulated_hevc_kernels_match_oracle(emu, oracle, group, bd):
    got = cases_hevc.run_group(emu, group, bd)
    want = cases_hevc.run_group(oracle, group, bd)
    assert want and set(got) == set(want), sorted(set(want) - set(got))[:10]
    bad = [k for k in got if got[k] != want[k]]
    assert not bad, bad[:20]
