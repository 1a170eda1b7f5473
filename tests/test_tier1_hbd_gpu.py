"""GPU: the 9- and 10-bit H.264 Tier-1 tables against the golden sha1s made by the reference's own BIT_DEPTH 9 / 10
template instantiations (tests/golden/h264dsp_hbd_ref_sha1.json; the generating script is tests/golden/make_golden.py)."""
import hashlib
import json

import pytest

import cases_h264_hbd as HB
from test_tier1_hbd_emu import GOLD

pytestmark = pytest.mark.gpu


# "wild" (every sample-reading entry on planes outside the bit depth's range) passed its first hardware run at the end of round 3
# (GPUTEST_r03: 2 xpassed): a plain test now, a regression fails.
_GROUPS = list(HB.GROUPS)


@pytest.mark.parametrize("bd", (9, 10))
@pytest.mark.parametrize("group", _GROUPS)
def test_hbd_tables_gpu_vs_golden(mi355, bd, group):
    gold = json.load(open(GOLD))[str(bd)]
    got = HB.run_group(mi355, group, bd)
    want = {k: v for k, v in gold.items() if k.startswith(group + ":")}
    assert len(got) == len(want) > 0
    bad = [k for k, v in got.items() if hashlib.sha1(v).hexdigest()[:20] != want[group + ":" + k]]
    assert not bad, "%d of %d cases differ, first: %s" % (len(bad), len(got), bad[:6])
