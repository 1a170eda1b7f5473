"""CPU: the 9- and 10-bit H.264 Tier-1 tables (libav_amd/csrc/h264_tier1_hbd.hip) under the SIMT emulator against
1. the REFERENCE's own BIT_DEPTH 9 / 10 instantiations (oracle/_ref/libref.so; only where /root/reference exists): every
   output buffer byte for byte,
2. the golden sha1s made from them (tests/golden/h264dsp_hbd_ref_sha1.json; anywhere)."""
import hashlib
import json
import os

import pytest

import cases_h264_hbd as HB
import providers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h264dsp_hbd_ref_sha1.json")


@pytest.mark.parametrize("bd", (9, 10))
@pytest.mark.parametrize("group", HB.GROUPS)
def test_hbd_tables_emulated_vs_reference(emu, bd, group):
    ref = providers.ref()
    if ref is None:
        pytest.skip("/root/reference not present")
    want = HB.run_group(ref, group, bd)
    got = HB.run_group(emu, group, bd)
    assert list(want) == list(got) and len(want) > 0, (len(want), len(got), sorted(set(want) ^ set(got))[:5])
    bad = [k for k in want if want[k] != got[k]]
    assert not bad, "%d of %d cases differ, first: %s" % (len(bad), len(want), bad[:6])


@pytest.mark.parametrize("bd", (9, 10))
def test_hbd_tables_emulated_vs_golden(emu, bd):
    gold = json.load(open(GOLD))[str(bd)]
    got = HB.run_all(emu, bd)
    assert set(got) == set(gold)
    bad = [k for k, v in got.items() if hashlib.sha1(v).hexdigest()[:20] != gold[k]]
    assert not bad, "%d cases differ, first: %s" % (len(bad), bad[:6])
