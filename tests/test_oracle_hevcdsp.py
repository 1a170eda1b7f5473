"""HEVC oracle pins (CPU): golden vectors from the reference's own objects, and the objects
themselves over more seeds where /root/reference exists.  Bit depths 8 and 10 (9 with the objects)."""
import hashlib
import json
import os

import pytest

import cases_hevc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hevcdsp_ref_sha1.json")


@pytest.mark.parametrize("bd", cases_hevc.DEPTHS)
@pytest.mark.parametrize("group", list(cases_hevc.GROUPS))
def test_oracle_matches_reference_golden(oracle, group, bd):
    gold = json.load(open(GOLD))
    res = cases_hevc.run_group(oracle, group, bd, gold["seed"])
    assert res
    for name, data in res.items():
        assert hashlib.sha1(data).hexdigest()[:20] == gold["cases"][name], name


@pytest.mark.parametrize("seed", [0x265, 5, 6])
def test_oracle_matches_reference_objects(oracle, ref, seed):
    want = cases_hevc.run_all(ref, seed, depths=(8, 9, 10))
    got = cases_hevc.run_all(oracle, seed, depths=(8, 9, 10))
    assert set(want) == set(got)
    bad = [k for k in want if want[k] != got[k]]
    assert not bad, bad[:20]
