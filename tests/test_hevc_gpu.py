"""GPU: checkasm-style parity of the HEVC Tier-1 pointer tables (HIP kernels behind the
reference's HEVCDSPContext / HEVCPredContext) against the oracle AND the golden vectors made by
the reference's own objects.  Bit-exact (integer kernels), bit depths 8 and 10 (+9 vs the oracle)."""
import hashlib
import json
import os

import pytest

import cases_hevc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hevcdsp_ref_sha1.json")


@pytest.mark.parametrize("bd", cases_hevc.DEPTHS)
@pytest.mark.parametrize("group", list(cases_hevc.GROUPS))
def test_gpu_hevc_matches_oracle_and_golden(mi355, oracle, group, bd):
    gold = json.load(open(GOLD))
    got = cases_hevc.run_group(mi355, group, bd, gold["seed"])
    want = cases_hevc.run_group(oracle, group, bd, gold["seed"])
    assert want and set(got) == set(want), sorted(set(want) - set(got))[:10]
    bad = [k for k in got if got[k] != want[k]]
    assert not bad, bad[:20]
    for name, data in got.items():
        assert hashlib.sha1(data).hexdigest()[:20] == gold["cases"][name], name


@pytest.mark.parametrize("seed", [11, 0xFEED])
def test_gpu_hevc_matches_oracle_other_seeds(mi355, oracle, seed):
    got = cases_hevc.run_all(mi355, seed, depths=(8, 9, 10))
    want = cases_hevc.run_all(oracle, seed, depths=(8, 9, 10))
    bad = [k for k in got if got[k] != want[k]]
    assert got and not bad, bad[:20]
