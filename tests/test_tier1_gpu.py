"""GPU: checkasm-style parity of the Tier-1 pointer tables (HIP kernels behind the
reference's H264DSPContext / H264QpelContext / H264ChromaContext / H264PredContext /
VideoDSPContext) against the oracle AND against the reference's golden vectors.
Bit-exact: these are integer kernels."""
import hashlib
import json
import os

import pytest

import cases_h264

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "h264dsp_ref_sha1.json")
EXPECT_MISSING = {"startcode"}


@pytest.mark.parametrize("group", list(cases_h264.GROUPS))
def test_gpu_matches_oracle_and_golden(mi355, oracle, group):
    gold = json.load(open(GOLD))
    got = cases_h264.run_group(mi355, group, gold["seed"])
    want = cases_h264.run_group(oracle, group, gold["seed"])
    if group in EXPECT_MISSING:
        assert not got
        return
    assert set(got) == set(want), sorted(set(want) - set(got))[:10]
    bad = [k for k in got if got[k] != want[k]]
    assert not bad, bad[:20]
    for name, data in got.items():
        assert hashlib.sha1(data).hexdigest()[:20] == gold["cases"][name], name


@pytest.mark.parametrize("seed", [7, 0xABCDEF])
def test_gpu_matches_oracle_other_seeds(mi355, oracle, seed):
    got = cases_h264.run_all(mi355, seed)
    want = cases_h264.run_all(oracle, seed)
    bad = [k for k in got if got[k] != want[k]]
    assert got and not bad, bad[:20]
