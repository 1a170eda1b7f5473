"""CPU: the oracle's picture-level HEVC deblocking driver (oracle_hevc_filter.c) against
1. the REFERENCE's own hevc_filter.c compiled in place (oracle/_ref/libhevcfilterref.so; only where /root/reference exists),
2. golden sha1s produced by it (tests/golden/hevc_filter_ref_sha1.json; runs anywhere)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import hevc_filter_cases as HC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "hevc_filter_ref_sha1.json")


def digest(planes):
    h = hashlib.sha1()
    for p in planes:
        h.update(p.tobytes())
    return h.hexdigest()[:20]


def _ref_lib():
    if not os.path.isdir("/root/reference/libavcodec"):
        pytest.skip("/root/reference not present")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libhevcfilterref.so"], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhevcfilterref.so"))
    lib.ref_hevc_deblock_picture.restype = C.c_int
    return lib


@pytest.mark.parametrize("name", list(HC.CASES))
def test_oracle_driver_matches_reference_driver(oracle, name):
    ref = _ref_lib()
    want, case = HC.run_host(ref.ref_hevc_deblock_picture, name)
    oracle.lib.oracle_hevc_deblock_picture.restype = None
    got, _ = HC.run_host(oracle.lib.oracle_hevc_deblock_picture, name)
    before = HC.Case(name).planes
    changed = sum(int((a != b).sum()) for a, b in zip(want, before))
    assert changed > (50 if case.w > 16 else 0), "the filter barely ran (%d bytes changed)" % changed
    for c in range(3):
        assert np.array_equal(want[c], got[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((want[c] != got[c]).sum()))


@pytest.mark.parametrize("name", list(HC.CASES))
def test_oracle_driver_matches_golden(oracle, name):
    gold = json.load(open(GOLD))
    oracle.lib.oracle_hevc_deblock_picture.restype = None
    got, _ = HC.run_host(oracle.lib.oracle_hevc_deblock_picture, name)
    assert digest(got) == gold["cases"][name]


# ---- boundary strengths ---------------------------------------------------------------------------------------------
import hevc_bs_cases as BC  # noqa: E402


@pytest.mark.parametrize("name", list(BC.CASES))
def test_oracle_boundary_strengths_match_reference_function(oracle, name):
    ref = _ref_lib()
    rv, rh, c = BC.run_reference(ref, name)
    ov, oh, _ = BC.run_oracle(oracle.lib, name)
    assert np.array_equal(rv, ov), "vertical_bs differs at %s" % np.flatnonzero(rv != ov)[:8]
    assert np.array_equal(rh, oh), "horizontal_bs differs at %s" % np.flatnonzero(rh != oh)[:8]
    assert len(set(rv.tolist())) == 3 and len(set(rh.tolist())) == 3 or c.w <= 32      # all of 0, 1, 2 occur


@pytest.mark.parametrize("name", list(BC.CASES))
def test_oracle_boundary_strengths_match_golden(oracle, name):
    gold = json.load(open(GOLD))
    ov, oh, _ = BC.run_oracle(oracle.lib, name)
    assert hashlib.sha1(ov.tobytes() + oh.tobytes()).hexdigest()[:20] == gold["bs_cases"][name]
