"""CPU: the oracle's HEVC intra_pred wrapper (oracle_hevc_intra.c) against
1. the REFERENCE's own HEVCPredContext.intra_pred[] compiled in place (oracle/_ref/libhevcfilterref.so; only where
   /root/reference exists),
2. golden sha1s produced by it (tests/golden/hevc_intra_ref_sha1.json; runs anywhere)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import hevc_intra_cases as IC
from test_oracle_hevc_filter import digest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "hevc_intra_ref_sha1.json")


def ref_lib():
    if not os.path.isdir("/root/reference/libavcodec"):
        pytest.skip("/root/reference not present")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libhevcfilterref.so"], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhevcfilterref.so"))


@pytest.mark.parametrize("name", list(IC.CASES))
def test_oracle_matches_reference_intra_pred(oracle, name):
    ref = ref_lib()
    want, c = IC.run_host(ref.ref_hevc_intra_pred_blocks, name)
    oracle.lib.oracle_hevc_intra_pred_blocks.restype = None
    got, _ = IC.run_host(oracle.lib.oracle_hevc_intra_pred_blocks, name)
    for k in range(3):
        assert np.array_equal(want[k], got[k]), "%s: plane %d differs (%d bytes)" % (name, k, int((want[k] != got[k]).sum()))
    assert len(c.launches) > 1


@pytest.mark.parametrize("name", list(IC.CASES))
def test_oracle_matches_golden(oracle, name):
    oracle.lib.oracle_hevc_intra_pred_blocks.restype = None
    got, c = IC.run_host(oracle.lib.oracle_hevc_intra_pred_blocks, name)
    assert digest(got) == json.load(open(GOLD))[name]
    # the guard rows stay untouched, and the case is not degenerate
    for k, pl in enumerate(got):
        assert (pl[0] == 0xA5).all() and (pl[-1] == 0xA5).all()
    fresh = IC.Case(name)
    assert any((a != b).any() for a, b in zip(got, fresh.planes))


def test_paths_are_reached():
    """the case list covers every branch family of the wrapper (counted on the host from the case data)"""
    seen = set()
    for name in IC.CASES:
        c = IC.Case(name)
        for x0, y0, l2, c_idx, mode, cand in c.blocks:
            seen.add(("size", l2)); seen.add(("plane", c_idx)); seen.add(("kind", min(mode, 2)))
            seen.add(("cand", cand))
            if c.cip:
                seen.add("cip")
            if x0 == 0:
                seen.add("x0")
            if y0 == 0:
                seen.add("y0")
    assert all(("size", k) in seen for k in (2, 3, 4, 5)) and all(("plane", k) in seen for k in (0, 1, 2))
    assert all(("kind", k) in seen for k in (0, 1, 2)) and {"cip", "x0", "y0"} <= seen
    assert ("cand", 0) in seen and ("cand", 31) in seen and sum(1 for k in seen if isinstance(k, tuple) and k[0] == "cand") >= 16
