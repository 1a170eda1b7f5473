"""GPU: mi355_hevc_deblock_pictures_dev vs the oracle and the golden sha1s made by the reference's hevc_filter.c."""
import json

import numpy as np
import pytest

import hevc_filter_cases as HC
from test_oracle_hevc_filter import GOLD, digest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(HC.CASES))
def test_deblock_pictures_gpu(mi355, oracle, name):
    oracle.lib.oracle_hevc_deblock_picture.restype = None
    want, _ = HC.run_host(oracle.lib.oracle_hevc_deblock_picture, name)
    outs, _ = HC.run_device(mi355.lib, name, npics=3)
    for got in outs:
        for c in range(3):
            assert np.array_equal(want[c], got[c]), "%s: plane %d differs (%d bytes)" % (name, c, int((want[c] != got[c]).sum()))
        assert digest(got) == json.load(open(GOLD))["cases"][name]
